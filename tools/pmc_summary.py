"""Per-kernel averages of the counters collected by tools/pmc_passes.sh (one row per kernel name and counter)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_' not in k:
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if r.get('Start_Timestamp'):
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k in sorted(acc):
    d = dur[k]
    print('==', k, ' launches/pass', len(next(iter(acc[k].values()))), ' avg ms %.3f' % (sum(d) / max(len(d), 1)))
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    for n in sorted(c):
        print('   %-36s %16.0f' % (n, c[n]))
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']
        print('   -> wait_any %.1f%%  wait_inst_any %.1f%%  active %.1f%% of wave cycles' % (
            100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc))
    if 'GRBM_GUI_ACTIVE' in c:
        print('   -> GRBM_GUI_ACTIVE / 8 = %.0f cycles' % (c['GRBM_GUI_ACTIVE'] / 8))
