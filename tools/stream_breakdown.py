"""Per-stream composition of the timed training steps, from a `rocprofv3 --kernel-trace` CSV of bench.py: which kernels sit on
which HIP stream, how long, and how much of the step's wall time the main stream spends in neither MFMA kernel.

    python tools/stream_breakdown.py <kernel_trace.csv>

Steps are cut at the weight_pack_batch_kernel launches (one per forward pass), the last three are printed."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace('void ', '')
    for pre in ('at::native::', '(anonymous namespace)::'):
        n = n.replace(pre, '')
    return n.split('(')[0].split('<')[0][:44]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', '?'), r.get('Queue_Id', '?')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith('weight_pack_batch_kernel')]
    marks = [i for j, i in enumerate(marks) if j == 0 or rows[i][0] - rows[marks[j - 1]][0] > 5_000_000]   # (round 6: up to three pack launches per pass)
    for a, b in list(zip(marks, marks[1:]))[-3:]:
        seq = rows[a:b]
        wall = (seq[-1][1] - seq[0][0]) / 1e6
        print('step: %d kernels, wall %.2f ms' % (len(seq), wall))
        by = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for s, e, n, st, q in seq:
            c = by[(st, q)][short(n)]
            c[0] += 1; c[1] += (e - s) / 1e6
        for key in sorted(by, key=lambda k: -sum(v[1] for v in by[k].values())):
            tot = sum(v[1] for v in by[key].values()); cnt = sum(v[0] for v in by[key].values())
            print('  stream %s queue %s: %d kernels, %.2f ms' % (key[0], key[1], cnt, tot))
            for n, (c, t) in sorted(by[key].items(), key=lambda kv: -kv[1][1])[:14]:
                print('      %-46s %4d  %7.3f ms' % (n, c, t))


if __name__ == '__main__':
    main()
