"""Condense a `rocprofv3 --kernel-trace --stats` run of bench.py into the table kept under profiles/.

    python tools/profile_summary.py <kernel_stats.csv> <bench json line file> [steps executed] > profiles/rNN_summary.md

bench.py's `roofline` object averages over every b2m_conv_fwd launch of a step (all template variants of
conv_fwd_kernel); rocprofv3 lists the variants separately, so the comparable figure is the call-weighted mean
over the variants, printed here next to bench.py's own HIP-event figure.
"""
import csv
import json
import re
import sys


def main():
    stats, bench = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(stats)))
    line = json.loads([l for l in open(bench) if l.startswith('{"metric"')][-1])
    # every step the process ran (set-up, warm-up, timed, and the H2D-inclusive repeat): bench.py reports the count
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else int(line['config']['steps_executed'])
    groups = {}
    total = 0.0
    for r in rows:
        name = re.sub(r'^void ', '', r['Name'])
        base = re.split(r'[<(]', name)[0]
        if base.startswith('conv_wgrad'):
            base = 'conv_wgrad_kernel'          # plain + flat-pipeline variants: one b2m_conv_wgrad entry
        if base.startswith('conv_fwd') or base.startswith('conv_1x1'):
            base = 'conv_fwd_kernel'            # likewise for b2m_conv_fwd
        if base.startswith('at::') or base.startswith('__amd') or 'Cijk' in base:
            base = 'torch / runtime kernels'
        g = groups.setdefault(base, [0, 0.0])
        g[0] += int(r['Calls'])
        g[1] += float(r['TotalDurationNs'])
        total += float(r['TotalDurationNs'])
    print('# rocprofv3 --kernel-trace --stats of `python3 bench.py` (%d steps in all: set-up, warm-up, timed, H2D-inclusive repeat)\n' % steps)
    print('bench line under the profiler: %.2f ms/step, %.2f scenes/s\n' % (line['ms_per_step'], line['value']))
    print('| kernel (all template variants) | launches/step | avg launch (ms) | ms/step | share |')
    print('|---|---:|---:|---:|---:|')
    for base, (calls, ns) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print('| %s | %.1f | %.4f | %.2f | %.1f %% |' % (base, calls / steps, ns / calls / 1e6, ns / steps / 1e6,
                                                          100 * ns / total))
    print('| **all kernels** | | | %.2f | |' % (total / steps / 1e6))
    print()
    for key, base in (('roofline', 'conv_fwd_kernel'), ('roofline_wgrad', 'conv_wgrad_kernel')):
        if key in line and base in groups:
            calls, ns = groups[base]
            print('- %s: rocprofv3 average %.4f ms over %d launches; bench.py HIP events %.4f ms '
                  '(%s launches/step) -> %.1f TFLOP/s, frac %.3f' % (
                      base, ns / calls / 1e6, calls, line[key]['avg_launch_ms'], line[key]['launches_per_step'],
                      line[key]['achieved'], line[key]['frac']))


if __name__ == '__main__':
    main()
