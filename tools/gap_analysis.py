"""Idle time between consecutive kernels of the timed steps, from a `rocprofv3 --kernel-trace` CSV of bench.py:
where does wall time exceed the sum of the kernel durations?

    python tools/gap_analysis.py <kernel_trace.csv> [steps_executed]

The last `steps` repetitions of the per-step kernel sequence are located by the weight_pack_batch_kernel launches (one per
forward pass).  Prints per step: wall, kernel time, idle, and the largest gaps with the kernels on either side."""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith('weight_pack_batch_kernel')]
    # (round 6: a training pass packs its images in up to three launches, the later ones on the side stream: a step begins at
    # the first pack launch that follows the previous one by more than 5 ms)
    marks = [i for j, i in enumerate(marks) if j == 0 or rows[i][0] - rows[marks[j - 1]][0] > 5_000_000]
    print('%d kernels, %d forward passes' % (len(rows), len(marks)))
    for a, b in list(zip(marks, marks[1:]))[-4:]:
        seq = rows[a:b]
        wall = (seq[-1][1] - seq[0][0]) / 1e6
        busy = sum(e - s for s, e, _ in seq) / 1e6
        # (kernels of the two streams overlap: a gap is the time NO kernel runs -- from the latest end so far to the next start)
        gaps, cur_end, last = [], seq[0][1], seq[0][2]
        for s_, e_, n_ in seq[1:]:
            gaps.append(((s_ - cur_end) / 1e3, last[:50], n_[:50]))
            if e_ > cur_end:
                cur_end, last = e_, n_
        idle = sum(max(g[0], 0) for g in gaps) / 1e3
        print('step: %d kernels  wall %.2f ms  kernels %.2f ms  idle %.2f ms' % (len(seq), wall, busy, idle))
        hist = [0, 0, 0, 0]
        for g in gaps:
            hist[0 if g[0] < 3 else 1 if g[0] < 10 else 2 if g[0] < 50 else 3] += max(g[0], 0) / 1e3
        print('   idle by gap size: <3us %.2f ms | 3-10us %.2f | 10-50us %.2f | >50us %.2f' % tuple(hist))
        for g in sorted(gaps, reverse=True)[:8]:
            print('   %8.1f us  after %-50s before %s' % g)


if __name__ == '__main__':
    main()
