"""Device idle time of the timed steps from a `rocprofv3 --kernel-trace` CSV of bench.py: wall minus the UNION of the
kernel intervals of all streams (tools/gap_analysis.py looks at consecutive kernels of the merged order instead).

    python tools/idle_analysis.py <kernel_trace.csv> [first step] [last step]     (indices of the forward passes; default -6 -2)

Steps are delimited by the weight_pack_batch_kernel launches (one per forward pass).  Per step: wall, union-busy, idle, idle
by gap length, and the longest idle gaps with the kernel that ended before and the one that started after."""
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
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else -6
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    for a, b in list(zip(marks, marks[1:]))[lo:hi]:
        seq = rows[a:b]
        t0, t1 = seq[0][0], max(e for _, e, _ in seq)
        gaps, cur_end, last = [], seq[0][1], seq[0][2]
        for s, e, n in seq[1:]:
            if s > cur_end:
                gaps.append(((s - cur_end) / 1e3, last[:44], n[:44]))
            if e > cur_end:
                cur_end, last = e, n
        idle = sum(g[0] for g in gaps) / 1e3
        hist = [0.0, 0.0, 0.0, 0.0]
        for g in gaps:
            hist[0 if g[0] < 3 else 1 if g[0] < 10 else 2 if g[0] < 50 else 3] += g[0] / 1e3
        print('step: %d kernels  wall %.2f ms  idle %.2f ms (%d gaps)  by gap length: <3us %.2f | 3-10us %.2f | 10-50us %.2f | >50us %.2f'
              % (len(seq), (t1 - t0) / 1e6, idle, len(gaps), *hist))
        for g in sorted(gaps, reverse=True)[:10]:
            print('   %8.1f us  after %-44s before %s' % g)


if __name__ == '__main__':
    main()
