"""Per-step launch census of a `rocprofv3 --kernel-trace` CSV of bench.py: launches, kernel time and the idle time IN FRONT of
each class of kernel (time no kernel of any stream runs, booked to the kernel that ends the gap), and the same split by the
duration of the kernel behind the gap -- where the launches are and whose gaps the idle time is.

    python tools/step_kernels.py <kernel_trace.csv> [first step] [last step]      (default -24 -20: the timed region)"""
import collections
import csv
import re
import sys


def klass(n):
    if n.startswith('void '):
        n = n[5:]
    m = re.match(r'_Z\d+([a-z_0-9]+?)(?:I|P|v|E)', n)          # a mangled name (kernels with _Float16 parameters are not demangled)
    if m:
        n = m.group(1)
    for pre, k in (('conv_wgrad', 'wgrad'), ('wgrad_narrow', 'wgrad'), ('conv_', 'conv_fwd'), ('bn_', 'batchnorm'),
                   ('reduce_final', 'batchnorm'), ('__amd_rocclr_copy', 'rt copy'), ('__amd_rocclr_fill', 'rt memset'),
                   ('at::', 'torch'), ('rocprim', 'torch'), ('deep_', 'deep program'), ('weight_pack', 'weight pack'), ('pool_', 'pooling')):
        if n.startswith(pre):
            return k
    return 'maps / other b2m'


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith('weight_pack_batch_kernel')]
    # (round 6: a training pass packs its images in up to three launches, the later ones on the side stream: a step begins at
    # the first pack launch that follows the previous one by more than 5 ms)
    marks = [i for j, i in enumerate(marks) if j == 0 or rows[i][0] - rows[marks[j - 1]][0] > 5_000_000]
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else -24
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else -20
    steps = list(zip(marks, marks[1:]))[lo:hi]
    cnt = collections.Counter(); busy = collections.Counter(); idle = collections.Counter(); idle_n = collections.Counter()
    by_len = collections.Counter(); by_len_n = collections.Counter(); queues = collections.Counter()
    per_q = collections.defaultdict(collections.Counter)
    wall = union = 0.0
    for a, b in steps:
        seq = rows[a:b]
        t0, cur_end = seq[0][0], seq[0][0]
        for s, e, n, q in seq:
            k = klass(n)
            cnt[k] += 1; busy[k] += e - s; queues[q] += 1; per_q[q][k] += e - s
            if s > cur_end:
                g = s - cur_end
                idle[k] += g; idle_n[k] += 1
                d = e - s
                c = '<10us' if d < 10e3 else '10-30us' if d < 30e3 else '30-100us' if d < 100e3 else '>=100us'
                by_len[c] += g; by_len_n[c] += 1
            else:
                union -= min(cur_end, e) - s          # overlap with what is already counted
            union += e - s
            cur_end = max(cur_end, e)
        wall += cur_end - t0
    ns = max(len(steps), 1)
    print('%d steps: wall %.2f ms/step, kernel sum %.2f, union %.2f, idle %.2f; launches/step %.0f; queues %s'
          % (ns, wall / ns / 1e6, sum(busy.values()) / ns / 1e6, union / ns / 1e6, (wall - union) / ns / 1e6, sum(cnt.values()) / ns,
             dict(queues)))
    print('%-18s %9s %9s %12s %9s' % ('class', 'launches', 'ms/step', 'idle before', 'gaps'))
    for k, _ in busy.most_common():
        print('%-18s %9.1f %9.3f %12.3f %9.1f' % (k, cnt[k] / ns, busy[k] / ns / 1e6, idle[k] / ns / 1e6, idle_n[k] / ns))
    print('per queue (stream): kernel time by class, ms/step')
    for qid in sorted(per_q):
        tot = sum(per_q[qid].values())
        print('   queue %-3s %8.3f  %s' % (qid, tot / ns / 1e6, '  '.join('%s %.2f' % (k, v / ns / 1e6) for k, v in per_q[qid].most_common(6))))
    print('idle in front of kernels by THEIR duration:')
    for c in ('<10us', '10-30us', '30-100us', '>=100us'):
        print('   %-9s %8.3f ms/step in %6.1f gaps' % (c, by_len[c] / ns / 1e6, by_len_n[c] / ns))


if __name__ == '__main__':
    main()
