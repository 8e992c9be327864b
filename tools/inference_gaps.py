"""Kernel time and idle time of the batch-size-1 forward passes in a rocprofv3 kernel trace of tools/inference_trace.py.
    python tools/inference_gaps.py <kernel_trace.csv>
A pass begins with its hilbert_keys_kernel launch (the row order of the scene's coordinate manager)."""
import csv, sys
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])))
marks = [i for i, r in enumerate(rows) if 'hilbert_keys_kernel' in r[2]]
if not marks:
    print(sorted(set(r[2][:60] for r in rows))[:40])
print('%d kernels, %d passes' % (len(rows), len(marks)))
tot = []
for a, b in list(zip(marks, marks[1:]))[3:]:
    seq = rows[a:b]
    period = (rows[b][0] - seq[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seq) / 1e6
    cur, idle = seq[0][1], 0.0
    for s_, e_, _ in seq[1:]:
        idle += max(s_ - cur, 0) / 1e6
        cur = max(cur, e_)
    conv = sum(e - s for s, e, n in seq if 'conv_' in n) / 1e6
    tot.append((period, busy, idle, conv, len(seq)))
k = len(tot)
print('per pass (mean of %d): period %.3f ms | kernels %.3f ms (convolutions %.3f) | no kernel running %.3f ms | %d launches'
      % (k, sum(t[0] for t in tot) / k, sum(t[1] for t in tot) / k, sum(t[3] for t in tot) / k, sum(t[2] for t in tot) / k, tot[0][4]))
