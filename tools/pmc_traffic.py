"""Per-kernel HBM traffic of one bench step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
--kernel-trace only), corrected as MI355X_MICROARCH.md's HBM section prescribes: both counters are in KiB, and on
gfx950 FETCH_SIZE tallies wide reads at half their bytes (x2).  The x2 is calibrated here on bn_apply2_kernel (two tensors
streamed in, one of the same size out; WRITE_SIZE is exact): the ratio is printed.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_traffic.json
"""
import collections
import csv
import json
import sys


def load(path):
    by = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        base = r['Kernel_Name'].replace('void ', '').split('<')[0].split('(')[0]
        if base.startswith('conv_wgrad'):
            base = 'conv_wgrad_kernel'          # plain + flat-pipeline variants: one b2m_conv_wgrad entry
        if base.startswith('conv_fwd') or base.startswith('conv_1x1'):
            base = 'conv_fwd_kernel'            # likewise for b2m_conv_fwd
        by[base][0] += float(r['Counter_Value']); by[base][1] += 1
    return by


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {'_units': 'bytes per launch; fetch = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024',
           '_source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 1 --warmup 1 '
                      '--cpu-baseline 0 --votes 1 --prepare 0 (two passes)'}
    # calibration of the x2: bn_apply2_kernel streams two tensors in and one of the same size out (WRITE_SIZE is exact), so
    # raw FETCH_SIZE / WRITE_SIZE = 1.0 when FETCH_SIZE counts half the streamed bytes (2.0 if it counted all of them)
    if 'bn_apply2_kernel' in fetch and 'bn_apply2_kernel' in write:
        cal = fetch['bn_apply2_kernel'][0] / max(write['bn_apply2_kernel'][0], 1e-9)
        out['_calibration'] = 'raw FETCH_SIZE / WRITE_SIZE of bn_apply2_kernel (reads two tensors, writes one of the same size) ' \
                              '= %.4f -> FETCH_SIZE counts %.3f of the streamed bytes, as the guide states' % (cal, cal / 2)
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, n = fetch[k]
        w = write.get(k, [0.0, 0])[0]
        if f + w < 1000:
            continue
        out[k] = {'launches': n, 'fetch_size_raw_kib': round(f / n, 1), 'write_size_kib': round(w / n, 1),
                  'fetch_bytes': round(2 * f * 1024 / n), 'write_bytes': round(w * 1024 / n),
                  'traffic_bytes': round((2 * f + w) * 1024 / n)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
