#!/usr/bin/env python3
"""Per-layer timing from a rocprofv3 --kernel-trace CSV of bench.py (workload c2).
usage: tools/trace_layers.py <kernel_trace.csv> [internal_pixels]"""
import collections
import csv
import sys

f = sys.argv[1]
px0 = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp']), int(r['End_Timestamp']))
      for r in rows if 'conv3x3' in r['Kernel_Name'] or 'pack_input' in r['Kernel_Name']]
ks = [k for k in ks] + [(r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp']), int(r['End_Timestamp']))
                        for r in rows if 'rdb_f16x2' in r['Kernel_Name']]
ks.sort(key=lambda k: k[2])
idx = [i for i, k in enumerate(ks) if 'pack_input' in k[0]]
i0 = idx[-2]
if any('rdb_f16x2' in k[0] for k in ks[i0:idx[-1]]):
    # fused dense blocks: pack_input, conv_first, 69 x rdb_f16x2_kernel, conv_body, up1, up2, hr, last
    seq = ks[i0:idx[-1]]
    tot = sum(k[1] for k in seq)
    span = seq[-1][3] - seq[0][2]
    print("forward: kernels %d, sum kernel %.3f ms, span %.3f ms, gaps %.3f ms" % (len(seq), tot / 1e6, span / 1e6, (span - tot) / 1e6))
    rdb = [k for k in seq if 'rdb_f16x2' in k[0]]
    d = [k[1] for k in rdb]
    gaps = [rdb[i + 1][2] - rdb[i][3] for i in range(len(rdb) - 1)]
    fl = 2 * 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) * px0
    print("rdb_f16x2_kernel x%d: avg %.2f us  min %.2f  max %.2f   %.1f TFLOP/s algorithmic; gap to the next launch avg %.2f us  min %.2f  max %.2f"
          % (len(rdb), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, fl / (sum(d) / len(d)) / 1e3, sum(gaps) / len(gaps) / 1e3, min(gaps) / 1e3, max(gaps) / 1e3))
    print("dense blocks: kernel sum %.3f ms, span %.3f ms" % (sum(d) / 1e6, (rdb[-1][3] - rdb[0][2]) / 1e6))
    for k in seq:
        if 'rdb_f16x2' not in k[0]:
            print("%-60s %8.1f us" % (k[0][:60], k[1] / 1e3))
    sys.exit(0)
seq = ks[i0:i0 + 352]
tot = sum(k[1] for k in seq)
span = seq[-1][3] - seq[0][2]
print("forward: kernels %d, sum kernel %.3f ms, span %.3f ms, gaps %.3f ms" % (len(seq), tot / 1e6, span / 1e6, (span - tot) / 1e6))
names = ['conv1', 'conv2', 'conv3', 'conv4', 'conv5']
agg = collections.defaultdict(list)
for j in range(345):
    agg[names[j % 5]].append(seq[2 + j][1])
macs = {'conv1': 64 * 32, 'conv2': 96 * 32, 'conv3': 128 * 32, 'conv4': 160 * 32, 'conv5': 192 * 64}
for n in names:
    v = agg[n]
    avg = sum(v) / len(v)
    print("%-9s avg %7.1f us  min %7.1f  max %7.1f   %6.1f TFLOP/s" % (n, avg / 1e3, min(v) / 1e3, max(v) / 1e3, 2 * 9 * macs[n] * px0 / avg / 1e3))
print("%-9s %7.1f us" % ("conv_first", seq[1][1] / 1e3))
for j, n in enumerate(['conv_body', 'up1', 'up2', 'hr', 'last']):
    d = seq[347 + j][1]
    px = {'conv_body': 1, 'up1': 4, 'up2': 16, 'hr': 16, 'last': 16}[n] * px0
    co = 3 if n == 'last' else 64
    print("%-9s %7.1f us   %6.1f TFLOP/s (algorithmic)" % (n, d / 1e3, 2 * 9 * 64 * co * px / d / 1e3))
body = sum(seq[2 + j][1] for j in range(345))
print("body: kernel sum %.3f ms, span %.3f ms" % (body / 1e6, (seq[346][3] - seq[2][2]) / 1e6))
