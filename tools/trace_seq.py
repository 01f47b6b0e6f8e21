#!/usr/bin/env python3
"""Per-shape kernel time from a rocprofv3 kernel-trace CSV of tools/ablate.py / conv_micro.py:
conv kernels in dispatch order, consecutive runs of `reps`, average of all but the first.
usage: tools/trace_seq.py <kernel_trace.csv> <reps> [name-substring=conv3x3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else "conv3x3"
rows = sorted((r for r in rows if sub in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
out = []
for i in range(0, len(rows) - reps + 1, reps):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i + 1:i + reps]]
    out.append("%.1f" % (sum(d) / len(d)))
print(" ".join(out))
