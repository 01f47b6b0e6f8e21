#!/usr/bin/env python3
"""Groups a rocprofv3 kernel-trace CSV by (kernel, grid size): count, avg/min us.
usage: tools/trace_groups.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("nesr::(anonymous namespace)::", "").split("(")[0][-44:]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"))
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-46s grid %-9s lds %-6s vgpr %-4s n %5d  avg %8.1f us  min %8.1f  share %5.1f%%" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, 100 * sum(v) / tot))
