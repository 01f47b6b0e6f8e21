#!/usr/bin/env python3
"""Writes profiles/pmc_traffic.json from two rocprofv3 --pmc passes (tools/gpu_pmc_traffic.sh):

    tools/pmc_traffic_json.py <key> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [--launches-per-frame N]

HBM bytes per conv launch = (sum FETCH_SIZE x 2 + sum WRITE_SIZE) x 1024 / conv dispatches, over every conv3x3 kernel of the
run: FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section).  The file is stamped with the hash of the kernel sources (bench.csrc_sha16) so that
bench.py reports `traffic: null` once the kernels have changed."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402


def total(d, counter):
    n, s = 0, 0.0
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "conv3x3" in r["Kernel_Name"]:
                n += 1
                s += float(r["Counter_Value"])
    return n, s


def main():
    key, dfetch, dwrite = sys.argv[1:4]
    nf, fetch = total(dfetch, "FETCH_SIZE")
    nw, write = total(dwrite, "WRITE_SIZE")
    assert nf == nw and nf > 0, (nf, nw)
    per_launch = int((2 * fetch + write) * 1024 / nf)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        d = {}
    sha = csrc_sha16()
    if d.get("csrc_sha16") != sha:
        d = {"csrc_sha16": sha}
    d[key] = per_launch
    json.dump(d, open(path, "w"))
    print(f"{key}: {per_launch} bytes per conv launch over {nf} dispatches (fetch x2 {2 * fetch * 1024 / nf:.0f} + write {write * 1024 / nf:.0f}); csrc {sha}")


if __name__ == "__main__":
    main()
