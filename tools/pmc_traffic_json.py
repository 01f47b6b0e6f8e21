#!/usr/bin/env python3
"""Writes profiles/pmc_traffic.json from two rocprofv3 --pmc passes (tools/gpu_pmc_traffic.sh):

    tools/pmc_traffic_json.py <key> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [--launches-per-frame N]

HBM bytes per launch = (sum FETCH_SIZE x 2 + sum WRITE_SIZE) x 1024 / dispatches, over the dominant kernel of the run (the
fused dense-block kernel, or every conv3x3 kernel when the run used per-layer launches): FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
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
    """(dispatches, sum) over the dominant kernel of the run: the fused dense-block kernel when the run used it (its 69
    launches per frame are the 345 dense-block convs -- the kernel bench.py's `roofline` is about), else every conv3x3 kernel."""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    fused = [r for r in rows if "rdb_f16x2" in r["Kernel_Name"] or "rdb_bf16_strip" in r["Kernel_Name"]]
    sel = fused if fused else [r for r in rows if "conv3x3" in r["Kernel_Name"]]
    return len(sel), sum(float(r["Counter_Value"]) for r in sel)


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
    print(f"{key}: {per_launch} bytes per launch over {nf} dispatches (fetch x2 {2 * fetch * 1024 / nf:.0f} + write {write * 1024 / nf:.0f}); csrc {sha}")


if __name__ == "__main__":
    main()
