#!/usr/bin/env python3
"""Times single conv layer shapes through the C ABI test hook (for rocprofv3 runs).
usage: tools/conv_micro.py [--dtype bf16|f32] [--n 8] [--hw 266] [--reps 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_enhanced_super_resolution_amd import conv3x3  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--hw", type=int, default=266)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--shapes", default="64x32,96x32,128x32,160x32,192x64,64x64")
a = ap.parse_args()
for sh in a.shapes.split(","):
    cin, cout = map(int, sh.split("x"))
    x = torch.randn(a.n, cin, a.hw, a.hw, device="cuda")
    w = torch.randn(cout, cin, 3, 3) * 0.05
    b = torch.zeros(cout)
    for _ in range(a.reps):
        y = conv3x3(x, w, b, lrelu=True, dtype=a.dtype)
    torch.cuda.synchronize()
    print(sh, "done", float(y.abs().mean()))
