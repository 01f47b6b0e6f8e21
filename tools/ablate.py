#!/usr/bin/env python3
"""Runs single conv layer shapes through the test hook of a GIVEN build of libnesr_hip.so (under
rocprofv3 --kernel-trace; tools/trace_seq.py then lists the kernel time per shape).
usage: tools/ablate.py <lib.so> [--dtype f32] [--n 1] [--hw 256] [--reps 6] [--shapes 64x32,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neural_enhanced_super_resolution_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--n", type=int, default=1)
ap.add_argument("--hw", type=int, default=256)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--shapes", default="64x32,96x32,128x32,160x32,192x64")
a = ap.parse_args()
_lib.LIB_PATH = os.path.abspath(a.lib)
from neural_enhanced_super_resolution_amd import conv3x3  # noqa: E402

for sh in a.shapes.split(","):
    cin, cout = map(int, sh.split("x"))
    x = torch.randn(a.n, cin, a.hw, a.hw, device="cuda")
    w = torch.randn(cout, cin, 3, 3) * 0.05
    b = torch.zeros(cout)
    for _ in range(a.reps):
        y = conv3x3(x, w, b, lrelu=True, dtype=a.dtype)
    torch.cuda.synchronize()
    print(sh, "done", float(y.abs().mean()))
    lib = _lib.load()
    if hasattr(lib, "nesr_debug_stamps"):      # diagnostic builds (-DNESR_ABL=64): in-kernel cycle stamps
        import ctypes
        buf = (ctypes.c_uint64 * 256)()
        lib.nesr_debug_stamps(buf, 256)
        t0 = buf[0]
        nch = -(-cin // 16)
        print("  stamps (cycles from kernel start): prologue issued %d, loop end %d, epilogue issued %d, stores done %d" %
              (buf[1] - t0, buf[2] - t0, buf[3] - t0, buf[4] - t0))
        if buf[204] > buf[200]:
            print("  in-kernel clock %.2f GHz (s_memtime / s_memrealtime at 100 MHz), workgroup lifetime %.2f us" %
                  ((buf[4] - buf[0]) / (buf[204] - buf[200]) * 0.1, (buf[204] - buf[200]) / 100.0))
        print("  setup: index %d, dma plan %d, operands/bias %d, prologue dma issue %d" % (buf[5] - t0, buf[6] - buf[5], buf[7] - buf[6], buf[1] - buf[7]))
        for c in range(nch):
            b = [buf[8 + 4 * c + i] for i in range(4)]
            prev = buf[11 + 4 * (c - 1)] if c else buf[1]
            print("  chunk %2d: dma wait %5d  barrier %5d  dma issue %5d  mfma phase %5d" % (c, b[0] - prev, b[1] - b[0], b[2] - b[1], b[3] - b[2]))
