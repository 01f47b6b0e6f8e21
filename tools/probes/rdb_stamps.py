#!/usr/bin/env python3
"""In-kernel timeline of the fused dense-block kernel (build with -DNESR_RDB_ABL=256): per step, cycles (s_memtime) that
MFMA wave 1 and DMA wave 0 of workgroup 77 spend waiting at the barrier / computing / issuing DMAs / in the epilogue.
usage: tools/probes/rdb_stamps.py build/v_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools.ab import load, make_ctx
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
lib = load(os.path.abspath(sys.argv[1]))
sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
ctx = make_ctx(lib, sd, 3)
x = torch.rand(1, 3, 512, 512, device="cuda")
y = torch.empty(1, 3, 1024, 1024, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    assert lib.nesr_forward(ctx, ctypes.c_void_p(x.data_ptr()), 1, 3, 512, 512, ctypes.c_void_p(y.data_ptr()), st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 8))()
lib.nesr_debug_rdb_stamps.argtypes = [ctypes.c_void_p]
assert lib.nesr_debug_rdb_stamps(buf) == 0
S = [[[buf[(r * 64 + s) * 8 + e] for e in range(8)] for s in range(64)] for r in range(2)]
t0 = min(S[0][0][0], S[1][0][0])
print("step | MFMA: arrive  wait  compute | DMA: arrive(vm wait) barrier-wait  issue  epilogue | step length")
for s in range(52):
    m, d = S[0][s], S[1][s]
    nm = S[0][s + 1][0] if s < 51 else m[2]
    nd = S[1][s + 1][0] if s < 51 else d[3]
    ep = ""
    if m[3] > m[1] and m[6] > m[3]:
        ep = f"  epilogue in the MFMA wave: start +{m[3]-m[1]} exchange/bias/act {m[4]-m[3]} split/regroup {m[5]-m[4]} stores {m[6]-m[5]}"
    print(f"{s:3d} | {m[0]-t0:8d} {m[1]-m[0]:6d} {m[2]-m[1]:6d} | {d[0]-t0:8d} ({d[1]-d[0]:5d}) {d[2]-d[1]:6d} {d[3]-d[2]:6d} {nd-d[3]:6d} | {nm-m[0]:6d}{ep}")
print("total cycles", S[0][51][2] - t0)
dr, dt = S[1][63][0] - S[1][62][0], S[1][63][1] - S[1][62][1]
print(f"steps 0..51: {dt} shader cycles in {dr} ticks of the 100 MHz clock = {dr / 100:.2f} us: shader clock {dt / dr * 100:.0f} MHz")

arr = (ctypes.c_ulonglong * (12 * 8 * 2))()
lib.nesr_debug_rdb_arrive.argtypes = [ctypes.c_void_p]
if lib.nesr_debug_rdb_arrive(arr) == 0:
    print("barrier arrival of every wave (MFMA 0-7, DMA 8-11) relative to the release of that barrier, steps 20..27")
    for st in range(8):
        rel = max(arr[(w * 8 + st) * 2 + 0] for w in range(12))
        print(20 + st, " ".join(f"{int(arr[(w * 8 + st) * 2 + 0]) - int(rel):6d}" for w in range(12)),
              "| release seen", " ".join(f"{int(arr[(w * 8 + st) * 2 + 1]) - int(rel):5d}" for w in range(12)))

tp = (ctypes.c_ulonglong * (2 * 8 * 8))()
if hasattr(lib, "nesr_debug_rdb_taps"):
    lib.nesr_debug_rdb_taps.argtypes = [ctypes.c_void_p]
    if lib.nesr_debug_rdb_taps(tp) == 0:
        print("tap-steps of MFMA waves 1 and 5 (the two of one SIMD), steps 20..27: start of tap-step 0..4 and end, relative to wave 1's tap-step 0")
        for st in range(8):
            b = int(tp[(0 * 8 + st) * 8 + 0])
            for w in range(2):
                print(20 + st, "wave", 1 + 4 * w, " ".join(f"{int(tp[(w * 8 + st) * 8 + i]) - b:6d}" for i in range(6)))
