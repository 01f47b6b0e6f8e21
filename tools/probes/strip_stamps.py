"""In-kernel timeline of rdb_bf16_strip_kernel (build with -DNESR_STRIP_ABL=256): position 5 of the first strip of workgroup 77.
MFMA waves: events 0 barrier arrival, 1 release, 2 end of the step's MFMAs, 3 end of the layer epilogue.
DMA waves: 0 end of the step's work (before the wait), 1 after vmcnt(0), 2 barrier release, 3 weights issued."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.ab import load, make_ctx  # noqa: E402
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict  # noqa: E402

os.environ["NESR_STRIP"] = "1"
lib = load(os.path.abspath(sys.argv[1]))
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 15
ctx = make_ctx(lib, synthetic_state_dict(seed=0, num_in_ch=3, scale=2), 1)
x = torch.rand(batch, 3, 532, 532, device="cuda")
y = torch.empty(batch, 3, 1064, 1064, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert lib.nesr_forward(ctx, ctypes.c_void_p(x.data_ptr()), batch, 3, 532, 532, ctypes.c_void_p(y.data_ptr()), st) == 0
torch.cuda.synchronize()
NW = 8
buf = (ctypes.c_ulonglong * (NW * 32 * 8))()
lib.nesr_debug_strip_stamps(buf)
s = [[[buf[(w * 32 + k) * 8 + e] for e in range(8)] for k in range(32)] for w in range(NW)]
t0 = s[0][0][0]
print("step | MFMA wave 0: arrive release(+wait) mfma epi | step period | DMA wave 0: work-done vmcnt-done release issued | DMA wave 1 work-done vmcnt-done")
for k in range(26):
    m = s[0][k]
    nxt = s[0][k + 1][1] if k < 25 else 0
    d = s[4][k]
    d1 = s[7][k]
    print(f"{k:2d} | arr {m[0]-t0:7d} wait {m[1]-m[0]:5d} mfma {m[2]-m[1]:5d} epi {max(0, m[3]-m[2]) if m[3] > m[2] else 0:5d} | period {nxt - m[1] if nxt else 0:5d} |"
          f" dma0: done {d[0]-t0:7d} vm {d[1]-d[0]:5d} rel {d[2]-d[1]:5d} issue {d[3]-d[2]:5d} | dma3: done {d1[0]-t0:7d} vm {d1[1]-d1[0]:5d}")
dr, dt = s[0][31][0] - s[0][30][0], s[0][31][1] - s[0][30][1]
print(f"position: {dt} memtime ticks in {dr} realtime ticks (100 MHz) = {dr / 100:.2f} us -> memtime rate {dt / max(dr, 1) * 100:.0f} MHz")
mf = sum(s[0][k][2] - s[0][k][1] for k in range(26)); wt = sum(s[0][k][1] - s[0][k][0] for k in range(26)); ep = sum(max(0, s[0][k][3] - s[0][k][2]) for k in range(26) if s[0][k][3] > s[0][k][2])
print(f"sums: mfma {mf} wait {wt} epilogue {ep} other {dt - mf - wt - ep}")
print("per-wave barrier arrival of steps 6..9 (relative to wave 0's):")
for k in (3, 6, 10, 14, 15, 19, 20, 21):
    print(k, [s[w][k][0] - s[0][k][0] for w in range(NW)], "release", [s[w][k][1 if w < 4 else 2] - s[0][k][0] for w in range(NW)])
