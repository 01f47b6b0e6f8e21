"""Probe: two 2160p bf16 frames (upstream's 512/10 tile grid each) in flight on one GPU vs one at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
ups = []
for i in range(2):
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, compute_dtype=dt), tile=512, tile_pad=10, pre_pad=0, device="cuda")
    up.pre_process(np.ascontiguousarray(synthetic_frame(2160, 3840, seed=i)[:, :, ::-1].astype(np.float32) / 255.0))
    ups.append(up)
streams = [torch.cuda.Stream() for _ in ups]
def one():
    ups[0].tile_process()
def two():
    for up, s in zip(ups, streams):
        with torch.cuda.stream(s):
            up.tile_process()
for name, fn, frames in (("one frame at a time", one, 1), ("two frames in flight", two, 2)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / 3
    print(f"{name}: {d * 1e3:.1f} ms per step, {frames / d:.2f} frames/s")
