#!/usr/bin/env python3
"""2160p bf16 frame through RealESRGANer(tile=512, tile_pad=10): all tiles in ragged batches on 1..4 streams against
one batch per tile shape on 3 streams; interleaved rounds in one process (the chip's clock drifts under sustained load:
back-to-back timings of different settings are not comparable)."""
import os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
dev = torch.device("cuda:0")
sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
H, W = 2160, 3840
frame = synthetic_frame(H, W, seed=0)
cfgs = {"groups": (0, 3), "ragged_1stream": (64, 1), "ragged_2streams": (64, 2), "ragged_3streams": (64, 3), "ragged_4streams": (64, 4)}
ups = {}
for name, (rb, ns) in cfgs.items():
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, compute_dtype="bf16"), tile=512, tile_pad=10, pre_pad=0, half=False, device=dev)
    up.ragged_tiles = rb > 0
    up.ragged_batch = max(rb, 1)
    up.tile_streams = ns
    up.pre_process(np.ascontiguousarray(frame[:, :, ::-1].astype(np.float32) / 255.0))
    up.tile_process()
    ups[name] = up
torch.cuda.synchronize()
times = {k: [] for k in cfgs}
for rnd in range(8):
    for name, up in ups.items():
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2): up.tile_process()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t) / 2)
for name in cfgs:
    m = statistics.median(times[name])
    print(f"{name:9s} median {1e3*m:.1f} ms per frame = {1/m:.2f} frames/s   min {1e3*min(times[name]):.1f} ms", flush=True)
ref = ups["groups"].output
for name in list(cfgs)[1:]:
    print(name, "== groups bitwise:", torch.equal(ups[name].output, ref))
