#!/usr/bin/env python3
"""Sweeps RealESRGANer.tile_streams / tile_batch on the 2160p workload (values are invariant).
usage: tools/sweep_c3.py [bf16|f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict

sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
frame = synthetic_frame(2160, 3840, seed=0)
up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, compute_dtype=(sys.argv[1] if len(sys.argv) > 1 else "bf16")), tile=512, tile_pad=10, pre_pad=0, device="cuda")
up.pre_process(np.ascontiguousarray(frame[:, :, ::-1].astype(np.float32) / 255.0))
for streams in (1, 2, 3):
    for batch in (1, 2, 3, 4, 8, 24):
        up.tile_streams, up.tile_batch = streams, batch
        for _ in range(2):
            up.tile_process()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            up.tile_process()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"streams {streams} batch<= {batch:2d}: {dt * 1e3:7.1f} ms/frame  {1 / dt:5.2f} fps", flush=True)
