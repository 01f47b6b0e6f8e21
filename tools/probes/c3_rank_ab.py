#!/usr/bin/env python3
"""What one rank of an N-rank sharded 2160p bf16 frame computes (sharded.plan_tiles: its share of the 40 tiles), timed
on this GPU as ragged batches against one batch per tile shape; interleaved rounds."""
import os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer, sharded
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
dev = torch.device("cuda:0")
sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
H, W = 2160, 3840
frame = synthetic_frame(H, W, seed=0)
x = torch.from_numpy(np.ascontiguousarray(frame[:, :, ::-1].astype(np.float32) / 255.0)).permute(2, 0, 1).unsqueeze(0).to(dev)
ups = {}
for name, cfg in {"old(3 streams, whole groups)": (False, 3, 0), "new(split, 5 streams)": (False, 3, 12), "new(split, 6 streams)": (False, 3, 12, 6), "ragged5": (True, 5, 0)}.items():
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, compute_dtype="bf16"), tile=512, tile_pad=10, pre_pad=0, half=False, device=dev)
    up.ragged_tiles = cfg[0]
    up.tile_streams = cfg[1]
    up.small_job_tiles = cfg[2]
    if len(cfg) > 3: up.small_job_streams = cfg[3]
    ups[name] = up
for world in (2, 4, 8):
    tiles, owner = sharded.plan_tiles(list(ups.values())[0], H, W, world)
    worst = {}
    for name, up in ups.items():
        per_rank = []
        for rank in range(world):
            mine = [(t.inp[0], t.inp[1], t.inp[2], t.inp[3], t) for t, o in zip(tiles, owner) if o == rank]
            sink = lambda payload, out: None
            up.run_tiles(x, mine, sink)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                up.run_tiles(x, mine, sink)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            per_rank.append(statistics.median(ts))
        worst[name] = max(per_rank)
        print(f"world {world} {name:30s}: slowest rank {1e3*max(per_rank):6.1f} ms  fastest {1e3*min(per_rank):6.1f} ms  tiles/rank {len(tiles)/world:.1f}", flush=True)
