"""CPU, world_size 2 and 3 over gloo: the multi-GPU tile-sharding path (plan, overlap-row exchange,
gather) gives bitwise the single-process enhance() result.  The network is injected (oracle RRDBNet /
a cheap stand-in), as in tests/test_host_logic.py; on the GPU box the same code runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neural_enhanced_super_resolution_amd import RealESRGANer  # noqa: E402
from neural_enhanced_super_resolution_amd import sharded  # noqa: E402
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_up(kind, tile, tile_pad, scale):
    from oracle.rrdbnet_ref import RRDBNetRef
    if kind == "rrdb":
        sd = synthetic_state_dict(seed=3, num_in_ch=3, scale=2 if scale == 2 else 4, num_block=1)
        return RealESRGANer(scale=scale, model_path={"params_ema": sd}, model=RRDBNetRef(3, 3, scale=2 if scale == 2 else 4, num_block=1),
                            tile=tile, tile_pad=tile_pad, pre_pad=0, device="cpu")
    from tests.test_host_logic import Nearest
    return RealESRGANer(scale=scale, model_path={"params": {"p": torch.zeros(1)}}, model=Nearest(scale), tile=tile,
                        tile_pad=tile_pad, pre_pad=0, device="cpu")


def _worker(rank, world, port, kind, hw, tile, tile_pad, scale, out_path):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        img = synthetic_frame(hw[0], hw[1], seed=42)
        up = _make_up(kind, tile, tile_pad, scale)
        band = sharded.scatter_rows(img, rank, world)
        got = sharded.enhance_sharded(up, band, hw)
        if rank == 0:
            want, _ = up.enhance(img)
            np.save(out_path, np.stack([got, want]))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,hw,tile,tile_pad,scale", [
    (2, "rrdb", (96, 140), 32, 10, 2),
    (2, "nearest", (130, 70), 32, 10, 4),
    (3, "nearest", (200, 64), 48, 6, 2),
    (2, "nearest", (64, 64), 0, 10, 2),          # tile=0: one unit, rank 1 only ships rows
])
def test_sharded_equals_single_process(tmp_path, world, kind, hw, tile, tile_pad, scale):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), kind, hw, tile, tile_pad, scale, out), nprocs=world, join=True)
    got, want = np.load(out)
    assert got.shape == (hw[0] * scale, hw[1] * scale, 3)
    assert np.array_equal(got, want)


def test_plan_covers_every_tile_once_and_balances_c3():
    up = RealESRGANer.__new__(RealESRGANer)
    up.scale, up.tile_size, up.tile_pad, up.pre_pad = 2, 512, 10, 0
    for world in (1, 2, 4, 8):
        tiles, owner = sharded.plan_tiles(up, 2160, 3840, world)
        assert len(tiles) == 40 and sorted(set(owner)) == list(range(world))
        assert owner == sorted(owner)                       # contiguous runs in upstream's order
        load = [sum(t.area for t, o in zip(tiles, owner) if o == r) for r in range(world)]
        assert max(load) <= 1.35 * (sum(load) / world)      # 40 ragged tiles over 8 ranks: within a tile


def test_exchange_moves_only_rows_outside_the_band():
    """4-way split of the 2160p frame: what crosses GPUs is the tile_pad overlap plus the rows the
    balanced assignment shifts across band boundaries -- never the whole frame."""
    up = RealESRGANer.__new__(RealESRGANer)
    up.scale, up.tile_size, up.tile_pad, up.pre_pad = 2, 512, 10, 0
    H, W, world = 2160, 3840, 4
    tiles, owner = sharded.plan_tiles(up, H, W, world)
    plan = sharded.exchange_plan(tiles, owner, world, H)
    for (s, d, lo, hi) in plan:
        b0, b1 = sharded.row_band(s, world, H)
        assert s != d and b0 <= lo < hi <= b1
        n0, n1 = sharded.rows_needed(tiles, owner, d)
        assert n0 <= lo and hi <= n1
    moved = sum(hi - lo for (_, _, lo, hi) in plan)
    assert 0 < moved < H                                   # less than one frame's worth of rows in total
    for d in range(world):                                  # every needed row is either owned or received
        n0, n1 = sharded.rows_needed(tiles, owner, d)
        b0, b1 = sharded.row_band(d, world, H)
        have = set(range(max(n0, b0), min(n1, b1)))
        for (s, dd, lo, hi) in plan:
            if dd == d:
                have |= set(range(lo, hi))
        assert have == set(range(n0, n1))


def test_sharded_rejects_unsupported_padding():
    up = RealESRGANer.__new__(RealESRGANer)
    up.scale, up.tile_size, up.tile_pad, up.pre_pad, up.device = 2, 32, 10, 10, torch.device("cpu")
    with pytest.raises(NotImplementedError):
        sharded.enhance_sharded(up, np.zeros((8, 8, 3), np.uint8), (8, 8))
