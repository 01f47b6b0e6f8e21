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
    (4, "nearest", (230, 150), 32, 10, 2),       # 4 ranks: 8 x 5 tiles, ragged last row and column
    (8, "nearest", (260, 96), 32, 8, 4),         # 8 ranks (the node size of BASELINE.json configs[3]): 9 x 3 tiles, some ranks get three, some four
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


@pytest.mark.parametrize("H,W,scale,tile,pad", [(2160, 3840, 2, 512, 10), (1080, 1920, 4, 512, 10), (300, 500, 2, 128, 10), (97, 64, 4, 32, 6),
                                                 (64, 64, 2, 0, 10)])
def test_c_abi_plan_is_the_python_plan(H, W, scale, tile, pad):
    """nesr_shard_plan (what nesr_forward_sharded_u8 runs on, host only) against plan_tiles / exchange_plan for 1 .. 8 ranks."""
    import ctypes
    import __graft_entry__ as g
    g.build()
    from neural_enhanced_super_resolution_amd import _lib
    lib = _lib.load()
    up = RealESRGANer.__new__(RealESRGANer)
    up.scale, up.tile_size, up.tile_pad, up.pre_pad = scale, tile, pad, 0
    for world in (1, 2, 3, 4, 8):
        tiles, owner = sharded.plan_tiles(up, H, W, world)
        moves = sharded.exchange_plan(tiles, owner, world, H)
        nt, nm = ctypes.c_int(), ctypes.c_int()
        t13 = (ctypes.c_int * (13 * 256))()
        m4 = (ctypes.c_int * (4 * 256))()
        assert lib.nesr_shard_plan(H, W, scale, tile, pad, world, t13, 256, ctypes.byref(nt), m4, 256, ctypes.byref(nm)) == 0
        assert nt.value == len(tiles) and nm.value == len(moves)
        for i, (t, o) in enumerate(zip(tiles, owner)):
            assert tuple(t13[13 * i:13 * i + 13]) == tuple(t.inp) + tuple(t.out) + tuple(t.crop) + (o,)
        assert [tuple(m4[4 * i:4 * i + 4]) for i in range(nm.value)] == [tuple(m) for m in moves]
    assert lib.nesr_shard_plan(0, 8, 2, 4, 1, 1, None, 0, ctypes.byref(nt), None, 0, ctypes.byref(nm)) < 0


def test_sharded_rejects_unsupported_padding():
    up = RealESRGANer.__new__(RealESRGANer)
    up.scale, up.tile_size, up.tile_pad, up.pre_pad, up.device = 2, 32, 10, 10, torch.device("cpu")
    with pytest.raises(NotImplementedError):
        sharded.enhance_sharded(up, np.zeros((8, 8, 3), np.uint8), (8, 8))
