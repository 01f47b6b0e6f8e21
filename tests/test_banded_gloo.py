"""CPU, world_size 2 and 3 over gloo: the exact (row-band) multi-GPU mode of banded.py -- band split, input apron
exchange, the per-RDB apron refresh and the gather -- driven through the same protocol the HIP engine serves, with
a CPU engine made of the oracle's functions.  The N-rank result must equal the single-process untiled evaluation
(CPU conv2d may pick another algorithm for another image size, hence a float tolerance, not bit equality; the GPU
test in test_gpu_banded.py checks bit equality of the HIP engine)."""
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

from neural_enhanced_super_resolution_amd import banded  # noqa: E402
from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict  # noqa: E402


class OracleBandEngine:
    """banded.py's engine interface on the CPU: the oracle's RDB / tail functions on the band image."""

    def __init__(self, sd, scale, num_block):
        self.sd, self.scale, self.num_block = sd, scale, num_block
        self.num_rdb = 3 * num_block
        self.unshuffle = {2: 2, 1: 4}.get(scale, 1)

    def band_begin(self, x):
        from oracle import rrdbnet_ref as R
        feat = R._conv(R.pixel_unshuffle(x.float(), self.scale) if self.unshuffle > 1 else x.float(), self.sd, "conv_first")
        self.bufs = [feat.clone(), None, None, feat.clone()]

    def band_rdb(self, i):
        from oracle import rrdbnet_ref as R
        b, r = divmod(i, 3)
        out = R.rdb_forward(self.bufs[r], self.sd, f"body.{b}.rdb{r + 1}")
        if r < 2:
            self.bufs[r + 1] = out
        else:
            self.bufs[0] = out * 0.2 + self.bufs[0]

    def band_tail(self):
        from oracle import rrdbnet_ref as R
        import torch.nn.functional as F
        sd = self.sd
        feat = self.bufs[3] + R._conv(self.bufs[0], sd, "conv_body")
        feat = R._lrelu(R._conv(F.interpolate(feat, scale_factor=2, mode="nearest"), sd, "conv_up1"))
        feat = R._lrelu(R._conv(F.interpolate(feat, scale_factor=2, mode="nearest"), sd, "conv_up2"))
        return R._conv(R._lrelu(R._conv(feat, sd, "conv_hr")), sd, "conv_last")

    def band_rows(self, buffer, row0, nrows):
        return self.bufs[buffer][:, :, row0:row0 + nrows].contiguous().view(-1).view(torch.uint8).clone()

    def band_set_rows(self, buffer, row0, rows):
        t = self.bufs[buffer]
        n = rows.numel() // (4 * t.shape[1] * t.shape[3])
        t[:, :, row0:row0 + n] = rows.view(torch.float32).view(1, t.shape[1], n, t.shape[3])


class _Up:   # the attributes enhance_banded reads from a RealESRGANer
    def __init__(self, engine, scale):
        self.model, self.scale, self.tile_size, self.pre_pad, self.device = engine, scale, 0, 0, torch.device("cpu")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _reference(sd, img, scale, num_block):
    from oracle.rrdbnet_ref import rrdbnet_forward
    x = torch.from_numpy(img[:, :, ::-1].copy()).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
    y = rrdbnet_forward(x, sd, scale=scale, num_block=num_block)
    return (y[0].clamp(0, 1).flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8).numpy(), y


def _worker(rank, world, port, hw, scale, num_block, out_path, overlap=True):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        sd = synthetic_state_dict(seed=5, num_in_ch=3, scale=scale, num_block=num_block)
        img = synthetic_frame(hw[0], hw[1], seed=7)
        eng = OracleBandEngine(sd, scale, num_block)
        band = banded.scatter_band(img, rank, world, eng.unshuffle)
        stats = {}
        got = banded.enhance_banded(_Up(eng, 2 if scale == 2 else 4), band, hw, overlap=overlap, stats=stats)
        if overlap:
            # the traffic model of DESIGN.md section 6: one step after conv_first + one per RDB (70 for 23 blocks), each
            # APRON rows x w x 64 channels x 4 bytes per neighbour -- nothing else crosses a band boundary
            w_int = hw[1] // eng.unshuffle
            assert stats["steps"] == 1 + eng.num_rdb <= 70
            assert set(stats["bytes_per_neighbour"]) == {banded.APRON * w_int * 64 * 4}, stats["bytes_per_neighbour"][:3]
        if rank == 0:
            np.save(out_path, got)
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,hw,scale,overlap", [(2, (64, 48), 2, True), (3, (96, 40), 2, True), (2, (32, 24), 4, True), (2, (64, 48), 2, False),
                                                    (4, (160, 40), 2, True), (8, (256, 24), 2, True)])
def test_banded_equals_untiled_single_process(tmp_path, world, hw, scale, overlap):
    out = str(tmp_path / "banded.npy")
    num_block = 2
    mp.spawn(_worker, args=(world, _free_port(), hw, scale, num_block, out, overlap), nprocs=world, join=True)
    got = np.load(out)
    sd = synthetic_state_dict(seed=5, num_in_ch=3, scale=scale, num_block=num_block)
    want, _ = _reference(sd, synthetic_frame(hw[0], hw[1], seed=7), scale, num_block)
    assert got.shape == want.shape
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3      # float-level differences may flip a rounding


def test_exchange_step_count_for_the_real_network():
    """23 blocks: 1 + 69 = 70 exchange steps per frame (SURVEY.md section 8(e) asked for ~75, round 1 had 71)."""
    assert 1 + 3 * 23 == 70 and [banded.out_buffer(i) for i in range(6)] == [1, 2, 0, 1, 2, 0]


def test_band_split_properties():
    for rows, world in [(1080, 8), (1080, 4), (540, 8), (128, 3), (14, 2)]:
        bands = banded.band_split(rows, world)
        assert bands[0][0] == 0 and bands[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        assert all(lo % 2 == 0 for lo, _ in bands) and all(hi - lo >= banded.APRON for lo, hi in bands)
    with pytest.raises(ValueError):
        banded.band_split(20, 4)


def test_single_rank_is_plain_forward():
    sd = synthetic_state_dict(seed=5, num_in_ch=3, scale=2, num_block=1)
    img = synthetic_frame(32, 40, seed=1)
    eng = OracleBandEngine(sd, 2, 1)
    got = banded.enhance_banded(_Up(eng, 2), img, (32, 40))
    want, _ = _reference(sd, img, 2, 1)
    assert np.array_equal(got, want)
