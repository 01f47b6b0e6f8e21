"""Multi-GPU ``enhance``: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI),
the tiles of upstream's tile grid sharded across ranks.

Why tiles: the reference's result at frame sizes above ``tile`` is *defined* by
``RealESRGANer.tile_process`` (``tile=512, tile_pad=10`` at standalone/direct_esrgan.py:122-123):
every tile is an independent network evaluation whose un-padded centre is pasted.  Tiles are
therefore the independent units of this path (SURVEY.md section 8(e) mode 1); no activation ever
crosses a GPU.  The input frame is row-scattered (rank r owns rows [r*H/N, (r+1)*H/N)); a rank
fetches only the rows its tiles read beyond its own band -- the ``tile_pad`` overlap rows plus
whatever the balanced tile assignment shifts across the band boundary -- from the owning ranks with
point-to-point sends (``batch_isend_irecv``).  There is no all-reduce on this path.  Output tiles are
quantised where they were computed and sent to rank 0 (``gather=True``) or kept (``gather=False``).

The per-tile arithmetic does not depend on which rank runs it, so the N-rank result is bitwise the
1-rank result of the same wrapper.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from .realesrganer import normalize_u8_on_device


@dataclass(frozen=True)
class Tile:
    index: int
    inp: tuple    # padded input window  (y0, y1, x0, x1) in frame coordinates
    out: tuple    # output window        (y0, y1, x0, x1) in output coordinates
    crop: tuple   # centre crop inside the tile's own output

    @property
    def area(self):
        return (self.inp[1] - self.inp[0]) * (self.inp[3] - self.inp[2])


def row_band(rank, world, height):
    """Rows [lo, hi) of the frame that rank `rank` holds when the frame is row-scattered."""
    return rank * height // world, (rank + 1) * height // world


def plan_tiles(up, height, width, world):
    """tile list (upstream order) and the per-rank assignment: contiguous runs in row-major order,
    balanced by padded input area (edge tiles are smaller)."""
    if up.tile_size > 0:
        grid = up.tile_grid(height, width)
    else:
        s = up.scale
        grid = [((0, height, 0, width), (0, height * s, 0, width * s), (0, height * s, 0, width * s))]
    tiles = [Tile(i, g[0], g[1], g[2]) for i, g in enumerate(grid)]
    total = sum(t.area for t in tiles)
    owner, acc, r = [], 0, 0
    for t in tiles:
        # move to the next rank when this tile's midpoint passes the rank's share
        while r < world - 1 and acc + t.area / 2 > (r + 1) * total / world:
            r += 1
        owner.append(r)
        acc += t.area
    return tiles, owner


def rows_needed(tiles, owner, rank):
    mine = [t for t, o in zip(tiles, owner) if o == rank]
    if not mine:
        return 0, 0
    return min(t.inp[0] for t in mine), max(t.inp[1] for t in mine)


def exchange_plan(tiles, owner, world, height):
    """[(src, dst, row_lo, row_hi)]: rows `src` owns that `dst` needs and does not own."""
    plan = []
    for d in range(world):
        n0, n1 = rows_needed(tiles, owner, d)
        for s in range(world):
            if s == d:
                continue
            b0, b1 = row_band(s, world, height)
            lo, hi = max(n0, b0), min(n1, b1)
            if lo < hi:
                plan.append((s, d, lo, hi))
    return plan


def _p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


@torch.no_grad()
def enhance_sharded(up, band, frame_hw, group=None, gather=True):
    """Distributed equivalent of ``up.enhance(img)`` for 8-bit BGR frames.

    up        : RealESRGANer (pre_pad must be 0; frame sides multiples of mod_scale)
    band      : this rank's rows of the frame, uint8 [rows, W, 3] (numpy or tensor), BGR
    frame_hw  : (H, W) of the whole frame
    returns   : on rank 0 (gather=True) the uint8 [H*s, W*s, 3] BGR result, else None;
                with gather=False a list of ((y0, y1, x0, x1), uint8 tile) for this rank's tiles.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    H, W = frame_hw
    s = up.scale
    ms = {2: 2, 1: 4}.get(s, 1)
    if up.pre_pad != 0 or H % ms or W % ms:
        raise NotImplementedError("enhance_sharded: pre_pad must be 0 and the frame sides multiples of mod_scale "
                                  "(the reflect pad lives on one rank's rows; use enhance() for such frames)")
    dev = up.device
    band = torch.as_tensor(band).to(dev)
    b0, b1 = row_band(rank, world, H)
    if band.dtype != torch.uint8 or tuple(band.shape) != (b1 - b0, W, 3):
        raise ValueError(f"rank {rank}: band must be uint8 [{b1 - b0}, {W}, 3], got {band.dtype} {tuple(band.shape)}")

    tiles, owner = plan_tiles(up, H, W, world)
    n0, n1 = rows_needed(tiles, owner, rank)
    local = torch.empty((max(n1 - n0, 0), W, 3), dtype=torch.uint8, device=dev)
    lo, hi = max(n0, b0), min(n1, b1)
    if lo < hi:
        local[lo - n0:hi - n0] = band[lo - b0:hi - b0]

    # ---- overlap-row exchange (point to point, neighbours in practice)
    ops, keep = [], []
    for (src, dst, r0, r1) in exchange_plan(tiles, owner, world, H):
        if src == rank:
            t = band[r0 - b0:r1 - b0].contiguous()
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, dst, group))
        elif dst == rank:
            ops.append(dist.P2POp(dist.irecv, local[r0 - n0:r1 - n0], src, group))
    _p2p(ops)

    # ---- this rank's tiles: same arithmetic as RealESRGANer.enhance / tile_process
    mine = [t for t, o in zip(tiles, owner) if o == rank]
    fused = bool(mine) and hasattr(up, "_u8_tiles_fused_ok") and up._u8_tiles_fused_ok(H, W) and up.tile_size > 0
    results = []
    if mine and fused:
        # cut / ragged forward / paste, one launch each per batch (RealESRGANer.tiles_u8_on_device): on rank 0 straight into the
        # frame's canvas, elsewhere into one packed buffer whose slices are sent
        windows = [(t.inp[0] - n0, t.inp[2], t.inp[1] - t.inp[0], t.inp[3] - t.inp[2]) for t in mine]
        if rank == 0 and gather:
            canvas = torch.empty((H * s, W * s, 3), dtype=torch.uint8, device=dev)
            pastes = [(t.crop[0], t.crop[2], t.crop[1] - t.crop[0], t.crop[3] - t.crop[2], (t.out[0] * W * s + t.out[2]) * 3, W * s * 3) for t in mine]
            up.tiles_u8_on_device(local, windows, pastes, canvas)
        else:
            sizes = [(t.out[1] - t.out[0]) * (t.out[3] - t.out[2]) * 3 for t in mine]
            offs = [0]
            for v in sizes:
                offs.append(offs[-1] + v)
            packed = torch.empty((offs[-1],), dtype=torch.uint8, device=dev)
            pastes = [(t.crop[0], t.crop[2], t.crop[1] - t.crop[0], t.crop[3] - t.crop[2], offs[i], (t.out[3] - t.out[2]) * 3) for i, t in enumerate(mine)]
            up.tiles_u8_on_device(local, windows, pastes, packed)
            results = [(t, packed[offs[i]:offs[i + 1]].view(t.out[1] - t.out[0], t.out[3] - t.out[2], 3)) for i, t in enumerate(mine)]
    elif mine:
        x = normalize_u8_on_device(local.permute(2, 0, 1).flip(0)).unsqueeze(0)   # BGR->RGB, /255, HWC->NCHW
        if up.half:
            x = x.half()
        def quantise(t, out):
            o = out[0, :, t.crop[0]:t.crop[1], t.crop[2]:t.crop[3]].float().clamp_(0, 1)
            q = (o.flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8).contiguous()   # RGB->BGR, CHW->HWC
            results.append((t, q))

        up.run_tiles(x, [(t.inp[0] - n0, t.inp[1] - n0, t.inp[2], t.inp[3], t) for t in mine], quantise)
    if not gather:
        return [(t.out, q) for t, q in results]

    # ---- gather quantised tiles on rank 0
    if rank == 0:
        if not (mine and fused):
            canvas = torch.zeros((H * s, W * s, 3), dtype=torch.uint8, device=dev)
            for t, q in results:
                canvas[t.out[0]:t.out[1], t.out[2]:t.out[3]] = q
        bufs, ops = [], []
        for t, o in zip(tiles, owner):
            if o != 0:
                buf = torch.empty((t.out[1] - t.out[0], t.out[3] - t.out[2], 3), dtype=torch.uint8, device=dev)
                bufs.append((t, buf))
                ops.append(dist.P2POp(dist.irecv, buf, o, group))
        _p2p(ops)
        for t, buf in bufs:
            canvas[t.out[0]:t.out[1], t.out[2]:t.out[3]] = buf
        if dev.type == "cuda":
            host = torch.empty(canvas.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(canvas, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            if hasattr(up, "_check_range"):
                up._check_range()
            return host.numpy()
        return canvas.cpu().numpy()
    results.sort(key=lambda r: r[0].index)
    _p2p([dist.P2POp(dist.isend, q.contiguous(), 0, group) for _, q in results])
    return None


def scatter_rows(img, rank, world):
    """The row band of `img` (uint8 HWC) that rank `rank` owns."""
    b0, b1 = row_band(rank, world, img.shape[0])
    return np.ascontiguousarray(img[b0:b1])
