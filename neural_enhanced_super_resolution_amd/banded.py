"""Exact (seamless) multi-GPU evaluation of ONE whole-frame network call: row bands of the frame, one per rank
(SURVEY.md section 8(e) mode 2).  This is the distributed form of what the NESR pipeline does with ``tile=0``
(nesr/nesr.py:224: ``RealESRGANer(..., tile=0, ...)`` then ``model(img)``, nesr/nesr.py:887-891): the result is the
untiled network output, not upstream's tile grid (that one is sharded.py).

A 3x3 conv needs one row of its neighbour band, and RRDBNet has 351 of them, so the bands are not exchanged
per layer.  A rank evaluates its band plus an APRON of neighbour rows as an independent image: every conv spoils one
more row at an edge that is not a frame edge, and before the spoiled rows could reach the band the apron rows of the
feature map about to be read are overwritten with the neighbours' true rows:

    conv_first                         on input rows band + apron
    for each of the 69 RDBs:           refresh the apron of x0 (64 ch) from each neighbour, run the RDB's 5 convs
                                       (they spoil 5 rows)
    conv_body .. conv_last:            refresh the aprons of the trunk output and of conv_first's output, run the
                                       tail (conv_body 1 row, up1 1/2, up2 1/4, hr 1/4, last 1/4 = 2.25 rows)

The whole apron (6 rows) is refreshed every time, so nothing stale survives a stage.  70 exchange steps per frame,
point to point with the two neighbours only (a line, not a ring; no all-reduce); 1920 px x 6 rows x 64 ch x 4 B =
2.9 MB per neighbour and step for a 2160p x2plus frame.  The extra arithmetic is
2*APRON / band rows (8 ranks, 1080 internal rows: +9 %).  Inside the band every pixel sees exactly the operands of the
single-GPU evaluation and the f32 kernels' per-pixel arithmetic does not depend on where a tile lies, so the N-rank
result is bitwise the 1-rank result (APRON and the band boundaries are even: the Winograd form depends on the
position inside its 2x2 tile; the bf16 path picks its kernel by image size, so there the guarantee is "same
operands", not "same bits").

The engine interface (``band_begin / band_rdb / band_tail / band_rows / band_set_rows / num_rdb / unshuffle``) is
RRDBNet's; tests drive the same protocol with a CPU engine built from the oracle.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .realesrganer import normalize_u8_on_device

APRON = 6        # internal rows of each neighbour a rank carries: >= 5 (an RDB's reach), even (Winograd tile parity)


def band_split(internal_rows, world):
    """[(lo, hi)] internal rows of every rank: even boundaries, every band at least APRON rows."""
    cuts = [2 * ((r * internal_rows // world) // 2) for r in range(world)] + [internal_rows]
    bands = [(cuts[r], cuts[r + 1]) for r in range(world)]
    if any(hi - lo < APRON for lo, hi in bands):
        raise ValueError(f"{internal_rows} internal rows over {world} ranks: bands shorter than the apron ({APRON} rows)")
    return bands


def forward_banded(engine, x_ext, top, bottom, exchange):
    """One rank's part of the banded forward.

    x_ext    : [1, C, H_ext, W] the rank's input rows including `top` / `bottom` internal apron rows (x unshuffle)
    exchange : callable(buffer, k) that overwrites the k apron rows next to the band, on both sides, of feature
               map `buffer` with the neighbours' band rows (no-op at frame edges)
    returns  : [1, C_out, 4 * band_internal_rows, 4 * w] the rank's rows of the network output
    """
    engine.band_begin(x_ext)
    for i in range(engine.num_rdb):
        exchange(i % 3, APRON)
        engine.band_rdb(i)
    exchange(0, APRON)
    exchange(3, APRON)
    y = engine.band_tail()
    return y[:, :, 4 * top: y.shape[2] - 4 * bottom]


def _p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def make_exchange(engine, rank, world, top, band_rows, bottom, group=None, via_cpu=False):
    """The distributed `exchange` of forward_banded: point to point with rank-1 and rank+1."""
    def exchange(buffer, k):
        sends, ops, recvs = [], [], []
        if rank > 0:          # upper neighbour: my first k band rows go up, its last k band rows land above my band
            sends.append((rank - 1, engine.band_rows(buffer, top, k)))
            recvs.append((rank - 1, top - k))
        if rank < world - 1:
            sends.append((rank + 1, engine.band_rows(buffer, top + band_rows - k, k)))
            recvs.append((rank + 1, top + band_rows))
        bufs = []
        for peer, t in sends:
            t = t.cpu() if via_cpu else t
            bufs.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer, group))
        landing = []
        for (peer, row0), (_, like) in zip(recvs, sends):
            t = torch.empty_like(like.cpu() if via_cpu else like)
            landing.append((row0, t))
            ops.append(dist.P2POp(dist.irecv, t, peer, group))
        _p2p(ops)
        for row0, t in landing:
            engine.band_set_rows(buffer, row0, t)
    return exchange


@torch.no_grad()
def enhance_banded(up, band, frame_hw, group=None, gather=True):
    """Distributed equivalent of ``up.enhance(img)`` for an UNTILED wrapper (``tile=0``) and 8-bit BGR frames.

    up        : RealESRGANer with tile=0, pre_pad=0; frame sides multiples of mod_scale
    band      : this rank's input rows, uint8 [rows, W, 3] BGR, rows = the band of band_split() x unshuffle
    frame_hw  : (H, W) of the whole frame
    returns   : on rank 0 (gather=True) the uint8 [H*s, W*s, 3] BGR frame, else this rank's output rows
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    net = up.model
    H, W = frame_hw
    u = net.unshuffle
    if up.tile_size != 0 or up.pre_pad != 0 or H % u or W % u:
        raise NotImplementedError("enhance_banded: tile=0, pre_pad=0 and frame sides multiples of the unshuffle factor")
    dev = up.device
    via_cpu = dist.is_initialized() and dist.get_backend(group) == "gloo"
    bands = band_split(H // u, world)
    lo, hi = bands[rank]
    band = torch.as_tensor(band).to(dev)
    if band.dtype != torch.uint8 or tuple(band.shape) != ((hi - lo) * u, W, 3):
        raise ValueError(f"rank {rank}: band must be uint8 [{(hi - lo) * u}, {W}, 3], got {band.dtype} {tuple(band.shape)}")
    top = APRON if rank > 0 else 0
    bottom = APRON if rank < world - 1 else 0

    # ---- input apron rows from the neighbours (APRON internal rows = APRON * unshuffle input rows)
    ext = torch.empty(((hi - lo + top + bottom) * u, W, 3), dtype=torch.uint8, device=dev)
    ext[top * u: top * u + band.shape[0]] = band
    ops, keep, landing = [], [], []
    def stage(t):
        t = t.contiguous()
        return t.cpu() if via_cpu else t
    if rank > 0:
        keep.append(stage(band[: APRON * u]))
        ops.append(dist.P2POp(dist.isend, keep[-1], rank - 1, group))
        landing.append((0, torch.empty((APRON * u, W, 3), dtype=torch.uint8, device="cpu" if via_cpu else dev)))
        ops.append(dist.P2POp(dist.irecv, landing[-1][1], rank - 1, group))
    if rank < world - 1:
        keep.append(stage(band[band.shape[0] - APRON * u:]))
        ops.append(dist.P2POp(dist.isend, keep[-1], rank + 1, group))
        landing.append(((top + hi - lo) * u, torch.empty((APRON * u, W, 3), dtype=torch.uint8, device="cpu" if via_cpu else dev)))
        ops.append(dist.P2POp(dist.irecv, landing[-1][1], rank + 1, group))
    _p2p(ops)
    for row0, t in landing:
        ext[row0: row0 + t.shape[0]] = t.to(dev)

    x = normalize_u8_on_device(ext.permute(2, 0, 1).flip(0)).unsqueeze(0)     # BGR->RGB, /255, HWC->NCHW
    y = forward_banded(net, x, top, bottom, make_exchange(net, rank, world, top, hi - lo, bottom, group, via_cpu))
    q = (y[0].float().clamp_(0, 1).flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8).contiguous()   # RGB->BGR, CHW->HWC
    if not gather or world == 1:
        return q.cpu().numpy() if world == 1 else q
    s = up.scale
    if rank == 0:
        canvas = torch.empty((H * s, W * s, 3), dtype=torch.uint8, device=dev)
        canvas[: q.shape[0]] = q
        ops, parts = [], []
        for r in range(1, world):
            blo, bhi = bands[r]
            t = torch.empty(((bhi - blo) * u * s, W * s, 3), dtype=torch.uint8, device="cpu" if via_cpu else dev)
            parts.append((blo * u * s, t))
            ops.append(dist.P2POp(dist.irecv, t, r, group))
        _p2p(ops)
        for row0, t in parts:
            canvas[row0: row0 + t.shape[0]] = t.to(dev)
        return canvas.cpu().numpy()
    _p2p([dist.P2POp(dist.isend, q.cpu() if via_cpu else q, 0, group)])
    return None


def scatter_band(img, rank, world, unshuffle):
    """The input rows of `img` (uint8 HWC) that rank `rank` owns under band_split()."""
    lo, hi = band_split(img.shape[0] // unshuffle, world)[rank]
    return np.ascontiguousarray(img[lo * unshuffle: hi * unshuffle])
