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

The whole apron (6 rows) is refreshed every time, so nothing stale survives a stage.  70 exchange steps per frame
(one after conv_first, whose rows also refresh the copy kept for the trunk skip, and one after each of the 69 RDBs),
point to point with the two neighbours only (a line, not a ring; no all-reduce); 1920 px x 6 rows x 64 ch x 4 B =
2.9 MB per neighbour and step for a 2160p x2plus frame.  The default protocol (forward_banded_overlapped) takes the
exchange off the critical path: an RDB first computes the rows its neighbours wait for, posts them (RCCL point-to-point
on a side stream, packed by one C-ABI call) and computes conv5 on the rest of the band -- 42 % of the block's
arithmetic -- while they travel.  The extra arithmetic is
2*APRON / band rows (8 ranks, 1080 internal rows: +9 %).  Inside the band every pixel sees exactly the operands of the
single-GPU evaluation and the f32 kernels' per-pixel arithmetic does not depend on where a tile lies, so the N-rank
result is bitwise the 1-rank result (APRON and the band boundaries are even: the Winograd form depends on the
position inside its 2x2 tile; the bf16 path picks its kernel by image size, so there the guarantee is "same
operands", not "same bits").

The engine interface (``band_begin / band_rdb / band_tail / band_rows / band_set_rows / num_rdb / unshuffle``) is
RRDBNet's; tests drive the same protocol with a CPU engine built from the oracle.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from .realesrganer import normalize_u8_on_device

_OVERLAP_CHECKED = set()      # (world size, transport) pairs whose overlapped exchange has been checked against the simple order in this process

APRON = 6        # internal rows of each neighbour a rank carries: >= 5 (an RDB's reach), even (Winograd tile parity)


def band_split(internal_rows, world):
    """[(lo, hi)] internal rows of every rank: even boundaries, every band at least APRON rows."""
    cuts = [2 * ((r * internal_rows // world) // 2) for r in range(world)] + [internal_rows]
    bands = [(cuts[r], cuts[r + 1]) for r in range(world)]
    if any(hi - lo < APRON for lo, hi in bands):
        raise ValueError(f"{internal_rows} internal rows over {world} ranks: bands shorter than the apron ({APRON} rows)")
    return bands


def out_buffer(i):
    """The dense-block buffer RDB i writes its result into (= the one RDB i + 1 reads): P, Q, R rotate."""
    return (i % 3 + 1) % 3


def forward_banded(engine, x_ext, top, bottom, exchange):
    """One rank's part of the banded forward, exchange AFTER every stage (the simple order; kept for engines and
    transports that cannot overlap -- tests drive it in lockstep for emulated ranks).

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


def forward_banded_overlapped(engine, x_ext, top, bottom, ex):
    """The same evaluation with the exchange off the critical path.  Per RDB i:

        phase 0   conv1..conv4, and conv5 on the APRON band rows next to each apron: the rows the neighbours wait for
        post      pack those rows of the buffer RDB i wrote and send / receive them (`ex`: asynchronously -- RCCL
                  point-to-point on a side stream)
        phase 1   conv5 on the band rows in between (42 % of the block's arithmetic) while the rows travel
        complete  the neighbours' rows -> the apron rows, before RDB i + 1 reads them

    conv_first's output needs one exchange before RDB 0; the received rows also refresh the copy kept for the trunk
    skip (buffer 3), so a frame takes 1 + num_rdb exchange steps (70 for 23 blocks), each 2 x APRON rows x w x
    num_feat values per neighbour.  Values are those of forward_banded: the phases are row ranges of the same kernel."""
    engine.band_begin(x_ext)
    ex.complete(ex.post(0), also=(3,))
    phased = hasattr(engine, "band_rdb_phase")
    for i in range(engine.num_rdb):
        if phased:
            engine.band_rdb_phase(i, 0, top, bottom, APRON)
        else:
            engine.band_rdb(i)
        handle = ex.post(out_buffer(i))
        if phased:
            engine.band_rdb_phase(i, 1, top, bottom, APRON)
        ex.complete(handle)
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


class EdgeExchange:
    """The asynchronous transport of forward_banded_overlapped: rank r <-> r - 1 and r + 1 (a line: no ring, no
    all-reduce).  `post` packs the APRON edge rows into persistent staging tensors (one C-ABI call on the HIP engine) and
    starts isend / irecv -- on a side stream for RCCL, so the caller's stream goes on with phase 1; `complete` waits for
    them and unpacks into the apron rows.  `stats` counts steps and bytes (tests assert the traffic model)."""

    def __init__(self, engine, rank, world, top, band_rows, bottom, group=None, via_cpu=False, device=None, stats=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.top, self.band_rows, self.bottom = top, band_rows, bottom
        self.group, self.via_cpu = group, via_cpu
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.stats = stats if stats is not None else {}
        self.stats.setdefault("steps", 0)
        self.stats.setdefault("bytes_per_neighbour", [])
        self.up, self.down = rank > 0, rank < world - 1
        self.fast = hasattr(engine, "band_pack_edges")
        self.side = torch.cuda.Stream(self.device) if (self.device.type == "cuda" and not via_cpu) else None
        self._bufs = None

    def _staging(self):
        if self._bufs is None:
            nbytes = APRON * int(self.engine.band_row_bytes()) if self.fast else None
            mk = (lambda: torch.empty(nbytes, dtype=torch.uint8, device=self.device)) if self.fast else (lambda: None)
            self._bufs = {k: (mk() if on else None) for k, on in (("send_up", self.up), ("send_down", self.down),
                                                                    ("recv_up", self.up), ("recv_down", self.down))}
        return self._bufs

    def post(self, buffer):
        b = self._staging()
        eng = self.engine
        main = torch.cuda.current_stream(self.device) if self.side is not None else None
        if self.side is not None:
            self.side.wait_stream(main)            # phase 0 wrote the rows that are packed now
        ctx = torch.cuda.stream(self.side) if self.side is not None else _Null()
        with ctx:
            if self.fast:
                eng.band_pack_edges(buffer, self.top, self.bottom, APRON, b["send_up"], b["send_down"])
                su, sd = b["send_up"], b["send_down"]
            else:                                   # engines with the row accessors only (the CPU oracle engine of the tests)
                su = eng.band_rows(buffer, self.top, APRON) if self.up else None
                sd = eng.band_rows(buffer, self.top + self.band_rows - APRON, APRON) if self.down else None
                b["recv_up"] = torch.empty_like(su) if self.up else None
                b["recv_down"] = torch.empty_like(sd) if self.down else None
            ru, rd = b["recv_up"], b["recv_down"]
            if self.via_cpu:
                su, sd = (t.cpu() if t is not None else None for t in (su, sd))
                ru, rd = (torch.empty_like(t) if t is not None else None for t in (su, sd))
            ops = []
            if self.up:
                ops += [dist.P2POp(dist.isend, su, self.rank - 1, self.group), dist.P2POp(dist.irecv, ru, self.rank - 1, self.group)]
            if self.down:
                ops += [dist.P2POp(dist.isend, sd, self.rank + 1, self.group), dist.P2POp(dist.irecv, rd, self.rank + 1, self.group)]
            works = dist.batch_isend_irecv(ops) if ops else []
        self.stats["steps"] += 1
        self.stats["bytes_per_neighbour"].append(int(su.numel()) if su is not None else (int(sd.numel()) if sd is not None else 0))
        return (buffer, works, ru, rd, (su, sd))

    def complete(self, handle, also=()):
        buffer, works, ru, rd, _keep = handle
        eng = self.engine
        ctx = torch.cuda.stream(self.side) if self.side is not None else _Null()
        with ctx:
            for w in works:
                w.wait()
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        if self.via_cpu:
            ru, rd = (t.to(self.device) if t is not None else None for t in (ru, rd))
        for buf in (buffer,) + tuple(also):
            if self.fast:
                eng.band_unpack_aprons(buf, self.top, self.bottom, APRON, ru, rd)
            else:
                if ru is not None:
                    eng.band_set_rows(buf, self.top - APRON, ru)
                if rd is not None:
                    eng.band_set_rows(buf, self.top + self.band_rows, rd)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@torch.no_grad()
def enhance_banded(up, band, frame_hw, group=None, gather=True, overlap=True, stats=None):
    """Distributed equivalent of ``up.enhance(img)`` for an UNTILED wrapper (``tile=0``) and 8-bit BGR frames.

    up        : RealESRGANer with tile=0, pre_pad=0; frame sides multiples of mod_scale
    band      : this rank's input rows, uint8 [rows, W, 3] BGR, rows = the band of band_split() x unshuffle
    frame_hw  : (H, W) of the whole frame
    overlap   : True = forward_banded_overlapped (edge rows first, exchange beside conv5's interior rows); False = the
                simple order (exchange, then the whole block).  Same values.
    stats     : optional dict that receives {"steps", "bytes_per_neighbour"} of the apron exchange
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
    if overlap:
        own = {} if stats is None else stats
        steps0 = own.get("steps", 0)
        ex = EdgeExchange(net, rank, world, top, hi - lo, bottom, group, via_cpu, dev, own)
        y = forward_banded_overlapped(net, x, top, bottom, ex)
        key = (world, "gloo" if via_cpu else "device")
        if world > 1 and key not in _OVERLAP_CHECKED and os.environ.get("NESR_BANDED_SELFCHECK", "1") != "0":
            # The overlapped protocol (side-stream exchange, persistent staging buffers, row-range conv5 launches) has only run
            # over gloo and in single-GPU lockstep emulation so far: the first multi-rank frame of a process also goes through the
            # simple order, and every rank must get bit for bit the same band and the step count of the traffic model.
            y_ref = forward_banded(net, x, top, bottom, make_exchange(net, rank, world, top, hi - lo, bottom, group, via_cpu))
            bad = torch.tensor([0 if (torch.equal(y, y_ref) and own.get("steps", 0) - steps0 == 3 * net.num_block + 1) else 1],
                               dtype=torch.int32, device="cpu" if via_cpu else dev)
            dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
            if int(bad.item()):
                raise RuntimeError(f"enhance_banded: the overlapped exchange disagrees with the simple order on {int(bad.item())} rank(s) "
                                   f"(this rank: equal={torch.equal(y, y_ref)}, steps={own.get('steps', 0) - steps0}); run with overlap=False")
            _OVERLAP_CHECKED.add(key)
    else:
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
