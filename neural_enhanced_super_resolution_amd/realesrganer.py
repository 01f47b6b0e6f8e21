"""``RealESRGANer`` drop-in: same constructor keywords, attributes and methods as
``realesrgan.RealESRGANer`` (realesrgan>=0.3.0, requirements.txt:9), as the reference uses it:

  nesr/nesr.py:220-229                   RealESRGANer(scale=int(upscale_factor), model_path=..., model=model,
                                         tile=0, tile_pad=0, pre_pad=0, half=False, device=self.device);
                                         afterwards only ``.model`` is used (nesr/nesr.py:887-891,930-935)
  standalone/direct_esrgan.py:118-127    RealESRGANer(scale, model_path, model, tile=512, tile_pad=10,
                                         pre_pad=0, half=False, device) ; ``.enhance(img)`` at :148
  standalone/superres_project.py:70-75   ctor defaults ; ``.enhance(bgr)`` at :282

Host logic only (padding, tile grid, cropping, colour order, quantisation); every network
evaluation goes through ``self.model``, which for this package is the HIP-backed
:class:`RRDBNet`.  cv2 is not needed: the colour conversions are plain numpy, and ``outscale != scale`` /
``alpha_upsampler != 'realesrgan'`` use imgproc.py's device-side restatement of cv2.resize (parity unpinned).
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import torch
from torch.nn import functional as F

from .rrdbnet import RRDBNet


# Integrity pins of the published checkpoints: the only ones the reference holds (nesr/utils/downloader.py:25-26, 33-34).
KNOWN_CHECKPOINTS = {
    "5db904e3e9f0dbf5c64b7ae665527e62": "RealESRGAN_x2plus.pth (v0.2.5.0 release, 67,010,191 bytes)",
    "94df4e7c584b55e2e9a5d2b8f161860e": "RealESRGAN_x4plus.pth (v0.1.0 release)",
}


def checkpoint_provenance(path):
    """md5 of a checkpoint file against the reference's table -> ("verified"|"unverified", description).
    A file NAMED like a published checkpoint whose digest differs is reported with a warning (it may be a fine-tune;
    it is not the file the reference downloads)."""
    import hashlib
    import os
    h = hashlib.md5()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    digest, size = h.hexdigest(), os.path.getsize(path)
    if digest in KNOWN_CHECKPOINTS:
        return "verified", f"{KNOWN_CHECKPOINTS[digest]}: md5 {digest} matches nesr/utils/downloader.py"
    base = os.path.basename(str(path))
    if base in ("RealESRGAN_x2plus.pth", "RealESRGAN_x4plus.pth"):
        warnings.warn(f"{path}: named like a published Real-ESRGAN checkpoint but md5 {digest} ({size} bytes) is not the "
                      "digest recorded in the reference (nesr/utils/downloader.py:25-26,33-34)")
    return "unverified", f"{base}: md5 {digest}, {size} bytes (not a digest the reference records)"


def _bgr2gray(img):
    # cv2.COLOR_BGR2GRAY on float32
    return (img[..., 0] * np.float32(0.114) + img[..., 1] * np.float32(0.587) + img[..., 2] * np.float32(0.299)).astype(np.float32)


_U8_LUT = np.arange(256, dtype=np.float32) / 255     # exactly enhance()'s `img.astype(float32) / 255`


def normalize_u8_on_device(x):
    """uint8 tensor -> float32 in [0,1] with numpy's correctly rounded division.  (torch divides by a
    Python scalar on the GPU by multiplying with its reciprocal, which is 1 ulp off for some of the 256
    values -- enough to flip a rounding tie 351 layers later.)"""
    lut = torch.from_numpy(_U8_LUT).to(x.device)
    return lut[x.long()]


def _gray2rgb(img):
    return np.repeat(img[:, :, None], 3, axis=2)


class RealESRGANer:
    """A helper class for upsampling images with RealESRGAN (MI355X/HIP backend).

    Args:
        scale (int): Upsampling scale factor used in the networks. It is usually 2 or 4.
        model_path (str | list[str] | dict): checkpoint path(s) ({'params_ema'|'params': state_dict}),
            or an already loaded checkpoint dict.
        dni_weight (list[float]): deep-network-interpolation weights when model_path is a list of two.
        model (nn.Module): the network (RRDBNet).
        tile (int): tile size; 0 = no tiling.
        tile_pad (int): pad size of each tile.  pre_pad (int): reflect pad before the network.
        half (bool): upstream's fp16 switch; here it selects the bf16 MFMA kernels.
        device: 'cuda' (= the ROCm GPU), torch.device or None (-> cuda if available).
    """

    def __init__(self, scale, model_path, dni_weight=None, model=None, tile=0, tile_pad=10, pre_pad=10,
                 half=False, device=None, gpu_id=None):
        self.scale = scale
        self.tile_size = tile
        self.tile_pad = tile_pad
        self.pre_pad = pre_pad
        self.mod_scale = None
        self.half = half
        self.tile_batch = 24  # upper bound on equal-shaped tiles per forward call (1 = upstream's serial loop)
        self.tile_streams = 3 # HIP streams (context replicas) the shape groups of one frame are spread over
        self.ragged_tiles = None  # bf16: all tiles of a frame, whatever their shapes, in ragged batches (see _run_tiles_ragged).  None = when the
                                  # model's dense blocks run as the LDS-resident strip kernel (rdb_bf16_strip.hip), whose schedule packs the
                                  # strips of ALL tiles onto the compute units; with the per-layer kernels ragged batches measured no faster
        self.ragged_batch = 64    # tiles per ragged batch
        self.small_job_tiles = 12   # a call with at most this many tiles (a rank's share of a sharded frame) ...
        self.small_job_streams = 5  # ... is spread over this many streams, its batches split until every stream has one

        if gpu_id:
            self.device = torch.device(f"cuda:{gpu_id}" if torch.cuda.is_available() else "cpu") if device is None else device
        else:
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu") if device is None else device
        self.device = torch.device(self.device)

        # where the weights came from: "verified" only for a file whose md5 is one the reference records
        self.weights_provenance = ("unverified", "in-memory state_dict (tests and benches use seeded synthetic weights: "
                                                 "no checkpoint ships with the reference)")
        if isinstance(model_path, list):
            assert len(model_path) == len(dni_weight), "model_path and dni_weight should have the save length."
            loadnet = self.dni(model_path[0], model_path[1], dni_weight)
            self.weights_provenance = ("unverified", "deep network interpolation of two checkpoints")
        elif isinstance(model_path, dict):
            loadnet = model_path
        else:
            if str(model_path).startswith("https://"):
                raise RuntimeError(f"{model_path}: downloading checkpoints is not supported (offline build); "
                                   "pass a local path to RealESRGAN_x2plus.pth / RealESRGAN_x4plus.pth")
            loadnet = torch.load(model_path, map_location=torch.device("cpu"), weights_only=True)
            self.weights_provenance = checkpoint_provenance(model_path)

        keyname = "params_ema" if "params_ema" in loadnet else "params"
        state = loadnet[keyname]
        self._adapt_declared_scale(model, state)
        model.load_state_dict(state, strict=True)
        model.eval()
        self.model = model.to(self.device)
        if self.half:
            self.model = self.model.half()
        if isinstance(self.model, RRDBNet) and self.tile_size > 0 and self.model.compute_dtype == "bf16":
            # a tiling wrapper batches tiles of different shapes (ragged batches): a tile's values must not depend on how it
            # was batched, nor on whether model(tile) was called directly -- kernels are chosen by arithmetic only
            self.model.size_independent = True

    @staticmethod
    def _adapt_declared_scale(model, state):
        """The reference declares RRDBNet(num_in_ch=3, ...) without scale=2 and then loads
        RealESRGAN_x2plus weights, whose conv_first has 12 input channels
        (standalone/direct_esrgan.py:104, standalone/download-x3-model.py:2-4 documents the
        resulting 'channel mismatch').  A genuine x2plus checkpoint is recognised by
        conv_first.weight having 4x the declared input channels and the model is switched to
        upstream's scale=2 form (pixel_unshuffle + 12-channel conv_first) instead of failing."""
        if not isinstance(model, RRDBNet) or "conv_first.weight" not in state:
            return
        cin = state["conv_first.weight"].shape[1]
        if model.scale == 4 and cin == model.num_in_ch * 4:
            warnings.warn("RRDBNet was declared without scale=2 but the checkpoint's conv_first takes "
                          f"{cin} channels: treating it as the scale=2 (pixel_unshuffle) network")
            model.set_scale(2)
        elif model.scale == 4 and cin == model.num_in_ch * 16:
            model.set_scale(1)

    def dni(self, net_a, net_b, dni_weight, key="params", loc="cpu"):
        """Deep network interpolation: weighted sum of two checkpoints' tensors."""
        if not isinstance(net_a, dict):
            net_a = torch.load(net_a, map_location=torch.device(loc), weights_only=True)
        if not isinstance(net_b, dict):
            net_b = torch.load(net_b, map_location=torch.device(loc), weights_only=True)
        for k, v_a in net_a[key].items():
            net_a[key][k] = dni_weight[0] * v_a + dni_weight[1] * net_b[key][k]
        return net_a

    # ------------------------------------------------------------------ pre / process / post
    def pre_process(self, img):
        """HWC float32 RGB [0,1] -> self.img [1,3,H,W] on device; reflect pre-pad; pad to mod_scale."""
        img = torch.from_numpy(np.transpose(img, (2, 0, 1))).float()
        self._pad_on_device(img.unsqueeze(0).to(self.device))

    def _pad_on_device(self, x):
        """The device-side half of upstream's pre_process: x is [1,3,H,W] float on self.device."""
        self.img = x
        if self.half:
            self.img = self.img.half()
        if self.pre_pad != 0:
            self.img = F.pad(self.img, (0, self.pre_pad, 0, self.pre_pad), "reflect")
        if self.scale == 2:
            self.mod_scale = 2
        elif self.scale == 1:
            self.mod_scale = 4
        if self.mod_scale is not None:
            self.mod_pad_h, self.mod_pad_w = 0, 0
            _, _, h, w = self.img.size()
            if h % self.mod_scale != 0:
                self.mod_pad_h = self.mod_scale - h % self.mod_scale
            if w % self.mod_scale != 0:
                self.mod_pad_w = self.mod_scale - w % self.mod_scale
            self.img = F.pad(self.img, (0, self.mod_pad_w, 0, self.mod_pad_h), "reflect")

    def process(self):
        self.output = self.model(self.img)

    def tile_grid(self, height, width):
        """The tile windows upstream's tile_process visits, in its order.  Each entry:
        (padded input window y0,y1,x0,x1 ; output window y0,y1,x0,x1 ; crop inside the tile's output y0,y1,x0,x1)."""
        tiles_x = math.ceil(width / self.tile_size)
        tiles_y = math.ceil(height / self.tile_size)
        s = self.scale
        grid = []
        for y in range(tiles_y):
            for x in range(tiles_x):
                ix0, iy0 = x * self.tile_size, y * self.tile_size
                ix1, iy1 = min(ix0 + self.tile_size, width), min(iy0 + self.tile_size, height)
                px0, px1 = max(ix0 - self.tile_pad, 0), min(ix1 + self.tile_pad, width)
                py0, py1 = max(iy0 - self.tile_pad, 0), min(iy1 + self.tile_pad, height)
                tw, th = ix1 - ix0, iy1 - iy0
                cx0, cy0 = (ix0 - px0) * s, (iy0 - py0) * s
                grid.append(((py0, py1, px0, px1), (iy0 * s, iy1 * s, ix0 * s, ix1 * s), (cy0, cy0 + th * s, cx0, cx0 + tw * s)))
        return grid

    def batch_for(self, th, tw, ntiles):
        """Tiles of th x tw input evaluated per forward call: at most ``tile_batch``, and on the HIP
        backend the count whose workgroups fill the CUs most evenly (values do not depend on it)."""
        nb = max(1, min(int(self.tile_batch), ntiles))
        if nb > 1 and isinstance(self.model, RRDBNet) and self.device.type == "cuda":
            nb = self.model.preferred_batch(self.device, th, tw, nb)
        return nb

    def run_tiles(self, img, tiles, sink):
        """Evaluates the network on windows of `img` ([1,C,H,W] on self.device).  `tiles` is a list of
        (y0, y1, x0, x1, payload); `sink(payload, out)` receives each window's output [1,C,h*s,w*s] (a view
        valid on the current stream).  Equal-shaped windows are batched (batch_for); on the HIP backend
        the shape groups are spread over `tile_streams` streams with their own context replicas, so the
        small edge-tile groups -- whose 351 launches are latency-bound -- overlap the large ones.
        Values do not depend on batching or stream assignment."""
        groups = {}
        for t in tiles:
            groups.setdefault((t[1] - t[0], t[3] - t[2]), []).append(t)
        order = sorted(groups.items(), key=lambda kv: -kv[0][0] * kv[0][1] * len(kv[1]))
        hip = isinstance(self.model, RRDBNet) and img.device.type == "cuda"
        if hip and img.shape[0] == 1 and self.model.compute_dtype == "bf16":
            ragged = self.model.strip_kernel_active() if self.ragged_tiles is None else bool(self.ragged_tiles)
            if ragged and (len(order) > 1 or self.ragged_tiles is None):
                return self._run_tiles_ragged(img, tiles, sink, single_stream=self.ragged_tiles is None)
        # Batches of equal-shaped windows, one shape group per stream.  A small job (a rank's share of a sharded frame: five
        # tiles of an 8-way split 4K frame, often of one shape) is spread wider: every batch its own unit, the largest
        # halved until each of `small_job_streams` streams has one -- one stream would run 351 launches of a few hundred
        # workgroups each, five overlap their prologues, epilogues and tails (measured on one GPU with an 8-rank share:
        # 25.6 -> 16.9 ms for the slowest rank)
        units = []                                    # a unit = the batches of one shape group, run in order on one stream
        for shape, ts in order:
            nb = self.batch_for(shape[0], shape[1], len(ts)) if img.shape[0] == 1 else 1
            units.append([ts[i:i + nb] for i in range(0, len(ts), nb)])
        nstreams = 1
        if hip and len(tiles) > 1:
            nstreams = max(1, int(self.tile_streams))
            if len(tiles) <= self.small_job_tiles:
                nstreams = min(max(nstreams, int(self.small_job_streams)), len(tiles))
                units = [[b] for u in units for b in u]
                while len(units) < nstreams:
                    k = max(range(len(units)), key=lambda i: len(units[i][0]))
                    b = units[k][0]
                    if len(b) < 2:
                        break
                    units[k:k + 1] = [[b[:(len(b) + 1) // 2]], [b[(len(b) + 1) // 2:]]]
            elif len(units) == 1:
                nstreams = 1

        def area(b):
            return (b[0][1] - b[0][0]) * (b[0][3] - b[0][2]) * len(b)

        def run_batch(chunk, slot):
            if len(chunk) == 1:
                inp = img[:, :, chunk[0][0]:chunk[0][1], chunk[0][2]:chunk[0][3]]
            else:
                inp = torch.cat([img[:, :, t[0]:t[1], t[2]:t[3]] for t in chunk], 0)
            with torch.no_grad():
                out = self.model(inp, slot=slot) if hip else self.model(inp)
            for j, t in enumerate(chunk):
                sink(t[4], out[j:j + 1] if len(chunk) > 1 else out)

        units.sort(key=lambda u: -sum(area(b) for b in u))
        if nstreams == 1:
            for u in units:
                for b in u:
                    run_batch(b, 0)
            return
        main = torch.cuda.current_stream(img.device)
        if not hasattr(self, "_side_streams") or len(self._side_streams) < nstreams - 1:
            self._side_streams = [torch.cuda.Stream(device=img.device) for _ in range(nstreams - 1)]
        streams = [main] + self._side_streams[:nstreams - 1]
        for s in streams[1:]:
            s.wait_stream(main)                       # img / the output canvas were produced on the main stream
        load = [0] * nstreams
        for u in units:                               # largest unit first, each to the least-loaded stream
            k = load.index(min(load))
            load[k] += sum(area(b) for b in u)
            with torch.cuda.stream(streams[k]):
                for b in u:
                    run_batch(b, k)
        for s in streams[1:]:
            main.wait_stream(s)

    def _run_tiles_ragged(self, img, tiles, sink, single_stream=False):
        """All windows of a frame, whatever their shapes, in `tile_streams` ragged batches that run side by side: every
        window lies in the top-left corner of an equal-sized slot and the kernels take each image's own size from the
        call (nesr_forward_ragged).  Against one batch per tile shape: the small edge-tile groups were latency-bound
        launches of a few workgroups -- and with the tiles sharded over ranks every group shrinks further -- while
        batches of mixed sizes keep every launch large; several batches on their own streams (context replicas) fill
        each other's prologues and epilogues as the shape groups did.  bf16 only; the model is size_independent, so a
        window's values are the ones model(window) gives it alone."""
        if not self.model.size_independent:
            self.model.size_independent = True
        cap = max(1, min(self.model.RAGGED_MAX, int(self.ragged_batch)))
        # (strip kernel: one batch after the other on one stream -- a persistent launch holds the whole device)
        nstreams = 1 if single_stream else max(1, min(int(self.tile_streams), len(tiles)))
        # largest first, each to the least-loaded batch; a batch that is full opens another one on the same stream
        ts = sorted(tiles, key=lambda t: -(t[1] - t[0]) * (t[3] - t[2]))
        lanes = [[[]] for _ in range(nstreams)]
        load = [0] * nstreams
        for t in ts:
            k = load.index(min(load))
            load[k] += (t[1] - t[0]) * (t[3] - t[2])
            if len(lanes[k][-1]) == cap:
                lanes[k].append([])
            lanes[k][-1].append(t)

        def run_batch(chunk, slot):
            H = max(t[1] - t[0] for t in chunk)
            W = max(t[3] - t[2] for t in chunk)
            x = img.new_zeros((len(chunk), img.shape[1], H, W))
            for j, t in enumerate(chunk):
                x[j, :, :t[1] - t[0], :t[3] - t[2]] = img[0, :, t[0]:t[1], t[2]:t[3]]
            with torch.no_grad():
                out = self.model.forward_ragged(x, [(t[1] - t[0], t[3] - t[2]) for t in chunk], slot=slot)
            s = out.shape[2] // H
            for j, t in enumerate(chunk):
                sink(t[4], out[j:j + 1, :, :(t[1] - t[0]) * s, :(t[3] - t[2]) * s])

        if nstreams == 1:
            for chunk in lanes[0]:
                run_batch(chunk, 0)
            return
        main = torch.cuda.current_stream(img.device)
        if not hasattr(self, "_side_streams") or len(self._side_streams) < nstreams - 1:
            self._side_streams = [torch.cuda.Stream(device=img.device) for _ in range(nstreams - 1)]
        streams = [main] + self._side_streams[:nstreams - 1]
        for st in streams[1:]:
            st.wait_stream(main)
        for k in range(nstreams):
            with torch.cuda.stream(streams[k]):
                for chunk in lanes[k]:
                    if chunk:
                        run_batch(chunk, k)
        for st in streams[1:]:
            main.wait_stream(st)

    def tile_process(self):
        """Runs the network on overlapping tiles and pastes the un-padded centres (upstream
        semantics, tile for tile); see run_tiles for the batching / stream spreading."""
        batch, channel, height, width = self.img.shape
        s = self.scale
        self.output = self.img.new_zeros((batch, channel, height * s, width * s))
        tiles = [(py0, py1, px0, px1, (o, c)) for ((py0, py1, px0, px1), o, c) in self.tile_grid(height, width)]

        def paste(payload, out):
            (oy0, oy1, ox0, ox1), (cy0, cy1, cx0, cx1) = payload
            self.output[:, :, oy0:oy1, ox0:ox1] = out[:, :, cy0:cy1, cx0:cx1]

        self.run_tiles(self.img, tiles, paste)

    def post_process(self):
        if self.mod_scale is not None:
            _, _, h, w = self.output.size()
            self.output = self.output[:, :, 0:h - self.mod_pad_h * self.scale, 0:w - self.mod_pad_w * self.scale]
        if self.pre_pad != 0:
            _, _, h, w = self.output.size()
            self.output = self.output[:, :, 0:h - self.pre_pad * self.scale, 0:w - self.pre_pad * self.scale]
        return self.output

    def _run(self):
        if self.tile_size > 0:
            self.tile_process()
        else:
            self.process()
        return self.post_process()

    # ------------------------------------------------------------------ enhance
    def _fused_u8_ok(self, img):
        """The fused u8 kernel path applies when the call reduces to one network evaluation of a
        plain 8-bit BGR frame: no tiling needed, no pre_pad / mod_pad, HIP-backed model."""
        if not isinstance(self.model, RRDBNet) or self.device.type != "cuda":
            return False
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3 or self.pre_pad != 0:
            return False
        h, w = img.shape[:2]
        if self.tile_size > 0 and (h > self.tile_size or w > self.tile_size):
            return False
        ms = {2: 2, 1: 4}.get(self.scale, 1)
        if h % ms or w % ms:
            return False
        return self.model.num_in_ch == 3 and self.model.num_out_ch == 3 and self.model.out_scale() == self.scale

    def _u8_on_device_ok(self, img):
        """8-bit BGR frames on the HIP backend (any tiling / padding): the uint8 frame is uploaded
        (4x fewer bytes than float), normalised, padded, tiled, clamped and quantised on the GPU with
        the same float32 operations enhance() performs in numpy, and only uint8 comes back."""
        return (isinstance(self.model, RRDBNet) and self.device.type == "cuda" and img.dtype == np.uint8
                and img.ndim == 3 and img.shape[2] == 3)

    def _u8_tiles_fused_ok(self, h, w):
        """8-bit frames larger than a tile whose tiles run as ragged batches (bf16, strip kernel): cut and paste are one launch
        each and the float canvas of the frame never exists.  Frames that need the reflect pre-pad / mod-pad keep the general path."""
        ms = {2: 2, 1: 4}.get(self.scale, 1)
        return (self.tile_size > 0 and self.pre_pad == 0 and h % ms == 0 and w % ms == 0 and isinstance(self.model, RRDBNet)
                and self.model.compute_dtype == "bf16" and self.model.strip_kernel_active() and self.ragged_tiles is None
                and self.model.out_scale() == self.scale)

    @torch.no_grad()
    def tiles_u8_on_device(self, frame_u8, windows, pastes, dst_u8):
        """frame_u8 [H, W, 3] uint8 on the device; windows [(y0, x0, h, w)] of the padded tiles; pastes [(crop_y, crop_x, h, w, dst byte
        offset, dst row pitch)] -> the tiles' quantised centres in dst_u8 (uint8, on the device).  Ragged batches of at most
        RAGGED_MAX tiles: cut (nesr_cut_tiles_u8), forward_ragged, paste (nesr_paste_tiles_u8)."""
        from .rrdbnet import cut_tiles_u8, paste_tiles_u8
        cap = max(1, min(self.model.RAGGED_MAX, int(self.ragged_batch)))
        for i in range(0, len(windows), cap):
            win, pst = windows[i:i + cap], pastes[i:i + cap]
            hs, ws = max(v[2] for v in win), max(v[3] for v in win)
            x = cut_tiles_u8(frame_u8, win, (hs, ws), flip_rgb=True, through_fp16=bool(self.half))
            out = self.model.forward_ragged(x, [(v[2], v[3]) for v in win])
            paste_tiles_u8(out, pst, dst_u8, flip_rgb=True, round_nearest=True, through_fp16=bool(self.half))

    @torch.no_grad()
    def _enhance_u8_tiles_fused(self, img):
        h, w = img.shape[:2]
        s = self.scale
        frame = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)          # H2D: uint8 HWC BGR
        canvas = torch.empty((h * s, w * s, 3), dtype=torch.uint8, device=self.device)
        windows, pastes = [], []
        for (py0, py1, px0, px1), (oy0, oy1, ox0, ox1), (cy0, cy1, cx0, cx1) in self.tile_grid(h, w):
            windows.append((py0, px0, py1 - py0, px1 - px0))
            pastes.append((cy0, cx0, cy1 - cy0, cx1 - cx0, (oy0 * w * s + ox0) * 3, w * s * 3))
        self.tiles_u8_on_device(frame, windows, pastes, canvas)
        host = torch.empty(canvas.shape, dtype=torch.uint8, pin_memory=True)          # (the caching host allocator recycles these)
        host.copy_(canvas, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        self._check_range()
        return host.numpy()

    @torch.no_grad()
    def _enhance_u8_on_device(self, img):
        if self._u8_tiles_fused_ok(img.shape[0], img.shape[1]) and (img.shape[0] > self.tile_size or img.shape[1] > self.tile_size):
            return self._enhance_u8_tiles_fused(img)
        x = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)            # H2D: uint8 HWC BGR
        x = normalize_u8_on_device(x.permute(2, 0, 1).flip(0)).unsqueeze(0)          # BGR->RGB, /255 (f32), HWC->NCHW
        self._pad_on_device(x)
        out = self._run()                                                            # [1,3,H*s,W*s] RGB
        out = out.data.squeeze(0).float().clamp_(0, 1)
        q = (out.flip(0).permute(1, 2, 0) * 255.0).round().to(torch.uint8)           # RGB->BGR, CHW->HWC, x255, round
        host = q.contiguous().cpu().numpy()
        self._check_range()
        return host

    def _check_range(self, slot=None):
        """After a device-to-host copy: an out-of-range forward of the f16-pair fp32 form raises here."""
        if isinstance(self.model, RRDBNet):
            self.model.check_range(slot)

    @torch.no_grad()
    def enhance_many(self, imgs, inflight=4):
        """``[self.enhance(img) for img in imgs]`` with up to `inflight` frames on the GPU at once.

        Not part of upstream's API: a frame whose network layers are only a few hundred workgroups (512x512:
        256-512 per layer in one round, all in the same phase) leaves the GPU idle a third of the time; a second frame on its own HIP
        stream and context replica fills it (bench.py's default `value`: 134 -> 160 / 167 / 170 MP/s with 2 / 3 / 4 frames
        in flight, no more beyond).  Frames
        that do not take the fused 8-bit path (tiling, padding, alpha, 16 bit) are processed one at a time.
        The results are identical to enhance()'s."""
        imgs = list(imgs)
        if inflight <= 1 or not imgs or not all(self._fused_u8_ok(i) for i in imgs):
            return [self.enhance(i) for i in imgs]
        streams = [torch.cuda.Stream(self.device) for _ in range(inflight)]
        caller = torch.cuda.current_stream(self.device)
        for st in streams:
            st.wait_stream(caller)     # the context workspaces (slot 0 is the caller's own) may still be in use there
        results, pending = [None] * len(imgs), []

        def finish(entry):
            idx, host, ev = entry
            ev.synchronize()
            with torch.cuda.stream(streams[idx % inflight]):
                self._check_range(idx % inflight)
            results[idx] = (host.numpy().copy(), "RGB")

        for i, img in enumerate(imgs):
            if len(pending) >= inflight:
                finish(pending.pop(0))
            k = i % inflight
            with torch.cuda.stream(streams[k]):
                x = torch.from_numpy(np.ascontiguousarray(img)).pin_memory().to(self.device, non_blocking=True)
                y = self.model.forward_u8(x, flip_rgb=True, round_nearest=True, slot=k)
                host = torch.empty(y.shape, dtype=torch.uint8, pin_memory=True)
                host.copy_(y, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            pending.append((i, host, ev))
        for entry in pending:
            finish(entry)
        for st in streams:
            caller.wait_stream(st)
        return results

    @torch.no_grad()
    def enhance_float(self, img, alpha_upsampler="realesrgan"):
        """Everything of enhance() up to (not including) quantisation: returns (HWC float32 in
        [0,1] in BGR(A)/gray order, img_mode, max_range)."""
        img = img.astype(np.float32)
        if np.max(img) > 256:  # 16-bit image
            max_range = 65535
        else:
            max_range = 255
        img = img / max_range
        if len(img.shape) == 2:  # gray image
            img_mode = "L"
            img = _gray2rgb(img)
        elif img.shape[2] == 4:  # RGBA image with alpha channel
            img_mode = "RGBA"
            alpha = img[:, :, 3]
            img = img[:, :, 0:3][:, :, ::-1]
            if alpha_upsampler == "realesrgan":
                alpha = _gray2rgb(alpha)
        else:
            img_mode = "RGB"
            img = img[:, :, ::-1]

        self.pre_process(np.ascontiguousarray(img))
        output_img = self._run()
        output_img = output_img.data.squeeze().float().cpu().clamp_(0, 1).numpy()
        self._check_range()
        output_img = np.transpose(output_img[[2, 1, 0], :, :], (1, 2, 0))
        if img_mode == "L":
            output_img = _bgr2gray(output_img)

        if img_mode == "RGBA":
            if alpha_upsampler == "realesrgan":
                self.pre_process(np.ascontiguousarray(alpha))
                output_alpha = self._run()
                output_alpha = output_alpha.data.squeeze().float().cpu().clamp_(0, 1).numpy()
                self._check_range()
                output_alpha = np.transpose(output_alpha[[2, 1, 0], :, :], (1, 2, 0))
                output_alpha = _bgr2gray(output_alpha)
            else:   # upstream: cv2.resize(alpha, (w * scale, h * scale), interpolation=cv2.INTER_LINEAR)
                from . import imgproc
                h, w = alpha.shape[0:2]
                a = torch.from_numpy(np.ascontiguousarray(alpha)).to(self.device)
                output_alpha = imgproc.linear_resize_f32(a, h * self.scale, w * self.scale).cpu().numpy()
            output_img = np.concatenate([output_img, output_alpha[:, :, None]], axis=2)
        return output_img, img_mode, max_range

    @torch.no_grad()
    def enhance(self, img, outscale=None, alpha_upsampler="realesrgan"):
        """img: HWC uint8/uint16 BGR | BGRA | gray ndarray -> (ndarray of the same kind, upscaled; img_mode).

        A forward whose persistent dense-block launch gave up waiting (another process's kernels kept its workgroups off the
        device: NesrHipError from the status check, never a silently wrong image) is evaluated once more: the context has
        switched to per-layer launches by then (include/nesr_hip.h, nesr_set_fused)."""
        from ._lib import NesrHipError, NesrRangeError
        try:
            return self._enhance_once(img, outscale, alpha_upsampler)
        except NesrRangeError:
            raise
        except NesrHipError as e:
            if "gave up waiting" not in str(e):
                raise
            warnings.warn(f"{e}; evaluating the frame again with per-layer launches")
            return self._enhance_once(img, outscale, alpha_upsampler)

    def _enhance_once(self, img, outscale=None, alpha_upsampler="realesrgan"):
        h_input, w_input = img.shape[0:2]
        plain_alpha = alpha_upsampler != "realesrgan" and img.ndim == 3 and img.shape[2] == 4

        if self._fused_u8_ok(img) and not plain_alpha:
            # /255, BGR->RGB, network, clamp, RGB->BGR, x255, round -- all inside the HIP path
            x = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
            output = self.model.forward_u8(x, flip_rgb=True, round_nearest=True).cpu().numpy()
            self._check_range(0)
            img_mode = "RGB"
        elif self._u8_on_device_ok(img):
            output = self._enhance_u8_on_device(img)
            img_mode = "RGB"
        else:
            output_img, img_mode, max_range = self.enhance_float(img, alpha_upsampler)
            if max_range == 65535:  # 16-bit image
                output = (output_img * 65535.0).round().astype(np.uint16)
            else:
                output = (output_img * 255.0).round().astype(np.uint8)

        if outscale is not None and outscale != float(self.scale):
            # upstream: cv2.resize(output, (int(w_input * outscale), int(h_input * outscale)), interpolation=cv2.INTER_LANCZOS4);
            # here OpenCV's algorithm restated on the device (imgproc.lanczos4_resize: PARITY UNPINNED, cv2 is not installed)
            from . import imgproc
            t = torch.from_numpy(np.ascontiguousarray(output if output.dtype == np.uint8 else output.astype(np.int32))).to(self.device)
            t = t[:, :, None] if t.dim() == 2 else t
            r = imgproc.lanczos4_resize(t, int(h_input * outscale), int(w_input * outscale)).cpu().numpy().astype(output.dtype)
            output = r[:, :, 0] if output.ndim == 2 else r
        return output, img_mode
