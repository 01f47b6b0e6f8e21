"""``RRDBNet`` drop-in: same constructor, state_dict key names and call signature as
``basicsr.archs.rrdbnet_arch.RRDBNet`` (basicsr>=1.4.2, requirements.txt:10), which the
reference builds at nesr/nesr.py:216, standalone/direct_esrgan.py:104 and
standalone/superres_project.py:69 and calls at nesr/nesr.py:891,935 (``model(img_12ch)``).

The module owns ordinary torch Parameters under upstream's names (so ``load_state_dict(strict=True)``,
``.parameters()``, ``.to()``, ``.eval()`` behave as the reference expects, nesr/nesr.py:888,962-973),
but its ``forward`` is the hand-written HIP path in libnesr_hip.so, reached through the C ABI.
There is no torch/CPU implementation of forward here: a non-CUDA input raises.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import torch
from torch import nn

from . import _lib


def conv_first_in_ch(num_in_ch: int, scale: int) -> int:
    """basicsr RRDBNet.__init__: scale 2 -> x4 channels (pixel_unshuffle 2), scale 1 -> x16."""
    return num_in_ch * {2: 4, 1: 16}.get(scale, 1)


def rrdbnet_state_dict_spec(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
    """Ordered {key: shape} of the 702 (for 23 blocks) tensors of an RRDBNet checkpoint."""
    spec = OrderedDict()

    def conv(name, cin, cout):
        spec[name + ".weight"] = (cout, cin, 3, 3)
        spec[name + ".bias"] = (cout,)

    conv("conv_first", conv_first_in_ch(num_in_ch, scale), num_feat)
    for b in range(num_block):
        for r in (1, 2, 3):
            for k in (1, 2, 3, 4):
                conv(f"body.{b}.rdb{r}.conv{k}", num_feat + (k - 1) * num_grow_ch, num_grow_ch)
            conv(f"body.{b}.rdb{r}.conv5", num_feat + 4 * num_grow_ch, num_feat)
    for name in ("conv_body", "conv_up1", "conv_up2", "conv_hr"):
        conv(name, num_feat, num_feat)
    conv("conv_last", num_feat, num_out_ch)
    return spec


class _ConvParams(nn.Module):
    """Parameter holder for one 3x3 conv (weight OIHW + bias).  Not callable: the arithmetic
    happens in the HIP library, never in torch."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, 3, 3), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("_ConvParams holds weights only; RRDBNet.forward runs in libnesr_hip.so")


class _RDBParams(nn.Module):
    def __init__(self, nf, gc):
        super().__init__()
        for k in range(1, 5):
            setattr(self, f"conv{k}", _ConvParams(nf + (k - 1) * gc, gc))
        self.conv5 = _ConvParams(nf + 4 * gc, nf)


class _RRDBParams(nn.Module):
    def __init__(self, nf, gc):
        super().__init__()
        self.rdb1 = _RDBParams(nf, gc)
        self.rdb2 = _RDBParams(nf, gc)
        self.rdb3 = _RDBParams(nf, gc)


class RRDBNet(nn.Module):
    """Networks consisting of Residual in Residual Dense Blocks (ESRGAN / Real-ESRGAN generator).

    Args mirror upstream: num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32.
    Extra keyword ``compute_dtype``:
      "f32" (default; the reference's half=False)  f32 in / out / accumulation; every conv operand is carried
                      as a pair of halves (x = hi + lo 2^-11) and each product is three f16 MFMAs (conv3x3_f16x2.hip);
                      whole-network max abs error vs an f64 evaluation 3e-6 (torch CPU f32: 1e-6).  Values beyond
                      +-65504 or non-finite do not fit: weights are refused at upload, activations turn the output
                      into NaN and raise NesrRangeError at the next check_range()/check_status()
      "f32-winograd"  f32 matrix cores, Winograd F(2x2,3x3) for the feature-map convs (error 2e-6)
      "f32-direct"    f32 matrix cores, direct implicit GEMM: bitwise a k-ordered fmaf chain
      "bf16"          bf16 storage and MFMA, f32 accumulation (upstream's half=True is fp16)
    """

    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                 compute_dtype="f32"):
        super().__init__()
        self.num_in_ch = num_in_ch
        self.num_out_ch = num_out_ch
        self.scale = scale
        self.num_feat = num_feat
        self.num_block = num_block
        self.num_grow_ch = num_grow_ch
        self.compute_dtype = compute_dtype
        self._build_params()
        self.calls = 0            # forward evaluations so far (callers assert on it: the reference's exception ladders
                                  # turn a dead backend into a silent bicubic resize, nesr/nesr.py:815-843)
        self._ctx = None          # (ctypes handle, device index, dtype code): slot 0
        self._dirty = True        # parameters changed since the last upload
        self._extra = {}          # slot -> (handle, device index, dtype code): replicas for concurrent streams

    # ------------------------------------------------------------------ parameters
    def _build_params(self):
        cin0 = conv_first_in_ch(self.num_in_ch, self.scale)
        self.conv_first = _ConvParams(cin0, self.num_feat)
        self.body = nn.Sequential(*[_RRDBParams(self.num_feat, self.num_grow_ch) for _ in range(self.num_block)])
        self.conv_body = _ConvParams(self.num_feat, self.num_feat)
        self.conv_up1 = _ConvParams(self.num_feat, self.num_feat)
        self.conv_up2 = _ConvParams(self.num_feat, self.num_feat)
        self.conv_hr = _ConvParams(self.num_feat, self.num_feat)
        self.conv_last = _ConvParams(self.num_feat, self.num_out_ch)

    def set_scale(self, scale):
        """Re-declares the network for another upstream ``scale`` (changes conv_first's input
        channels); used by RealESRGANer to accept genuine x2plus weights for a model that was
        declared without ``scale=2`` (the reference does that: SURVEY.md Appendix A)."""
        if scale == self.scale:
            return
        self.scale = scale
        dev = self.conv_body.weight.device
        self.conv_first = _ConvParams(conv_first_in_ch(self.num_in_ch, scale), self.num_feat).to(dev)
        self._release()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._dirty = True
        return out

    def half(self):
        """Upstream's fp16 switch (RealESRGANer(half=True) calls model.half()).  Here it selects the bf16 MFMA
        kernels; the parameters stay float32, so the bf16 weights are rounded once from the checkpoint's values
        (not float32 -> fp16 -> bf16)."""
        self.compute_dtype = "bf16"
        self._dirty = True
        return self

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._dirty = True
        p = self.conv_body.weight
        if p.dtype in (torch.float16, torch.bfloat16):
            self.compute_dtype = "bf16"   # .half(): upstream's fp16 switch selects the bf16 MFMA kernels here
        return out

    # ------------------------------------------------------------------ HIP context
    def _release(self):
        if self._ctx is not None:
            _lib.load().nesr_destroy(self._ctx[0])
            self._ctx = None
        for h in getattr(self, "_extra", {}).values():
            _lib.load().nesr_destroy(h[0])
        self._extra = {}
        self._dirty = True

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _dtype_code(self):
        if self.compute_dtype in ("f32", "fp32", torch.float32, "f32-split", "f32-f16x2", "split"):
            return _lib.DTYPE_F32_SPLIT         # default f32 algorithm: operands as (hi, lo) half pairs on the f16 matrix cores
        if self.compute_dtype in ("f32-winograd", "f32w", "winograd"):
            return _lib.DTYPE_F32_WINOGRAD      # f32 matrix cores, Winograd F(2x2,3x3) for the feature-map convs
        if self.compute_dtype in ("f32-direct", "direct"):
            return _lib.DTYPE_F32               # direct implicit GEMM everywhere (bitwise a k-ordered fmaf chain)
        if self.compute_dtype in ("bf16", torch.bfloat16, "half", torch.float16):
            return _lib.DTYPE_BF16
        raise ValueError(f"compute_dtype {self.compute_dtype!r}: expected 'f32' or 'bf16'")

    def _upload(self, handle):
        lib = _lib.load()
        for key, t in self.state_dict().items():
            arr = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * arr.dim())(*arr.shape)
            _lib.check(lib.nesr_load_weight(handle, key.encode(), ctypes.c_void_p(arr.data_ptr()), shape, arr.dim()),
                       f"nesr_load_weight({key})")
        _lib.check(lib.nesr_finalize_weights(handle), "nesr_finalize_weights")

    def _create(self, index, code):
        handle = ctypes.c_void_p()
        unshuffle = {2: 2, 1: 4}.get(self.scale, 0)
        _lib.check(_lib.load().nesr_create(ctypes.byref(handle), index, conv_first_in_ch(self.num_in_ch, self.scale),
                                           unshuffle, self.num_feat, self.num_block, self.num_grow_ch, self.num_out_ch, code),
                   "nesr_create")
        if getattr(self, "_size_independent", False):
            _lib.check(_lib.load().nesr_set_size_independent(handle, 1), "nesr_set_size_independent")
        return handle

    def _context(self, device: torch.device, slot: int = 0):
        """HIP context of `slot`.  Slot 0 is the model's own; further slots are replicas (own packed
        weights and workspace) so independent forward calls can run concurrently on different streams."""
        lib = _lib.load()
        index = device.index if device.index is not None else torch.cuda.current_device()
        code = self._dtype_code()
        if self._ctx is not None and (self._ctx[1] != index or self._ctx[2] != code):
            self._release()
        if self._dirty and self._extra:
            for h in self._extra.values():
                lib.nesr_destroy(h[0])
            self._extra = {}
        if slot != 0:
            self._context(device, 0)   # slot 0 first: settles device / dtype / dirty state
            if slot not in self._extra:
                handle = self._create(index, code)
                self._upload(handle)
                self._extra[slot] = (handle, index, code)
                # replicas exist to run beside each other: tell every context of the model
                for h in [self._ctx] + list(self._extra.values()):
                    _lib.check(lib.nesr_set_concurrent(h[0], 1), "nesr_set_concurrent")
            return self._extra[slot][0]
        if self._ctx is None:
            self._ctx = (self._create(index, code), index, code)
            self._dirty = True
        if self._dirty:
            self._upload(self._ctx[0])
            self._dirty = False
        return self._ctx[0]

    RAGGED_MAX = 64          # images per forward_ragged call (nesr::RAG_MAX)

    @property
    def size_independent(self):
        """True: kernels are chosen by arithmetic only, never by image size, so an image has the same bits alone, in an
        equal-shape batch and in a ragged batch (include/nesr_hip.h: nesr_set_size_independent)."""
        return getattr(self, "_size_independent", False)

    @size_independent.setter
    def size_independent(self, on):
        self._size_independent = bool(on)
        handles = ([self._ctx] if self._ctx is not None else []) + list(self._extra.values())
        for h in handles:
            _lib.check(_lib.load().nesr_set_size_independent(h[0], 1 if on else 0), "nesr_set_size_independent")

    def strip_kernel_active(self):
        """True when bf16 dense blocks of a size-independent model run as the LDS-resident strip kernel (rdb_bf16_strip.hip;
        NESR_STRIP=0 turns it off): the tiling wrapper then hands all tiles of a frame over as one ragged batch."""
        import os
        return (self.compute_dtype == "bf16" and self.size_independent and self.num_feat == 64 and self.num_grow_ch == 32
                and self.num_block > 0 and os.environ.get("NESR_STRIP", "-1") != "0")

    # ------------------------------------------------------------------ forward
    def out_scale(self):
        """Output size / input size of forward(): 4 / unshuffle factor."""
        return {2: 2, 1: 1}.get(self.scale, 4)

    def _require_cuda(self, x):
        if x.device.type != "cuda":
            raise RuntimeError(
                "RRDBNet.forward runs only on an AMD GPU through libnesr_hip.so; got a tensor on "
                f"{x.device}. There is no CPU/PyTorch fallback for this path.")

    @torch.no_grad()
    def forward(self, x, slot: int = 0):
        """x: [N, num_in_ch, H, W] float on a ROCm device -> [N, num_out_ch, H*s, W*s].
        `slot` selects a context replica (see _context); work is enqueued on torch's current stream."""
        self._require_cuda(x)
        if x.dim() != 4:
            raise ValueError(f"expected NCHW input, got shape {tuple(x.shape)}")
        in_dtype = x.dtype
        xf = x.to(torch.float32).contiguous()
        n, c, h, w = xf.shape
        u = {2: 2, 1: 4}.get(self.scale, 1)
        if h % u or w % u:
            raise AssertionError(f"hh({h}) and hw({w}) must be divisible by {u}")  # upstream pixel_unshuffle asserts
        s = self.out_scale()
        self.calls += 1
        with torch.cuda.device(xf.device):
            ctx = self._context(xf.device, slot)
            y = torch.empty((n, self.num_out_ch, h * s, w * s), dtype=torch.float32, device=xf.device)
            stream = torch.cuda.current_stream(xf.device).cuda_stream
            _lib.check(_lib.load().nesr_forward(ctx, ctypes.c_void_p(xf.data_ptr()), n, c, h, w,
                                                ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(stream)), "nesr_forward")
        return y if in_dtype == torch.float32 else y.to(in_dtype)

    @torch.no_grad()
    def forward_ragged(self, x, sizes, slot: int = 0):
        """Images of different sizes in one batch (the tiles of a frame: realesrgan's tile_process as
        standalone/direct_esrgan.py:118-127 configures it cuts interior tiles of 532 x 532 and smaller edge tiles).
        x: [N, num_in_ch, H, W] float on a ROCm device, image i in the top-left sizes[i] = (h_i, w_i) pixels of slot i
        (the rest of a slot is ignored); returns [N, num_out_ch, H*s, W*s] whose slot i holds image i's output in its
        top-left h_i*s x w_i*s pixels (the rest is unspecified).  Every image gets the values forward() gives it alone
        on a model with size_independent = True.  compute_dtype "bf16" only; N <= RAGGED_MAX."""
        self._require_cuda(x)
        if x.dim() != 4 or len(sizes) != x.shape[0]:
            raise ValueError(f"expected NCHW input and one (h, w) per image, got {tuple(x.shape)} and {len(sizes)} sizes")
        in_dtype = x.dtype
        xf = x.to(torch.float32).contiguous()
        n, c, h, w = xf.shape
        hw = (ctypes.c_int * (2 * n))(*[int(v) for pair in sizes for v in pair])
        s = self.out_scale()
        self.calls += 1
        with torch.cuda.device(xf.device):
            ctx = self._context(xf.device, slot)
            y = torch.empty((n, self.num_out_ch, h * s, w * s), dtype=torch.float32, device=xf.device)
            stream = torch.cuda.current_stream(xf.device).cuda_stream
            _lib.check(_lib.load().nesr_forward_ragged(ctx, ctypes.c_void_p(xf.data_ptr()), n, c, h, w, hw,
                                                       ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(stream)), "nesr_forward_ragged")
        return y if in_dtype == torch.float32 else y.to(in_dtype)

    @torch.no_grad()
    def forward_u8(self, img_hwc_u8, flip_rgb=True, round_nearest=True, slot: int = 0):
        """Fused image path: u8 HWC [H,W,3] device tensor -> u8 HWC [H*s,W*s,3] (`slot`: context replica, see forward).

        flip_rgb/round_nearest = (True, True) reproduces RealESRGANer.enhance's /255, BGR<->RGB,
        clamp, x255, round; (False, False) reproduces nesr/nesr.py:851-857,894-898 (truncation)."""
        self._require_cuda(img_hwc_u8)
        if img_hwc_u8.dtype != torch.uint8 or img_hwc_u8.dim() != 3 or img_hwc_u8.shape[2] != 3:
            raise ValueError("expected a uint8 [H, W, 3] tensor")
        x = img_hwc_u8.contiguous()
        h, w, _ = x.shape
        s = self.out_scale()
        self.calls += 1
        with torch.cuda.device(x.device):
            ctx = self._context(x.device, slot)
            y = torch.empty((h * s, w * s, 3), dtype=torch.uint8, device=x.device)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(_lib.load().nesr_forward_u8(ctx, ctypes.c_void_p(x.data_ptr()), h, w, ctypes.c_void_p(y.data_ptr()),
                                                   1 if flip_rgb else 0,
                                                   _lib.ROUND_NEAREST if round_nearest else _lib.ROUND_TRUNC,
                                                   ctypes.c_void_p(stream)), "nesr_forward_u8")
        return y

    # ------------------------------------------------------------------ sharded frames through the C ABI (RCCL below Python)
    def comm_init(self, device, rank, nranks, unique_id: bytes):
        """ncclCommInitRank for this model's context on `device` (include/nesr_hip.h: nesr_comm_init); `unique_id` = the 128 bytes
        rank 0 got from comm_unique_id(), distributed by the caller."""
        ctx = self._context(torch.device(device))
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.check(_lib.load().nesr_comm_init(ctx, int(rank), int(nranks), buf), "nesr_comm_init")

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().nesr_comm_unique_id(buf), "nesr_comm_unique_id")
        return buf.raw

    @torch.no_grad()
    def forward_sharded_u8(self, band_u8, frame_hw, tile, tile_pad, through_fp16=False, rank=0):
        """This rank's rows of a uint8 HWC BGR frame -> (rank 0) the whole upscaled uint8 frame on the device, the tiles of
        upstream's grid dealt to the ranks of the communicator (nesr_forward_sharded_u8).  Without comm_init: the one-rank case."""
        self._require_cuda(band_u8)
        b = band_u8.contiguous()
        H, W = int(frame_hw[0]), int(frame_hw[1])
        s = self.out_scale()
        self.calls += 1
        with torch.cuda.device(b.device):
            ctx = self._context(b.device)
            out = torch.empty((H * s, W * s, 3), dtype=torch.uint8, device=b.device) if rank == 0 else None
            stream = torch.cuda.current_stream(b.device).cuda_stream
            _lib.check(_lib.load().nesr_forward_sharded_u8(ctx, ctypes.c_void_p(b.data_ptr()), H, W, int(tile), int(tile_pad), 1 if through_fp16 else 0,
                                                           ctypes.c_void_p(out.data_ptr() if out is not None else 0), ctypes.c_void_p(stream)),
                       "nesr_forward_sharded_u8")
        return out

    # ------------------------------------------------------------------ measurement helpers
    def forward_flops(self, n, h, w):
        """Algorithmic FLOPs of one forward on [n, *, h, w]."""
        macs = 0
        u = {2: 2, 1: 4}.get(self.scale, 1)
        px = n * (h // u) * (w // u)
        nf, gc = self.num_feat, self.num_grow_ch
        rdb = sum(9 * (nf + k * gc) * gc for k in range(4)) + 9 * (nf + 4 * gc) * nf
        macs += 9 * conv_first_in_ch(self.num_in_ch, self.scale) * nf + self.num_block * 3 * rdb + 9 * nf * nf
        macs += 4 * 9 * nf * nf + 16 * 9 * nf * nf * 2 + 16 * 9 * nf * self.num_out_ch
        return 2.0 * macs * px

    def set_kernel_timing(self, device, enable=True):
        handles = [self._context(torch.device(device))] + [h[0] for h in self._extra.values()]
        for ctx in handles:
            _lib.check(_lib.load().nesr_set_kernel_timing(ctx, 1 if enable else 0), "nesr_set_kernel_timing")

    # ---- banded evaluation (banded.py: one row band of the frame per rank, SURVEY.md section 8(e) mode 2) ----
    @property
    def num_rdb(self):
        return 3 * self.num_block

    @property
    def unshuffle(self):
        """Input rows per internal (trunk) row: 2 for scale=2, 4 for scale=1, else 1."""
        return {2: 2, 1: 4}.get(self.scale, 1)

    @torch.no_grad()
    def band_begin(self, x):
        """pixel_unshuffle + conv_first on this rank's rows (band + apron): x [1, num_in_ch, H, W] float32 on a ROCm device."""
        self._require_cuda(x)
        if x.dim() != 4 or x.shape[0] != 1:
            raise ValueError(f"expected [1, C, H, W], got {tuple(x.shape)}")
        xf = x.to(torch.float32).contiguous()
        _, c, h, w = xf.shape
        with torch.cuda.device(xf.device):
            ctx = self._context(xf.device)
            stream = torch.cuda.current_stream(xf.device).cuda_stream
            _lib.check(_lib.load().nesr_band_begin(ctx, ctypes.c_void_p(xf.data_ptr()), c, h, w, ctypes.c_void_p(stream)), "nesr_band_begin")
        self._band = (xf.device, h // self.unshuffle, w // self.unshuffle)

    def _band_call(self):
        if getattr(self, "_band", None) is None or self._ctx is None:
            raise RuntimeError("band_begin has not run")
        dev = self._band[0]
        return self._ctx[0], dev, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    @torch.no_grad()
    def band_rdb(self, index):
        """The five convs of RDB `index` (0 .. num_rdb-1) on the band image."""
        ctx, dev, stream = self._band_call()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nesr_band_rdb(ctx, int(index), stream), "nesr_band_rdb")

    @torch.no_grad()
    def band_rdb_phase(self, index, phase, top, bottom, edge_rows):
        """Phase 0: conv1..conv4 of RDB `index` and conv5 on the `edge_rows` band rows next to each apron (what the
        neighbours wait for); phase 1: conv5 on the rows in between.  Same values as band_rdb."""
        ctx, dev, stream = self._band_call()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nesr_band_rdb_phase(ctx, int(index), int(phase), int(top), int(bottom), int(edge_rows), stream),
                       "nesr_band_rdb_phase")

    def band_row_bytes(self):
        ctx, _, _ = self._band_call()
        return int(_lib.load().nesr_band_row_bytes(ctx))

    @torch.no_grad()
    def band_pack_edges(self, buffer, top, bottom, nrows, top_dst, bottom_dst):
        """The first / last `nrows` BAND rows (the rows the neighbours need) of `buffer` -> two preallocated uint8 tensors
        (either may be None), in one C-ABI call."""
        ctx, dev, stream = self._band_call()
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()   # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nesr_band_pack_edges(ctx, int(buffer), int(top), int(bottom), int(nrows), ptr(top_dst), ptr(bottom_dst), stream),
                       "nesr_band_pack_edges")

    @torch.no_grad()
    def band_unpack_aprons(self, buffer, top, bottom, nrows, top_src, bottom_src):
        """The neighbours' rows -> the `nrows` apron rows next to the band on each side (either source may be None)."""
        ctx, dev, stream = self._band_call()
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()   # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nesr_band_unpack_aprons(ctx, int(buffer), int(top), int(bottom), int(nrows), ptr(top_src), ptr(bottom_src), stream),
                       "nesr_band_unpack_aprons")

    @torch.no_grad()
    def band_tail(self):
        """conv_body .. conv_last -> [1, num_out_ch, 4 h, 4 w] float32 (h, w = internal size of the band image)."""
        ctx, dev, stream = self._band_call()
        _, h, w = self._band
        with torch.cuda.device(dev):
            y = torch.empty((1, self.num_out_ch, 4 * h, 4 * w), dtype=torch.float32, device=dev)
            _lib.check(_lib.load().nesr_band_tail(ctx, ctypes.c_void_p(y.data_ptr()), stream), "nesr_band_tail")
        return y

    @torch.no_grad()
    def band_rows(self, buffer, row0, nrows):
        """Internal rows [row0, row0+nrows) of the num_feat-channel slice of `buffer` (0..2 dense-block buffers,
        3 = conv_first output) as an opaque uint8 tensor (the context's own element layout)."""
        ctx, dev, stream = self._band_call()
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty(int(nrows) * int(lib.nesr_band_row_bytes(ctx)), dtype=torch.uint8, device=dev)
            _lib.check(lib.nesr_band_rows(ctx, int(buffer), int(row0), int(nrows), ctypes.c_void_p(out.data_ptr()), 0, stream), "nesr_band_rows")
        return out

    @torch.no_grad()
    def band_set_rows(self, buffer, row0, rows):
        """Inverse of band_rows: overwrites the rows with another rank's band_rows() bytes."""
        ctx, dev, stream = self._band_call()
        lib = _lib.load()
        rb = int(lib.nesr_band_row_bytes(ctx))
        if rb == 0:
            raise RuntimeError("band_set_rows: no banded evaluation is active (band_begin has not run, or a whole-frame forward reused the workspace)")
        rows = rows.to(dev).contiguous()
        if rows.dtype != torch.uint8 or rows.numel() % rb:
            raise ValueError("rows must be the uint8 tensor band_rows() returned on the sending rank")
        with torch.cuda.device(dev):
            _lib.check(lib.nesr_band_rows(ctx, int(buffer), int(row0), rows.numel() // rb, ctypes.c_void_p(rows.data_ptr()), 1, stream), "nesr_band_rows")

    def _handles(self):
        return ([self._ctx] if self._ctx is not None else []) + list(self._extra.values())

    def set_fused(self, on: bool):
        """Persistent (fused) dense-block launches on / off for every context of this model (include/nesr_hip.h: nesr_set_fused).
        They switch themselves off after a forward that gave up waiting (NesrHipError at check_range / check_status)."""
        for h in self._handles():
            _lib.check(_lib.load().nesr_set_fused(h[0], 1 if on else 0), "nesr_set_fused")

    def fused_state(self, slot=0):
        """(persistent launches enabled, forwards that gave up so far) of a context."""
        h = self._ctx if slot == 0 else self._extra.get(slot)
        if h is None:
            return False, 0
        v = int(_lib.load().nesr_fused_state(h[0]))
        return bool(v & 1), v >> 1

    def debug_fault(self, drop_workgroups=1, slot=0):
        """TEST HOOK: the next persistent launch of the context leaves out its last workgroups (nesr_debug_fault)."""
        h = self._ctx if slot == 0 else self._extra.get(slot)
        _lib.check(_lib.load().nesr_debug_fault(h[0], int(drop_workgroups)), "nesr_debug_fault")

    def set_concurrent(self, concurrent: bool):
        """Hint for kernel selection: forwards of this model's contexts run beside each other on several streams
        (set automatically when a context replica is created; clear it to time one forward alone)."""
        lib = _lib.load()
        for h in ([self._ctx] if self._ctx is not None else []) + list(self._extra.values()):
            _lib.check(lib.nesr_set_concurrent(h[0], 1 if concurrent else 0), "nesr_set_concurrent")

    def preferred_batch(self, device, h, w, max_batch):
        """Tiles of h x w input per forward call that fill the GPU's CUs most evenly (<= max_batch)."""
        ctx = self._context(torch.device(device))
        return max(1, int(_lib.load().nesr_preferred_batch(ctx, h, w, max_batch)))

    def check_status(self):
        """Synchronises the device and raises if asynchronous work of this model failed."""
        if self._ctx is not None:
            _lib.check(_lib.load().nesr_check_status(self._ctx[0]), "nesr_check_status")
        self.check_range()

    def check_range(self, slot=None):
        """Raises NesrRangeError if a forward enqueued so far (on torch's current stream) met an input or activation
        the f16-pair fp32 form cannot carry (non-finite or beyond +-65504): its float output is NaN and an 8-bit
        output is invalid.  Waits for the current stream only; a no-op for the other compute dtypes.  The wrappers
        call it after every device-to-host copy (the reference would have returned NaN pixels, nesr/nesr.py:891-898)."""
        handles = ([self._ctx] if self._ctx is not None else []) + list(self._extra.values()) if slot is None else \
                  [self._ctx if slot == 0 else self._extra.get(slot)]
        lib = _lib.load()
        for h in handles:
            if h is None or h[2] not in (_lib.DTYPE_F32_SPLIT, _lib.DTYPE_BF16):     # the forms with a range word or persistent launches
                continue
            dev = torch.device("cuda", h[1])
            with torch.cuda.device(dev):
                _lib.check(lib.nesr_check_range(h[0], ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "nesr_check_range")

    def kernel_time(self):
        """(total ms, launches, algorithmic flops) of the dense-block convs since the last call."""
        if self._ctx is None:
            return 0.0, 0, 0.0
        tot_ms, tot_n, tot_fl = 0.0, 0, 0.0
        # replicas run on concurrent streams: their brackets overlap in wall time, so the sum of the
        # bracketed times is an upper bound of the busy time (the derived TFLOP/s a lower bound)
        for h in [self._ctx] + list(self._extra.values()):
            ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            _lib.check(_lib.load().nesr_kernel_time_ms(h[0], ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)),
                       "nesr_kernel_time_ms")
            tot_ms, tot_n, tot_fl = tot_ms + ms.value, tot_n + n.value, tot_fl + fl.value
        return tot_ms, tot_n, tot_fl


def cut_tiles_u8(frame_u8, windows, slot_hw, flip_rgb=True, through_fp16=False):
    """All tiles of a frame in one launch: u8 HWC [H, W, 3] device tensor -> float32 [n, 3, Hs, Ws], tile i =
    frame[y0:y0+h, x0:x0+w] / 255 (BGR -> RGB if flip_rgb) in the top-left of slot i, zeros elsewhere -- what
    RealESRGANer.enhance + tile_process feed the network (include/nesr_hip.h: nesr_cut_tiles_u8).  windows: [(y0, x0, h, w)]."""
    if frame_u8.device.type != "cuda" or frame_u8.dtype != torch.uint8 or frame_u8.dim() != 3 or frame_u8.shape[2] != 3:
        raise ValueError("cut_tiles_u8: a uint8 [H, W, 3] tensor on the ROCm device")
    f = frame_u8.contiguous()
    n = len(windows)
    hs, ws = int(slot_hw[0]), int(slot_hw[1])
    x = torch.empty((n, 3, hs, ws), dtype=torch.float32, device=f.device)
    arr = (ctypes.c_int * (4 * n))(*[int(v) for win in windows for v in win])
    index = f.device.index if f.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(f.device):
        stream = torch.cuda.current_stream(f.device).cuda_stream
        _lib.check(_lib.load().nesr_cut_tiles_u8(index, ctypes.c_void_p(f.data_ptr()), f.shape[0], f.shape[1], 1 if flip_rgb else 0,
                                                 1 if through_fp16 else 0, arr, n, hs, ws, ctypes.c_void_p(x.data_ptr()),
                                                 ctypes.c_void_p(stream)), "nesr_cut_tiles_u8")
    return x


def paste_tiles_u8(tiles, descs, dst_u8, flip_rgb=True, round_nearest=True, through_fp16=False):
    """The un-padded centres of all tiles in one launch: tiles float32 [n, 3, Hs, Ws] (network outputs in their slots) ->
    clamp(0, 1), RGB -> BGR, x255, round, into the uint8 device tensor dst_u8 (a frame's output canvas [H, W, 3], or any flat
    buffer).  descs: [(crop_y, crop_x, h, w, dst_byte_offset, dst_row_pitch_bytes)] (nesr_paste_tiles_u8)."""
    if tiles.device.type != "cuda" or tiles.dtype != torch.float32 or tiles.dim() != 4 or tiles.shape[1] != 3 or not tiles.is_contiguous():
        raise ValueError("paste_tiles_u8: a contiguous float32 [n, 3, Hs, Ws] tensor on the ROCm device")
    if dst_u8.dtype != torch.uint8 or not dst_u8.is_contiguous() or dst_u8.device != tiles.device:
        raise ValueError("paste_tiles_u8: a contiguous uint8 destination on the same device")
    n = tiles.shape[0]
    arr = (ctypes.c_int64 * (6 * n))(*[int(v) for d in descs for v in d])
    index = tiles.device.index if tiles.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(tiles.device):
        stream = torch.cuda.current_stream(tiles.device).cuda_stream
        _lib.check(_lib.load().nesr_paste_tiles_u8(index, ctypes.c_void_p(tiles.data_ptr()), n, tiles.shape[2], tiles.shape[3], arr,
                                                   ctypes.c_void_p(dst_u8.data_ptr()), dst_u8.numel(), 1 if flip_rgb else 0,
                                                   _lib.ROUND_NEAREST if round_nearest else _lib.ROUND_TRUNC, 1 if through_fp16 else 0,
                                                   ctypes.c_void_p(stream)),
                   "nesr_paste_tiles_u8")


def conv3x3(x, weight, bias, lrelu=False, upsample=False, dtype="f32"):
    """Single fused layer through the C ABI (test hook): conv3x3(pad 1) + bias [+ LeakyReLU(0.2)],
    optionally on the nearest-x2 upsample of x.  x NCHW float32 on a ROCm device."""
    if x.device.type != "cuda":
        raise RuntimeError("conv3x3 runs only on an AMD GPU through libnesr_hip.so (no CPU fallback)")
    lib = _lib.load()
    x = x.to(torch.float32).contiguous()
    n, cin, h, w = x.shape
    wt = weight.detach().to("cpu", torch.float32).contiguous()
    bs = bias.detach().to("cpu", torch.float32).contiguous()
    cout = wt.shape[0]
    up = 1 if upsample else 0
    y = torch.empty((n, cout, h << up, w << up), dtype=torch.float32, device=x.device)
    index = x.device.index if x.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        # "f32" is what RRDBNet(compute_dtype="f32") runs: the f16-pair kernel
        code = {"bf16": _lib.DTYPE_BF16, "f32-winograd": _lib.DTYPE_F32_WINOGRAD, "f32": _lib.DTYPE_F32_SPLIT, "f32-split": _lib.DTYPE_F32_SPLIT,
                "f32-direct": _lib.DTYPE_F32}[dtype]
        _lib.check(lib.nesr_conv3x3(index, code,
                                    ctypes.c_void_p(x.data_ptr()), n, cin, h, w, ctypes.c_void_p(wt.data_ptr()),
                                    ctypes.c_void_p(bs.data_ptr()), cout, 1 if lrelu else 0, up,
                                    ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(stream)), "nesr_conv3x3")
    return y
