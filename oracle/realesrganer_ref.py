"""Oracle (test infrastructure): torch-CPU restatement of realesrgan 0.3.0 ``RealESRGANer``.

PARITY UNPINNED (see oracle/__init__.py): realesrgan is not vendored in the reference.
Restates upstream ``realesrgan/utils.py`` (pre_process / process / tile_process /
post_process / enhance) as called by the reference at

  standalone/direct_esrgan.py:118-127,148   RealESRGANer(scale, model_path, model, tile=512,
                                            tile_pad=10, pre_pad=0, half=False, device) ; .enhance(img)
  nesr/nesr.py:220-229                      RealESRGANer(scale=int(upscale_factor), ..., tile=0,
                                            tile_pad=0, pre_pad=0, half=False, device)
  standalone/superres_project.py:70-75,282  ctor defaults ; .enhance(bgr)

cv2 is absent from this image; the three colour conversions upstream uses
(COLOR_BGR2RGB, COLOR_GRAY2RGB, COLOR_BGR2GRAY on float32) are restated in numpy.
``outscale`` (cv2.resize INTER_LANCZOS4) and ``alpha_upsampler != 'realesrgan'``
(cv2.resize INTER_LINEAR) are not restated: they raise.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def bgr2gray_f32(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(float32 BGR, COLOR_BGR2GRAY): 0.114 B + 0.587 G + 0.299 R."""
    return (img[..., 0] * np.float32(0.114) + img[..., 1] * np.float32(0.587) + img[..., 2] * np.float32(0.299)).astype(np.float32)


class RealESRGANerRef:
    """Upstream RealESRGANer, CPU only.  ``model`` is any nn.Module with upstream's
    RRDBNet state_dict keys (oracle.rrdbnet_ref.RRDBNetRef).  ``model_path`` may be a path
    to a torch checkpoint ({'params_ema'|'params': state_dict}) or such a dict itself."""

    def __init__(self, scale, model_path, dni_weight=None, model=None, tile=0, tile_pad=10,
                 pre_pad=10, half=False, device=None, gpu_id=None):
        self.scale = scale
        self.tile_size = tile
        self.tile_pad = tile_pad
        self.pre_pad = pre_pad
        self.mod_scale = None
        self.half = half
        self.device = torch.device("cpu")
        if isinstance(model_path, list):
            assert len(model_path) == len(dni_weight), "model_path and dni_weight should have the save length."
            loadnet = self.dni(model_path[0], model_path[1], dni_weight)
        elif isinstance(model_path, dict):
            loadnet = model_path
        else:
            loadnet = torch.load(model_path, map_location=torch.device("cpu"), weights_only=True)
        keyname = "params_ema" if "params_ema" in loadnet else "params"
        model.load_state_dict(loadnet[keyname], strict=True)
        model.eval()
        self.model = model.to(self.device)

    def dni(self, net_a, net_b, dni_weight, key="params", loc="cpu"):
        """Upstream deep-network-interpolation of two checkpoints."""
        if not isinstance(net_a, dict):
            net_a = torch.load(net_a, map_location=torch.device(loc), weights_only=True)
        if not isinstance(net_b, dict):
            net_b = torch.load(net_b, map_location=torch.device(loc), weights_only=True)
        for k, v_a in net_a[key].items():
            net_a[key][k] = dni_weight[0] * v_a + dni_weight[1] * net_b[key][k]
        return net_a

    def pre_process(self, img):
        img = torch.from_numpy(np.transpose(img, (2, 0, 1))).float()
        self.img = img.unsqueeze(0).to(self.device)
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

    def tile_process(self):
        batch, channel, height, width = self.img.shape
        output_shape = (batch, channel, height * self.scale, width * self.scale)
        self.output = self.img.new_zeros(output_shape)
        tiles_x = math.ceil(width / self.tile_size)
        tiles_y = math.ceil(height / self.tile_size)
        for y in range(tiles_y):
            for x in range(tiles_x):
                ofs_x = x * self.tile_size
                ofs_y = y * self.tile_size
                input_start_x = ofs_x
                input_end_x = min(ofs_x + self.tile_size, width)
                input_start_y = ofs_y
                input_end_y = min(ofs_y + self.tile_size, height)
                input_start_x_pad = max(input_start_x - self.tile_pad, 0)
                input_end_x_pad = min(input_end_x + self.tile_pad, width)
                input_start_y_pad = max(input_start_y - self.tile_pad, 0)
                input_end_y_pad = min(input_end_y + self.tile_pad, height)
                input_tile_width = input_end_x - input_start_x
                input_tile_height = input_end_y - input_start_y
                input_tile = self.img[:, :, input_start_y_pad:input_end_y_pad, input_start_x_pad:input_end_x_pad]
                with torch.no_grad():
                    output_tile = self.model(input_tile)
                output_start_x = input_start_x * self.scale
                output_end_x = input_end_x * self.scale
                output_start_y = input_start_y * self.scale
                output_end_y = input_end_y * self.scale
                output_start_x_tile = (input_start_x - input_start_x_pad) * self.scale
                output_end_x_tile = output_start_x_tile + input_tile_width * self.scale
                output_start_y_tile = (input_start_y - input_start_y_pad) * self.scale
                output_end_y_tile = output_start_y_tile + input_tile_height * self.scale
                self.output[:, :, output_start_y:output_end_y, output_start_x:output_end_x] = \
                    output_tile[:, :, output_start_y_tile:output_end_y_tile, output_start_x_tile:output_end_x_tile]

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

    @torch.no_grad()
    def enhance_float(self, img, alpha_upsampler="realesrgan"):
        """The float image upstream enhance() holds just before quantisation (HWC, BGR order,
        clamped to [0,1]) -- what the 1e-3 criterion is measured on."""
        img = img.astype(np.float32)
        max_range = 65535 if np.max(img) > 256 else 255
        img = img / max_range
        if img.ndim == 2:
            img_mode = "L"
            img = np.repeat(img[:, :, None], 3, axis=2)          # COLOR_GRAY2RGB
        elif img.shape[2] == 4:
            img_mode = "RGBA"
            alpha = img[:, :, 3]
            img = img[:, :, 0:3][:, :, ::-1]                      # COLOR_BGR2RGB
            if alpha_upsampler == "realesrgan":
                alpha = np.repeat(alpha[:, :, None], 3, axis=2)   # COLOR_GRAY2RGB
        else:
            img_mode = "RGB"
            img = img[:, :, ::-1]                                 # COLOR_BGR2RGB
        self.pre_process(np.ascontiguousarray(img))
        out = self._run().data.squeeze().float().cpu().clamp_(0, 1).numpy()
        out = np.transpose(out[[2, 1, 0], :, :], (1, 2, 0))
        if img_mode == "L":
            out = bgr2gray_f32(out)
        if img_mode == "RGBA" and alpha_upsampler != "realesrgan":
            from oracle import cv2_ref
            h, w = alpha.shape[0:2]
            oa = cv2_ref.resize_linear_f32(np.ascontiguousarray(alpha), h * self.scale, w * self.scale)   # cv2.resize(alpha, ..., INTER_LINEAR)
            out = np.concatenate([out, oa[:, :, None]], axis=2)
        elif img_mode == "RGBA":
            self.pre_process(np.ascontiguousarray(alpha))
            oa = self._run().data.squeeze().float().cpu().clamp_(0, 1).numpy()
            oa = bgr2gray_f32(np.transpose(oa[[2, 1, 0], :, :], (1, 2, 0)))
            out = np.concatenate([out, oa[:, :, None]], axis=2)   # COLOR_BGR2BGRA then [:,:,3]=alpha
        return out, img_mode, max_range

    @torch.no_grad()
    def enhance(self, img, outscale=None, alpha_upsampler="realesrgan"):
        h_input, w_input = img.shape[0:2]
        out, img_mode, max_range = self.enhance_float(img, alpha_upsampler)
        if max_range == 65535:
            output = (out * 65535.0).round().astype(np.uint16)
        else:
            output = (out * 255.0).round().astype(np.uint8)
        if outscale is not None and outscale != float(self.scale):
            from oracle import cv2_ref                         # cv2.resize(..., INTER_LANCZOS4), restated: parity unpinned
            o3 = output[:, :, None] if output.ndim == 2 else output
            r = cv2_ref.resize_lanczos4(np.ascontiguousarray(o3), int(h_input * outscale), int(w_input * outscale))
            output = r[:, :, 0] if output.ndim == 2 else r
        return output, img_mode
