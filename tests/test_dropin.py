"""CPU: the reference's own import lines and probes resolve to the drop-in classes
(nesr/nesr.py:153-162, standalone/direct_esrgan.py:92-93)."""
import importlib
import importlib.util
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def dropin_path(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "dropin"))
    for m in [m for m in sys.modules if m.split(".")[0] in ("basicsr", "realesrgan")]:
        monkeypatch.delitem(sys.modules, m)
    importlib.invalidate_caches()


def test_find_spec_and_imports(dropin_path):
    assert importlib.util.find_spec("basicsr") is not None          # nesr/nesr.py:153
    assert importlib.util.find_spec("realesrgan") is not None       # nesr/nesr.py:157
    from basicsr.archs.rrdbnet_arch import RRDBNet                   # nesr/nesr.py:161
    from realesrgan import RealESRGANer                              # nesr/nesr.py:162
    import neural_enhanced_super_resolution_amd as pkg
    assert RRDBNet is pkg.RRDBNet and RealESRGANer is pkg.RealESRGANer


def test_reference_constructor_calls(dropin_path):
    from basicsr.archs.rrdbnet_arch import RRDBNet
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    from realesrgan import RealESRGANer
    # nesr/nesr.py:216-229 verbatim argument lists (2 blocks to keep the test light)
    model = RRDBNet(num_in_ch=12, num_out_ch=3, num_feat=64, num_block=2, num_grow_ch=32)
    sd = synthetic_state_dict(seed=0, num_in_ch=12, scale=4, num_block=2)
    up = RealESRGANer(scale=int(2.0), model_path={"params_ema": sd}, model=model, tile=0, tile_pad=0, pre_pad=0,
                      half=False, device="cpu")
    assert up.scale == 2 and up.tile_size == 0 and up.tile_pad == 0 and up.pre_pad == 0 and up.half is False
    assert up.mod_scale is None
    m = up.model
    m.eval()                                                         # nesr/nesr.py:888
    assert next(m.parameters()).device.type == "cpu"                 # nesr/nesr.py:962
    assert m.to("cpu") is m                                          # nesr/nesr.py:963
    assert m.out_scale() == 4                                        # the 12-ch quirk is a 4x network


def test_state_dict_layout_matches_upstream_names():
    from neural_enhanced_super_resolution_amd import RRDBNet, rrdbnet_state_dict_spec
    from oracle.rrdbnet_ref import state_dict_spec
    net = RRDBNet(3, 3, scale=2)
    sd = net.state_dict()
    assert list(sd) == list(rrdbnet_state_dict_spec(3, 3, 2)) == list(state_dict_spec(3, 3, 2))
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(state_dict_spec(3, 3, 2))
    assert len(sd) == 702 and sum(v.numel() for v in sd.values()) == 16_703_171
    assert net.forward_flops(1, 512, 512) == 2 * 17_932_032 * 256 * 256


def test_synthetic_weights_are_reproducible():
    import hashlib
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    a = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1)
    b = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1)
    assert all(torch.equal(a[k], b[k]) for k in a)
    h = hashlib.sha256(a["body.0.rdb1.conv5.weight"].numpy().tobytes()).hexdigest()
    assert h == hashlib.sha256(b["body.0.rdb1.conv5.weight"].numpy().tobytes()).hexdigest()
    f = synthetic_frame(32, 48, seed=2)
    assert f.shape == (32, 48, 3) and f.dtype.name == "uint8" and 20 < f.std() < 90
