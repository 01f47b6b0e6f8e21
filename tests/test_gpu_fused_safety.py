"""GPU: the persistent dense-block kernels (rdb_f16x2_kernel: small f32 frames; rdb_bf16_strip_kernel: bf16 tile batches) beside
other work and under a fault.  Their workgroups wait for one another, so (a) two contexts that launch them on two streams are
serialised per device and both get the per-layer values, (b) a launch whose workgroups are not all there (test hook
nesr_debug_fault: the last workgroup never starts) gives up within the wall-clock bound, raises NESR_ERR_HIP at the next status
check instead of hanging or handing back a wrong image, switches the context to per-layer launches, and the next forward is
right; (c) RealESRGANer.enhance evaluates such a frame again by itself.

Reference call behind all of it: `self.model(img)` in RealESRGANer.process / tile_process (standalone/direct_esrgan.py:148) and
nesr/nesr.py:887-891."""
import os
import time
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(sd, dtype, nb, env):
    from neural_enhanced_super_resolution_amd import RRDBNet
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        n = RRDBNet(3, 3, scale=2, num_block=nb, compute_dtype=dtype)
        n.load_state_dict(sd)
        n.eval().to("cuda:0")
        n(torch.zeros(1, 3, 16, 16, device="cuda:0"))
        n.check_status()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return n


def test_two_models_on_two_streams_share_the_device(cuda_device):
    """f32, 512 x 512 frames (256 tiles = every CU): two contexts, two streams, interleaved forwards."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=3)
    ref = _net(sd, "f32", 3, {"NESR_RDB_FUSE": "0"})
    a, b = _net(sd, "f32", 3, {"NESR_RDB_FUSE": "-1"}), _net(sd, "f32", 3, {"NESR_RDB_FUSE": "-1"})
    xs = [torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(i)).to(cuda_device) for i in range(2)]
    want = [ref(x) for x in xs]
    s1, s2 = torch.cuda.Stream(cuda_device), torch.cuda.Stream(cuda_device)
    torch.cuda.synchronize()
    outs = []
    for _ in range(6):
        with torch.cuda.stream(s1):
            ya = a(xs[0])
        with torch.cuda.stream(s2):
            yb = b(xs[1])
        outs.append((ya, yb))
    torch.cuda.synchronize()
    a.check_status()
    b.check_status()
    for ya, yb in outs:
        assert torch.equal(ya, want[0]) and torch.equal(yb, want[1])
    assert a.fused_state() == (True, 0) and b.fused_state() == (True, 0)


@pytest.mark.parametrize("dtype,hw,n", [("f32", (512, 512), 1), ("bf16", (120, 200), 6)])
def test_a_launch_with_a_missing_workgroup_gives_up_and_the_context_recovers(cuda_device, dtype, hw, n):
    from neural_enhanced_super_resolution_amd._lib import NesrHipError
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=2)
    env = {"NESR_FUSED_TIMEOUT_MS": "20", "NESR_STRIP": "1", "NESR_RDB_FUSE": "-1"}
    net = _net(sd, dtype, 2, env)
    per_layer = _net(sd, dtype, 2, {"NESR_STRIP": "0", "NESR_RDB_FUSE": "0"})
    x = torch.rand(n, 3, *hw, generator=torch.Generator().manual_seed(3)).to(cuda_device)
    good = net(x)
    net.check_status()
    assert net.fused_state() == (True, 0)
    net.debug_fault(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net(x)                                     # its first dense block runs without its last workgroup
    with pytest.raises(NesrHipError, match="gave up waiting"):
        net.check_status()
    dt = time.perf_counter() - t0
    assert dt < 2.0, dt                        # bounded: the 20 ms limit once, every later wait of the forward skipped
    on, aborts = net.fused_state()
    assert not on and aborts == 1              # per-layer launches from now on
    again = net(x)
    net.check_status()
    assert torch.equal(again, per_layer(x))    # ... with the per-layer kernels' values
    if dtype == "f32":
        assert torch.equal(again, good)        # (f32: those are the fused kernel's bits too)
    net.set_fused(True)
    back = net(x)
    net.check_status()
    assert torch.equal(back, good)


def test_enhance_evaluates_a_frame_again_after_a_launch_gave_up(cuda_device):
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    os.environ["NESR_FUSED_TIMEOUT_MS"] = "20"
    try:
        sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=2)
        up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), tile=0, tile_pad=10,
                          pre_pad=0, half=False, device="cuda:0")
        img = synthetic_frame(512, 512, seed=4)
        want, _ = up.enhance(img)
        up.model.debug_fault(1)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got, mode = up.enhance(img)
        assert any("per-layer launches" in str(x.message) for x in w)
        assert mode == "RGB" and np.array_equal(got, want)
        assert up.model.fused_state() == (False, 1)
    finally:
        os.environ.pop("NESR_FUSED_TIMEOUT_MS", None)
