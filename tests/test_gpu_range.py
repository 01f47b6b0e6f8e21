"""GPU: what the default fp32 form (operands as f16 (hi, lo) pairs, conv3x3_f16x2.hip) does outside the comfortable
range of the bench weights -- every case is either inside the north-star tolerance (1e-3 max abs on the [0,1]
image, BASELINE.json) against the torch-CPU oracle, or a loud error; never a silently saturated image.

The pair holds x = hi + lo * 2^-11 with hi = f16(x), lo = f16((x - hi) * 2^11): 22 significant bits for
6.1e-5 <= |x| <= 65504, an absolute 2^-35 below that, nothing above it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _nets(sd, num_block, compute_dtype="f32", num_in_ch=3, scale=2):
    from neural_enhanced_super_resolution_amd import RRDBNet
    from oracle.rrdbnet_ref import RRDBNetRef
    ours = RRDBNet(num_in_ch, 3, scale=scale, num_block=num_block, compute_dtype=compute_dtype)
    ours.load_state_dict(sd, strict=True)
    ours.eval().to("cuda:0")
    ref = RRDBNetRef(num_in_ch, 3, scale=scale, num_block=num_block)
    ref.load_state_dict(sd, strict=True)
    return ours, ref


def _x(h=48, w=64, seed=1):
    return torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(seed))


def _scaled_trunk(sd, gain):
    """Same function of the input up to the biases: conv_first x gain, conv_last / gain (weights only), so the
    activations between them are `gain` times larger while the image stays O(1)."""
    sd = {k: v.clone() for k, v in sd.items()}
    sd["conv_first.weight"] *= gain
    sd["conv_first.bias"] *= gain
    sd["conv_last.weight"] /= gain
    return sd


@pytest.mark.parametrize("gain", [1e2, 1e-4])
def test_activations_scaled_full_depth(cuda_device, gain):
    """Trunk activations x100 (thousands: the upper part of the f16 range) and x1e-4 (hi halves subnormal)."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = _scaled_trunk(synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23), gain)
    ours, ref = _nets(sd, 23)
    x = _x()
    want = ref(x)
    got = ours(x.to(cuda_device)).cpu()
    ours.check_range()
    err = (got - want).abs().max().item()
    print(f"trunk activations x{gain:g}: max abs err {err:.3e} (output range {want.min().item():.2f} .. {want.max().item():.2f})")
    assert err < TOL, err


def test_activations_x1000_overflow_is_reported_not_saturated(cuda_device):
    """x1000 pushes the 23-block trunk of the bench weights past 65504: that must be an error (and NaN), and the
    f32 matrix-core forms must carry the same data inside the tolerance (relative to the image range)."""
    from neural_enhanced_super_resolution_amd._lib import NesrRangeError
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = _scaled_trunk(synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23), 1e3)
    ours, ref = _nets(sd, 23)
    x = _x()
    want = ref(x)
    got = ours(x.to(cuda_device)).cpu()
    with pytest.raises(NesrRangeError):
        ours.check_range()
    assert torch.isnan(got).all()
    wino, _ = _nets(sd, 23, compute_dtype="f32-winograd")
    err = (wino(x.to(cuda_device)).cpu() - want).abs().max().item()
    assert err < TOL * max(1.0, want.abs().max().item()), err


def test_heavy_tailed_weights_full_depth(cuda_device):
    """Student-t (3 degrees of freedom) dense-block weights at the bench weights' scale: a few weights are 10-50x the
    typical one, so single products dominate their sums."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    rng = np.random.default_rng(5)
    for k, v in sd.items():
        if k.startswith("body.") and k.endswith(".weight"):
            t = rng.standard_t(3, size=tuple(v.shape)).astype(np.float32) / np.float32(np.sqrt(3.0))   # unit variance
            sd[k] = torch.from_numpy(t) * v.std()
    ours, ref = _nets(sd, 23)
    x = _x(seed=2)
    want = ref(x)
    got = ours(x.to(cuda_device)).cpu()
    ours.check_range()
    err = (got - want).abs().max().item()
    print(f"heavy-tailed weights: max abs err {err:.3e}, max |w| {max(v.abs().max().item() for v in sd.values()):.2f}")
    assert err < TOL, err


def test_tiny_weights_keep_relative_precision(cuda_device):
    """Weights mostly below 1e-4 (lo halves would be f16 subnormals without the 2^11 scale: an absolute 3e-8 floor,
    i.e. 1e-3 relative at |w| ~ 3e-5).  One layer, error relative to the result's magnitude."""
    from neural_enhanced_super_resolution_amd import conv3x3
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 64, 24, 40, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) * 3e-5
    b = torch.zeros(32)
    assert (w.abs() < 1e-4).float().mean().item() > 0.99
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    got = conv3x3(x.to(cuda_device), w, b, dtype="f32").cpu().double()
    rel = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"tiny weights: max error / max|result| = {rel:.3e}")
    assert rel < 2e-6, rel
    # and tiny activations against ordinary weights
    x2 = x * 1e-5
    w2 = torch.randn(32, 64, 3, 3, generator=g) * 0.05
    ref2 = F.conv2d(x2.double(), w2.double(), b.double(), padding=1)
    got2 = conv3x3(x2.to(cuda_device), w2, b, dtype="f32").cpu().double()
    rel2 = ((got2 - ref2).abs().max() / ref2.abs().max()).item()
    # |x| ~ 5e-6 is below the f16 normal range: the pair resolves 2^-35 = 2.9e-11 absolute there, i.e. ~6e-6 of such a
    # value (without the scaled lo it would be 6e-8 absolute: 1e-2 relative)
    print(f"tiny activations: max error / max|result| = {rel2:.3e}")
    assert rel2 < 1e-5, rel2


def test_tiny_weights_full_depth(cuda_device):
    """Whole network with every dense-block weight scaled to ~1e-5 .. 1e-4."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    for k in sd:
        if k.startswith("body.") and k.endswith(".weight"):
            sd[k] = sd[k] * 2e-3
    assert np.mean([(v.abs() < 1e-4).float().mean().item() for k, v in sd.items() if k.startswith("body.") and k.endswith(".weight")]) > 0.9
    ours, ref = _nets(sd, 23)
    x = _x(seed=4)
    err = (ours(x.to(cuda_device)).cpu() - ref(x)).abs().max().item()
    assert err < 2e-5, err


def test_activation_overflow_is_loud(cuda_device):
    """Activations beyond 65504 cannot be carried: the float output is NaN (not a saturated picture) and the next
    range check raises; the 8-bit path raises out of enhance().  The strict-f32 forms compute the same data."""
    from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet
    from neural_enhanced_super_resolution_amd._lib import NesrRangeError
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    base = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=2)
    sd = _scaled_trunk(base, 3e5)                            # conv_first weights stay below 65504, its outputs do not
    ours, ref = _nets(sd, 2)
    x = _x()
    got = ours(x.to(cuda_device)).cpu()
    assert torch.isnan(got).all(), "an out-of-range forward must not return a plausible image"
    with pytest.raises(NesrRangeError):
        ours.check_range()
    ours.check_range()                                       # reported once; the flag is cleared
    ok = ours(x.to(cuda_device) * 1e-9).cpu()                # the context recovers: the next in-range forward is clean
    ours.check_range()
    assert torch.isfinite(ok).all()
    # the same weights on the strict-f32 kernels are inside the tolerance
    direct, _ = _nets(sd, 2, compute_dtype="f32-direct")
    want = ref(x)
    assert (direct(x.to(cuda_device)).cpu() - want).abs().max().item() < TOL * max(1.0, want.abs().max().item())
    # 8-bit wrapper path
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), tile=0, pre_pad=0,
                      half=False, device=cuda_device)
    with pytest.raises(NesrRangeError):
        up.enhance(synthetic_frame(32, 48, seed=1))
    with pytest.raises(NesrRangeError):
        up.enhance(synthetic_frame(33, 47, seed=1))          # padded frame: the device-side float path


def test_range_flag_is_scoped_to_one_forward(cuda_device):
    """A caller that uses model(x) directly (as nesr/nesr.py:887-891 does with upsampler.model) and never asks: an out-of-range
    frame must not turn every later, valid frame of that context into NaN; the unreported condition is latched and the next
    check reports it once, naming it as an earlier forward's."""
    from neural_enhanced_super_resolution_amd._lib import NesrRangeError
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = _scaled_trunk(synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=2), 3e5)
    ours, _ = _nets(sd, 2)
    x = _x().to(cuda_device)
    assert torch.isnan(ours(x).cpu()).all()                  # out of range, never checked
    ok = ours(x * 1e-9).cpu()                                # a valid frame on the same context
    assert torch.isfinite(ok).all(), "an earlier frame's range condition leaked into a valid frame"
    with pytest.raises(NesrRangeError, match="EARLIER forward"):
        ours.check_range()
    ours.check_range()                                       # reported once
    assert torch.isnan(ours(x).cpu()).all()
    with pytest.raises(NesrRangeError) as e:
        ours.check_range()                                   # the latest forward's own condition
    assert "EARLIER" not in str(e.value)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 1e5])
def test_bad_input_is_loud(cuda_device, bad):
    """The reference propagates a NaN/Inf input to NaN pixels (model(img), nesr/nesr.py:891); so does this path,
    and it says so."""
    from neural_enhanced_super_resolution_amd._lib import NesrRangeError
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    ours, _ = _nets(synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1), 1)
    x = _x()
    x[0, 1, 7, 9] = bad
    got = ours(x.to(cuda_device)).cpu()
    assert torch.isnan(got).all()
    with pytest.raises(NesrRangeError):
        ours.check_status()


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 7e4])
def test_bad_weights_are_refused(cuda_device, bad):
    from neural_enhanced_super_resolution_amd import RRDBNet
    from neural_enhanced_super_resolution_amd._lib import NesrRangeError
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1)
    sd["body.0.rdb2.conv3.weight"][5, 17, 1, 2] = bad
    net = RRDBNet(3, 3, scale=2, num_block=1)
    net.load_state_dict(sd)
    net.to(cuda_device)
    with pytest.raises(NesrRangeError, match="body.0.rdb2.conv3.weight"):
        net(_x().to(cuda_device))
    if np.isfinite(bad):                                     # a large finite weight is fine for the f32 matrix-core forms
        net2 = RRDBNet(3, 3, scale=2, num_block=1, compute_dtype="f32-direct")
        net2.load_state_dict(sd)
        net2.to(cuda_device)
        assert torch.isfinite(net2(_x().to(cuda_device))).all()
