"""Per-layer parity of the HIP 3x3 conv (through the C ABI entry nesr_conv3x3) against torch CPU
ops -- the primitives the oracle is made of.  Shapes: every (Cin, Cout) RRDBNet uses
(SURVEY.md section 8(c) golden item 1)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(3, 64), (12, 64), (64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64), (64, 3)]
F32_TOL = 2e-5   # f32 MFMA is a k-ordered fmaf chain; oneDNN sums in another order


def _case(cin, cout, h, w, seed, n=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wgt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    return x, wgt, b


@pytest.mark.parametrize("cin,cout", SHAPES)
@pytest.mark.parametrize("lrelu", [False, True])
def test_conv3x3_f32_matches_torch(cuda_device, cin, cout, lrelu):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, w, b = _case(cin, cout, 24, 40, seed=cin * 100 + cout)
    ref = F.conv2d(x, w, b, padding=1)
    if lrelu:
        ref = F.leaky_relu(ref, 0.2)
    got = conv3x3(x.to(cuda_device), w, b, lrelu=lrelu).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < F32_TOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("h,w", [(1, 1), (7, 5), (8, 16), (9, 17), (33, 47), (16, 130)])
def test_conv3x3_f32_ragged_sizes(cuda_device, h, w):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, wgt, b = _case(64, 32, h, w, seed=h * 1000 + w, n=2)
    ref = F.conv2d(x, wgt, b, padding=1)
    got = conv3x3(x.to(cuda_device), wgt, b).cpu()
    assert (got - ref).abs().max().item() < F32_TOL * max(1.0, ref.abs().max().item())


def test_conv3x3_f32_upsample_fused(cuda_device):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, wgt, b = _case(64, 64, 13, 21, seed=5)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wgt, b, padding=1), 0.2)
    got = conv3x3(x.to(cuda_device), wgt, b, lrelu=True, upsample=True).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < F32_TOL * max(1.0, ref.abs().max().item())


def test_conv3x3_asymmetric_taps(cuda_device):
    """One-hot weights: catches swapped dy/dx, transposed C/D maps and channel permutations exactly."""
    from neural_enhanced_super_resolution_amd import conv3x3
    cin, cout = 16, 32
    x = torch.arange(cin * 6 * 7, dtype=torch.float32).reshape(1, cin, 6, 7) % 251
    for tap in range(9):
        w = torch.zeros(cout, cin, 3, 3)
        for o in range(cout):
            w[o, (o * 5 + tap) % cin, tap // 3, tap % 3] = 1.0
        b = torch.arange(cout, dtype=torch.float32)
        ref = F.conv2d(x, w, b, padding=1)
        got = conv3x3(x.to(cuda_device), w, b).cpu()
        assert torch.equal(got, ref), f"tap {tap}"


@pytest.mark.parametrize("cin,cout", [(12, 64), (64, 32), (160, 32), (192, 64), (64, 3)])
def test_conv3x3_bf16_matches_bf16_rounded_reference(cuda_device, cin, cout):
    """bf16 kernel: operands are bf16-rounded, accumulation is f32 -> compare against the torch
    conv of the bf16-rounded operands (tolerance = output bf16 rounding + summation order)."""
    from neural_enhanced_super_resolution_amd import conv3x3
    x, w, b = _case(cin, cout, 24, 40, seed=cin + cout)
    xr, wr = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(xr, wr, b, padding=1)
    got = conv3x3(x.to(cuda_device), w, b, dtype="bf16").cpu()
    scale = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 2 ** -7 * scale   # output stored as bf16 (8 bits of mantissa)


BIG = (136, 160)   # >= 128*128 trunk pixels: the bf16 path takes the large-tile LDS-DMA kernel


@pytest.mark.parametrize("cin,cout", [(16, 64), (64, 32), (96, 32), (160, 32), (192, 64), (64, 64), (64, 3)])
def test_conv3x3_bf16_xl_kernel(cuda_device, cin, cout):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, w, b = _case(cin, cout, BIG[0], BIG[1], seed=3 * cin + cout, n=2)
    ref = F.leaky_relu(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, padding=1), 0.2)
    got = conv3x3(x.to(cuda_device), w, b, lrelu=True, dtype="bf16").cpu()
    scale = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 2 ** -7 * scale


def test_conv3x3_bf16_xl_one_hot_taps(cuda_device):
    """Exact one-hot check of the large-tile kernel's tap / row-reuse / channel bookkeeping
    (small integers are exact in bf16)."""
    from neural_enhanced_super_resolution_amd import conv3x3
    cin, cout = 32, 64
    h, w = 130, 131
    x = ((torch.arange(cin * h * w, dtype=torch.float32).reshape(1, cin, h, w) * 7) % 61).contiguous()
    for tap in range(9):
        wt = torch.zeros(cout, cin, 3, 3)
        for o in range(cout):
            wt[o, (o * 5 + tap) % cin, tap // 3, tap % 3] = 1.0
        b = torch.arange(cout, dtype=torch.float32)
        ref = F.conv2d(x, wt, b, padding=1)
        got = conv3x3(x.to(cuda_device), wt, b, dtype="bf16").cpu()
        assert torch.equal(got, ref), f"tap {tap}"


def test_conv3x3_bf16_xl_upsample(cuda_device):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, wgt, b = _case(64, 64, 70, 90, seed=8)
    ref = F.conv2d(F.interpolate(x.bfloat16().float(), scale_factor=2, mode="nearest"), wgt.bfloat16().float(), b, padding=1)
    got = conv3x3(x.to(cuda_device), wgt, b, upsample=True, dtype="bf16").cpu()
    assert (got - ref).abs().max().item() < 2 ** -7 * max(1.0, ref.abs().max().item())


WINO_TOL = 2e-5   # relative to max|ref|: Winograd's transforms add a few ulps to the direct form


@pytest.mark.parametrize("cin,cout", [(16, 64), (64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)])
@pytest.mark.parametrize("hw", [(24, 40), (9, 17), (33, 47), (1, 1), (2, 3), (7, 5)])
def test_conv3x3_f32_winograd_matches_torch(cuda_device, cin, cout, hw):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, w, b = _case(cin, cout, hw[0], hw[1], seed=cin * 7 + cout + hw[0], n=2)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    got = conv3x3(x.to(cuda_device), w, b, lrelu=True, dtype="f32-winograd").cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < WINO_TOL * max(1.0, ref.abs().max().item())


def test_conv3x3_f32_winograd_one_hot_taps_and_upsample(cuda_device):
    from neural_enhanced_super_resolution_amd import conv3x3
    cin, cout = 16, 32
    x = (torch.arange(cin * 10 * 12, dtype=torch.float32).reshape(1, cin, 10, 12) * 3) % 127
    for tap in range(9):
        w = torch.zeros(cout, cin, 3, 3)
        for o in range(cout):
            w[o, (o * 5 + tap) % cin, tap // 3, tap % 3] = 1.0
        b = torch.arange(cout, dtype=torch.float32)
        ref = F.conv2d(x, w, b, padding=1)
        got = conv3x3(x.to(cuda_device), w, b, dtype="f32-winograd").cpu()
        assert (got - ref).abs().max().item() < 1e-3, f"tap {tap}"      # small integers: transforms are exact up to halves
    xs, ws, bs = _case(64, 64, 13, 21, seed=5)
    ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), ws, bs, padding=1)
    got = conv3x3(xs.to(cuda_device), ws, bs, upsample=True, dtype="f32-winograd").cpu()
    assert (got - ref).abs().max().item() < WINO_TOL * max(1.0, ref.abs().max().item())


def test_winograd_vs_direct_random_shapes(cuda_device):
    """Seeded sweep over ragged shapes / batches / channel counts: the Winograd and the direct kernels
    implement the same convolution, so they must agree to a few ulps of the accumulation everywhere
    (catches tile-edge, odd-size and batch indexing slips that fixed shapes can miss)."""
    from neural_enhanced_super_resolution_amd import conv3x3
    rng = np.random.default_rng(123)
    for _ in range(24):
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        cin = int(rng.choice([3, 8, 12, 16, 64, 96, 192]))
        cout = int(rng.choice([3, 32, 64]))
        up = bool(rng.integers(0, 2)) and h * w < 400
        lrelu = bool(rng.integers(0, 2))
        x, wt, b = _case(cin, cout, h, w, seed=int(rng.integers(1 << 30)), n=n)
        xd = x.to(cuda_device)
        a = conv3x3(xd, wt, b, lrelu=lrelu, upsample=up, dtype="f32-direct").cpu()
        c = conv3x3(xd, wt, b, lrelu=lrelu, upsample=up, dtype="f32-winograd").cpu()
        assert a.shape == c.shape
        tol = WINO_TOL * max(1.0, a.abs().max().item())
        assert (a - c).abs().max().item() < tol, (n, cin, cout, h, w, up, lrelu)


# ---------------------------------------------------------------- f32 on the f16 matrix cores (f16 pairs)
SPLIT_TOL = 2e-5   # relative to max|ref|: operands carry 22-23 bits, products lose w_lo*x_lo (2^-22)


@pytest.mark.parametrize("cin,cout", SHAPES)
@pytest.mark.parametrize("hw", [(24, 40), (1, 1), (7, 5), (9, 17), (33, 47), (16, 130)])
def test_conv3x3_f32_split_matches_torch(cuda_device, cin, cout, hw):
    from neural_enhanced_super_resolution_amd import conv3x3
    x, w, b = _case(cin, cout, hw[0], hw[1], seed=cin * 11 + cout + hw[1], n=2)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    got = conv3x3(x.to(cuda_device), w, b, lrelu=True, dtype="f32-split").cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < SPLIT_TOL * max(1.0, ref.abs().max().item())


def test_conv3x3_f32_split_one_hot_taps_upsample_and_small_values(cuda_device):
    from neural_enhanced_super_resolution_amd import conv3x3
    cin, cout = 16, 32
    x = (torch.arange(cin * 10 * 12, dtype=torch.float32).reshape(1, cin, 10, 12) * 3) % 127
    for tap in range(9):
        w = torch.zeros(cout, cin, 3, 3)
        for o in range(cout):
            w[o, (o * 5 + tap) % cin, tap // 3, tap % 3] = 1.0
        b = torch.arange(cout, dtype=torch.float32)
        ref = F.conv2d(x, w, b, padding=1)
        got = conv3x3(x.to(cuda_device), w, b, dtype="f32-split").cpu()
        assert torch.equal(got, ref), f"tap {tap}"      # small integers are exact in every plane
    xs, ws, bs = _case(64, 64, 13, 21, seed=5)
    ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), ws, bs, padding=1)
    got = conv3x3(xs.to(cuda_device), ws, bs, upsample=True, dtype="f32-split").cpu()
    assert (got - ref).abs().max().item() < SPLIT_TOL * max(1.0, ref.abs().max().item())
    # values whose lo halves are f16 subnormals (|x| ~ 1e-3): the matrix core must not flush them
    xt, wt, bt = _case(64, 32, 20, 36, seed=9)
    xt, bt = xt * 1e-3, bt * 0
    ref = F.conv2d(xt, wt, bt, padding=1)
    got = conv3x3(xt.to(cuda_device), wt, bt, dtype="f32-split").cpu()
    assert (got - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_conv3x3_f32_split_big_tile_variant(cuda_device):
    """>= 2048 tiles of 16x32 px selects the 4-rows-per-wave geometry (one workgroup per CU)."""
    from neural_enhanced_super_resolution_amd import conv3x3
    x, w, b = _case(64, 64, 500, 509, seed=77, n=2)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    got = conv3x3(x.to(cuda_device), w, b, lrelu=True, dtype="f32-split").cpu()
    assert (got - ref).abs().max().item() < SPLIT_TOL * max(1.0, ref.abs().max().item())


def test_split_vs_direct_random_shapes(cuda_device):
    from neural_enhanced_super_resolution_amd import conv3x3
    rng = np.random.default_rng(321)
    for _ in range(24):
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 41)), int(rng.integers(1, 80))
        cin = int(rng.choice([3, 8, 12, 16, 64, 96, 192]))
        cout = int(rng.choice([3, 32, 64]))
        up = bool(rng.integers(0, 2)) and h * w < 400
        lrelu = bool(rng.integers(0, 2))
        x, wt, b = _case(cin, cout, h, w, seed=int(rng.integers(1 << 30)), n=n)
        xd = x.to(cuda_device)
        a = conv3x3(xd, wt, b, lrelu=lrelu, upsample=up, dtype="f32-direct").cpu()
        c = conv3x3(xd, wt, b, lrelu=lrelu, upsample=up, dtype="f32-split").cpu()
        assert a.shape == c.shape
        tol = SPLIT_TOL * max(1.0, a.abs().max().item())
        assert (a - c).abs().max().item() < tol, (n, cin, cout, h, w, up, lrelu)
