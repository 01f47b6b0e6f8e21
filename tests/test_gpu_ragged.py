"""GPU: images of different sizes in one batch (nesr_forward_ragged), the form in which a tiling RealESRGANer evaluates
all tiles of a frame -- realesrgan's tile_process, driven as standalone/direct_esrgan.py:118-127 does (tile=512,
tile_pad=10), cuts interior tiles of 532 x 532 and smaller edge and corner tiles, each an independent model() call.
Every image of a ragged batch must carry exactly the bits model(image) gives it alone (on a size_independent model:
kernels chosen by arithmetic, not by image size), whatever else is in the batch and wherever its slot is."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(scale, num_block=2, dtype="bf16"):
    from neural_enhanced_super_resolution_amd import RRDBNet
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    net = RRDBNet(3, 3, scale=scale, num_block=num_block, compute_dtype=dtype)
    net.load_state_dict(synthetic_state_dict(seed=0, num_in_ch=3, scale=scale, num_block=num_block))
    net.eval().to("cuda:0")
    return net


@pytest.mark.parametrize("scale", [2, 4])
def test_ragged_batch_equals_each_image_alone(cuda_device, scale):
    net = _net(scale)
    net.size_independent = True
    sizes = [(132, 200), (40, 64), (132, 36), (2, 2), (66, 200), (130, 198)]
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    g = torch.Generator().manual_seed(5)
    imgs = [torch.rand(1, 3, h, w, generator=g).to(cuda_device) for h, w in sizes]
    x = torch.full((len(sizes), 3, H, W), 7.0, device=cuda_device)          # what lies outside an image must not matter
    for j, (im, (h, w)) in enumerate(zip(imgs, sizes)):
        x[j, :, :h, :w] = im[0]
    out = net.forward_ragged(x, sizes)
    s = net.out_scale()
    assert out.shape == (len(sizes), 3, H * s, W * s)
    for j, (im, (h, w)) in enumerate(zip(imgs, sizes)):
        alone = net(im)
        assert torch.equal(out[j:j + 1, :, :h * s, :w * s], alone), (j, (out[j:j + 1, :, :h * s, :w * s] - alone).abs().max().item())
    # a different company and slot order change nothing
    perm = [3, 0, 5]
    x2 = torch.zeros((len(perm), 3, H, W), device=cuda_device)
    for j, k in enumerate(perm):
        x2[j, :, :sizes[k][0], :sizes[k][1]] = imgs[k][0]
    out2 = net.forward_ragged(x2, [sizes[k] for k in perm])
    for j, k in enumerate(perm):
        h, w = sizes[k]
        assert torch.equal(out2[j, :, :h * s, :w * s], out[k, :, :h * s, :w * s])
    net.check_status()


def test_tiled_enhance_ragged_equals_shape_groups(cuda_device):
    """1100 x 1300 with tile 512 / pad 10: nine tile shapes.  One ragged batch against the shape-group batches."""
    from neural_enhanced_super_resolution_amd import RealESRGANer, RRDBNet
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=2)
    frame = synthetic_frame(1100, 1300, seed=2)
    outs = []
    for ragged in (True, False):
        up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=2, num_block=2), tile=512, tile_pad=10,
                          pre_pad=0, half=True, device=cuda_device)
        assert up.model.size_independent
        up.ragged_tiles = ragged
        calls0 = up.model.calls
        out, _ = up.enhance(frame)
        outs.append((out, up.model.calls - calls0))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] <= 3 and outs[1][1] >= 9, (outs[0][1], outs[1][1])        # one network call per stream vs one per shape group


def test_ragged_is_bf16_only_and_validates_sizes(cuda_device):
    from neural_enhanced_super_resolution_amd._lib import NesrHipError
    x = torch.rand(2, 3, 32, 32, device=cuda_device)
    with pytest.raises(NesrHipError, match="bf16"):
        _net(2, dtype="f32").forward_ragged(x, [(32, 32), (16, 16)])
    net = _net(2)
    with pytest.raises(NesrHipError, match="image 1"):
        net.forward_ragged(x, [(32, 32), (34, 16)])                           # taller than its slot
    with pytest.raises(NesrHipError, match="image 0"):
        net.forward_ragged(x, [(31, 32), (16, 16)])                           # not a multiple of the unshuffle factor
    with pytest.raises(ValueError):
        net.forward_ragged(x, [(32, 32)])
    big = torch.zeros(net.RAGGED_MAX + 1, 3, 4, 4, device=cuda_device)
    with pytest.raises(NesrHipError, match="images per call"):
        net.forward_ragged(big, [(4, 4)] * (net.RAGGED_MAX + 1))


def test_size_independent_small_frame_matches_batched_evaluation(cuda_device):
    """Without the switch a small frame takes the small-frame kernel (another summation order: close, not equal); with it,
    the frame alone, in an equal-shape batch and in a ragged batch are the same bits."""
    net = _net(2)
    x = torch.rand(3, 3, 64, 96, generator=torch.Generator().manual_seed(9)).to(cuda_device)
    default = net(x[:1])
    net.size_independent = True
    alone = net(x[:1])
    batch = net(x)
    assert torch.equal(batch[:1], alone)
    assert (alone - default).abs().max().item() < 0.05        # bf16: both within the format's error of each other
    net.size_independent = False
    assert torch.equal(net(x[:1]), default)
