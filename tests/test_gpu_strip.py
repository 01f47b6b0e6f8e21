"""GPU: the LDS-resident bf16 dense-block kernel (rdb_bf16_strip_kernel, csrc/rdb_bf16_strip.hip) on the shapes that stress its
indexing -- the form in which the tiles of a frame are evaluated (realesrgan's tile_process behind upscaler.enhance(img),
standalone/direct_esrgan.py:118-127, 148).  A strip is 16 columns wide and is swept in positions of 12 rows, layer m lagging
m - 1 rows; strips exchange edge columns.  So: images narrower than a strip and lower than a position, widths / heights one off
the multiples, one-strip images (no neighbour at all), many small images per workgroup, the full ragged batch (64 images), both
network scales, and the third block of an RRDB (second residual).

The checker is the CPU oracle (f32) and the per-layer bf16 path of the same library: the strip kernel rounds at the same
points (x1..x4 and the block output to bf16, f32 accumulation) but sums in another order, so it is not the per-layer path's
bits; its error against the f32 oracle must be the per-layer path's (PSNR within 0.5 dB) and the two must agree to bf16
resolution.  What IS bitwise: repeatability, and independence of an image's values from its company and its slot."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

NB = 2      # dense-block groups (RRDBs): 6 strip launches per forward, every third with the second residual


def _psnr(a, b):
    return float(10.0 * torch.log10(1.0 / ((a.double() - b.double()) ** 2).mean()))


def _net(scale, strip, sd, seg=None):
    from neural_enhanced_super_resolution_amd import RRDBNet
    old, old_seg = os.environ.get("NESR_STRIP"), os.environ.get("NESR_STRIP_SEG")
    os.environ["NESR_STRIP"] = strip          # read when the device context is created (first forward)
    if seg is not None:
        os.environ["NESR_STRIP_SEG"] = seg
    try:
        net = RRDBNet(3, 3, scale=scale, num_block=NB, compute_dtype="bf16")
        net.load_state_dict(sd)
        net.eval().to("cuda:0")
        net.size_independent = True
        net(torch.zeros(1, 3, 4 * (4 // scale), 4 * (4 // scale), device="cuda:0"))      # creates the context under the switch
    finally:
        for k, v in (("NESR_STRIP", old), ("NESR_STRIP_SEG", old_seg)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return net


def _oracle(scale, sd):
    from oracle.rrdbnet_ref import RRDBNetRef
    ref = RRDBNetRef(3, 3, scale=scale, num_block=NB)
    ref.load_state_dict(sd, strict=True)
    return ref


def _ragged(net, imgs):
    sizes = [tuple(im.shape[-2:]) for im in imgs]
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    x = torch.full((len(imgs), 3, H, W), 3.0, device="cuda:0")          # what lies outside an image must not matter
    for j, im in enumerate(imgs):
        x[j, :, :sizes[j][0], :sizes[j][1]] = im[0].to("cuda:0")
    out = net.forward_ragged(x, sizes)
    net.check_status()
    s = net.out_scale()
    return [out[j:j + 1, :, :h * s, :w * s].cpu() for j, (h, w) in enumerate(sizes)]


# trunk sizes (after the x2 model's pixel-unshuffle: input / 2): below one strip / one position, one off the multiples of 16
# and 12, exactly the multiples, one strip wide, long and thin
SHAPES_X2 = [(2, 2), (8, 30), (22, 32), (24, 34), (26, 64), (46, 66), (48, 96), (50, 98), (200, 20), (20, 200), (130, 198)]
SHAPES_X4 = [(1, 1), (5, 15), (11, 16), (12, 17), (13, 33), (23, 47), (24, 48), (25, 49), (100, 9), (9, 100)]


@pytest.mark.parametrize("scale", [2, 4])
def test_strip_kernel_against_oracle_and_per_layer_path_on_edge_shapes(cuda_device, scale):
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=scale, num_block=NB)
    strip, layer, ref = _net(scale, "1", sd), _net(scale, "0", sd), _oracle(scale, sd)
    shapes = SHAPES_X2 if scale == 2 else SHAPES_X4
    g = torch.Generator().manual_seed(11)
    imgs = [torch.rand(1, 3, h, w, generator=g) for h, w in shapes]
    for net in (strip, layer):
        net.set_kernel_timing("cuda:0", True)
        net.kernel_time()
    got, per = _ragged(strip, imgs), _ragged(layer, imgs)
    # which path ran: one launch per dense block (3 per RRDB) against five
    assert strip.kernel_time()[1] == 3 * NB, "the ragged batch did not run the LDS-resident kernel"
    assert layer.kernel_time()[1] == 15 * NB
    again = _ragged(strip, imgs)
    for j, im in enumerate(imgs):
        with torch.no_grad():
            want = ref(im)
        assert got[j].shape == want.shape
        assert torch.equal(got[j], again[j]), f"{shapes[j]}: not repeatable"
        d = (got[j] - per[j]).abs().max().item()
        assert d < 4e-3 * max(1.0, want.abs().max().item()), (shapes[j], d)          # bf16 resolution of an O(1) output
        if im.numel() >= 3 * 64:          # PSNR of a handful of pixels says nothing
            ps, pl = _psnr(got[j], want), _psnr(per[j], want)
            assert ps > pl - 0.5, (shapes[j], ps, pl)
            assert ps > 55.0, (shapes[j], ps)
        assert torch.isfinite(got[j]).all()


def test_strip_values_do_not_depend_on_company_or_slot(cuda_device):
    """The packing decides which workgroup runs which strips in which order; an image's values must not depend on it."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=NB)
    net = _net(2, "1", sd)
    g = torch.Generator().manual_seed(3)
    shapes = [(132, 200), (40, 64), (132, 36), (66, 200), (130, 198), (24, 34)]
    imgs = [torch.rand(1, 3, h, w, generator=g) for h, w in shapes]
    full = _ragged(net, imgs)
    for j in (0, 2, 5):
        alone = _ragged(net, [imgs[j]])
        assert torch.equal(alone[0], full[j]), (shapes[j], (alone[0] - full[j]).abs().max().item())
    perm = [4, 1, 0]
    part = _ragged(net, [imgs[k] for k in perm])
    for j, k in enumerate(perm):
        assert torch.equal(part[j], full[k])
    # plain batched forward of equal images = the same images in a ragged batch
    same = torch.cat([imgs[1], imgs[1].flip(-1)], 0).to(cuda_device)
    eq = net(same).cpu()
    net.check_status()
    rg = _ragged(net, [imgs[1], imgs[1].flip(-1)])
    assert torch.equal(eq[0:1], rg[0]) and torch.equal(eq[1:2], rg[1])


def test_strip_full_ragged_batch_of_64(cuda_device):
    """RAGGED_MAX images, several items per workgroup: every image equals its evaluation alone."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1)
    from neural_enhanced_super_resolution_amd import RRDBNet
    old = os.environ.get("NESR_STRIP")
    os.environ["NESR_STRIP"] = "1"
    try:
        net = RRDBNet(3, 3, scale=2, num_block=1, compute_dtype="bf16")
        net.load_state_dict(sd)
        net.eval().to(cuda_device)
        net.size_independent = True
        g = torch.Generator().manual_seed(9)
        shapes = [(2 * (3 + (7 * i) % 40), 2 * (5 + (11 * i) % 90)) for i in range(64)]
        imgs = [torch.rand(1, 3, h, w, generator=g) for h, w in shapes]
        net.set_kernel_timing(cuda_device, True)
        net.kernel_time()
        full = _ragged(net, imgs)
        assert net.kernel_time()[1] == 3
        for j in (0, 13, 31, 63):
            alone = _ragged(net, [imgs[j]])
            assert torch.equal(alone[0], full[j]), shapes[j]
    finally:
        if old is None:
            os.environ.pop("NESR_STRIP", None)
        else:
            os.environ["NESR_STRIP"] = old


@pytest.mark.parametrize("scale", [2, 4])
def test_row_segments_give_the_bits_of_the_whole_sweep(cuda_device, scale):
    """A strip cut into row segments (what the packer does when a batch has fewer strips than the device has compute units:
    one rank's share of a sharded frame, a small frame) stores, row for row, the bits of the uncut sweep: every segment
    starts one position early and leaves the rows of that warm-up position to the segment above."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=scale, num_block=NB)
    u = 4 // scale if scale in (2, 1) else 1
    u = 2 if scale == 2 else 1
    g = torch.Generator().manual_seed(21)
    shapes = [(266 * u, 40 * u), (130 * u, 72 * u), (61 * u, 33 * u), (25 * u, 16 * u), (13 * u, 50 * u)]      # trunk heights 266 (23 positions) ... 13 (2)
    imgs = [torch.rand(1, 3, h, w, generator=g) for h, w in shapes]
    whole = _ragged(_net(scale, "1", sd, seg="-1"), imgs)
    for seg in ("2", "3", "7", "0"):          # at most 2 / 3 / 7 positions per segment; the packer's own choice
        net = _net(scale, "1", sd, seg=seg)
        net.set_kernel_timing("cuda:0", True)
        net.kernel_time()
        cut = _ragged(net, imgs)
        assert net.kernel_time()[1] == 3 * NB
        for j in range(len(imgs)):
            assert torch.equal(cut[j], whole[j]), (seg, shapes[j], (cut[j] - whole[j]).abs().max().item())
