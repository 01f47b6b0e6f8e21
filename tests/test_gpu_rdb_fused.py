"""GPU: the fused dense-block kernel (rdb_f16x2_kernel: conv1..conv5 of an RDB in one launch, neighbouring tiles
synchronised through progress words) against the per-layer launches of the same arithmetic (NESR_RDB_FUSE=0): the two
must agree bit for bit on every shape -- whole tiles, a partial tile column, tile rows that end below the image (MFMA
waves without a row), both network scales, the 12-channel nesr form -- run after run (the progress words are never reset:
an epoch per launch), and the fused path must be the one that ran (its launch count is a fifth of the other's).

Reference semantics: the dense block of basicsr's RRDBNet (un-vendored; restated in oracle/rrdbnet_ref.py), called at
nesr/nesr.py:887-891 and through RealESRGANer.enhance at standalone/direct_esrgan.py:148."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3     # BASELINE.json north_star: max abs on the [0, 1] image against the CPU oracle


def _net(sd, num_block, fuse, num_in_ch=3, scale=2):
    from neural_enhanced_super_resolution_amd import RRDBNet
    old = os.environ.get("NESR_RDB_FUSE")
    os.environ["NESR_RDB_FUSE"] = "-1" if fuse else "0"
    try:
        n = RRDBNet(num_in_ch, 3, scale=scale, num_block=num_block)
        n.load_state_dict(sd)
        n.eval().to("cuda:0")
        n(torch.zeros(1, num_in_ch, 16, 16, device="cuda:0"))        # the context is created with the switch in force
    finally:
        if old is None:
            os.environ.pop("NESR_RDB_FUSE", None)
        else:
            os.environ["NESR_RDB_FUSE"] = old
    return n


def _launches(net, x):
    net.set_kernel_timing(x.device, True)
    net.kernel_time()                                    # clear
    net(x)
    torch.cuda.synchronize()
    _, launches, _ = net.kernel_time()
    net.set_kernel_timing(x.device, False)
    return launches


@pytest.mark.parametrize("num_block,hw,scale,num_in_ch", [
    (1, (64, 96), 2, 3),          # 4 x 2 whole tiles
    (2, (200, 264), 2, 3),        # internal 100 x 132: last tile row has 4 of 8 rows, last tile column 4 of 32 pixels
    (3, (512, 448), 2, 3),        # 256 x 224 internal: 32 x 7 tiles
    (2, (510, 512), 2, 3),        # odd rows after the pad
    (2, (40, 72), 4, 3),          # x4plus: no unshuffle, internal = input size
    (1, (96, 64), 2, 12),         # the nesr form: RRDBNet(num_in_ch=12) with x2plus weights, no unshuffle, x4 (nesr/nesr.py:216)
])
def test_fused_equals_per_layer_bitwise(cuda_device, num_block, hw, scale, num_in_ch):
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    if num_in_ch == 12:
        sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=num_block)     # x2plus shapes: conv_first takes 12
        kw = dict(num_in_ch=12, scale=4)
    else:
        sd = synthetic_state_dict(seed=0, num_in_ch=num_in_ch, scale=scale, num_block=num_block)
        kw = dict(num_in_ch=num_in_ch, scale=scale)
    x = torch.rand(1, kw["num_in_ch"], *hw, generator=torch.Generator().manual_seed(1)).to(cuda_device)
    per_layer = _net(sd, num_block, False, **kw)
    fused = _net(sd, num_block, True, **kw)
    want = per_layer(x)
    per_layer.check_status()
    got = fused(x)
    fused.check_status()
    assert torch.equal(got, want), (got - want).abs().max().item()
    for _ in range(3):                                   # epochs advance, progress words persist
        assert torch.equal(fused(x), want)
    fused.check_status()
    # the fused kernel is what ran: one launch per dense block instead of five
    n_f, n_p = _launches(fused, x), _launches(per_layer, x)
    assert n_p - n_f == 4 * 3 * num_block, (n_f, n_p)


def test_fused_against_the_cpu_oracle(cuda_device):
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    from oracle.rrdbnet_ref import RRDBNetRef
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    net = _net(sd, 23, True)
    ref = RRDBNetRef(3, 3, scale=2, num_block=23)
    ref.load_state_dict(sd, strict=True)
    x = torch.rand(1, 3, 72, 136, generator=torch.Generator().manual_seed(3))
    err = (net(x.to(cuda_device)).cpu() - ref(x)).abs().max().item()
    net.check_status()
    assert err < TOL, err


def test_frame_too_large_for_the_cus_takes_the_per_layer_path(cuda_device):
    """More 8x32 tiles than CUs: the workgroups could not all be resident, so the fused kernel must not be chosen."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=1)
    net = _net(sd, 1, True)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    x_small = torch.rand(1, 3, 64, 64, device=cuda_device)
    h = 2 * 8 * (cus // 8 + 1)                           # internal rows x 256 px = more tiles than CUs
    x_large = torch.rand(1, 3, h, 512, device=cuda_device)
    assert ((h // 2 + 7) // 8) * 8 > cus
    n_small, n_large = _launches(net, x_small), _launches(net, x_large)
    assert n_large - n_small == 4 * 3, (n_small, n_large)
    net.check_status()


def test_fused_full_frame_repeats_are_bitwise_stable(cuda_device):
    """The 512 x 512 bench frame: 256 tiles, one per compute unit, every tile boundary an inter-workgroup hand-off (x1..x4
    halos through sc1 stores, a progress word, sc1 LDS-DMA loads) and one row of tiles in eight an inter-XCD one.  A stale
    halo read would show as a run that differs from the per-layer evaluation: 150 forwards, each compared bit for bit."""
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2, num_block=23)
    x = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(11)).to(cuda_device)
    want = _net(sd, 23, False)(x)
    fused = _net(sd, 23, True)
    assert _launches(fused, x) == 3 * 23
    bad = 0
    for i in range(150):
        if i % 7 == 0:
            torch.cuda.synchronize()                      # vary what is in flight when a forward starts
        got = fused(x)
        bad += 0 if torch.equal(got, want) else 1
    fused.check_status()
    assert bad == 0, f"{bad} of 150 fused forwards differ from the per-layer evaluation"
