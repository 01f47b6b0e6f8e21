"""GPU (one rank): nesr_forward_sharded_u8 -- the sharded frame below Python, RCCL point to point -- gives RealESRGANer.enhance's
bytes for a tiled 8-bit frame (standalone/direct_esrgan.py:118-127,148), with and without a communicator.  More ranks need
more GPUs than the test box has: the plan is tested against the Python protocol for 1..8 ranks on the CPU
(tests/test_sharded_gloo.py::test_c_abi_plan_is_the_python_plan), the protocol itself over gloo with 2, 3, 4 and 8 ranks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scale,hw,tile,half", [(2, (300, 420), 128, True), (4, (150, 200), 64, False)])
def test_one_rank_sharded_c_abi_equals_enhance(cuda_device, scale, hw, tile, half):
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=scale, num_block=2)
    netscale = {2: 2}.get(scale, 4)
    up = RealESRGANer(scale=netscale, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=scale, num_block=2, compute_dtype="bf16"), tile=tile,
                      tile_pad=10, pre_pad=0, half=half, device=cuda_device)
    frame = synthetic_frame(hw[0], hw[1], seed=6)
    want, _ = up.enhance(frame)
    band = torch.from_numpy(frame).to(cuda_device)
    got = up.model.forward_sharded_u8(band, hw, tile, 10, through_fp16=half)
    torch.cuda.synchronize()
    up.model.check_status()
    assert np.array_equal(got.cpu().numpy(), want)
    # the same through a one-rank RCCL communicator (librccl.so is loaded here, not before)
    up.model.comm_init(cuda_device, 0, 1, RRDBNet.comm_unique_id())
    got2 = up.model.forward_sharded_u8(band, hw, tile, 10, through_fp16=half)
    torch.cuda.synchronize()
    assert torch.equal(got2, got)
