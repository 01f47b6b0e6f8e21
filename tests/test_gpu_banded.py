"""The exact (row-band) multi-GPU mode on the HIP engine: the C ABI's nesr_band_* stages driven in lockstep for 2
and 3 emulated ranks inside one process (one context per rank; the row exchange is banded.py's, with an in-process
transport).  The band rows of every rank must be BITWISE the rows of the whole-frame forward: inside a band every
pixel sees the same operands, and the f32 kernels' per-pixel arithmetic does not depend on tile position."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _banded(nets, x, world):
    from neural_enhanced_super_resolution_amd import banded
    u = nets[0].unshuffle
    bands = banded.band_split(x.shape[2] // u, world)
    tops = [banded.APRON if r > 0 else 0 for r in range(world)]
    bots = [banded.APRON if r < world - 1 else 0 for r in range(world)]
    for r, net in enumerate(nets):
        lo, hi = bands[r]
        net.band_begin(x[:, :, (lo - tops[r]) * u:(hi + bots[r]) * u].contiguous())

    def exchange(buffer, k):   # everybody reads, then everybody writes: what simultaneous sends/receives do
        moves = []
        for r, net in enumerate(nets):
            lo, hi = bands[r]
            if r > 0:
                moves.append((r - 1, buffer, tops[r - 1] + (bands[r - 1][1] - bands[r - 1][0]), net.band_rows(buffer, tops[r], k)))
            if r < world - 1:
                moves.append((r + 1, buffer, tops[r + 1] - k, net.band_rows(buffer, tops[r] + (hi - lo) - k, k)))
        for dst, buf, row0, rows in moves:
            nets[dst].band_set_rows(buf, row0, rows)

    for i in range(nets[0].num_rdb):
        exchange(i % 3, banded.APRON)
        for net in nets:
            net.band_rdb(i)
    exchange(0, banded.APRON)
    exchange(3, banded.APRON)
    outs = []
    for r, net in enumerate(nets):
        y = net.band_tail()
        outs.append(y[:, :, 4 * tops[r]: y.shape[2] - 4 * bots[r]])
    return torch.cat(outs, 2)


def _banded_overlapped(nets, x, world):
    """The default protocol (banded.forward_banded_overlapped) for emulated ranks in lockstep: phase 0, pack the edge rows
    (one C-ABI call per rank), phase 1, unpack the neighbours' rows (one call per rank)."""
    from neural_enhanced_super_resolution_amd import banded
    A = banded.APRON
    u = nets[0].unshuffle
    bands = banded.band_split(x.shape[2] // u, world)
    tops = [A if r > 0 else 0 for r in range(world)]
    bots = [A if r < world - 1 else 0 for r in range(world)]
    for r, net in enumerate(nets):
        lo, hi = bands[r]
        net.band_begin(x[:, :, (lo - tops[r]) * u:(hi + bots[r]) * u].contiguous())
    nbytes = A * nets[0].band_row_bytes()
    mk = lambda: torch.empty(nbytes, dtype=torch.uint8, device=x.device)   # noqa: E731
    send_up = [mk() if r > 0 else None for r in range(world)]
    send_down = [mk() if r < world - 1 else None for r in range(world)]

    def pack(buffer):
        for r, net in enumerate(nets):
            net.band_pack_edges(buffer, tops[r], bots[r], A, send_up[r], send_down[r])

    def unpack(buffers):
        for r, net in enumerate(nets):
            for b in buffers:
                net.band_unpack_aprons(b, tops[r], bots[r], A, send_down[r - 1] if r > 0 else None, send_up[r + 1] if r < world - 1 else None)

    pack(0)
    unpack((0, 3))
    steps = 1
    for i in range(nets[0].num_rdb):
        for r, net in enumerate(nets):
            net.band_rdb_phase(i, 0, tops[r], bots[r], A)
        pack(banded.out_buffer(i))
        for r, net in enumerate(nets):
            net.band_rdb_phase(i, 1, tops[r], bots[r], A)
        unpack((banded.out_buffer(i),))
        steps += 1
    outs = []
    for r, net in enumerate(nets):
        y = net.band_tail()
        outs.append(y[:, :, 4 * tops[r]: y.shape[2] - 4 * bots[r]])
    assert steps == 1 + nets[0].num_rdb
    return torch.cat(outs, 2)


def _nets(n, algo, scale, num_block=2):
    from neural_enhanced_super_resolution_amd import RRDBNet
    from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(seed=9, num_in_ch=3, scale=scale, num_block=num_block)
    out = []
    for _ in range(n):
        net = RRDBNet(3, 3, scale=scale, num_block=num_block, compute_dtype=algo)
        net.load_state_dict(sd)
        out.append(net.eval().to("cuda:0"))
    return out


@pytest.mark.parametrize("algo", ["f32", "f32-winograd", "f32-direct"])
@pytest.mark.parametrize("world,scale,hw", [(2, 2, (96, 80)), (3, 2, (132, 72)), (2, 4, (40, 56))])
def test_banded_bitwise_equals_whole_frame(cuda_device, algo, world, scale, hw):
    nets = _nets(world + 1, algo, scale)
    x = torch.rand(1, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(4)).to(cuda_device)
    want = nets[-1](x)
    got = _banded(nets[:world], x, world)
    assert got.shape == want.shape
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("algo", ["f32", "f32-winograd", "f32-direct"])
@pytest.mark.parametrize("world,scale,hw", [(2, 2, (96, 80)), (3, 2, (132, 72)), (2, 4, (40, 56)), (2, 2, (560, 544))])
def test_overlapped_protocol_bitwise_equals_whole_frame(cuda_device, algo, world, scale, hw):
    """Edge rows first, conv5's interior rows while they travel (row-range launches for the default f32 form; whole-block
    phase 0 for the others): still bit for bit the whole-frame forward.  560x544: bands of several tile rows, and a
    whole frame that is evaluated by per-layer launches (more tiles than CUs) -- the small frames compare against the fused
    dense-block kernel."""
    if hw[0] > 200 and algo != "f32":
        pytest.skip("large frame: default form only")
    nets = _nets(world + 1, algo, scale)
    x = torch.rand(1, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(4)).to(cuda_device)
    want = nets[-1](x)
    got = _banded_overlapped(nets[:world], x, world)
    assert got.shape == want.shape
    assert torch.equal(got, want), float((got - want).abs().max())
    for n in nets:
        n.check_status()


def test_banded_bf16_same_operands(cuda_device):
    nets = _nets(3, "bf16", 2)
    x = torch.rand(1, 3, 96, 80, generator=torch.Generator().manual_seed(4)).to(cuda_device)
    want = nets[-1](x)
    got = _banded(nets[:2], x, 2)
    assert (got - want).abs().max().item() < 2e-2      # bf16: the kernel may differ with the image size


def test_band_api_state_errors(cuda_device):
    from neural_enhanced_super_resolution_amd._lib import NesrHipError
    net = _nets(1, "f32", 2, num_block=1)[0]
    with pytest.raises(RuntimeError):
        net.band_rdb(0)                                  # band_begin has not run
    x = torch.rand(1, 3, 32, 32).to(cuda_device)
    net.band_begin(x)
    with pytest.raises(NesrHipError):
        net.band_rdb(3)                                  # only 3 RDBs
    with pytest.raises(NesrHipError):
        net.band_rows(0, 10, 10)                         # 16 internal rows
    net(x)                                               # a whole-frame forward takes the workspace
    with pytest.raises(NesrHipError):
        net.band_tail()
