"""GPU probe: rdb_bf16_strip_kernel (NESR_STRIP=1) against the per-layer bf16 kernels (NESR_STRIP=0) and the f32 oracle."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_enhanced_super_resolution_amd import RRDBNet  # noqa: E402
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict  # noqa: E402


def net(sd, nb, strip, num_in_ch=3, scale=2):
    os.environ["NESR_STRIP"] = "1" if strip else "0"
    n = RRDBNet(num_in_ch, 3, scale=scale, num_block=nb, compute_dtype="bf16")
    n.load_state_dict(sd)
    n.eval().to("cuda:0")
    n(torch.zeros(1, num_in_ch, 16, 16, device="cuda:0"))
    n.check_status()
    return n


def psnr(a, b):
    return float(10 * torch.log10(1.0 / ((a - b) ** 2).mean().clamp_min(1e-30)))


def main():
    from oracle.rrdbnet_ref import RRDBNetRef
    cases = [(1, 1, (64, 96), 2), (1, 1, (200, 264), 2), (2, 3, (120, 72), 2), (2, 1, (40, 72), 4), (1, 2, (26, 34), 2)]
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        cases = [(23, 6, (532, 532), 2)]
    for nb, n, hw, scale in cases:
        sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=scale, num_block=nb)
        x = torch.rand(n, 3, *hw, generator=torch.Generator().manual_seed(1))
        a, b = net(sd, nb, True, scale=scale), net(sd, nb, False, scale=scale)
        xd = x.cuda()
        ya = a(xd); a.check_status()
        yb = b(xd); b.check_status()
        d = (ya - yb).abs()
        line = f"nb {nb} n {n} hw {hw} scale {scale}: strip vs per-layer max {d.max().item():.3e} mean {d.mean().item():.3e}"
        if nb <= 2 and hw[0] * hw[1] <= 30000:
            ref = RRDBNetRef(3, 3, scale=scale, num_block=nb)
            ref.load_state_dict(sd)
            with torch.no_grad():
                yr = ref(x)
            line += f" | psnr vs oracle: strip {psnr(ya.cpu(), yr):.2f} per-layer {psnr(yb.cpu(), yr):.2f}"
        ya2 = a(xd); a.check_status()
        line += f" | repeat bitwise {bool(torch.equal(ya, ya2))}"
        for m, name in ((a, "strip"), (b, "per-layer")):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                m(xd)
            torch.cuda.synchronize()
            line += f" | {name} {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    main()
