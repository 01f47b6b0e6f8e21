import os, statistics, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neural_enhanced_super_resolution_amd import RRDBNet
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
dev = torch.device("cuda:0")
net = RRDBNet(3, 3, scale=2, compute_dtype="bf16")
net.load_state_dict(synthetic_state_dict(seed=0, num_in_ch=3, scale=2))
net.eval().to(dev)
net.size_independent = True
x = torch.rand(8, 3, 532, 532, device=dev)
sizes = [(532, 532)] * 8
half = [(532, 532)] * 4 + [(266, 532)] * 4
y0 = net(x); y1 = net.forward_ragged(x, sizes)
print("equal-size ragged == batch:", torch.equal(y0, y1))
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3
res = {"batch": [], "ragged equal": [], "ragged 4 full + 4 half-height": [], "batch of 6 (same valid area)": []}
x6 = x[:6].contiguous()
for r in range(6):
    res["batch"].append(t(lambda: net(x)))
    res["ragged equal"].append(t(lambda: net.forward_ragged(x, sizes)))
    res["ragged 4 full + 4 half-height"].append(t(lambda: net.forward_ragged(x, half)))
    res["batch of 6 (same valid area)"].append(t(lambda: net(x6)))
for k, v in res.items():
    print(f"{k:32s} {1e3*statistics.median(v):.2f} ms")
