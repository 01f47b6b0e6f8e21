import sys, time, torch
sys.path.insert(0, '/root/repo')
from neural_enhanced_super_resolution_amd import RRDBNet
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict
sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
net = RRDBNet(3, 3, scale=2); net.load_state_dict(sd); net.eval().to('cuda:0')
for ns in (1, 2, 3, 4):
    xs = [torch.rand(1, 3, 512, 512, device='cuda') for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def step():
        for i in range(ns):
            with torch.cuda.stream(streams[i]):
                net(xs[i], slot=i)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{ns} concurrent frames: {dt*1e3:.2f} ms per step, {ns * 1.048576 / dt:.1f} MP/s")
# batch in one launch
for nb in (2, 3, 4):
    x = torch.rand(nb, 3, 512, 512, device='cuda')
    for _ in range(3): net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"batch {nb}: {dt*1e3:.2f} ms, {nb * 1.048576 / dt:.1f} MP/s")
