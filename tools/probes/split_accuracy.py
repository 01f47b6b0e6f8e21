"""f16-pair (hi, lo) emulation of the f32 convs on CPU vs an f64 evaluation of the same network:
the accuracy argument for conv3x3_f16x2.hip (run from the repo root; needs no GPU)."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import oracle.rrdbnet_ref as R
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict, synthetic_frame
torch.set_num_threads(8)
sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
x = torch.from_numpy(synthetic_frame(64, 64, seed=1)).float().permute(2, 0, 1).unsqueeze(0) / 255.0
orig_conv = R._conv

def split16(t, dt):
    hi = t.to(dt).float()
    lo = (t - hi).to(dt).float()
    return hi, lo

def make_conv(mode, dt=torch.float16):
    def conv(x, sd, name):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        if mode == "f64":
            return F.conv2d(x.double(), w.double(), b.double(), 1, 1)
        xh, xl = split16(x, dt); wh, wl = split16(w, dt)
        # "stored activations are hi+lo": the kernel never sees x itself
        y = F.conv2d(xh, wh, None, 1, 1) + (F.conv2d(xh, wl, None, 1, 1) + F.conv2d(xl, wh, None, 1, 1))
        if mode == "s4":
            y = y + F.conv2d(xl, wl, None, 1, 1)
        return y + b.view(1, -1, 1, 1)
    return conv

def run(convfn, dtype=torch.float32):
    R._conv = convfn
    try:
        sdd = {k: v.to(dtype) for k, v in sd.items()}
        with torch.no_grad():
            return R.rrdbnet_forward(x.to(dtype), sdd, scale=2, num_block=23)
    finally:
        R._conv = orig_conv

y64 = run(make_conv("f64"), torch.float64)
y32 = run(orig_conv)
print("f32 oracle vs f64: max abs %.3e  rms %.3e" % ((y32.double() - y64).abs().max(), (y32.double() - y64).pow(2).mean().sqrt()))
for mode in ("s3", "s4"):
    y = run(make_conv(mode))
    print("f16x2 %s vs f64: max abs %.3e rms %.3e ; vs f32 oracle max abs %.3e" % (mode, (y.double() - y64).abs().max(), (y.double() - y64).pow(2).mean().sqrt(), (y - y32).abs().max()))
y = run(make_conv("s3", torch.bfloat16))
print("bf16x2 s3 vs f64: max abs %.3e" % (y.double() - y64).abs().max())
print("out range", float(y64.min()), float(y64.max()))
