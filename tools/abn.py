#!/usr/bin/env python3
"""In-process timing of N builds of libnesr_hip.so, interleaved rounds on one device (cdna_hip_programming.md rule 24).

    tools/abn.py A.so B.so [C.so ...] [--dtype direct|bf16|wino|split] [--hw 512] [--batch 1] [--rounds 12]
                 [--env "K=V,K2=V2;K=V;..."]      (one ;-separated entry per library, applied through its first forward)
"""
import argparse
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from neural_enhanced_super_resolution_amd import _lib  # noqa: E402
from neural_enhanced_super_resolution_amd.synth import synthetic_state_dict  # noqa: E402
from tools.ab import load, make_ctx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--dtype", default="split")
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--env", default="")
    ap.add_argument("--c3", action="store_true", help="the 40 ragged tiles of a 3840x2160 frame cut 512/10 (nesr_forward_ragged, bf16)")
    ap.add_argument("--share", type=int, default=1, help="with --c3: only every share-th tile (what one of `share` ranks of a sharded frame evaluates)")
    args = ap.parse_args()
    code = {"f32": 0, "direct": 0, "bf16": 1, "wino": 2, "split": 3}[args.dtype]
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
    n = len(args.libs)
    envs = (args.env.split(";") + [""] * n)[:n]
    x = torch.rand(args.batch, 3, args.hw, args.hw, device="cuda")
    y = [torch.empty(args.batch, 3, 2 * args.hw, 2 * args.hw, device="cuda") for _ in range(n)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs, ctxs = [], []

    hw = None
    if args.c3:
        def spans(n):
            out = []
            for t in range((n + 511) // 512):
                a, b = t * 512, min((t + 1) * 512, n)
                out.append(min(b + 10, n) - max(a - 10, 0))
            return out
        sizes = [(hh, ww) for hh in spans(2160) for ww in spans(3840)][::args.share]
        args.batch, args.hw = len(sizes), 532
        hw = (ctypes.c_int * (2 * len(sizes)))(*[v for pr in sizes for v in pr])
        x = torch.rand(args.batch, 3, args.hw, args.hw, device="cuda")
        y = [torch.zeros(args.batch, 3, 2 * args.hw, 2 * args.hw, device="cuda") for _ in range(n)]

    def run(i):
        if hw is not None:
            libs[i].nesr_forward_ragged.argtypes = _lib.SIGNATURES["nesr_forward_ragged"][1]
            rc = libs[i].nesr_forward_ragged(ctxs[i], ctypes.c_void_p(x.data_ptr()), args.batch, 3, args.hw, args.hw, hw,
                                             ctypes.c_void_p(y[i].data_ptr()), stream)
            assert rc == 0, libs[i].nesr_last_error()
            return
        rc = libs[i].nesr_forward(ctxs[i], ctypes.c_void_p(x.data_ptr()), args.batch, 3, args.hw, args.hw,
                                  ctypes.c_void_p(y[i].data_ptr()), stream)
        assert rc == 0, libs[i].nesr_last_error()

    for i, (path, env) in enumerate(zip(args.libs, envs)):
        kvs = [kv.split("=") for kv in env.split(",") if kv]
        for k, v in kvs:
            os.environ[k] = v
        lib = load(os.path.abspath(path))
        libs.append(lib)
        ctxs.append(make_ctx(lib, sd, code))
        run(i)
        torch.cuda.synchronize()
        for k, _ in kvs:
            os.environ.pop(k, None)
    for i in range(n):
        run(i)
    torch.cuda.synchronize()
    times = [[] for _ in range(n)]
    for r in range(args.rounds):
        order = list(range(n))
        order = order[r % n:] + order[:r % n]
        for i in order:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                run(i)
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / args.iters * 1e3)
    base = statistics.median(times[0])
    for i, name in enumerate(args.libs):
        t = times[i]
        eq = torch.equal(y[0], y[i])
        print(f"{i} {os.path.basename(name):28s} {envs[i]:24s} median {statistics.median(t):8.3f} ms  min {min(t):8.3f}  ratio {statistics.median(t) / base:.4f}  "
              f"== lib0: {eq} (max diff {float((y[0] - y[i]).abs().max()):.2e})", flush=True)


if __name__ == "__main__":
    main()
