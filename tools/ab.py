#!/usr/bin/env python3
"""In-process A/B timing of two builds of libnesr_hip.so (cdna_hip_programming.md rule 24: perf
deltas come from interleaved rounds in ONE process on ONE device).

    tools/ab.py A.so B.so [--dtype direct|bf16|wino|split] [--dtype-b ...] [--hw 512] [--batch 1] [--rounds 12] [--env-b K=V,K2=V2]
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


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def make_ctx(lib, sd, dtype):
    h = ctypes.c_void_p()
    assert lib.nesr_create(ctypes.byref(h), 0, 12, 2, 64, 23, 32, 3, dtype) == 0, lib.nesr_last_error()
    for k, t in sd.items():
        t = t.contiguous()
        shape = (ctypes.c_int64 * t.dim())(*t.shape)
        assert lib.nesr_load_weight(h, k.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()) == 0
    assert lib.nesr_finalize_weights(h) == 0, lib.nesr_last_error()
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--dtype", default="wino")
    ap.add_argument("--dtype-b", default=None)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--env-a", default="")
    ap.add_argument("--env-b", default="")
    args = ap.parse_args()
    codes = {"f32": 0, "direct": 0, "bf16": 1, "wino": 2, "split": 3}
    dts = [codes[args.dtype], codes[args.dtype_b or args.dtype]]
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=2)
    x = torch.rand(args.batch, 3, args.hw, args.hw, device="cuda")
    y = [torch.empty(args.batch, 3, 2 * args.hw, 2 * args.hw, device="cuda") for _ in range(2)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs, ctxs = [], []

    def run(i):
        rc = libs[i].nesr_forward(ctxs[i], ctypes.c_void_p(x.data_ptr()), args.batch, 3, args.hw, args.hw,
                                  ctypes.c_void_p(y[i].data_ptr()), stream)
        assert rc == 0, libs[i].nesr_last_error()

    # the libraries read their NESR_* switches at context creation / first launch: keep each
    # variant's environment set through its first forward (use two copies of the file to compare
    # switches of one build -- dlopen returns the same handle for the same path)
    for i, (path, env) in enumerate(((args.a, args.env_a), (args.b, args.env_b))):
        kvs = [kv.split("=") for kv in env.split(",") if kv]
        for k, v in kvs:
            os.environ[k] = v
        lib = load(os.path.abspath(path))
        libs.append(lib)
        ctxs.append(make_ctx(lib, sd, dts[i]))
        run(i)
        torch.cuda.synchronize()
        for k, _ in kvs:
            os.environ.pop(k, None)

    for i in (0, 1):
        run(i)
    torch.cuda.synchronize()
    same = torch.equal(y[0], y[1])
    maxdiff = float((y[0] - y[1]).abs().max())
    times = [[], []]
    for r in range(args.rounds):
        for i in ((0, 1) if r % 2 == 0 else (1, 0)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                run(i)
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / args.iters * 1e3)
    for i, name in enumerate((args.a, args.b)):
        t = times[i]
        print(f"{'AB'[i]} {os.path.basename(name):24s} median {statistics.median(t):8.3f} ms   min {min(t):8.3f}   max {max(t):8.3f}")
    print(f"B/A median ratio {statistics.median(times[1]) / statistics.median(times[0]):.4f}   outputs bitwise equal: {same} (max abs diff {maxdiff:.3e})")


if __name__ == "__main__":
    main()
