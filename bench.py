#!/usr/bin/env python3
"""Headline bench: output megapixels/s of RealESRGAN_x2plus x2 upscaling on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1 directly; N>1 under torch.distributed.run)

Workload (BASELINE.json configs[1], "c2"): synthetic 512x512 -> 1024x1024 RealESRGAN_x2plus, fp32,
single tile, one frame per step per GPU, input already resident in HBM as the NCHW float tensor
RealESRGANer.process() hands to ``self.model`` (the drop-in boundary).  Frames (= tiles
of the reference's tile grid) are independent units, so N GPUs run N frames per step with no
data-path collective ("scaling": "weak").  Other workloads: --workload c3 (2160p, bf16, reference
tile grid 512/10) and c4 (1080p x4plus).

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel family (the 345
dense-block 3x3 convs = 92 % of the FLOPs): algorithmic FLOPs / HIP-event time on the launch
stream, against the dense MFMA peak of the dtype.  `cpu_baseline` is the torch-CPU oracle
(oracle/, a restatement: the reference's own RRDBNet lives in the absent basicsr package) timed on
this host's cores on a bounded crop of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip-level table)
# f32 workloads: which conv algorithm (RRDBNet compute_dtype); --f32-algo or NESR_F32_ALGO=split|winograd|direct
F32_ALGOS = {
    "split": dict(compute_dtype="f32", mfma="f16", executed_per_algorithmic=3.0,
                  dtype="f32 (operands as f16 hi+lo pairs, 3 f16 MFMAs per product, f32 accumulate)",
                  algorithm="direct implicit GEMM, v_mfma_f32_16x16x32_f16, three products per MAC (hi*hi + hi*lo + lo*hi)"),
    "winograd": dict(compute_dtype="f32-winograd", mfma="f32", executed_per_algorithmic=1 / 2.25, dtype="f32",
                     algorithm="winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32"),
    "direct": dict(compute_dtype="f32-direct", mfma="f32", executed_per_algorithmic=1.0, dtype="f32",
                   algorithm="direct implicit GEMM, v_mfma_f32_32x32x2_f32"),
}

WORKLOADS = {
    # name: (H, W, num_in_ch, upstream scale, netscale, dtype, tile, tile_pad)
    "c2": dict(h=512, w=512, scale=2, dtype="f32", tile=0, tile_pad=10, desc="512x512->1024x1024 RealESRGAN_x2plus fp32 single tile"),
    "c2-bf16": dict(h=512, w=512, scale=2, dtype="bf16", tile=0, tile_pad=10, desc="512x512->1024x1024 RealESRGAN_x2plus bf16 single tile"),
    "c3": dict(h=2160, w=3840, scale=2, dtype="bf16", tile=512, tile_pad=10, desc="3840x2160->7680x4320 RealESRGAN_x2plus bf16, tile 512/10"),
    "c3-stream": dict(h=2160, w=3840, scale=2, dtype="bf16", tile=512, tile_pad=10, independent=True,
                      desc="3840x2160->7680x4320 RealESRGAN_x2plus bf16, tile 512/10, one independent frame per GPU (a video stream: weak scaling)"),
    "c3-f32": dict(h=2160, w=3840, scale=2, dtype="f32", tile=512, tile_pad=10, desc="3840x2160->7680x4320 RealESRGAN_x2plus fp32, tile 512/10"),
    "c3-exact": dict(h=2160, w=3840, scale=2, dtype="f32", tile=0, tile_pad=10, banded=True,
                     desc="3840x2160->7680x4320 RealESRGAN_x2plus fp32, untiled (tile=0, as nesr/nesr.py:224); N>1: row bands + RCCL apron exchange per RDB"),
    "c4": dict(h=1080, w=1920, scale=4, dtype="bf16", tile=512, tile_pad=10, desc="1920x1080->7680x4320 RealESRGAN_x4plus bf16, tile 512/10"),
}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads(cap=16):
    """Threads for the CPU baseline: affinity mask, capped by the cgroup CPU quota and by `cap`
    (the GPU box gives one GPU's job a 16-CPU share; more threads than that only thrash)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def pmc_traffic(workload):
    """HBM bytes per conv launch (average over the launches of one frame) from the committed rocprofv3 PMC
    passes (profiles/r01/split_c2_pmc_fetch_write.txt; Winograd: final_c2_pmc_fetch_write.txt; collected
    with tools/gpu_pmc_traffic.sh): FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md, HBM
    section) + WRITE_SIZE, two separate --pmc passes.  PMC counters cannot be collected inside this
    process, so the figure comes from profiles/ (null if absent)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except Exception:
        return None


def cpu_baseline(sd, frame, scale, crop, threads):
    """Times the oracle on a crop of the frame; returns (dict, oracle float output of the crop)."""
    import numpy as np
    import torch
    from oracle.rrdbnet_ref import RRDBNetRef
    torch.set_num_threads(threads)
    ref = RRDBNetRef(3, 3, scale=scale)
    ref.load_state_dict(sd, strict=True)
    x = torch.from_numpy(np.ascontiguousarray(frame[:crop, :crop, ::-1].transpose(2, 0, 1))).float().div(255.0).unsqueeze(0)
    with torch.no_grad():
        ref(x[:, :, :32, :32])           # warm the thread pool / oneDNN primitives
        t0 = time.perf_counter()
        y = ref(x)
        dt = time.perf_counter() - t0
    netscale = {2: 2, 1: 1}.get(scale, 4)
    mp = (crop * netscale) ** 2 / 1e6
    return dict(value=round(mp / dt, 5), unit="MP/s", cores=threads, kind="port",
                sample=f"{crop}x{crop} crop of the bench frame, one RRDBNet forward, {dt:.1f} s on {threads} torch threads"), x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-crop", type=int, default=512, help="side of the crop the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--inflight", type=int, default=None,
                    help="frames in flight per GPU for the single-tile workloads (each on its own HIP stream and context "
                         "replica); default 4 for c2 / c2-bf16 (2: -5 %, 1: -20 %, 6-8: no more), 1 otherwise")
    ap.add_argument("--f32-algo", default=None, choices=["split", "winograd", "direct"],
                    help="conv algorithm of the fp32 workloads (default split: f16 hi+lo operand pairs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    wl = WORKLOADS[args.workload]
    scale, dtype = wl["scale"], wl["dtype"]
    netscale = {2: 2, 1: 1}.get(scale, 4)
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=scale)
    frame = synthetic_frame(wl["h"], wl["w"], seed=rank)            # uint8 HWC BGR, one per rank
    algo = None
    if dtype == "f32":
        key = {"s": "split", "w": "winograd", "d": "direct"}[(args.f32_algo or os.environ.get("NESR_F32_ALGO") or "split").lower()[0]]
        algo = F32_ALGOS[key]
        os.environ.pop("NESR_F32_ALGO", None)   # the choice travels as compute_dtype
    net = RRDBNet(3, 3, scale=scale, compute_dtype=algo["compute_dtype"] if algo else dtype)
    up = RealESRGANer(scale=netscale, model_path={"params_ema": sd}, model=net, tile=wl["tile"], tile_pad=wl["tile_pad"],
                      pre_pad=0, half=False, device=dev)
    model = up.model

    # the boundary tensor: what enhance() -> pre_process() puts on the device
    up.pre_process(np.ascontiguousarray(frame[:, :, ::-1].astype(np.float32) / 255.0))
    x = up.img

    banded_frame = bool(wl.get("banded")) and world > 1
    if banded_frame:
        # one untiled frame for the whole job: row bands, apron rows exchanged point-to-point before every RDB
        # (banded.py), result gathered on rank 0 (strong scaling); bitwise the single-GPU untiled result
        from neural_enhanced_super_resolution_amd import banded
        frame = synthetic_frame(wl["h"], wl["w"], seed=0)
        band = torch.from_numpy(banded.scatter_band(frame, rank, world, net.unshuffle)).to(dev)
    sharded_frame = wl["tile"] > 0 and world > 1 and not wl.get("independent")
    if sharded_frame:
        # one frame for the whole job: tiles of upstream's grid sharded over the ranks, overlap rows
        # exchanged point-to-point over RCCL, result gathered on rank 0 (strong scaling)
        from neural_enhanced_super_resolution_amd import sharded
        frame = synthetic_frame(wl["h"], wl["w"], seed=0)
        band = torch.from_numpy(sharded.scatter_rows(frame, rank, world)).to(dev)

    # single-tile workloads: `inflight` independent frames per step, each on its own stream with its own context
    # replica -- a 512x512 frame's layers are one-round launches of 256-512 workgroups (all in their prologue, then
    # all in their epilogue, 1.3 us between dependent kernels), so a second frame's kernels fill the gaps and the
    # idle half of the LDS / wave slots (tools/probes/concurrent_frames.py, tools/probes/launch_floor.hip)
    inflight = args.inflight if args.inflight else (4 if (wl["tile"] == 0 and not wl.get("banded")) else 1)
    if wl["tile"] > 0 or wl.get("banded"):
        inflight = 1
    xs, streams = [x], [torch.cuda.current_stream(dev)]
    for i in range(1, inflight):
        up.pre_process(np.ascontiguousarray(synthetic_frame(wl["h"], wl["w"], seed=1000 * i + rank)[:, :, ::-1].astype(np.float32) / 255.0))
        xs.append(up.img)
        streams.append(torch.cuda.Stream(dev))
    for st in streams[1:]:
        st.wait_stream(streams[0])      # xs[i] were produced on the main stream

    def step():
        if inflight > 1:
            ys = []
            for i in range(inflight):
                with torch.cuda.stream(streams[i]):
                    ys.append(model(xs[i], slot=i))
            return ys[0]
        if banded_frame:
            return banded.enhance_banded(up, band, (wl["h"], wl["w"]))
        if sharded_frame:
            return sharded.enhance_sharded(up, band, (wl["h"], wl["w"]))
        if wl["tile"] > 0:
            up.tile_process()
            return up.output
        return model(x)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    log(f"workload {args.workload} rank {rank}/{world}: model ready, warming up")
    for _ in range(args.warmup):
        y = step()
    torch.cuda.synchronize(dev)
    log("warmup done")
    # HIP-event brackets around the dense-block convs, on the launch stream.  With several frames in flight the
    # brackets of the streams overlap, so the roofline leg is measured on a second timed region below (one frame,
    # one stream: the kernel by itself, which is also what the committed rocprofv3 summary shows)
    model.set_kernel_timing(dev, inflight == 1)
    model.kernel_time()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches, k_flops = model.kernel_time()
    model.set_kernel_timing(dev, False)
    single = None
    if inflight > 1:
        model.set_concurrent(False)          # one frame alone: the device is this context's
        model.set_kernel_timing(dev, True)
        model.kernel_time()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model(x)
        torch.cuda.synchronize(dev)
        single = (time.perf_counter() - t1) / args.steps
        k_ms, k_launches, k_flops = model.kernel_time()
        model.set_kernel_timing(dev, False)
        model.set_concurrent(True)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out_mp = wl["h"] * netscale * wl["w"] * netscale / 1e6
    frames_per_step = 1 if (sharded_frame or banded_frame) else world * inflight
    value = frames_per_step * args.steps * out_mp / elapsed
    frame_flops = net.forward_flops(1, wl["h"], wl["w"])

    result = {
        "metric": "output megapixels/sec, RealESRGAN_x2plus x2 upscale" if scale == 2 else "output megapixels/sec, RealESRGAN_x4plus x4 upscale",
        "value": round(value, 3), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "strong" if (sharded_frame or banded_frame) else "weak",
        "vs_baseline": None, "dtype": algo["dtype"] if algo else dtype, "data": "synthetic (seeded frames, seeded random-init weights)",
        "config": {"workload": f"{args.workload}: {wl['desc']}",
                   "frames_per_step": frames_per_step, "frames_in_flight_per_gpu": inflight,
                   "partition": ("tiles of the 512/10 grid sharded over ranks, RCCL p2p overlap rows, gather on rank 0" if sharded_frame
                                 else "row bands of one untiled frame, RCCL p2p apron rows before every RDB, gather on rank 0" if banded_frame
                                 else "independent frames, no data-path collective"),
                   "boundary": "RRDBNet.forward on the device-resident NCHW f32 tensor of RealESRGANer.pre_process",
                   "tflop_per_frame": round(frame_flops / 1e12, 4)},
        "frames_per_s": round(frames_per_step * args.steps / elapsed, 4),
        "tflops_whole_net": round(frames_per_step * args.steps * frame_flops / elapsed / 1e12, 2),
    }
    multi_stream = wl["tile"] > 0 and getattr(up, "tile_streams", 1) > 1
    if k_ms > 0:
        # concurrent streams: per-stream event brackets overlap, so divide the trunk FLOPs by the wall
        # time of the timed region instead (includes the non-trunk kernels: a lower bound)
        achieved = k_flops / ((elapsed if multi_stream else k_ms * 1e-3)) / 1e12
        # `achieved` / `frac` use the ALGORITHMIC (direct-conv) FLOPs as the contract asks, against the dense
        # peak of the MFMA type that runs; `executed_frac` is the matrix-core utilisation of what actually
        # executes: 3 f16 MFMA-FLOPs per algorithmic FLOP for the f16-pair form, 1/2.25 for Winograd.
        peak = PEAK_TFLOPS[algo["mfma"] if algo else dtype]
        per_alg = algo["executed_per_algorithmic"] if algo else 1.0
        if multi_stream:
            result["roofline_note"] = ("tile groups run on concurrent streams (overlapping event brackets): `achieved` = "
                                       "trunk FLOPs / wall time of the timed region, a lower bound")
        result["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                              "frac": round(achieved / peak, 4), "traffic": pmc_traffic(args.workload if (algo is None or algo is F32_ALGOS["split"]) else args.workload + "-" + key),
                              "algorithm": algo["algorithm"] if algo else "direct implicit GEMM, v_mfma_f32_32x32x16_bf16",
                              "executed_frac": round(achieved * per_alg / peak, 4),
                              "kernel": "conv3x3_f16x2_kernel / conv3x3_wino_f32_kernel / conv3x3_mfma_kernel / conv3x3_bf16_xl_kernel (the 345 dense-block convs per frame)",
                              "avg_launch_us": round(1e3 * k_ms / max(k_launches, 1), 2), "launches": int(k_launches)}
        if inflight > 1:
            result["roofline"]["measured_on"] = (f"a second timed region of {args.steps} single-frame forwards on one stream "
                                                 "(the kernel by itself; `value` is the throughput with frames in flight)")
    if single is not None:
        # one frame alone, same process, same device: the latency figure (and what `value` is with --inflight 1)
        result["single_frame"] = {"ms": round(1e3 * single, 3), "mp_s": round(out_mp / single, 3)}
    if rank == 0:
        # host-to-host enhance() (PCIe + quantisation inclusive), reported beside the metric, never as `value`
        log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
        up.enhance(frame)                                  # warm (workspace for the u8 path)
        t1 = time.perf_counter()
        out_u8, _ = up.enhance(frame)
        torch.cuda.synchronize(dev)
        result["enhance_host_to_host_mp_s"] = round(out_mp / (time.perf_counter() - t1), 3)
        if args.cpu_crop > 0 and not args.no_parity:
            crop = min(args.cpu_crop, wl["h"], wl["w"])
            crop -= crop % 2
            threads = host_threads()
            log(f"cpu baseline: oracle on a {crop}x{crop} crop with {threads} threads")
            base, xc, yref = cpu_baseline(sd, frame, scale, crop, threads)
            log("cpu baseline done")
            result["cpu_baseline"] = base
            ygpu = model(xc.to(dev)).float().cpu()
            err = (ygpu - yref).abs().max().item()
            mse = ((ygpu - yref) ** 2).mean().item()
            result["parity_vs_cpu_oracle"] = {"max_abs": float(f"{err:.3e}"), "psnr_db": round(10 * np.log10(1.0 / max(mse, 1e-30)), 2),
                                             "sample": base["sample"].split(",")[0], "tolerance": 1e-3 if dtype == "f32" else None,
                                             "note": "oracle = torch CPU f32; an f64 evaluation of the same net differs from it by 1e-6"}
            result["gpu_over_cpu"] = round(value / world / base["value"], 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
