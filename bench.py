#!/usr/bin/env python3
"""Headline bench: output megapixels/s of RealESRGAN_x2plus x2 upscaling on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 may be started plainly (this process then spawns `python -m torch.distributed.run --nproc-per-node N` on
127.0.0.1 before it touches a GPU, and relays rank 0's JSON line) or under torch.distributed.run as the driver does
(RANK / LOCAL_RANK / WORLD_SIZE in the environment): one process per GPU, RCCL ("nccl") process group.

Workload (BASELINE.json configs[1], "c2"): synthetic 512x512 -> 1024x1024 RealESRGAN_x2plus, fp32, single tile,
input already resident in HBM as the NCHW float tensor RealESRGANer.process() hands to ``self.model`` (the drop-in
boundary).  Frames are independent units, so N GPUs run their own frames with no data-path
collective ("scaling": "weak").  `value` is one frame at a time per GPU -- the reference's batch-1 call
(standalone/direct_esrgan.py:148), whose dense blocks run as rdb_f16x2_kernel -- and the `roofline` object describes that same
timed region and kernel.  `frames_in_flight` is the extra: THROUGHPUT with 4 independent frames per GPU, each on its own HIP stream
and context replica (per-layer kernels: the persistent kernel wants the device to itself), timed right after.

The line also carries
  strict_f32   the same frame on the f32 matrix cores (Winograd F(2x2,3x3) and direct implicit GEMM): ms, MP/s,
               fraction of the 157.3 TFLOP/s f32 MFMA peak, max abs vs the CPU oracle -- the default `dtype` is f32 in /
               out / accumulate with operands carried as f16 (hi, lo) pairs, so the strict forms are timed beside it;
  c4_stream    BASELINE.json configs[3] (1920x1080 -> 7680x4320 RealESRGAN_x4plus, bf16, tile 512/10), one frame per GPU;
  c3 / c3_stream   BASELINE.json configs[2] (3840x2160 -> 7680x4320, bf16, RealESRGANer(tile=512, tile_pad=10)): one
               frame for the whole job, tiles sharded over the ranks (sharded.py: RCCL point-to-point overlap rows,
               gather on rank 0), and one independent frame per GPU; frames/s of the whole job;
  cpu_baseline the torch-CPU oracle (oracle/, a restatement: the reference's own RRDBNet lives in the absent basicsr
               package) timed on this host's cores on the same frame, and the parity of the GPU result against it.

Other workloads: --workload c2-bf16 | c3 | c3-stream | c3-f32 | c3-exact | c4 | c5.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip-level table)
# f32 workloads: which conv algorithm (RRDBNet compute_dtype); --f32-algo or NESR_F32_ALGO=split|winograd|direct
F32_ALGOS = {
    "split": dict(compute_dtype="f32", mfma="f16", executed_per_algorithmic=3.0 * 14 / 13.5,
                  dtype="f32 (operands as f16 hi+lo pairs, 3 f16 MFMAs per product, f32 accumulate)",
                  algorithm="direct implicit GEMM, v_mfma_f32_16x16x32_f16, three products per MAC (hi*hi + hi*lo + lo*hi; 14 MFMAs where 13.5 are ideal)"),
    "winograd": dict(compute_dtype="f32-winograd", mfma="f32", executed_per_algorithmic=1 / 2.25, dtype="f32",
                     algorithm="winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32"),
    "direct": dict(compute_dtype="f32-direct", mfma="f32", executed_per_algorithmic=1.0, dtype="f32",
                   algorithm="direct implicit GEMM, v_mfma_f32_32x32x2_f32"),
}

WORKLOADS = {
    "c2": dict(h=512, w=512, scale=2, dtype="f32", tile=0, tile_pad=10, desc="512x512->1024x1024 RealESRGAN_x2plus fp32 single tile"),
    "c2-bf16": dict(h=512, w=512, scale=2, dtype="bf16", tile=0, tile_pad=10, desc="512x512->1024x1024 RealESRGAN_x2plus bf16 single tile"),
    "c3": dict(h=2160, w=3840, scale=2, dtype="bf16", tile=512, tile_pad=10, desc="3840x2160->7680x4320 RealESRGAN_x2plus bf16, tile 512/10"),
    "c3-stream": dict(h=2160, w=3840, scale=2, dtype="bf16", tile=512, tile_pad=10, independent=True,
                      desc="3840x2160->7680x4320 RealESRGAN_x2plus bf16, tile 512/10, one independent frame per GPU (a video stream: weak scaling)"),
    "c3-f32": dict(h=2160, w=3840, scale=2, dtype="f32", tile=512, tile_pad=10, desc="3840x2160->7680x4320 RealESRGAN_x2plus fp32, tile 512/10"),
    "c3-exact": dict(h=2160, w=3840, scale=2, dtype="f32", tile=0, tile_pad=10, banded=True,
                     desc="3840x2160->7680x4320 RealESRGAN_x2plus fp32, untiled (tile=0, as nesr/nesr.py:224); N>1: row bands + RCCL apron exchange per RDB"),
    "c4": dict(h=1080, w=1920, scale=4, dtype="bf16", tile=512, tile_pad=10, desc="1920x1080->7680x4320 RealESRGAN_x4plus bf16, tile 512/10"),
    "c5": dict(h=512, w=512, scale=4, dtype="f32", tile=0, tile_pad=0, nesr=True,
               desc="nesr pipeline ESRGAN stage, iterations=3 upscale_factor=2.0 no diffusion (nesr/nesr.py:516-633): 512^2 -> 2048^2 -> 8192^2 -> 16384^2, fp32"),
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


def csrc_sha16():
    """Hash of the kernel sources: stamps profiles/pmc_traffic.json so that a stale PMC figure is not reported."""
    d = os.path.join(ROOT, "neural_enhanced_super_resolution_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(key):
    """HBM bytes per conv launch (average over the launches of one frame) from the committed rocprofv3 PMC passes
    (tools/gpu_pmc_traffic.sh + tools/pmc_traffic_json.py: FETCH_SIZE x 2, the gfx950 correction of
    MI355X_MICROARCH.md's HBM section, + WRITE_SIZE, two separate --pmc passes).  PMC counters cannot be collected
    inside this process, so the figure comes from profiles/pmc_traffic.json -- and only if that file was collected on
    these kernel sources (its `csrc_sha16` stamp); otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d.get(key) if d.get("csrc_sha16") == csrc_sha16() else None
    except Exception:
        return None


def net_input(frame_bgr_u8, crop=None):
    """HWC uint8 BGR -> [1,3,H,W] float32 RGB in [0,1] (numpy's correctly rounded /255, as enhance() computes it)."""
    import numpy as np
    import torch
    f = frame_bgr_u8 if crop is None else frame_bgr_u8[:crop, :crop]
    return torch.from_numpy(np.ascontiguousarray((f[:, :, ::-1].astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))[None])


def cpu_baseline(sd, frame, scale, crop, threads):
    """Times the oracle on a crop of the frame; returns (dict, input, oracle float output of the crop)."""
    import torch
    from oracle.rrdbnet_ref import RRDBNetRef
    torch.set_num_threads(threads)
    ref = RRDBNetRef(3, 3, scale=scale)
    ref.load_state_dict(sd, strict=True)
    x = net_input(frame, crop)
    with torch.no_grad():
        ref(x[:, :, :32, :32])           # warm the thread pool / oneDNN primitives
        t0 = time.perf_counter()
        y = ref(x)
        dt = time.perf_counter() - t0
    netscale = {2: 2, 1: 1}.get(scale, 4)
    mp = (crop * netscale) ** 2 / 1e6
    return dict(value=round(mp / dt, 5), unit="MP/s", cores=threads, kind="port",
                sample=f"{crop}x{crop} crop of the bench frame, one RRDBNet forward, {dt:.1f} s on {threads} torch threads"), x, y


# ----------------------------------------------------------------------------------------------------- launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args, argv):
    """--gpus N > 1 started plainly: spawn N fresh worker processes (torch.distributed.run, one per GPU) BEFORE this
    process makes any GPU call, relay rank 0's JSON line, exit with the workers' status.  Never exec: a process
    that has initialised the GPU must not be replaced, and this one simply stays a parent."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL / tensor sharing across processes needs it here
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads(64) // args.gpus)))
    log(f"launcher: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is None:
        raise SystemExit(f"launcher: the workers printed no result line (exit status {rc})")
    res = json.loads(line)
    res["launcher"] = "bench.py spawned torch.distributed.run"
    print(json.dumps(res), flush=True)
    raise SystemExit(rc)


def dry_worker(args):
    """Launcher rehearsal without GPUs (tests/test_bench_launcher.py): the rendezvous, barrier and max-over-ranks
    timing of the real worker over gloo on the CPU; no kernel runs and the line says so."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the launcher (no GPU work)", "value": None, "unit": "MP/s", "n_gpus": world,
                          "world_size_seen": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * float(t.item()), 3), "backend": "gloo"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------- worker
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-crop", type=int, default=512, help="side of the crop the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the strict_f32 / c3 / c3_stream objects of the default c2 line")
    ap.add_argument("--inflight", type=int, default=None,
                    help="frames in flight per GPU for the `frames_in_flight` extra of the single-tile workloads (each on its own HIP "
                         "stream and context replica); default 4 for c2 / c2-bf16, 0 = skip; `value` is always one frame at a time")
    ap.add_argument("--f32-algo", default=None, choices=["split", "winograd", "direct"],
                    help="conv algorithm of the fp32 workloads (default split: f16 hi+lo operand pairs)")
    ap.add_argument("--filters", action="store_true", help="c5: also run the reference's cv2 pre / post filters (imgproc.py: NL-means + CLAHE, adaptive unsharp)")
    ap.add_argument("--dry-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        launch(args, sys.argv[1:])                       # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        return dry_worker(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict

    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(fn, steps, warmup):
        """W untimed steps, then exactly `steps` steps between barrier + synchronize brackets; max over ranks (s)."""
        for _ in range(warmup):
            fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    wl = WORKLOADS[args.workload]
    if wl.get("nesr"):
        return bench_c5(args, wl, dev, rank, world, timed)
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
    extra_inflight = (4 if args.inflight is None else args.inflight) if (wl["tile"] == 0 and not wl.get("banded")) else 0
    inflight = 1

    def step():
        if banded_frame:
            return banded.enhance_banded(up, band, (wl["h"], wl["w"]))
        if sharded_frame:
            return sharded.enhance_sharded(up, band, (wl["h"], wl["w"]))
        if wl["tile"] > 0:
            up.tile_process()
            return up.output
        return model(x)

    log(f"workload {args.workload} rank {rank}/{world}: model ready, warming up")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    log("warmup done")
    # HIP-event brackets around the dense blocks, on the launch stream, over the timed region itself
    model.set_kernel_timing(dev, True)
    model.kernel_time()
    elapsed = timed(step, args.steps, 0)
    k_ms, k_launches, k_flops = model.kernel_time()
    model.set_kernel_timing(dev, False)
    model.check_status()
    flight = None
    if extra_inflight > 1:
        # the throughput extra: `extra_inflight` independent frames per step, each on its own stream with its own context replica --
        # a 512x512 frame's per-layer launches are one-round grids of 256-512 workgroups (all in their prologue, then all in their
        # epilogue, 1.3 us between dependent kernels), so a second frame's kernels fill the gaps (tools/probes/concurrent_frames.py)
        xs, streams = [x], [torch.cuda.current_stream(dev)]
        for i in range(1, extra_inflight):
            up.pre_process(np.ascontiguousarray(synthetic_frame(wl["h"], wl["w"], seed=1000 * i + rank)[:, :, ::-1].astype(np.float32) / 255.0))
            xs.append(up.img)
            streams.append(torch.cuda.Stream(dev))
        for st in streams[1:]:
            st.wait_stream(streams[0])      # xs[i] were produced on the main stream

        def step_flight():
            ys = []
            for i in range(extra_inflight):
                with torch.cuda.stream(streams[i]):
                    ys.append(model(xs[i], slot=i))
            return ys[0]

        for _ in range(2):
            step_flight()
        torch.cuda.synchronize(dev)
        flight = timed(step_flight, args.steps, 0)
        model.check_status()
        model.set_concurrent(False)          # back to one frame alone: the device is this context's
        del xs

    out_mp = wl["h"] * netscale * wl["w"] * netscale / 1e6
    frames_per_step = 1 if (sharded_frame or banded_frame) else world
    value = frames_per_step * args.steps * out_mp / elapsed
    frame_flops = net.forward_flops(1, wl["h"], wl["w"])
    name = "RealESRGAN_x2plus x2" if scale == 2 else "RealESRGAN_x4plus x4"
    multi = ""

    result = {
        "metric": f"output megapixels/sec, {name} upscale{multi}",
        "value": round(value, 3), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "strong" if (sharded_frame or banded_frame) else "weak",
        "vs_baseline": None, "dtype": algo["dtype"] if algo else dtype, "data": "synthetic (seeded frames, seeded random-init weights)",
        "config": {"workload": f"{args.workload}: {wl['desc']}",
                   "frames_per_step": frames_per_step, "frames_in_flight_per_gpu": inflight, "world_size": world,
                   "partition": ("tiles of the 512/10 grid sharded over ranks, RCCL p2p overlap rows, gather on rank 0" if sharded_frame
                                 else "row bands of one untiled frame, RCCL p2p apron rows before every RDB, gather on rank 0" if banded_frame
                                 else "independent frames, no data-path collective"),
                   "boundary": "RRDBNet.forward on the device-resident NCHW f32 tensor of RealESRGANer.pre_process",
                   "tflop_per_frame": round(frame_flops / 1e12, 4)},
        "frames_per_s": round(frames_per_step * args.steps / elapsed, 4),
        "tflops_whole_net": round(frames_per_step * args.steps * frame_flops / elapsed / 1e12, 2),
    }
    strip = dtype == "bf16" and wl["tile"] > 0 and model.strip_kernel_active()
    multi_stream = wl["tile"] > 0 and getattr(up, "tile_streams", 1) > 1 and not strip
    if k_ms > 0:
        # concurrent streams: per-stream event brackets overlap, so divide the trunk FLOPs by the wall
        # time of the timed region instead (includes the non-trunk kernels: a lower bound)
        achieved = k_flops / ((elapsed if multi_stream else k_ms * 1e-3)) / 1e12
        # `achieved` / `frac` use the ALGORITHMIC (direct-conv) FLOPs as the contract asks, against the dense
        # peak of the MFMA type that runs; `executed_frac` is the matrix-core utilisation of what actually
        # executes: 3.11 f16 MFMA-FLOPs per algorithmic FLOP for the f16-pair form, 1/2.25 for Winograd.
        peak = PEAK_TFLOPS[algo["mfma"] if algo else dtype]
        per_alg = algo["executed_per_algorithmic"] if algo else 1.0
        if multi_stream:
            result["roofline_note"] = ("tile groups run on concurrent streams (overlapping event brackets): `achieved` = "
                                       "trunk FLOPs / wall time of the timed region, a lower bound")
        tkey = args.workload if (algo is None or algo is F32_ALGOS["split"]) else args.workload + "-" + key
        result["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                              "frac": round(achieved / peak, 4), "traffic": pmc_traffic(tkey),
                              "algorithm": algo["algorithm"] if algo else "direct implicit GEMM, v_mfma_f32_32x32x16_bf16",
                              "executed_frac": round(achieved * per_alg / peak, 4),
                              "kernel": ("rdb_bf16_strip_kernel (conv1..conv5 of a dense block of ALL tiles of the frame in one persistent launch, working set in LDS: 69 launches per frame)"
                                         if strip else
                                         "rdb_f16x2_kernel (conv1..conv5 of a dense block in one launch: 69 launches = the 345 dense-block convs of a frame)"
                                         if (algo is F32_ALGOS["split"] and k_launches and k_flops / k_launches > 2e10) else
                                         "conv3x3_f16x2_kernel / conv3x3_wino_f32_kernel / conv3x3_mfma_kernel / conv3x3_bf16_xl_kernel (the 345 dense-block convs per frame)"),
                              "avg_launch_us": round(1e3 * k_ms / max(k_launches, 1), 2), "launches": int(k_launches)}
        result["roofline"]["measured_on"] = "the timed region of `value` (HIP events on the launch stream around the dense blocks of every frame)"
    if flight is not None:
        result["frames_in_flight"] = {"frames_per_gpu": extra_inflight, "mp_s": round(world * extra_inflight * args.steps * out_mp / flight, 3),
                                      "ms_per_step": round(1e3 * flight / args.steps, 3),
                                      "note": "throughput with independent frames on their own HIP streams and context replicas (per-layer kernels); "
                                              "not the headline: `value` and `roofline` are one frame at a time"}

    extras = args.workload == "c2" and not args.no_extras
    yref = xc = None
    if rank == 0:
        # host-to-host enhance() (PCIe + quantisation inclusive), reported beside the metric, never as `value`
        log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
        up.enhance(frame)                                  # warm (workspace for the u8 path)
        t1 = time.perf_counter()
        up.enhance(frame)
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
            model.check_status()
            err = (ygpu - yref).abs().max().item()
            mse = ((ygpu - yref) ** 2).mean().item()
            result["parity_vs_cpu_oracle"] = {"max_abs": float(f"{err:.3e}"), "psnr_db": round(10 * np.log10(1.0 / max(mse, 1e-30)), 2),
                                             "sample": base["sample"].split(",")[0], "tolerance": 1e-3 if dtype == "f32" else None,
                                             "note": "oracle = torch CPU f32; an f64 evaluation of the same net differs from it by 1e-6"}
            result["gpu_over_cpu"] = round(value / world / base["value"], 1)
    if extras:
        if rank == 0:
            log("strict f32 forms")
            result["strict_f32"] = strict_f32(args, sd, x, xc, yref, dev, out_mp, frame_flops)
        del up, model, net
        torch.cuda.empty_cache()
        log("c3 objects")
        # The 4K objects are extras of this line: whatever happens in them -- an exception on this rank, or (N > 1) a rank
        # that never comes back from the sharded frame's point-to-point exchange -- the headline measured above is still
        # printed, with the failure named where the object would have been.
        import threading
        state = {"printed": False}
        plock = threading.Lock()
        limit = 300.0

        def give_up():
            # the headline is printed (once) with the hang named in it, and EVERY rank leaves with a failure status: a hung
            # exchange must not read as success
            if rank == 0:
                with plock:
                    if not state["printed"]:
                        state["printed"] = True
                        result.setdefault("c3", {"error": f"no result after {limit:.0f} s: a rank did not come back from the sharded 4K frame "
                                                           "(RCCL point-to-point exchange / gather)"})
                        print(json.dumps(result), flush=True)
            os._exit(4 if rank == 0 else 3)

        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        if world > 1:
            watchdog.start()
        try:
            result.update(c3_objects(args, dev, rank, world, timed))
        except Exception as e:      # noqa: BLE001 -- reported in the line, not swallowed
            log(f"c3 objects failed: {e!r}")
            result["c3"] = {"error": repr(e)}
    else:
        watchdog = None
    if rank == 0:
        if extras:
            with plock:
                if not state["printed"]:
                    state["printed"] = True
                    print(json.dumps(result), flush=True)
        else:
            print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        if watchdog is not None:
            watchdog.cancel()
        dist.destroy_process_group()


def strict_f32(args, sd, x, xc, yref, dev, out_mp, frame_flops):
    """The C2 frame on the f32 matrix cores (no reduced-precision operand format anywhere): one frame at a time."""
    import torch
    from neural_enhanced_super_resolution_amd import RRDBNet
    out = {"note": "f32 storage, f32 MFMA, f32 accumulate; one 512x512 frame at a time on one stream; `frac` = algorithmic "
                   "(direct-convolution) FLOPs of the 345 dense-block convs / HIP-event time / 157.3 TFLOP/s (Winograd executes 1/2.25 "
                   "of them, so it can exceed 1)"}
    for name, cd in (("winograd", "f32-winograd"), ("direct", "f32-direct")):
        net = RRDBNet(3, 3, scale=2, compute_dtype=cd)
        net.load_state_dict(sd, strict=True)
        net.eval().to(dev)
        for _ in range(2):
            net(x)
        net.set_kernel_timing(dev, True)
        net.kernel_time()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / args.steps
        k_ms, k_n, k_fl = net.kernel_time()
        net.set_kernel_timing(dev, False)
        ent = {"ms": round(1e3 * dt, 3), "mp_s": round(out_mp / dt, 3), "tflops_whole_net": round(frame_flops / dt / 1e12, 2),
               "trunk_tflops": round(k_fl / (k_ms * 1e-3) / 1e12, 2), "frac": round(k_fl / (k_ms * 1e-3) / 1e12 / PEAK_TFLOPS["f32"], 4),
               "peak": PEAK_TFLOPS["f32"], "algorithm": F32_ALGOS[name]["algorithm"]}
        if yref is not None:
            ent["max_abs_vs_cpu_oracle"] = float(f"{(net(xc.to(dev)).float().cpu() - yref).abs().max().item():.3e}")
        out[name] = ent
        del net
        torch.cuda.empty_cache()
    return out


def stream_object(args, dev, rank, world, timed, scale, H, W, key, name):
    """One independent frame per GPU through RealESRGANer(tile=512, tile_pad=10), bf16, device-resident input: frames/s of the
    whole job and a `roofline` object for its dominant kernel (the dense blocks), HIP events around them on the launch stream
    over the timed region.  Returns (RealESRGANer, dict)."""
    import numpy as np
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    netscale = {2: 2, 1: 1}.get(scale, 4)
    sd = synthetic_state_dict(seed=0, num_in_ch=3, scale=scale)
    up = RealESRGANer(scale=netscale, model_path={"params_ema": sd}, model=RRDBNet(3, 3, scale=scale, compute_dtype="bf16"), tile=512,
                      tile_pad=10, pre_pad=0, half=False, device=dev)
    steps, warm = max(2, min(args.steps, 6)), 2
    flops = up.model.forward_flops(1, H, W)
    mine = synthetic_frame(H, W, seed=rank)
    up.pre_process(np.ascontiguousarray(mine[:, :, ::-1].astype(np.float32) / 255.0))
    for _ in range(warm):
        up.tile_process()
    up.model.set_kernel_timing(dev, True)
    up.model.kernel_time()
    el = timed(up.tile_process, steps, 0)
    k_ms, k_n, k_fl = up.model.kernel_time()
    up.model.set_kernel_timing(dev, False)
    up.model.check_status()
    out_mp = netscale * netscale * H * W / 1e6
    strip = up.model.strip_kernel_active()
    obj = {"workload": f"one independent {W}x{H}->{W * netscale}x{H * netscale} frame per GPU, {name}, bf16, tile 512/10 (device-resident input)",
           "frames_per_s": round(world * steps / el, 3), "mp_s": round(world * steps * out_mp / el, 2),
           "ms_per_frame": round(1e3 * el / steps, 2), "tflops_whole_net": round(world * steps * flops / el / 1e12, 1),
           "frac_of_bf16_peak_per_gpu": round(steps * flops / el / 1e12 / PEAK_TFLOPS["bf16"], 4),
           "steps": steps, "scaling": "weak", "n_gpus": world}
    if k_ms > 0 and k_n > 0:
        ach = k_fl / (k_ms * 1e-3) / 1e12
        obj["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_TFLOPS["bf16"], 4), "traffic": pmc_traffic(key),
                           "kernel": ("rdb_bf16_strip_kernel: conv1..conv5 of a dense block over ALL tiles of the frame in one persistent launch, "
                                      "x0..x4 of a workgroup's strip resident in LDS" if strip else
                                      "conv3x3_bf16_xl_kernel (per-layer launches)"),
                           "avg_launch_us": round(1e3 * k_ms / k_n, 2), "launches": int(k_n),
                           "algorithmic_gflop_per_launch": round(k_fl / k_n / 1e9, 2),
                           "note": "`achieved` = algorithmic FLOPs of the tiles' own pixels in the dense blocks / HIP-event time around them; "
                                   "`traffic` = bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE), null when stale"}
    return up, obj


def c3_objects(args, dev, rank, world, timed):
    """BASELINE.json configs[2] and [3] beside the headline: one 2160p x2plus frame sharded over the ranks (strong scaling), one
    independent 2160p x2plus frame per GPU and one independent 1080p x4plus frame per GPU (weak scaling), bf16,
    RealESRGANer(tile=512, tile_pad=10)."""
    import torch
    from neural_enhanced_super_resolution_amd import sharded
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame
    H, W = 2160, 3840
    up, c3s = stream_object(args, dev, rank, world, timed, 2, H, W, "c3", "RealESRGAN_x2plus")
    out = {"c3_stream": c3s}
    steps, warm = max(2, min(args.steps, 6)), 2
    flops = up.model.forward_flops(1, H, W)
    frame = synthetic_frame(H, W, seed=0)
    band = torch.from_numpy(sharded.scatter_rows(frame, rank, world)).to(dev)
    el = timed(lambda: sharded.enhance_sharded(up, band, (H, W)), steps, warm)
    out["c3"] = {"workload": "one 3840x2160->7680x4320 frame for the whole job, bf16, tile 512/10, tiles sharded over ranks "
                             "(uint8 rows in, RCCL p2p overlap rows, uint8 frame gathered to rank 0's host)",
                 "frames_per_s": round(steps / el, 3), "mp_s": round(steps * 4 * H * W / 1e6 / el, 2), "ms_per_frame": round(1e3 * el / steps, 2),
                 "tflops_whole_net": round(steps * flops / el / 1e12, 1), "steps": steps, "scaling": "strong", "n_gpus": world}
    up.model.check_status()
    del up
    torch.cuda.empty_cache()
    up4, c4s = stream_object(args, dev, rank, world, timed, 4, 1080, 1920, "c4", "RealESRGAN_x4plus")
    out["c4_stream"] = c4s
    del up4
    torch.cuda.empty_cache()
    return out


def bench_c5(args, wl, dev, rank, world, timed):
    """BASELINE.json configs[4]: the ESRGAN stage of SuperResolutionPipeline.enhance_image, three iterations
    (nesr_adapter.enhance_iterations; the reference's cv2 pre/post filters need cv2 and stay off)."""
    import torch
    from neural_enhanced_super_resolution_amd import RRDBNet, RealESRGANer, nesr_adapter
    from neural_enhanced_super_resolution_amd.synth import synthetic_frame, synthetic_state_dict
    sd = synthetic_state_dict(seed=0, num_in_ch=12, scale=4)
    up = RealESRGANer(scale=2, model_path={"params_ema": sd}, model=RRDBNet(12, 3), tile=0, tile_pad=0, pre_pad=0, half=False,
                      device=dev)                                   # nesr/nesr.py:216-229
    img = synthetic_frame(wl["h"], wl["w"], seed=rank)[:, :, ::-1].copy()
    trace = []
    cfg = {"iterations": 3, "upscale_factor": 2.0}
    last = {}

    def step():
        trace.clear()
        last["out"] = nesr_adapter.enhance_iterations(up, img, cfg, "cuda", trace=trace, filters=args.filters)

    steps = max(1, min(args.steps, 3))
    el = timed(step, steps, min(args.warmup, 1))
    up.model.check_status()
    out_mp = last["out"].shape[0] * last["out"].shape[1] / 1e6
    flops = sum(t["net_input_px"] for t in trace) * up.model.forward_flops(1, 1, 1)      # tile overlaps included: executed, not algorithmic
    if rank == 0:
        print(json.dumps({
            "metric": "output megapixels/sec, nesr pipeline ESRGAN stage x3 iterations (x2plus tensor shapes, 12-channel 4x quirk), host uint8 in -> host uint8 out",
            "value": round(world * steps * out_mp / el, 3), "unit": "MP/s", "n_gpus": world, "steps": steps, "warmup": min(args.warmup, 1),
            "ms_per_step": round(1e3 * el / steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": F32_ALGOS["split"]["dtype"], "data": "synthetic (seeded frame, seeded random-init weights)",
            "config": {"workload": f"c5: {wl['desc']}",
                       "routes": [{k: t[k] for k in ("in_shape", "out_shape", "tiled", "three_channel", "model_calls")} for t in trace],
                       "filters": bool(args.filters),
                       "note": ("with the reference's pre / post filters (NL-means + CLAHE, adaptive unsharp) restated on the device, parity unpinned"
                                if args.filters else "cv2 pre/post filters off (--filters turns their device-side restatement on)")
                               + "; network evaluations are the reference's, tile for tile"},
            "tflops_whole_net": round(world * steps * flops / el / 1e12, 1)}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
