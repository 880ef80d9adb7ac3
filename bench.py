#!/usr/bin/env python3
"""Headline benchmark: Depth-Anything-V2-Small f16, 518x518, images/sec (BASELINE.json).
(`--workload esrgan` measures the next row to the same contract: Real-ESRGAN-4x f16, 256^2 -> 1024^2, batch 16,
BASELINE.json configs[2].)

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started without RANK in the environment, `--gpus N` launches the second form itself as a
CHILD process (before this process has imported torch or loaded the HIP library; never an exec) and exits with its code.

One process per GPU. A step = one pass of the hot path (depthany_compute semantics per image:
pre-process, DINOv2-S encoder, DPT neck/head, min-max normalise) over one batch of 32 synthetic
518x518x3 uint8 images that are already resident in HBM. Images are independent units, so the
batch is sharded across ranks with no data-path collective (weak scaling: 32 images per GPU);
RCCL is used once, at load, to broadcast the packed weight arena from rank 0.

Rank 0 prints ONE JSON line. `roofline` is measured live with HIP events on the compute stream
(per kernel group, non-graph pass); `cpu_baseline` times the CPU oracle (oracle/, a port of the
reference's ggml CPU path) on a bounded sample on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path



def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start N fresh rank processes through
    torch.distributed.run as a child of this (still GPU-free) process and return its exit code. The ranks inherit stdout,
    so rank 0's JSON line is this command's one line of output."""
    n = 1
    for i, a in enumerate(sys.argv[1:]):
        if a == "--gpus" and i + 2 < len(sys.argv):
            n = int(sys.argv[i + 2])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "RANK" in os.environ:
        return None
    import socket
    import subprocess

    with socket.socket() as so:  # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    _rc = _self_launch()
    if _rc is not None:
        raise SystemExit(_rc)

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import _lib as L  # noqa: E402
from visioncpp_amd import synth, vision  # noqa: E402

GFLOP_PER_IMAGE = 115.27  # BASELINE.md section 2 (2 x MACs of matmuls/convs)
PEAK_MFMA_F16 = 2.5e15    # dense f16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def pipeline_leg(model, imgs, B, W, H, n_pipe):
    """upload + compute + download per batch, as the reference's benchmark times a call (tests/benchmark.cpp:55-91), through the overlapped host
    pipeline (pinned staging, H2D / compute / D2H on three streams): never `value`."""
    pipe = vision.DepthPipeline(model, B, W, H, n_slots=3)
    tickets = []
    for _ in range(3):  # warm-up incl. first-touch of the pinned buffers
        pipe.input_view()[...] = imgs
        tickets.append(pipe.submit(None))
        if len(tickets) == 3:
            pipe.wait(tickets.pop(0), copy=False)
    while tickets:
        pipe.wait(tickets.pop(0), copy=False)
    t0 = time.perf_counter()  # the fill (one upload + one step + one download before the first result) is amortised over the run
    for i in range(n_pipe):
        tickets.append(pipe.submit(imgs))  # pageable numpy batch -> pinned staging (host memcpy) -> H2D
        if len(tickets) == 3:
            pipe.wait(tickets.pop(0), copy=False)
    last = None
    while tickets:
        last = pipe.wait(tickets.pop(0), copy=False)  # the result stays in pinned memory: a fresh 34 MB numpy copy is page faults, not pipeline
    dt = time.perf_counter() - t0
    assert np.isfinite(last).all() and last.min() >= 0 and last.max() <= 1 + 1e-6
    pipe.close()
    return {"value": round(B * n_pipe / dt, 2), "ms_per_step": round(1e3 * dt / n_pipe, 3), "steps": n_pipe,
            "note": "rank 0; host numpy batch -> pinned staging -> H2D -> forward -> D2H -> pinned, 3 slots in flight (visp_depthany_pipeline_*)"}


def pipeline_child(args):
    """`bench.py --pipeline-child`: the host-pipeline leg in a process of its own that never imports torch (the parent waits, idle)."""
    B, W, H = args.batch or 32, 518, 518
    tmp = Path(tempfile.gettempdir()) / f"visp_bench_da_v2_small_f16_child_{os.getpid()}.gguf"
    synth.write_gguf(tmp, synth.SMALL, seed=0)
    dev = vision.Device.init(index=args.device or 0)
    model = vision.Model.load(tmp, dev, vision.Arch.depth_anything)
    tmp.unlink()
    imgs = synth.images(min(B, 8), W, H, seed=1234)
    imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
    model.use_graph(True)
    rgb = vision.DeviceBuffer.from_numpy(imgs)
    out = vision.DeviceBuffer(B * W * H * 4)
    for _ in range(5):
        model.compute_batch_device(rgb.ptr, B, W, H, out.ptr)
    t0 = time.perf_counter()
    for _ in range(30):
        model.compute_batch_device(rgb.ptr, B, W, H, out.ptr)  # blocking call per step (no caller stream): the resident step as this process sees it
    resident = (time.perf_counter() - t0) / 30
    res = pipeline_leg(model, imgs, B, W, H, args.steps)
    res["resident_ms_per_step"] = round(1e3 * resident, 3)
    maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
    res["runtime"] = "child process without torch (what a C / ctypes caller gets): " + ", ".join(maps)
    assert "torch" not in sys.modules
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["depthany", "esrgan", "sam", "swin", "birefnet"], default="depthany",
                    help="depthany = the headline metric (BASELINE.json configs[1]); esrgan = configs[2], the next SURVEY section 8 row")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 32 for depthany, 16 for esrgan, 128 for sam, 8 for swin and birefnet)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--min-seconds", type=float, default=10.0, help="soak: after the K timed steps keep stepping for this long and report that rate too (0 = skip)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the upload + compute + download (host pipeline) figure")
    ap.add_argument("--schedule", type=int, default=-1, help="depthany encoder schedule: -1 = the library's default (block kernel), 0 = GEMM launches, 1 = token-stationary block kernel")
    ap.add_argument("--fp8", action="store_true", help="sam workload only, opt-in: the transformer stages' MLPs on the e4m3 matrix instruction (BASELINE.json configs[4]); reports the embedding error next to the rate")
    ap.add_argument("--cpu-images", type=int, default=32, help="bounded CPU-baseline sample (about 10 s at 16 threads)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the CPU baseline (one GPU's share of the host)")
    ap.add_argument("--profile-groups", action="store_true", help="print the per-kernel-group table to stderr")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="force this HIP device for every rank (gloo rehearsal only)")
    ap.add_argument("--pipeline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pipeline_child:
        return pipeline_child(args)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    device_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(args.dist_backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    api = L.get_lib()
    if args.workload in ("esrgan", "sam", "swin", "birefnet"):
        {"esrgan": run_esrgan, "sam": run_sam, "swin": run_swin, "birefnet": run_birefnet}[args.workload](args, torch, dist, rank, world, device_index, barrier, api)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    B, W, H = args.batch or 32, 518, 518
    cfg = synth.SMALL

    # ---- load: rank 0 reads the GGUF and uploads; other ranks allocate and receive the arena over RCCL
    tmp = Path(tempfile.gettempdir()) / f"visp_bench_da_v2_small_f16_{os.environ.get('MASTER_PORT', '0')}.gguf"
    if rank == 0:
        synth.write_gguf(tmp, cfg, seed=0)
    barrier()
    dev = vision.Device.init(index=device_index)
    model = vision.Model.load(tmp, dev, vision.Arch.depth_anything, no_upload=(rank != 0))
    if world > 1:
        broadcast_arena(model, torch, dist, rank, api)

    # ---- inputs resident in HBM before the timed region
    imgs = synth.images(min(B, 8), W, H, seed=1234 + 100 * rank)
    imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
    rgb = torch.from_numpy(imgs).cuda()
    out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    compute_stream = torch.cuda.Stream()  # a real (non-null) stream: required for hipGraph capture
    stream = compute_stream.cuda_stream
    model.reserve(B, W, H)
    if args.schedule >= 0:
        model.set_schedule(args.schedule)

    def step():
        model.compute_batch_device(rgb.data_ptr(), B, W, H, out.data_ptr(), None, stream)

    # ---- per-kernel-group timing (direct launches, HIP events on the compute stream), untimed
    groups = []
    if rank == 0:
        step()
        torch.cuda.synchronize()
        model.enable_timing(True)
        acc: dict[str, dict] = {}
        reps = 3
        for _ in range(reps):
            step()
            torch.cuda.synchronize()
            for t in model.read_timing():
                a = acc.setdefault(t["name"], dict(name=t["name"], ms=0.0, launches=0, flops=0.0, bytes=0.0))
                a["ms"] += t["ms"] / reps
                a["launches"] = t["launches"]
                a["flops"] = t["flops"]
                a["bytes"] = t["bytes"]
        # the same table at the launch shapes the timed step issues: sub-batches on parallel streams, each launch timed on its own
        # stream while the other streams' kernels share the chip (direct launches; the hipGraph replays the same shapes)
        model.enable_timing(2)
        acc_split: dict[str, dict] = {}
        for _ in range(reps):
            step()
            torch.cuda.synchronize()
            for t in model.read_timing():
                a = acc_split.setdefault(t["name"], dict(name=t["name"], ms=0.0, launches=0, flops=0.0))
                a["ms"] += t["ms"] / reps
                a["launches"] = t["launches"]
                a["flops"] = t["flops"]
        model.enable_timing(False)
        groups = sorted(acc.values(), key=lambda g: -g["ms"])
        if args.profile_groups:
            tot = sum(g["ms"] for g in groups)
            print(f"{'group':16s} {'ms':>8s} {'%':>6s} {'launch':>6s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for g in groups:
                print(f"{g['name']:16s} {g['ms']:8.3f} {100 * g['ms'] / tot:6.1f} {g['launches']:6d} "
                      f"{g['flops'] / g['ms'] / 1e9:9.1f} {g['bytes'] / g['ms'] / 1e6:9.1f}", file=sys.stderr)
            print(f"{'total':16s} {tot:8.3f}", file=sys.stderr)

    if not args.no_graph:
        model.use_graph(True)
    for _ in range(args.warmup):
        step()
    ev = StepEvents(api, stream, args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.mark()
        step()
    ev.mark()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = ev.step_ms()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- soak: the same step for --min-seconds (the reference's benchmark runs >= 10 s; the shader clock settles only after a
    # few hundred launches), outside the K timed steps the contract defines
    soak = None
    if args.min_seconds > 0:
        barrier()
        t0 = time.perf_counter()
        n_soak = 0
        while True:
            for _ in range(20):
                step()
            n_soak += 20
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= args.min_seconds:
                break
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt / n_soak], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item()) * n_soak
        soak = {"seconds": round(dt, 2), "steps": n_soak, "ms_per_step": round(1e3 * dt / n_soak, 3), "value": round(world * B * n_soak / dt, 2)}

    # ---- upload + compute + download per batch, as the reference's benchmark times a call (tests/benchmark.cpp:55-91), through
    # the overlapped host pipeline (pinned staging, H2D / compute / D2H on three streams): never `value`
    incl = None
    if not args.no_pipeline and rank == 0:
        incl = pipeline_leg(model, imgs, B, W, H, max(args.steps, 60))
        incl["runtime"] = "this process (torch loaded first: the library is bound to the HIP runtime torch bundles)"
        # the same leg from a process that never loads torch -- what a C or ctypes caller of the library gets: the system ROCm runtime. The 0.90 x of
        # the resident rate measured in round 3 was the bundled runtime's copy path, not the pipeline.
        try:
            if world > 1:
                raise ValueError("N > 1: no extra process next to the ranks")
            r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--pipeline-child", "--batch", str(B), "--steps", str(max(args.steps, 60)), "--device", str(device_index)],
                               capture_output=True, text=True, timeout=600)
            child = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
        except (subprocess.TimeoutExpired, ValueError):
            child = None
        if child:
            incl = {"value": child["value"], "ms_per_step": child["ms_per_step"], "steps": child["steps"], "note": child["note"], "runtime": child["runtime"],
                    "resident_ms_per_step_same_process": child["resident_ms_per_step"], "in_torch_process": {k: incl[k] for k in ("value", "ms_per_step", "runtime")}}
    model.use_graph(False)

    # sanity: the timed output is a valid normalised depth batch
    o = out.cpu().numpy()
    # timing-only ablations exist only in diagnostic builds of the library (make ABLATE=1); there VISP_ABLATE_LAUNCHES="a-b,c" skips launches of the step and the results are invalid
    ablated = bool(os.environ.get("VISP_ABLATE_LAUNCHES"))
    assert ablated or (np.isfinite(o).all() and o.min() >= 0 and o.max() <= 1 + 1e-6), "invalid output"

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        res = {
            "metric": "images/sec, Depth-Anything-V2-Small 518x518 f16",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "step_ms_device": {"mean": round(float(np.mean(step_ms)), 3), "std": round(float(np.std(step_ms)), 3), "min": round(float(np.min(step_ms)), 3),
                               "note": "rank 0, HIP events at the step boundaries of the same timed region"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "Depth-Anything-V2-Small f16 (DINOv2-S ViT) 518x518 batch=32 per MI355X (BASELINE.json configs[1])",
                       "images_per_gpu_per_step": B, "global_batch": world * B, "weights": "random-init synthetic GGUF (seed 0)",
                       "parallelism": f"dp{world} (image shards, no data-path collective)", "hip_graph": not args.no_graph,
                       "encoder_schedule": "graph executor: one launch per epilogue-fused node (GEMM launches)" if args.schedule == 0 else "graph executor: node groups lowered onto attention + token-stationary block kernel per layer, LDS-ring convs, head kernel; 3 sub-batch graphs on parallel streams"},
            "value_soak": soak,
            "value_incl_h2d_d2h": incl,
            "model_tflops": round(value * GFLOP_PER_IMAGE / 1e3, 2),
            "mfma_frac_whole_model": round(value * GFLOP_PER_IMAGE * 1e9 / (world * PEAK_MFMA_F16), 4),
        }
        if ablated:
            res["INVALID_ablation"] = os.environ["VISP_ABLATE_LAUNCHES"]
        if groups:
            dom = groups[0]
            per_launch_ms = dom["ms"] / max(dom["launches"], 1)
            if dom["flops"] > 0 and dom["flops"] / max(dom["bytes"], 1) > PEAK_MFMA_F16 / PEAK_HBM / 4:
                ach = dom["flops"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e12
                res["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F16 / 1e12,
                                   "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_MFMA_F16, 4), "traffic": None,
                                   "avg_launch_ms": round(per_launch_ms, 4), "launches_per_step": dom["launches"]}
            else:
                ach = dom["bytes"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e9
                res["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM / 1e9,
                                   "unit": "GB/s", "frac": round(ach * 1e9 / PEAK_HBM, 4), "traffic": None,
                                   "avg_launch_ms": round(per_launch_ms, 4), "launches_per_step": dom["launches"]}
            # HBM traffic and MFMA-busy % per launch of that kernel from the PMC passes (rocprofv3 --pmc, one counter set per
            # pass, on tools/bench_block.py --pmc = the same launch shapes; units and corrections in tools/pmc_summary.py)
            pmc = next((q for q in (ROOT / "profiles" / d / "traffic.json" for d in ("r04_pmc", "r03_pmc", "r02_pmc", "r01_pmc")) if q.exists()), ROOT / "profiles" / "none")
            if pmc.exists():
                ks = json.loads(pmc.read_text())["kernels"]
                # one entry per launch shape ("block@343wg"): this object describes the unsplit batch-32 launch (the timing pass runs
                # the batch on one stream), i.e. the entry with the most workgroups
                shaped = [v for n, v in ks.items() if n.startswith(dom["name"] + "@")]
                k = ks.get(dom["name"]) or (max(shaped, key=lambda v: v.get("workgroups", 0)) if shaped else None)
                if k:
                    res["roofline"]["launch_shape"] = (f"batch {B} on one stream ({k.get('workgroups', '?')} workgroups), as in the per-group timing pass; the timed "
                                                       "step issues the same work as 3 sub-batch launches on parallel streams (kernel_stats_by_grid in profiles/)")
                    res["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
                    res["roofline"]["traffic_source"] = f"profiles/{pmc.parent.name}/traffic.json (FETCH_SIZE{' x2' if k.get('fetch_doubled', True) else ''} + WRITE_SIZE)"
                    if "mfma_busy_pct" in k:
                        res["roofline"]["mfma_busy_pct"] = k["mfma_busy_pct"]
                        res["roofline"]["mfma_busy_source"] = f"profiles/{pmc.parent.name}/ (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))"
            sp = acc_split.get(dom["name"])
            if sp and sp["ms"] > 0 and sp["launches"] > 0:
                ach_s = sp["flops"] / (sp["ms"] * 1e-3) / 1e12
                res["roofline"]["in_graph"] = {
                    "launches_per_step": sp["launches"], "avg_launch_ms": round(sp["ms"] / sp["launches"], 4), "achieved": round(ach_s, 2),
                    "frac": round(ach_s * 1e12 / PEAK_MFMA_F16, 4),
                    "note": "the same kernel at the launch shapes of the timed step (3 sub-batches on parallel streams), measured live: HIP events around "
                            "each launch on its own stream while the other streams' kernels share the chip, so a launch's duration includes what it "
                            "waits for them; frac = the kernel's FLOPs per step / the sum of its launch durations / peak"}
            res["roofline"]["measured_live"] = ["achieved", "frac", "avg_launch_ms", "in_graph"]
            res["roofline"]["read_from_files"] = [k for k in ("traffic", "mfma_busy_pct") if k in res["roofline"] and res["roofline"][k] is not None]
            res["kernel_groups_ms"] = {g["name"]: round(g["ms"], 3) for g in groups}
            res["kernel_groups_ms_in_graph_shapes"] = {k: round(v["ms"], 3) for k, v in sorted(acc_split.items(), key=lambda kv: -kv[1]["ms"]) if not k.startswith("__end")}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, imgs, o, args.cpu_images, args.cpu_threads)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


class StepEvents:
    """HIP events on the compute stream at every step boundary of the timed region (no host sync inside it): gives the
    per-step spread (SURVEY section 8d asks for mean +- stdev) next to the wall-clock figure the contract defines."""

    def __init__(self, api, stream, n):
        import ctypes

        self.api, self.stream, self.c = api, stream, ctypes
        self.ev = [ctypes.c_void_p() for _ in range(n + 1)]
        for e in self.ev:
            L.vx_check(api.vx_event_create(ctypes.byref(e)))
        self.i = 0

    def mark(self):
        L.vx_check(self.api.vx_event_record(self.ev[self.i], self.stream))
        self.i += 1

    def step_ms(self):
        out = []
        for a, b in zip(self.ev[:self.i - 1], self.ev[1:self.i]):
            ms = self.c.c_float()
            L.vx_check(self.api.vx_event_elapsed_ms(a, b, self.c.byref(ms)))
            out.append(ms.value)
        for e in self.ev:
            self.api.vx_event_destroy(e)
        return out


def broadcast_arena(model, torch, dist, rank, api):
    """Rank 0 read the GGUF and uploaded; the other ranks receive the packed weight arena over RCCL."""
    ptr, nbytes = model.weights_arena()
    staging = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    if rank == 0:
        L.vx_check(api.vx_memcpy_d2d(staging.data_ptr(), ptr, nbytes, None))
        L.vx_check(api.vx_stream_sync(None))
    dist.broadcast(staging, src=0)
    torch.cuda.synchronize()
    if rank != 0:
        L.vx_check(api.vx_memcpy_d2d(ptr, staging.data_ptr(), nbytes, None))
        L.vx_check(api.vx_stream_sync(None))
        model.weights_ready()


ESRGAN_GFLOP_PER_IMAGE = 2349.7  # RRDBNet 23 blocks, 256^2 -> 1024^2, untiled (2 x MACs)


def run_esrgan(args, torch, dist, rank, world, device_index, barrier, api):
    """A step = esrgan_compute semantics (vision.cpp:220-253: 224/16 tiling, RRDBNet per tile, blend, rgba_u8) for a
    batch of 16 synthetic 256x256 RGB images resident in HBM; all 64 tiles of the batch go through the network
    together. Images are independent: ranks take whole images, no data-path collective."""
    B, W, H = args.batch or 16, 256, 256
    cfg = synth.ESRGAN_X4
    tmp = Path(tempfile.gettempdir()) / f"visp_bench_realesrgan_x4_f16_{os.environ.get('MASTER_PORT', '0')}.gguf"
    if rank == 0:
        synth.write_esrgan_gguf(tmp, cfg, seed=1)
    barrier()
    dev = vision.Device.init(index=device_index)
    model = vision.Model.load(tmp, dev, vision.Arch.esrgan, no_upload=(rank != 0))
    if world > 1:
        broadcast_arena(model, torch, dist, rank, api)
    imgs = synth.images(B, W, H, seed=4321 + 100 * rank)
    src = torch.from_numpy(imgs).cuda()
    s = cfg.scale
    out = torch.empty((B, H * s, W * s, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream().cuda_stream

    def step():
        model.upscale_batch_device(src.data_ptr(), B, W, H, out.data_ptr(), vision.ImageFormat.rgb_u8, stream)

    groups = []
    if rank == 0:
        step()
        torch.cuda.synchronize()
        model.enable_timing(True)
        step()  # (the timing pass runs every tile group on one stream: its first step builds that shape's graph -- not part of what is timed)
        torch.cuda.synchronize()
        step()
        torch.cuda.synchronize()
        groups = sorted(model.read_timing(), key=lambda g: -g["ms"])
        model.enable_timing(False)
        if args.profile_groups:
            tot = sum(g["ms"] for g in groups)
            print(f"{'group':16s} {'ms':>8s} {'%':>6s} {'launch':>6s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for g in groups:
                print(f"{g['name']:16s} {g['ms']:8.3f} {100 * g['ms'] / tot:6.1f} {g['launches']:6d} "
                      f"{g['flops'] / g['ms'] / 1e9:9.1f} {g['bytes'] / g['ms'] / 1e6:9.1f}", file=sys.stderr)
            print(f"{'total':16s} {tot:8.3f}", file=sys.stderr)
    for _ in range(args.warmup):
        step()
    ev = StepEvents(api, stream, args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.mark()
        step()
    ev.mark()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = ev.step_ms()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    o = out.cpu().numpy()
    assert (o[..., 3] == 255).all() and o[..., :3].std() > 1, "invalid output"
    if rank != 0:
        return
    value = world * B * args.steps / elapsed
    tiled_gflop = sum(g["flops"] for g in groups) / B / 1e9 if groups else None
    res = {
        "metric": "images/sec, Real-ESRGAN-4x (RRDBNet 23 blocks) 256x256 -> 1024x1024 f16",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "step_ms_device": {"mean": round(float(np.mean(step_ms)), 3), "std": round(float(np.std(step_ms)), 3), "min": round(float(np.min(step_ms)), 3),
                           "note": "rank 0, HIP events at the step boundaries of the same timed region"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "Real-ESRGAN-4x f16 (RRDB conv stack) 256x256 -> 1024x1024 batch=16 per MI355X (BASELINE.json configs[2])",
                   "images_per_gpu_per_step": B, "global_batch": world * B, "weights": "random-init synthetic GGUF (seed 1)",
                   "tiling": "224 max / overlap 16 / align 16 as the reference: 4 tiles of 144x144 per image, 64 tiles per step",
                   "parallelism": f"dp{world} (image shards, no data-path collective)"},
        "model_tflops": round(value * ESRGAN_GFLOP_PER_IMAGE / 1e3, 2),
        "model_tflops_incl_tile_overlap": round(value * tiled_gflop / 1e3, 2) if tiled_gflop else None,
        "mfma_frac_whole_model": round(value * (tiled_gflop or ESRGAN_GFLOP_PER_IMAGE) * 1e9 / (world * PEAK_MFMA_F16), 4),
    }
    if groups:
        dom = groups[0]
        per_launch_ms = dom["ms"] / max(dom["launches"], 1)
        ach = dom["flops"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e12
        res["roofline"] = {"kernel": f"dconv3x3_kernel ({dom['name']})", "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F16 / 1e12,
                           "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_MFMA_F16, 4), "traffic": None,
                           "avg_launch_ms": round(per_launch_ms, 4), "launches_per_step": dom["launches"]}
        pmc = ROOT / "profiles" / "r01_pmc" / "traffic_esrgan.json"
        if pmc.exists():
            k = json.loads(pmc.read_text())["kernels"].get(dom["name"])
            if k:
                res["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
                res["roofline"]["traffic_source"] = "profiles/r01_pmc/traffic_esrgan.json (FETCH_SIZE x2 + WRITE_SIZE)"
        res["kernel_groups_ms"] = {g["name"]: round(g["ms"], 3) for g in groups}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle

        sd = synth.esrgan_state_dict(cfg, 1)
        tensors, conv2d = synth.esrgan_gguf_tensors(sd)
        om = oracle.Model(tensors, conv2d, "whcn")
        oracle.set_num_threads(args.cpu_threads)
        n = max(1, min(2, args.cpu_images // 16))
        t0 = time.perf_counter()
        diffs = []
        for i in range(n):
            want = oracle.esrgan_compute(om, cfg.scale, cfg.num_blocks, imgs[i], oracle.RGB_U8)
            diffs.append(np.abs(o[i].astype(np.int32) - want.astype(np.int32)))
        dt = time.perf_counter() - t0
        d = np.stack(diffs)
        res["cpu_baseline"] = {"value": round(n / dt, 4), "unit": "images/s", "cores": args.cpu_threads, "kind": "port",
                               "sample": f"{n} image(s) 256x256 -> 1024x1024 (4 tiles each), OpenMP {args.cpu_threads} threads, oracle/libvisp_oracle.so",
                               "u8_mean_abs_diff_gpu_vs_cpu": round(float(d.mean()), 4), "u8_max_abs_diff_gpu_vs_cpu": int(d.max())}
    print(json.dumps(res), flush=True)


def run_sam(args, torch, dist, rank, world, device_index, barrier, api):
    """A step = sam_encode's graph (mobile-sam.cpp:20-215: TinyViT-5M, 1024x1024 -> embedding [64, 64, 256]) for a batch of
    synthetic rgb_u8 images resident in HBM. Images are independent: ranks take whole images, no data-path collective."""
    B, S = args.batch or 128, 1024  # configs[4]: a 1k-image batch over 8 GPUs = 128 images per GPU and step
    cfg = synth.TINYVIT_5M
    tmp = Path(tempfile.gettempdir()) / f"visp_bench_mobile_sam_f16_{os.environ.get('MASTER_PORT', '0')}.gguf"
    if rank == 0:
        synth.write_mobile_sam_gguf(tmp, cfg, enc_sd=synth.tinyvit_state_dict(cfg, 3), dec_sd=synth.sam_decoder_state_dict(1003))
    barrier()
    dev = vision.Device.init(index=device_index)
    model = vision.Model.load(tmp, dev, vision.Arch.sam, no_upload=(rank != 0))
    if world > 1:
        broadcast_arena(model, torch, dist, rank, api)
    imgs = synth.images(min(B, 4), S, S, seed=99 + 100 * rank)
    imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
    src = torch.from_numpy(imgs).cuda()
    out = torch.empty((B, 64, 64, 256), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream().cuda_stream

    def step():
        model.sam_encode_batch_device(src.data_ptr(), B, out.data_ptr(), stream)

    fp8_err = None
    if args.fp8:  # opt-in (configs[4] "fp8 GGUF weights on CDNA4 fp8 MFMA"): the stage MLPs on the e4m3 matrix instruction; the f16 result first, for the error
        step()
        torch.cuda.synchronize()
        ref = out[:8].clone()
        model.sam_set_fp8_mlp(True)
        step()
        torch.cuda.synchronize()
        fp8_err = float((out[:8] - ref).abs().mean() / ref.abs().mean())
    groups = []
    if rank == 0:
        step()
        torch.cuda.synchronize()
        model.enable_timing(True)
        step()
        torch.cuda.synchronize()
        groups = sorted(model.read_timing(), key=lambda g: -g["ms"])
        model.enable_timing(False)
        if args.profile_groups:
            tot = sum(g["ms"] for g in groups)
            print(f"{'group':16s} {'ms':>8s} {'%':>6s} {'launch':>6s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for g in groups:
                print(f"{g['name']:16s} {g['ms']:8.3f} {100 * g['ms'] / tot:6.1f} {g['launches']:6d} "
                      f"{g['flops'] / g['ms'] / 1e9:9.1f} {g['bytes'] / g['ms'] / 1e6:9.1f}", file=sys.stderr)
            print(f"{'total':16s} {tot:8.3f}", file=sys.stderr)
    for _ in range(args.warmup):
        step()
    ev = StepEvents(api, stream, args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.mark()
        step()
    ev.mark()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = ev.step_ms()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    o = out.cpu().numpy()
    assert np.isfinite(o).all() and o.std() > 0.1, "invalid output"
    if rank != 0:
        return
    value = world * B * args.steps / elapsed
    gflop = sum(g["flops"] for g in groups) / B / 1e9 if groups else None
    res = {
        "metric": "images/sec, MobileSAM TinyViT-5M image encoder 1024x1024 f16",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "step_ms_device": {"mean": round(float(np.mean(step_ms)), 3), "std": round(float(np.std(step_ms)), 3), "min": round(float(np.min(step_ms)), 3),
                           "note": "rank 0, HIP events at the step boundaries of the same timed region"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (stage MLPs e4m3)" if args.fp8 else "f16", "data": "synthetic",
        "config": {"workload": f"MobileSAM TinyViT encoder 1024x1024 f16, batch={B} per MI355X (BASELINE.json configs[4] names fp8 weights and a "
                               "1k-image batch over 8 GPUs: this row is the f16 encoder, image shards per rank)" + (
                                   " -- with --fp8: the transformer stages' MLPs on the block-scaled e4m3 matrix instruction (opt-in, never the default)" if args.fp8 else ""),
                   "images_per_gpu_per_step": B, "global_batch": world * B, "weights": "random-init synthetic GGUF (seed 3)",
                   "parallelism": f"dp{world} (image shards, no data-path collective)"},
        "model_gflop_per_image": round(gflop, 2) if gflop else None,
        "model_tflops": round(value * gflop / 1e3, 2) if gflop else None,
    }
    if fp8_err is not None:
        res["fp8_mlp"] = {"embedding_mean_abs_diff_over_mean_abs_vs_f16": round(fp8_err, 4),
                          "note": "e4m3 weights per output channel + e4m3 activations per token in fc1 / fc2 of every transformer block; tests/test_fp8_decision.py: mask IoU below the bar"}
    if groups:
        tot = sum(g["ms"] for g in groups)
        # the profile is flat (no group above 12 %): the roofline object describes the step's longest single launch among the
        # groups that matter (>= 5 % of the step), which is one kernel at one shape -- the MBConv depthwise conv
        dom = max((g for g in groups if g["ms"] >= 0.05 * tot), key=lambda g: g["ms"] / max(g["launches"], 1))
        hbm_bound = dom["flops"] / max(dom["bytes"], 1) < PEAK_MFMA_F16 / PEAK_HBM
        if hbm_bound:
            ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            res["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                               "frac": round(ach * 1e9 / PEAK_HBM, 4), "traffic": None}
        else:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            res["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F16 / 1e12, "unit": "TFLOP/s",
                               "frac": round(ach * 1e12 / PEAK_MFMA_F16, 4), "traffic": None}
        res["roofline"]["avg_launch_ms"] = round(dom["ms"] / max(dom["launches"], 1), 4)
        res["roofline"]["launches_per_step"] = dom["launches"]
        res["roofline"]["share_of_step"] = round(dom["ms"] / tot, 3)
        res["roofline"]["algorithmic_bytes_per_launch"] = round(dom["bytes"] / max(dom["launches"], 1))
        pmc = ROOT / "profiles" / "r01_pmc" / "traffic_sam.json"
        key = {"depthwise_mbconv": "depthwise", "mbconv_dw_pw": "mbconv_dw_pw"}.get(dom["name"])
        if key and pmc.exists():  # counters were collected at batch 16; the kernel's traffic is linear in the batch
            k = json.loads(pmc.read_text())["kernels"].get(key)
            if k:
                res["roofline"]["traffic"] = round(k["hbm_bytes_per_launch"] * B / 16)
                res["roofline"]["traffic_source"] = "profiles/r01_pmc/traffic_sam.json (FETCH_SIZE x2 + WRITE_SIZE, collected at batch 16, scaled by batch)"
        res["kernel_groups_ms"] = {g["name"]: round(g["ms"], 3) for g in groups}
    # configs[0] of BASELINE.json (encode + decode of one 1024x1024 image), outside the timed region: latency of the reference
    # API calls sam_encode / sam_compute from host buffers (decode = prompt encoder + mask decoder on the GPU + the reference's
    # host-side mask resize / threshold)
    model.sam_encode(imgs[0])
    model.sam_compute([300, 300])  # first calls allocate the staging / decoder scratch
    t0 = time.perf_counter()
    for k in range(5):
        model.sam_encode(imgs[k % len(imgs)])
    t1 = time.perf_counter()
    for k in range(10):
        mask = model.sam_compute([400 + 40 * k, 500])
    t2 = time.perf_counter()
    assert mask.shape == (S, S) and set(np.unique(mask)) <= {0, 255}
    res["single_image_latency_ms"] = {"sam_encode": round(1e3 * (t1 - t0) / 5, 2), "sam_compute": round(1e3 * (t2 - t1) / 10, 2),
                                      "note": "steady state, host buffers in, u8 mask out; batch 1; not part of value"}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle

        tensors, conv2d = synth.tinyvit_gguf_tensors(synth.tinyvit_state_dict(cfg, 3))
        om = oracle.Model(tensors, conv2d, "whcn")
        oracle.set_num_threads(args.cpu_threads)
        params = oracle.tinyvit_params(cfg.img_size, cfg.layers())
        mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
        n = max(1, min(4, args.cpu_images // 8))
        t0 = time.perf_counter()
        errs = []
        for i in range(n):
            want = oracle.tinyvit_encode(om, params, (imgs[i].astype(np.float32) / np.float32(255.0) - mean) / std)
            errs.append(np.abs(o[i] - want))
        dt = time.perf_counter() - t0
        e = np.stack(errs)
        res["cpu_baseline"] = {"value": round(n / dt, 4), "unit": "images/s", "cores": args.cpu_threads, "kind": "port",
                               "sample": f"{n} image(s) 1024x1024, batch-1 sequential, OpenMP {args.cpu_threads} threads, oracle/libvisp_oracle.so",
                               "mean_abs_diff_gpu_vs_cpu": round(float(e.mean()), 5), "max_abs_diff_gpu_vs_cpu": round(float(e.max()), 4)}
    print(json.dumps(res), flush=True)


def run_swin(args, torch, dist, rank, world, device_index, barrier, api):
    """A step = birefnet_process_input's normalisation + swin_encode (swin.cpp:237-262: SWIN-T, 1024x1024 -> four normed stage
    maps) for a batch of synthetic rgb_u8 images resident in HBM: the encoder half of BASELINE.json configs[3] (BiRefNet-lite,
    batch 64 over 8 GPUs = 8 images per GPU and step; --workload birefnet adds the decoder). Ranks take whole images, no collective."""
    B, S = args.batch or 8, 1024
    cfg = synth.SWIN_T
    tmp = Path(tempfile.gettempdir()) / f"visp_bench_swin_t_f16_{os.environ.get('MASTER_PORT', '0')}.gguf"
    if rank == 0:
        synth.write_swin_gguf(tmp, cfg, seed=4)
    barrier()
    dev = vision.Device.init(index=device_index)
    model = vision.SwinEncoder.load(tmp, dev)  # 55 MB of weights: every rank reads the file
    imgs = synth.images(min(B, 2), S, S, seed=77 + 100 * rank)
    imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
    src = torch.from_numpy(imgs).cuda()
    dims = model.output_dims(S, S)
    outs = [torch.empty((B, h, w, c), dtype=torch.float32, device="cuda") for (w, h, c) in dims]
    ptrs = [o.data_ptr() for o in outs]
    stream = torch.cuda.Stream().cuda_stream

    def step():
        model.encode_batch_device(src.data_ptr(), B, S, S, ptrs, stream)

    groups = []
    if rank == 0:
        step()
        torch.cuda.synchronize()
        model.enable_timing(True)
        step()
        torch.cuda.synchronize()
        groups = sorted(model.read_timing(), key=lambda g: -g["ms"])
        model.enable_timing(False)
        if args.profile_groups:
            tot = sum(g["ms"] for g in groups)
            print(f"{'group':16s} {'ms':>8s} {'%':>6s} {'launch':>6s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for g in groups:
                print(f"{g['name']:16s} {g['ms']:8.3f} {100 * g['ms'] / tot:6.1f} {g['launches']:6d} "
                      f"{g['flops'] / g['ms'] / 1e9:9.1f} {g['bytes'] / g['ms'] / 1e6:9.1f}", file=sys.stderr)
            print(f"{'total':16s} {tot:8.3f}", file=sys.stderr)
    for _ in range(args.warmup):
        step()
    ev = StepEvents(api, stream, args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.mark()
        step()
    ev.mark()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = ev.step_ms()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    o3 = outs[3].cpu().numpy()
    assert np.isfinite(o3).all() and o3.std() > 0.1, "invalid output"
    if rank != 0:
        return
    value = world * B * args.steps / elapsed
    gflop = sum(g["flops"] for g in groups) / B / 1e9 if groups else None
    res = {
        "metric": "images/sec, SWIN-T encoder of BiRefNet-lite 1024x1024 f16",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "step_ms_device": {"mean": round(float(np.mean(step_ms)), 3), "std": round(float(np.std(step_ms)), 3), "min": round(float(np.min(step_ms)), 3),
                           "note": "rank 0, HIP events at the step boundaries of the same timed region"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"SWIN-T encoder (the backbone of BiRefNet-lite, BASELINE.json configs[3]) 1024x1024 f16, batch={B} per MI355X; "
                               "encoder only (the whole BiRefNet is --workload birefnet)",
                   "images_per_gpu_per_step": B, "global_batch": world * B, "weights": "random-init synthetic GGUF (seed 4)",
                   "parallelism": f"dp{world} (image shards, no data-path collective)"},
        "model_gflop_per_image": round(gflop, 2) if gflop else None,
        "model_tflops": round(value * gflop / 1e3, 2) if gflop else None,
    }
    if groups:
        tot = sum(g["ms"] for g in groups)
        dom = groups[0]
        hbm_bound = dom["flops"] / max(dom["bytes"], 1) < PEAK_MFMA_F16 / PEAK_HBM
        if hbm_bound:
            ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            res["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                               "frac": round(ach * 1e9 / PEAK_HBM, 4), "traffic": None}
        else:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            res["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F16 / 1e12, "unit": "TFLOP/s",
                               "frac": round(ach * 1e12 / PEAK_MFMA_F16, 4), "traffic": None}
        res["roofline"]["avg_launch_ms"] = round(dom["ms"] / max(dom["launches"], 1), 4)
        res["roofline"]["launches_per_step"] = dom["launches"]
        res["roofline"]["share_of_step"] = round(dom["ms"] / tot, 3)
        res["kernel_groups_ms"] = {g["name"]: round(g["ms"], 3) for g in groups}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle

        tensors, conv2d = synth.swin_gguf_tensors(synth.swin_state_dict(cfg, 4))
        om = oracle.Model(tensors, conv2d)
        P = oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)
        oracle.set_num_threads(args.cpu_threads)
        mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
        n = 2
        t0 = time.perf_counter()
        errs = []
        for i in range(n):
            want = oracle.swin_encode(om, P, ((imgs[i].astype(np.float32) / 255.0 - mean) / std).astype(np.float32))
            errs.append(max(float(np.abs(outs[k][i].cpu().numpy() - want[k]).max() / np.abs(want[k]).max()) for k in range(4)))
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(n / dt, 4), "unit": "images/s", "cores": args.cpu_threads, "kind": "port",
                               "sample": f"{n} images 1024x1024, OpenMP {args.cpu_threads} threads, oracle/libvisp_oracle.so (vo_swin_encode)",
                               "max_rel_err_gpu_vs_cpu": round(max(errs), 5)}
    print(json.dumps(res), flush=True)


def run_birefnet(args, torch, dist, rank, world, device_index, barrier, api):
    """A step = birefnet_process_input's normalisation + birefnet_predict (birefnet.cpp:252-260: two-scale SWIN-T encode, squeeze
    block, deformable-conv decoder -> sigmoid mask) for a batch of synthetic rgb_u8 images resident in HBM: BASELINE.json
    configs[3] (BiRefNet-lite, 1024x1024, batch 64 over 8 GPUs = 8 images per GPU and step). Ranks take whole images, no collective."""
    import dataclasses

    B, S = args.batch or 8, 1024
    cfg = dataclasses.replace(synth.SWIN_T, image_size=S)
    tmp = Path(tempfile.gettempdir()) / f"visp_bench_birefnet_lite_f16_{os.environ.get('MASTER_PORT', '0')}.gguf"
    if rank == 0:
        synth.write_birefnet_gguf(tmp, cfg, seed=4)
    barrier()
    dev = vision.Device.init(index=device_index)
    model = vision.Model.load(tmp, dev, vision.Arch.birefnet)  # 100 MB of weights: every rank reads the file
    imgs = synth.images(min(B, 2), S, S, seed=77 + 100 * rank)
    imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
    src = torch.from_numpy(imgs).cuda()
    out = torch.empty((B, S, S), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream().cuda_stream

    def step():
        model.segment_batch_device(src.data_ptr(), B, S, S, out.data_ptr(), stream)

    groups = []
    if rank == 0:
        step()
        torch.cuda.synchronize()
        model.enable_timing(True)
        step()
        torch.cuda.synchronize()
        groups = sorted(model.read_timing(), key=lambda g: -g["ms"])
        model.enable_timing(False)
        if args.profile_groups:
            tot = sum(g["ms"] for g in groups)
            print(f"{'group':16s} {'ms':>8s} {'%':>6s} {'launch':>6s} {'TFLOP/s':>9s} {'GB/s':>9s}", file=sys.stderr)
            for g in groups:
                print(f"{g['name']:16s} {g['ms']:8.3f} {100 * g['ms'] / tot:6.1f} {g['launches']:6d} "
                      f"{g['flops'] / g['ms'] / 1e9:9.1f} {g['bytes'] / g['ms'] / 1e6:9.1f}", file=sys.stderr)
            print(f"{'total':16s} {tot:8.3f}", file=sys.stderr)
    for _ in range(args.warmup):
        step()
    ev = StepEvents(api, stream, args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.mark()
        step()
    ev.mark()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = ev.step_ms()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    o = out.cpu().numpy()
    assert np.isfinite(o).all() and o.min() >= 0 and o.max() <= 1 and o.std() > 1e-3, "invalid output"
    if rank != 0:
        return
    value = world * B * args.steps / elapsed
    gflop = sum(g["flops"] for g in groups) / B / 1e9 if groups else None
    res = {
        "metric": "images/sec, BiRefNet-lite (SWIN-T) 1024x1024 f16",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "step_ms_device": {"mean": round(float(np.mean(step_ms)), 3), "std": round(float(np.std(step_ms)), 3), "min": round(float(np.min(step_ms)), 3),
                           "note": "rank 0, HIP events at the step boundaries of the same timed region"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"BiRefNet-lite f16 (SWIN-T shifted-window attention, deformable-conv decoder) 1024x1024, batch={B} per MI355X "
                               "(BASELINE.json configs[3]: batch 64 over 8 GPUs)",
                   "images_per_gpu_per_step": B, "global_batch": world * B, "weights": "random-init synthetic GGUF (seed 4)",
                   "parallelism": f"dp{world} (image shards, no data-path collective)"},
        "model_gflop_per_image": round(gflop, 2) if gflop else None,
        "model_tflops": round(value * gflop / 1e3, 2) if gflop else None,
    }
    if groups:
        tot = sum(g["ms"] for g in groups)
        dom = groups[0]
        hbm_bound = dom["flops"] / max(dom["bytes"], 1) < PEAK_MFMA_F16 / PEAK_HBM
        if hbm_bound:
            ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            res["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                               "frac": round(ach * 1e9 / PEAK_HBM, 4), "traffic": None}
        else:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            res["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F16 / 1e12, "unit": "TFLOP/s",
                               "frac": round(ach * 1e12 / PEAK_MFMA_F16, 4), "traffic": None}
        res["roofline"]["avg_launch_ms"] = round(dom["ms"] / max(dom["launches"], 1), 4)
        res["roofline"]["launches_per_step"] = dom["launches"]
        res["roofline"]["share_of_step"] = round(dom["ms"] / tot, 3)
        res["kernel_groups_ms"] = {g["name"]: round(g["ms"], 3) for g in groups}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle

        tensors, conv2d = synth.birefnet_gguf_tensors(synth.birefnet_state_dict(cfg, 4))
        om = oracle.Model(tensors, conv2d)
        P = oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)
        oracle.set_num_threads(args.cpu_threads)
        mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
        t0 = time.perf_counter()
        want = oracle.birefnet_predict(om, P, ((imgs[0].astype(np.float32) / 255.0 - mean) / std).astype(np.float32))
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(1 / dt, 4), "unit": "images/s", "cores": args.cpu_threads, "kind": "port",
                               "sample": f"1 image 1024x1024, OpenMP {args.cpu_threads} threads, oracle/libvisp_oracle.so (vo_birefnet_predict)",
                               "mask_mae_gpu_vs_cpu": round(float(np.abs(o[0] - want).mean()), 6), "mask_max_abs_diff_gpu_vs_cpu": round(float(np.abs(o[0] - want).max()), 5)}
    print(json.dumps(res), flush=True)


def cpu_baseline(cfg, imgs, gpu_out, n_images, n_threads):
    """Times the CPU oracle (a port of the reference's ggml CPU path: f32 math, f16-rounded weights,
    batch-1 sequential like vision.cpp:155) on a bounded sample and reports the parity of the timed
    GPU output against it."""
    from oracle import oracle

    sd = synth.state_dict(cfg, 0)
    tensors, conv2d = synth.gguf_tensors(sd)
    om = oracle.Model(tensors, conv2d, "whcn")
    params = oracle.make_params(cfg.patch_size, cfg.embed_dim, cfg.n_layers, cfg.n_heads, cfg.image_size, 14, cfg.feature_layers)
    # measured on the MI355X box (tools/oracle_scaling.py): 16 threads is the fastest setting of this port
    # (0.33 s/image; 128 threads: 2.9 s/image) and is one GPU's share of the 256-thread host
    oracle.set_num_threads(n_threads)
    cores = n_threads
    om.compute(params, imgs[0])  # warm-up (page-in, LUT init)
    t0 = time.perf_counter()
    maes = []
    for i in range(n_images):
        want, _ = om.compute(params, imgs[i % len(imgs)])
        maes.append(float(np.abs(gpu_out[i % len(imgs)] - want).mean()))
    dt = time.perf_counter() - t0
    return {"value": round(n_images / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_images} images 518x518, batch-1 sequential, OpenMP {cores} threads, oracle/libvisp_oracle.so",
            "mae_gpu_vs_cpu": round(max(maes), 6)}


if __name__ == "__main__":
    main()
