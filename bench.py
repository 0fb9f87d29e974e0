#!/usr/bin/env python3
"""Benchmark of the Diff-UNet hot path on MI355X: BASELINE.json config 2.

Workload: DiffUNet(in=1, out=16), one 96^3 patch per GPU, DDPM ancestral sampling
(`diffusion.p_sample_loop` semantics) -- a "step" is ONE reverse-diffusion step = one denoiser
evaluation (18 conv3x3x3 + 4 deconv + 1x1 head, 1.0564 TFLOP) + the sampler update, with inputs
resident in HBM, in-kernel Philox noise, replayed from a captured HIP graph.  The conditioning
encoder pass (once per patch, 0.28 TFLOP) runs before the timed region like x_T generation.
Metric: denoised voxel-steps / s  (= N_gpus * 96^3 * K / wall time of K steps).

Multi-GPU (--gpus N under torch.distributed.run): config 2 does not shard (1000 strictly sequential
steps on one tensor) => N independent replicas, weak scaling, no data-path
collective; only the timing barrier/all-reduce(MAX) touches RCCL.

Extra objects in the JSON line:
  roofline     -- for the dominant kernel (conv3d_k3_v2_kernel, all 18 launches of a step): algorithmic
                  FLOPs per launch / average launch duration, measured with HIP events on the launch
                  stream: each of the step's 18 launches replayed back to back between one event pair
                  (split-K layers without their finish kernel); peak = dense fp16 MFMA.
  cpu_baseline -- the CPU oracle (oracle/unet_ref.py, "port") timed on this box's host cores on a
                  bounded sample (1 warm-up + 2 denoiser evaluations at the same 96^3 x 16 shape).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOX = 96 ** 3
CLASSES = 16
FEATURES = (64, 64, 128, 256, 512, 64)
PEAK_F16_TFLOPS = 2500.0      # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3


def conv3_flops(plan):
    """Algorithmic FLOPs of every conv3x3x3 launch of one denoiser evaluation (2*Cin*Cout*27*voxels,
    true channel counts: the first layer counts 17 inputs, not its padded 24)."""
    out = []
    for l, pair in enumerate(plan.den):
        v = plan.S[l][0] * plan.S[l][1] * plan.S[l][2]
        for c in pair:
            out.append(2.0 * c.cin * c.cout * 27 * v * plan.N)
    for l in (3, 2, 1, 0):
        v = plan.S[l][0] * plan.S[l][1] * plan.S[l][2]
        for c in plan.dec[l]:
            out.append(2.0 * c.cin * c.cout * 27 * v * plan.N)
    return out


def time_conv_launches(plan, reps):
    """HIP events (torch.cuda.Event on the launch stream) around the conv3d_k3 kernel: one eager denoiser evaluation
    records the arguments of its 18 launches; each launch is then replayed `reps` times back to back between ONE event
    pair (an event pair around a single ~30 us launch reads the command-processor gaps as kernel time).  Split-K layers
    are replayed without their finish kernel (dua_set_option(2, 1)), so the figure is conv3d_k3_v2_kernel alone, as the
    rocprofv3 summary lists it.  Returns (avg ms per launch, launches/step, ms by launch)."""
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd import ops
    real = ops.conv3d_k3
    calls = []

    def wrapped(*a, **k):
        calls.append((a, k))
        return real(*a, **k)

    ops.conv3d_k3 = wrapped
    try:
        plan.denoiser_body()
        torch.cuda.synchronize()
    finally:
        ops.conv3d_k3 = real
    by_launch = []
    nv.check(nv.lib().dua_set_option(2, 1), "dua_set_option")
    try:
        for a, k in calls:
            for _ in range(3):
                real(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                real(*a, **k)
            e1.record()
            torch.cuda.synchronize()
            by_launch.append(e0.elapsed_time(e1) / reps)
    finally:
        nv.check(nv.lib().dua_set_option(2, 0), "dua_set_option")
    return sum(by_launch) / len(by_launch), len(calls), by_launch


def host_threads():
    """Threads for the CPU baseline: this process's CPU share (a one-GPU box grants 16 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(net_state, threads):
    """Oracle (CPU restatement of the reference path) on the host cores: bounded sample."""
    from oracle.unet_ref import RefDiffUNet
    torch.set_num_threads(threads)
    ref = RefDiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES).eval()
    ref.load_state_dict(net_state)
    g = torch.Generator().manual_seed(1)
    image = torch.rand(1, 1, 96, 96, 96, generator=g)
    x = torch.randn(1, CLASSES, 96, 96, 96, generator=g)
    t = torch.tensor([500])
    with torch.no_grad():
        emb = ref.embed_model(image)
        ref.model(x, t, image=image, embeddings=emb)        # warm-up
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            ref.model(x, t, image=image, embeddings=emb)
        dt = (time.perf_counter() - t0) / n
    return {"value": VOX / dt, "unit": "voxel-steps/s", "cores": threads, "kind": "port",
            "sample": f"1 warm-up + {n} timed denoiser evaluations (torch CPU fp32 oracle) at 96^3 x 16 classes, "
                      f"{dt:.2f} s/step; sampler update excluded (<1%)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="patches per GPU (BASELINE config 2 is 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # one rank per GPU on a real node.  Rehearsal on a one-GPU box only: DUA_BENCH_BACKEND=gloo lets several ranks
    # share device 0 (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
    backend = os.environ.get("DUA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd import ops
    from diff_unet_amos_amd.diff_unet import DiffUNet

    dtype = torch.float16 if args.dtype == "f16" else torch.float32
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES, compute_dtype=dtype).to(dev).eval()
    state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    B = args.batch
    image = torch.rand(B, 1, 96, 96, 96, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    plan = net._rt.plan(B, (96, 96, 96), dev)
    diffusion = net.diffusion                      # 1000-step process; we time K of its steps
    with torch.no_grad():
        net.embed_model(image)                     # encoder once per patch (not in the timed region)
        x_T = torch.randn(B, CLASSES, 96, 96, 96, device=dev)
        ops.to_channels_last(x_T, plan.x_state, 0, plan.cx)
        ops.to_channels_last(x_T, plan.xin, 0, plan.C)
        plan.x_sum.zero_()
        plan.refresh_weights()
        T = diffusion.num_timesteps
        order = list(range(T))[::-1]
        coef_table = diffusion.ddpm_coef(torch.tensor(order)).to(dev).contiguous()
        row_of_step = torch.tensor(order, dtype=torch.int32, device=dev)
        plan.counter.zero_()
        plan.new_seed(3 + rank)

        def one_step():
            ops.step_begin(B, plan.temb_table, plan.cur_add, row_of_step=row_of_step, counter=plan.counter,
                           coef_table=coef_table, cur_coef=plan.cur_coef, step_word=plan.step_word)
            plan.denoiser_body()
            plan.tail(nv.MODE_DDPM, noise=None, use_sum=False)

        assert args.warmup + args.steps + 2 <= T, "the 1000-step process bounds warmup+steps"
        one_step()
        torch.cuda.synchronize()
        if args.no_graph:
            run = one_step
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step()
            run = g.replay
        for _ in range(args.warmup):
            run()

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        barrier()
        finite = bool(torch.isfinite(plan.x_state).all())

        roof = None
        if rank == 0 and not args.no_roofline:
            fl = conv3_flops(plan)
            avg_ms, per_step, by_launch = time_conv_launches(plan, 20)
            assert per_step == len(fl)
            flops_per_launch = sum(fl) / len(fl)
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            peak = PEAK_F16_TFLOPS if args.dtype == "f16" else PEAK_F32_TFLOPS
            traffic = None       # HBM bytes per launch from the committed PMC passes (profiles/), fp16 path only
            tj = os.path.join(ROOT, "profiles", "r1_conv_traffic.json")
            if args.dtype == "f16" and os.path.exists(tj):
                traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
            roof = {"bound": "mfma", "kernel": "conv3d_k3_v2_kernel", "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "launches_per_step": per_step, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_gflop_per_launch": round(flops_per_launch / 1e9, 2),
                    "conv_ms_per_step": round(avg_ms * per_step, 3),
                    "by_launch_us": [round(x * 1e3, 1) for x in by_launch],
                    "by_launch_tflops": [round(f / (x * 1e-3) / 1e12) for f, x in zip(fl, by_launch)],
                    "largest_launch": {"layer": "upcat_1.convs.conv_0 128->64 @96^3",
                                       "tflops": round(max(fl) / (by_launch[fl.index(max(fl))] * 1e-3) / 1e12, 2)}}

    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {
            "metric": "denoised voxel-steps/sec on 96^3 16-class AMOS patches",
            "value": world * B * VOX * args.steps / dt, "unit": "voxel-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "DiffUNet 96^3 patch, 16 classes, DDPM p_sample steps of the 1000-step process "
                                   "(BASELINE.json configs[1]); one patch per GPU, replicas only",
                       "patch": [96, 96, 96], "classes": CLASSES, "batch_per_gpu": B, "graph_replay": not args.no_graph,
                       "noise": "in-kernel Philox4x32-10", "weights": "torch.manual_seed(0) default init"},
            "step_tflops": B * 1.0564e12 / (ms * 1e-3) / 1e12, "finite": finite,
        }
        if roof is not None:
            line["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(state, host_threads())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()          # rank 0 also ran the instrumented roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
