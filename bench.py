#!/usr/bin/env python3
"""Benchmark of the Diff-UNet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

--config 2 (default, the headline: BASELINE.json configs[1]): DiffUNet(in=1, out=16), one 96^3 patch per GPU, DDPM
  ancestral sampling (`diffusion.p_sample_loop` semantics) -- a "step" is ONE reverse-diffusion step = one denoiser
  evaluation (18 conv3x3x3 + 4 deconv + 1x1 head, 1.0564 TFLOP algorithmic) + the sampler update, with inputs resident in
  HBM, in-kernel Philox noise, replayed from a captured HIP graph.  The conditioning encoder pass (once per patch,
  0.28 TFLOP) runs before the timed region like x_T generation.  Metric: denoised voxel-steps / s.  HEADLINE (value,
  ms_per_step) = the WHOLE 1000-step loop through the public API, net.diffusion.p_sample_loop(net.model, shape, model_kwargs):
  the loop the parity test covers, the two exact-fp32 finishing steps of an fp16 plan included, second call, bracketed by
  the ranks' barrier, maximum over the ranks; the --warmup / --steps replays of the captured fp16 step are reported beside
  it (replayed_step_ms, replayed_value).  Config 2 does not shard (1000 strictly sequential steps on one tensor, SURVEY.md
  8(e)) => N independent replicas, weak scaling, no data-path collective.
--config 3 (BASELINE.json configs[2]): sliding-window DDIM inference of a 256x256x192 volume (48 windows of 96^3,
  50 steps), windows sharded over the ranks, ONE RCCL all_gather_into_tensor of the per-window outputs, identical
  blend on every rank.  A "step" is one whole volume; reports seconds per volume and the all-gather share.
--config 4 (BASELINE.json configs[3]): DDP training step (q_sample + denoise + mse/bce/dice + backward + AdamW) on
  synthetic 96^3 x 16-class batches, 2 samples per GPU, gradients averaged over RCCL (DDP buckets overlapped with
  backward); reports samples/s and the share a flat all-reduce of the 153.6 MB of gradients would take.
--config 5 (BASELINE.json configs[4]): the diff_swin_unetr variant -- DiffSwinUNETR(in=1, out=16, feature_size 48), one
  96^3 patch per GPU; a "step" is one reverse-diffusion step = one SwinUNETRDenoiser evaluation (8 shifted-window
  attention blocks, 4 patch mergings, 10 UnetResBlocks, 5 transposed convolutions, 1x1x1 head) + the sampler update,
  replayed from a captured HIP graph; same metric as config 2.

Launch: with no WORLD_SIZE in the environment and --gpus N > 1 this script starts its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) BEFORE touching the GPU
and relays their output; under torch.distributed.run it is one rank (RANK / LOCAL_RANK / WORLD_SIZE from the env).

Extra objects in the JSON line (config 2, rank 0, N = 1):
  roofline     -- for the DOMINANT kernel: the one with the largest share of the step's convolution time
                  (fp16: conv3d_k3_wide_kernel, the three 96^3 layers = 74 % of the step's FLOPs; fp32:
                  conv3d_k3_v2_kernel).  achieved = algorithmic FLOPs of its launches / their duration, avg_launch_ms =
                  the per-kernel average the rocprofv3 --kernel-trace --stats summary of the same command shows.
                  Durations come from HIP events on the launch stream: every launch replayed back to back between one
                  event pair (split-K layers without their finish kernel); launches >= 100 us are timed a second time
                  INSIDE a replayed step (event pairs around the launch as the step runs it: caches and clocks as in
                  the step) and that figure is used -- both are listed (by_launch_us / by_launch_back_to_back_us).
                  other_kernels = the same summary for the other convolution kernels of the step, all_conv_launches =
                  the figure over all 18 launches (comparable across rounds whatever the kernel split; .back_to_back = the
                  same with every launch timed back to back, the only method of rounds 1-3).  FLOPs are ALGORITHMIC (the
                  reference's layers); the folded up-convolution (upconv_k3_kernel, round 5) executes fewer multiply-adds
                  than its two layers' count: executed_* fields give what it multiplies.
                  peak = dense fp16 MFMA (2.5 PFLOP/s nominal: at this part's 1400 W cap the matrix pipes alone sustain
                  1.5-1.8 PFLOP/s on random data, measured_mfma_ceiling / frac_of_ceiling).
                  traffic = HBM bytes per launch of that kernel from two rocprofv3 --pmc child passes of this script
                  (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE), or null when the profiler is not available.
  full_loop    -- config 2 end to end whatever --steps says: the whole 1000-step DDPM loop through the public API
                  (net.diffusion.p_sample_loop(net.model, shape, model_kwargs=...)), second call (the first one captures
                  the step graph and is reported as first_call_seconds).
  f32          -- the parity dtype on the same workload: a short replayed pass of an fp32 plan, ms_per_step and the
                  convolution kernel's fraction of the fp32 MFMA peak (157.3 TFLOP/s).
  cpu_baseline -- the CPU oracle (oracle/unet_ref.py, "port") timed on this box's host cores on a bounded sample
                  (1 warm-up + 2 denoiser evaluations at the same 96^3 x 16 shape).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOX = 96 ** 3
CLASSES = 16
FEATURES = (64, 64, 128, 256, 512, 64)
PEAK_F16_TFLOPS = 2500.0      # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3
CONV_KERNEL = "conv3d_k3_v2_kernel"          # the 4x8x8 / 2x8x8 / split-K forms (every layer below 96^3; every fp32 layer)
WIDE_KERNEL = "conv3d_k3_wide_kernel"        # round 4: the 8x8x8-tile form of the fp16 96^3 layers, 70 % of a step's FLOPs


UPCONV_KERNEL = "upconv_k3_kernel"            # round 5: UpCat's transposed convolution folded into its first convolution


def conv_kernel_of(call):
    """Which kernel the launcher picks for a recorded ops.conv3d_k3 call (dua_conv3d_k3_kernel_kind: the launcher's own rule)."""
    from diff_unet_amos_amd import ops
    a, k = call
    if k.get("__upconv__"):
        return UPCONV_KERNEL
    x, cin, cout = a[0], a[1], a[5]
    N, D, H, W = x.shape[:4]
    kind = ops.conv3_kernel_kind(x.dtype, N, D, H, W, cin, x.shape[-1], cout, fused=k.get("norm") is not None,
                                 tap_channel=k.get("tap_channel"), background=bool(k.get("background")))
    return {ops.KIND_V2: CONV_KERNEL, ops.KIND_FIRST: "conv3d_k3_first_kernel", ops.KIND_WIDE: WIDE_KERNEL}[kind]


def conv3_flops(plan, executed=False):
    """Algorithmic FLOPs of every conv3x3x3 launch of one denoiser evaluation (2*Cin*Cout*27*voxels,
    true channel counts: the first layer counts 17 inputs, not its padded 24).  ``executed``: what the launch multiplies --
    the same, except for the folded up-convolution (8 parents x Cu channels for the upsampled half instead of 27 taps x Cmid)."""
    out = []
    for l, pair in enumerate(plan.den):
        v = plan.S[l][0] * plan.S[l][1] * plan.S[l][2]
        for c in pair:
            out.append(2.0 * c.cin * c.cout * 27 * v * plan.N)
    for l in (3, 2, 1, 0):
        v = plan.S[l][0] * plan.S[l][1] * plan.S[l][2]
        for i, c in enumerate(plan.dec[l]):
            fl = 2.0 * c.cin * c.cout * 27 * v * plan.N
            if i == 0 and plan._fold_level(l):
                # the folded launch also does the transposed convolution's work (2 * Cin * Cout * 8 * coarse voxels
                # ... * 8 children = 2 * Cin * Cout * fine voxels); ALGORITHMIC figures of the reference's two layers, not the
                # 3.4x fewer multiply-adds the regrouped upsampled half executes
                d = plan.deconv[l]
                if executed:
                    cs = plan.f[l]
                    fl = 2.0 * v * plan.N * c.cout * (27 * cs + 8 * d.weight.shape[0])
                else:
                    fl += 2.0 * d.weight.shape[0] * d.weight.shape[1] * v * plan.N
            out.append(fl)
    return out


_LAST_CONV_CALLS = []
_CONV_VARIANT = 0          # --conv-variant (ops.CONV_POLICY): 0 = the launchers' automatic choice


def time_conv_launches(plan, reps):
    """HIP events (torch.cuda.Event on the launch stream) around the conv3d_k3 kernel: one eager denoiser evaluation
    records the arguments of its 18 launches; each launch is then replayed `reps` times back to back between ONE event
    pair (an event pair around a single ~30 us launch reads the command-processor gaps as kernel time).  Split-K layers
    are replayed without their finish kernel (dua_conv3_desc.policy bit DUA_POLICY_NO_FINISH), so the figure is the conv kernel alone, as the
    rocprofv3 summary lists it.  Returns (avg ms per launch, launches/step, ms by launch)."""
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd import ops
    real_conv, real_up = ops.conv3d_k3, ops.upconv_k3
    calls = []
    global _LAST_CONV_CALLS
    _LAST_CONV_CALLS = calls

    def real(*a, **k):
        k = dict(k)
        return (real_up if k.pop("__upconv__", False) else real_conv)(*a, **k)

    def wrapped(*a, **k):
        calls.append((a, k))
        return real_conv(*a, **k)

    def wrapped_up(*a, **k):
        calls.append((a, dict(k, __upconv__=True)))
        return real_up(*a, **k)

    ops.conv3d_k3, ops.upconv_k3 = wrapped, wrapped_up
    try:
        plan.denoiser_body()
        torch.cuda.synchronize()
    finally:
        ops.conv3d_k3, ops.upconv_k3 = real_conv, real_up
    by_launch = []
    ops.CONV_POLICY |= nv.POLICY_NO_FINISH
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
        ops.CONV_POLICY &= ~nv.POLICY_NO_FINISH
    # Launches of >= 100 us are ALSO timed where they run: event pairs around them inside eager passes of the whole evaluation
    # (the pair's command-processor gap is ~1 % there).  Replayed back to back, a 200-350 us MFMA-bound launch holds the chip at
    # its lowest clock and reads 5-8 % longer than the same launch between the step's light kernels, which is what rocprofv3
    # lists for it; the in-step figure is the one reported (by_launch_us), the back-to-back one is kept beside it.
    global _BACK_TO_BACK_MS
    _BACK_TO_BACK_MS = list(by_launch)
    big = [i for i, ms in enumerate(by_launch) if ms >= 0.1]
    if big:
        pairs = {i: [] for i in big}
        idx = [0]

        def timed_with(fn):
            def timed(*a, **k):
                i = idx[0]
                idx[0] += 1
                if i in pairs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(*a, **k)
                    e1.record()
                    pairs[i].append((e0, e1))
                else:
                    fn(*a, **k)
            return timed

        ops.conv3d_k3, ops.upconv_k3 = timed_with(real_conv), timed_with(real_up)
        try:
            for _ in range(1 + max(3, reps // 4)):
                idx[0] = 0
                plan.denoiser_body()
            torch.cuda.synchronize()
        finally:
            ops.conv3d_k3, ops.upconv_k3 = real_conv, real_up
        for i in big:
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in pairs[i][1:])
            by_launch[i] = ts[len(ts) // 2]
    return sum(by_launch) / len(by_launch), len(calls), by_launch


_BACK_TO_BACK_MS = []


def conv_roofline(plan, dtype_flag, reps=20):
    """roofline object for the 3x3x3 convolution of one denoiser evaluation of `plan` (HIP events, see time_conv_launches);
    FLOPs are algorithmic (2 * Cin * Cout * 27 * voxels * batch).  The object describes the DOMINANT kernel -- the one with
    the largest share of the step: conv3d_k3_wide_kernel in fp16 (the three 96^3 layers it runs are 74 % of the step's FLOPs),
    conv3d_k3_v2_kernel in fp32 -- so that `avg_launch_ms` is the per-kernel average of the rocprofv3 summary; the other
    convolution kernels of the step are listed under `other_kernels`, and `all_conv_launches` is the figure over all 18 launches
    (comparable across rounds whatever the kernel split)."""
    fl = conv3_flops(plan)
    _, per_step, by_launch = time_conv_launches(plan, reps)
    assert per_step == len(fl)
    kern = [conv_kernel_of(c) for c in _LAST_CONV_CALLS]
    peak = PEAK_F16_TFLOPS if dtype_flag == "f16" else PEAK_F32_TFLOPS
    out = roofline_by_kernel(fl, by_launch, kern, peak, "upcat_1.convs.conv_0 128->64 @96^3 (+ upsample.deconv when folded)")
    if UPCONV_KERNEL in kern:
        # `achieved` / `frac` everywhere are ALGORITHMIC (the reference's layers: Conv3d over the 128-channel concat + ConvTranspose3d);
        # the folded launch executes fewer multiply-adds than that -- both figures are given, and the matrix pipes are priced by
        # the executed one
        ex = conv3_flops(plan, executed=True)
        idx = [i for i, k_ in enumerate(kern) if k_ == UPCONV_KERNEL]
        ms = sum(by_launch[i] for i in idx)
        u = out["other_kernels"].get(UPCONV_KERNEL) or out
        u["executed_gflop_per_launch"] = round(sum(ex[i] for i in idx) / len(idx) / 1e9, 2)
        u["executed_tflops"] = round(sum(ex[i] for i in idx) / (ms * 1e-3) / 1e12, 2)
        u["executed_frac"] = round(u["executed_tflops"] / peak, 4)
        all_ex = sum(ex) / (sum(by_launch) * 1e-3) / 1e12
        out["all_conv_launches"]["executed_tflops"] = round(all_ex, 2)
        out["all_conv_launches"]["executed_frac"] = round(all_ex / peak, 4)
        big = fl.index(max(fl))
        if kern[big] == UPCONV_KERNEL:
            out["largest_launch"]["executed_tflops"] = round(ex[big] / (by_launch[big] * 1e-3) / 1e12, 2)
        out["note"] = ("upconv_k3_kernel = ConvTranspose3d(k2,s2) folded into the 3x3x3 convolution behind it (csrc/upconv.hip): same "
                       "function of the same parameters with 3.4x fewer multiply-adds on the upsampled half; achieved / frac count the "
                       "reference's layers (algorithmic), executed_* what the kernel multiplies")
    return out


def roofline_by_kernel(fl, by_launch, kern, peak, largest_name):
    """The roofline object from per-launch FLOPs, durations (ms) and kernel names (see conv_roofline)."""
    groups = {}
    for name, f, ms in zip(kern, fl, by_launch):
        g = groups.setdefault(name, [0, 0.0, 0.0])
        g[0] += 1; g[1] += f; g[2] += ms
    main = max(groups, key=lambda n_: groups[n_][2])

    def summary(g):
        tf = g[1] / (g[2] * 1e-3) / 1e12
        return {"launches_per_step": g[0], "avg_launch_ms": round(g[2] / g[0], 4), "algorithmic_gflop_per_launch": round(g[1] / g[0] / 1e9, 2),
                "ms_per_step": round(g[2], 4), "achieved": round(tf, 2), "frac": round(tf / peak, 4)}

    m = summary(groups[main])
    big = fl.index(max(fl))
    out = {"bound": "mfma", "kernel": main, "achieved": m["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": m["frac"],
           "traffic": None, "launches_per_step": m["launches_per_step"], "avg_launch_ms": m["avg_launch_ms"],
           "algorithmic_gflop_per_launch": m["algorithmic_gflop_per_launch"], "kernel_ms_per_step": m["ms_per_step"],
           "conv_ms_per_step": round(sum(by_launch), 3),
           "by_launch_us": [round(x * 1e3, 1) for x in by_launch],
           "by_launch_back_to_back_us": [round(x * 1e3, 1) for x in _BACK_TO_BACK_MS],
           "by_launch_tflops": [round(f / (x * 1e-3) / 1e12) for f, x in zip(fl, by_launch)],
           "by_launch_kernel": [{"conv3d_k3_first_kernel": "first", WIDE_KERNEL: "wide", CONV_KERNEL: "v2", UPCONV_KERNEL: "upconv"}[k_] for k_ in kern],
           "largest_launch": {"layer": largest_name, "kernel": kern[big],
                              "tflops": round(max(fl) / (by_launch[big] * 1e-3) / 1e12, 2)},
           "other_kernels": {n_: summary(g) for n_, g in groups.items() if n_ != main}}
    # every 3x3x3 launch of the step, whichever kernel runs it
    all_tf = sum(fl) / (sum(by_launch) * 1e-3) / 1e12
    out["all_conv_launches"] = {"launches_per_step": len(fl), "avg_launch_ms": round(sum(by_launch) / len(by_launch), 4),
                                "achieved": round(all_tf, 2), "frac": round(all_tf / peak, 4)}
    if len(_BACK_TO_BACK_MS) == len(fl):
        # the same figure with EVERY launch timed back to back (the only method of rounds 1-3; the in-step medians used above for
        # launches >= 100 us read 5-8 % faster): the one to compare across rounds
        b2b = sum(fl) / (sum(_BACK_TO_BACK_MS) * 1e-3) / 1e12
        out["all_conv_launches"]["back_to_back"] = {"avg_launch_ms": round(sum(_BACK_TO_BACK_MS) / len(fl), 4),
                                                    "achieved": round(b2b, 2), "frac": round(b2b / peak, 4)}
    return out


def full_loop_config2(net, image, shape, D=None):
    """BASELINE config 2 as it is worded: the whole 1000-step DDPM loop through the public API,
    ``net.diffusion.p_sample_loop(net.model, shape, model_kwargs=...)`` (gaussian_diffusion.py:441-485 of the reference),
    in-kernel noise, the fp32 finishing steps of an fp16 plan included (engine.Plan.sample_loop: what the parity test of this
    loop covers), encoder pass included in neither figure.  The first call captures the step graph of this (plan, process)
    pair; the second is what every later patch costs -- bracketed by the ranks' barrier, maximum over the ranks."""
    out = {}
    with torch.no_grad():
        kw = {"image": image, "embeddings": net.embed_model(image)}
        for key in ("first_call_seconds", "seconds"):
            if D is not None:
                D.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sample = net.diffusion.p_sample_loop(net.model, shape, model_kwargs=kw)
            torch.cuda.synchronize()
            out[key] = time.perf_counter() - t0
            if D is not None:
                out[key] = D.max_over_ranks(out[key])
                D.barrier()
    T = net.diffusion.num_timesteps
    out.update(steps=T, ms_per_step=out["seconds"] / T * 1e3, voxel_steps_per_s=shape[0] * VOX * T / out["seconds"],
               finite=bool(torch.isfinite(sample).all()), api="net.diffusion.p_sample_loop(net.model, shape, model_kwargs)")
    return out


def f32_pass_config2(state, image, dev, steps=20, warmup=3):
    """The parity dtype on the same workload: `steps` replayed DDPM steps of an fp32 plan (exact-fp32 MFMA 32x32x2) and the
    convolution kernel's roofline fraction against the fp32 MFMA peak."""
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd import ops
    from diff_unet_amos_amd.diff_unet import DiffUNet
    net = DiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES, compute_dtype=torch.float32)
    net.load_state_dict(state)
    net = net.to(dev).eval()
    B = image.shape[0]
    plan = net._rt.plan(B, (96, 96, 96), dev)
    with torch.no_grad():
        net.embed_model(image)
        x_T = torch.randn(B, CLASSES, 96, 96, 96, device=dev)
        ops.to_channels_last(x_T, plan.x_state, 0, plan.cx)
        ops.to_channels_last(x_T, plan.xin, 0, plan.C)
        plan.x_sum.zero_()
        plan.refresh_weights()
        T = net.diffusion.num_timesteps
        order = list(range(T))[::-1]
        coef_table = net.diffusion.ddpm_coef(torch.tensor(order)).to(dev).contiguous()
        row_of_step = torch.tensor(order, dtype=torch.int32, device=dev)
        plan.counter.zero_()
        plan.new_seed(5)

        def one_step():
            plan.native_step(nv.MODE_DDPM, row_of_step=row_of_step, coef_table=coef_table, use_sum=False)

        one_step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_step()
        for _ in range(warmup):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        roof = conv_roofline(plan, "f32", reps=5)
    return {"ms_per_step": ms, "steps": steps, "frac": roof["frac"], "achieved": roof["achieved"], "peak": roof["peak"],
            "unit": "TFLOP/s", "kernel": roof["kernel"], "largest_launch_tflops": roof["largest_launch"]["tflops"],
            "finite": bool(torch.isfinite(plan.x_state).all())}


def mfma_ceiling(dev, seconds=2.0):
    """What the matrix pipes of this device sustain on dense random fp16 operands after `seconds` of back-to-back launches
    (dua_mfma_probe: one wave per SIMD, v_mfma_f32_32x32x16_f16 on register operands), and the shader clock held meanwhile
    (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups)."""
    from diff_unet_amos_amd import _native as nv
    L = nv.lib()
    wgs, iters = 256, 12000
    sink = torch.zeros(1, device=dev)
    stamps = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
    s = nv.stream_ptr()

    def go(n, st=None):
        for _ in range(n):
            nv.check(L.dua_mfma_probe(wgs, iters, nv.ptr(sink), nv.ptr(st), s), "dua_mfma_probe")

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        go(20)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    go(reps - 1)
    go(1, stamps)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = wgs * 4.0 * iters * 16 * 32768
    st = stamps.view(wgs, 2).double()
    clock = float((st[:, 0] / (st[:, 1] * 10.0)).median())          # cycles per ns = GHz
    return flops / (ms * 1e-3) / 1e12, clock


def wgrad_roofline(step_fn, reps=10):
    """roofline object for the weight-gradient kernel of the training step (the kernel with the largest share of config 4):
    one eager step records the arguments of its conv3d_k3_wgrad launches, each is then replayed `reps` times between one HIP
    event pair on the launch stream.  FLOPs are algorithmic: 2 * Cin * Cout * 27 * voxels * batch per launch."""
    from diff_unet_amos_amd import ops
    real = ops.conv3d_k3_wgrad
    calls = []

    def wrapped(*a, **k):
        calls.append((a, k))
        return real(*a, **k)

    ops.conv3d_k3_wgrad = wrapped
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.conv3d_k3_wgrad = real
    if not calls:
        return None
    fl, by_launch = [], []
    for a, k in calls:
        x, cin, dy, cout, dw = a[0], a[1], a[3], a[4], a[6]
        scratch = torch.zeros_like(dw)               # the launch accumulates: never into the live gradient
        a2 = a[:6] + (scratch,) + a[7:]
        real(*a2, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            real(*a2, **k)
        e1.record()
        torch.cuda.synchronize()
        by_launch.append(e0.elapsed_time(e1) / reps)
        fl.append(2.0 * dw.shape[1] * cout * 27 * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3])
    achieved = sum(fl) / (sum(by_launch) * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv3d_k3_wgrad_fo_kernel (+ wgrad_fo_reduce_kernel)", "achieved": round(achieved, 2),
            "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": None,
            "launches_per_step": len(calls), "avg_launch_ms": round(sum(by_launch) / len(by_launch), 4),
            "algorithmic_gflop_per_launch": round(sum(fl) / len(fl) / 1e9, 2), "wgrad_ms_per_step": round(sum(by_launch), 3),
            "by_launch_us": [round(x * 1e3, 1) for x in by_launch],
            "by_launch_tflops": [round(f / (x * 1e-3) / 1e12) for f, x in zip(fl, by_launch)]}


def cpu_training_baseline(net_state, threads):
    """Oracle training step (oracle/train_ref.ref_training_step: q_sample + denoise + mse/bce/dice, backward through torch
    autograd) on the host cores: bounded sample, ONE 96^3 x 16-class sample, one warm-up + one timed step."""
    from oracle.train_ref import RefLoss, ref_training_step
    from oracle.unet_ref import RefDiffUNet
    torch.set_num_threads(threads)
    ref = RefDiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES)
    ref.load_state_dict(net_state)
    g = torch.Generator().manual_seed(1)
    image = torch.rand(1, 1, 96, 96, 96, generator=g)
    labels = (torch.rand(1, CLASSES, 96, 96, 96, generator=g) > 0.8).float()
    noise = torch.randn(1, CLASSES, 96, 96, 96, generator=g)
    crit = RefLoss("mse,bce,dice", "sum")
    dt = None
    for i in range(2):
        t0 = time.perf_counter()
        loss = ref_training_step(ref, image, labels, crit, noise, torch.tensor([500]))
        loss.backward()
        dt = time.perf_counter() - t0
        ref.zero_grad(set_to_none=True)
    return {"value": 1.0 / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"1 warm-up + 1 timed training step (forward + loss + backward, no optimizer) of the torch CPU fp32 oracle on "
                      f"ONE 96^3 x 16-class sample, {dt:.2f} s/sample"}


def measure_traffic(dtype_flag, kernel=None, config=2, extra=()):
    """HBM bytes per conv launch from the PMC counters, as MI355X_MICROARCH.md (HBM / rocprofv3 sections) prescribes:
    FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes over a short eager run of this script (a child
    process: the profiler must start the program itself), FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request),
    both in KiB; averaged over every launch of the conv kernel.  Returns bytes per launch or None."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dua_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-graph",
                   "--no-cpu-baseline", "--no-roofline", "--dtype", dtype_flag, "--config", str(config)] + list(extra)
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if (kernel or CONV_KERNEL) in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            out[counter] = sum(vals) / len(vals)
        except Exception:       # noqa: BLE001 -- no profiler, no permission, timeout: report null, never fail the bench
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0


def host_threads():
    """Threads for the CPU baseline: this process's CPU share (a one-GPU box grants 16 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(net_state, threads, swin=False):
    """Oracle (CPU restatement of the reference path) on the host cores: bounded sample."""
    torch.set_num_threads(threads)
    if swin:
        from oracle.swin_ref import make_ref_diff_swin_unetr
        ref = make_ref_diff_swin_unetr(1, CLASSES, 48).eval()
    else:
        from oracle.unet_ref import RefDiffUNet
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


def self_launch(args):
    """No WORLD_SIZE and --gpus N > 1: start the N ranks ourselves, before any GPU call in this process (a process that
    has initialised the GPU must never exec another program; this one only waits for its children)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Dist:
    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        assert self.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={self.world}"
        # one rank per GPU on a real node.  Rehearsal on a one-GPU box only: DUA_BENCH_BACKEND=gloo lets several ranks
        # share device 0 (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
        self.backend = os.environ.get("DUA_BENCH_BACKEND", "nccl")
        if self.backend != "nccl":
            self.local = self.local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        if self.world > 1:
            import torch.distributed as dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def finish(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def run_config2(args, D):
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd import ops
    from diff_unet_amos_amd.diff_unet import DiffUNet

    dev, rank, world = D.dev, D.rank, D.world
    dtype = torch.float16 if args.dtype == "f16" else torch.float32
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES, compute_dtype=dtype).to(dev).eval()
    if args.fold_min_tiles is not None:
        net.upconv_min_tiles = args.fold_min_tiles
    net.fold_upconv = not args.no_fold
    net.fold_residual = not args.no_fold_residual
    state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    B = args.batch or 1
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

        def one_step():          # dua_denoiser_step: step begin + 18 conv + 4 deconv + materialise passes + fused tail
            plan.native_step(nv.MODE_DDPM, row_of_step=row_of_step, coef_table=coef_table, use_sum=False)

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
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        dt = D.max_over_ranks(time.perf_counter() - t0)
        D.barrier()
        finite = bool(torch.isfinite(plan.x_state).all())

    # config 2 end to end, whatever --steps says: the 1000-step loop through the public sampler API, on every rank (replicas).
    # THIS is the headline: the loop the parity test covers (test_thousand_step_p_sample_loop_matches_oracle), with the two
    # exact-fp32 finishing steps of an fp16 plan inside the timed region; the K replayed steps above are kept beside it.  It runs
    # BEFORE rank 0's per-kernel measurements (the bare-MFMA probe holds the chip at its power cap for two seconds).
    loop = None
    if not args.no_graph and not args.no_full_loop:
        loop = full_loop_config2(net, image, (B, CLASSES, 96, 96, 96), D)
    with torch.no_grad():
        roof = None
        if rank == 0 and not args.no_roofline:
            ops.to_channels_last(x_T, plan.xin, 0, plan.C)       # a defined input for the per-launch timings
            roof = conv_roofline(plan, args.dtype)
    if rank == 0 and roof is not None:
        roof["traffic"] = getattr(args, "traffic", None)   # measured by main() before this process touched the GPU
        if args.dtype == "f16":
            # the nominal 2.5 PFLOP/s is not reachable on dense random operands: the chip lowers its clock under the load.
            # `frac` stays against the nominal peak; this is what the matrix pipes alone sustain on this device right now.
            ceil_tf, clock = mfma_ceiling(dev)
            roof["measured_mfma_ceiling"] = round(ceil_tf, 1)
            roof["measured_mfma_ceiling_clock_ghz"] = round(clock, 3)
            roof["frac_of_ceiling"] = round(roof["achieved"] / ceil_tf, 4)
            # the matrix pipes run what is EXECUTED: the folded launch is priced by its executed figure against the ceiling
            roof["largest_launch"]["frac_of_ceiling"] = round(roof["largest_launch"].get("executed_tflops", roof["largest_launch"]["tflops"]) / ceil_tf, 4)
    if rank == 0:
        replayed_ms = dt / args.steps * 1e3
        ms = loop["seconds"] / loop["steps"] * 1e3 if loop is not None else replayed_ms
        line = {
            "metric": "denoised voxel-steps/sec on 96^3 16-class AMOS patches",
            "value": world * B * VOX / (ms * 1e-3), "unit": "voxel-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "DiffUNet 96^3 patch, 16 classes, 1000-step DDPM p_sample_loop (BASELINE.json configs[1]); "
                                   "one patch per GPU, replicas only",
                       "patch": [96, 96, 96], "classes": CLASSES, "batch_per_gpu": B, "graph_replay": not args.no_graph,
                       "noise": "in-kernel Philox4x32-10", "weights": "torch.manual_seed(0) default init"},
            "headline": ("full_loop: value / ms_per_step = the whole 1000-step net.diffusion.p_sample_loop through the API (second "
                         "call; fp32 finishing steps included), max over ranks; replayed_step_ms = the --steps replays of the "
                         "captured fp16 step after --warmup replays") if loop is not None else "replayed steps (--no-full-loop / --no-graph)",
            "replayed_step_ms": replayed_ms, "replayed_value": world * B * VOX * args.steps / dt,
            "step_tflops": B * 1.0564e12 / (ms * 1e-3) / 1e12, "finite": finite,
        }
        if loop is not None:
            line["full_loop"] = loop
        if roof is not None:
            line["roofline"] = roof
        if world == 1 and not args.no_roofline and not args.no_graph and not args.no_full_loop:
            if args.dtype == "f16" and not args.no_f32:
                line["f32"] = f32_pass_config2(state, image, dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(state, host_threads())
        print(json.dumps(line), flush=True)


def run_config3(args, D):
    """Sliding-window DDIM inference of one synthetic volume, windows sharded over the ranks, one all-gather."""
    from diff_unet_amos_amd import inference
    from diff_unet_amos_amd.diff_unet import DiffUNet
    dev, rank, world = D.dev, D.rank, D.world
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES, sample_steps=50).to(dev).eval()
    state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    vol = torch.rand(1, 1, *args.volume, generator=torch.Generator().manual_seed(1)).to(dev)      # same volume on every rank
    nwin = len(inference._plan(vol, (96, 96, 96), 0.25)[4])
    swb = args.sw_batch
    timings = {}

    def one_volume():
        if world > 1:
            out = inference.sharded_sliding_window_inference(vol, (96, 96, 96), swb, net, 0.25, pred_type="ddim_sample",
                                                            gather_dtype=torch.float16 if args.gather_fp16 else None,
                                                            timings=timings)
        else:
            out = inference.sliding_window_inference(vol, (96, 96, 96), swb, net, 0.25, pred_type="ddim_sample")
        return inference.binarise(out)

    with torch.no_grad():
        # pack weights and capture the step graph of every batch size this rank's sampler passes will have (sharded: the rank's
        # windows in balanced calls, e.g. 6 windows at sw_batch_size 4 -> 3 + 3; single process: slices of sw_batch_size + tail)
        mine = len(range(rank, nwin, world))
        sizes = set(inference.balanced_batches(mine, swb)) if world > 1 else {min(swb, nwin), nwin % swb or min(swb, nwin)}
        for b in sorted(sizes or {1}):
            net(vol[:, :, :96, :96, :96].contiguous().repeat(b, 1, 1, 1, 1), pred_type="ddim_sample")
        for _ in range(args.warmup):
            one_volume()
        D.barrier()
        timings.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            seg = one_volume()
        torch.cuda.synchronize()
        dt = D.max_over_ranks(time.perf_counter() - t0)
        D.barrier()
        roof = None
        if rank == 0 and not args.no_roofline:      # the conv kernel over one denoiser evaluation of the sampler's own plan
            roof = conv_roofline(net._rt.plan(swb, (96, 96, 96), dev), "f16")
            roof["note"] = f"one denoiser evaluation at batch {swb} (the windows of one sampler pass)"
            roof["traffic"] = getattr(args, "traffic", None)       # bytes per launch of the wide kernel at that batch (PMC child passes)
    if rank == 0:
        per = dt / args.steps
        gather_s = timings.get("all_gather_s", 0.0) / max(1, args.steps)
        line = {
            "metric": "sliding-window DDIM inference: denoised voxel-steps/sec over a full volume", "value": nwin * VOX * 50 / per,
            "unit": "voxel-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"AMOS-sized volume {'x'.join(map(str, args.volume))}, 96^3 windows (overlap 0.25 -> {nwin}), 50-step DDIM, windows "
                                   "sharded over the ranks, one all_gather_into_tensor of the per-window sums (BASELINE.json configs[2])",
                       "windows": nwin, "sw_batch_size": swb, "gather_dtype": "f16" if args.gather_fp16 else "f32"},
            "seconds_per_volume": per, "all_gather_seconds": gather_s, "all_gather_share": gather_s / per if per > 0 else None,
            "gathered_bytes": timings.get("gathered_bytes"), "foreground_fraction": float(seg.mean())}
        if roof is not None:
            line["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(state, host_threads())
            line["cpu_baseline"]["sample"] += "; a volume is 48 windows x (50 such evaluations + one encoder pass)"
        print(json.dumps(line), flush=True)


def run_config4(args, D):
    """DDP training step on synthetic batches; gradients averaged over RCCL."""
    from diff_unet_amos_amd.diff_unet import DiffUNet
    from diff_unet_amos_amd.training import NativeConvTrainer, allreduce_mean_
    dev, rank, world = D.dev, D.rank, D.world
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=CLASSES, features=FEATURES).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    B = args.batch or 2                            # default: cfg/amos/train.yaml, batch 10 over 5 GPUs; --batch 1 = BASELINE.md section 4
    tr = NativeConvTrainer(net, dtype=torch.float16, overlap=not args.flat_allreduce, graph=args.train_graph)
    g = torch.Generator(device=dev).manual_seed(10 + rank)
    image = torch.rand(B, 1, 96, 96, 96, device=dev, generator=g)
    labels = (torch.rand(B, CLASSES, 96, 96, 96, device=dev, generator=g) > 0.8).float()
    for _ in range(max(1, args.warmup)):
        tr.step(image, labels)
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(image, labels)
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    D.barrier()
    # what one flat all-reduce of all gradients costs on its own (upper bound of the share when it is not overlapped)
    grads = [torch.zeros_like(p) for p in net.parameters()]
    allreduce_mean_(grads)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        allreduce_mean_(grads)
    torch.cuda.synchronize()
    ar = D.max_over_ranks((time.perf_counter() - t1) / 5) if world > 1 else 0.0
    roof = None
    if rank == 0 and not args.no_roofline:
        probe = tr if not args.train_graph else NativeConvTrainer(net, dtype=torch.float16, overlap=False, graph=False)
        roof = wgrad_roofline(lambda: probe.step(image, labels))
    if rank == 0:
        per = dt / args.steps
        line = {
            "metric": "DDP training samples/sec on synthetic 96^3 16-class batches", "value": world * B / per, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"q_sample + denoise + mse/bce/dice + backward + AdamW, 96^3, 16 classes, {B} sample(s) per GPU "
                                   "(BASELINE.json configs[3])", "batch_per_gpu": B,
                       "gradient_sync": "flat all-reduce after backward" if args.flat_allreduce else "DDP reducer, 32 MB buckets overlapped with backward",
                       "graph": bool(args.train_graph)},
            "flat_allreduce_seconds": ar, "flat_allreduce_share": ar / per if per > 0 else None,
            "gradient_bytes": sum(p.numel() for p in net.parameters()) * 4, "loss": float(loss),
            "step_tflops": B * 3 * (1.0564e12 + 0.2798e12) / per / 1e12}
        if roof is not None:
            roof["traffic"] = getattr(args, "traffic", None)       # HBM bytes per launch of conv3d_k3_wgrad_fo_kernel (PMC child passes)
            line["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_training_baseline(state, host_threads())
        print(json.dumps(line), flush=True)


def run_config5(args, D):
    """DiffSwinUNETR reverse-diffusion steps (replicas only, like config 2)."""
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd import ops
    from diff_unet_amos_amd.diff_swin_unetr import DiffSwinUNETR
    dev, rank, world = D.dev, D.rank, D.world
    dtype = torch.float16 if args.dtype == "f16" else torch.float32
    torch.manual_seed(0)
    net = DiffSwinUNETR(in_channels=1, out_channels=CLASSES, feature_size=48, compute_dtype=dtype).to(dev).eval()
    net.fold_upconv = not args.no_fold
    net.fold_residual = not args.no_fold_residual
    state = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    B = args.batch or 1
    image = torch.rand(B, 1, 96, 96, 96, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    plan = net._rt.plan(B, (96, 96, 96), dev)
    diffusion = net.diffusion
    with torch.no_grad():
        net.embed_model(image)                     # encoder once per patch (not in the timed region)
        x_T = torch.randn(B, CLASSES, 96, 96, 96, device=dev)
        ops.to_channels_last(x_T, plan.x_state, 0, plan.cx)
        ops.to_channels_last(x_T, plan.xin, 0, plan.C)
        T = diffusion.num_timesteps
        order = list(range(T))[::-1]
        coef_table = diffusion.ddpm_coef(torch.tensor(order)).to(dev).contiguous()
        row_of_step = torch.tensor(order, dtype=torch.int32, device=dev)
        plan.counter.zero_()
        plan.new_seed(3 + rank)

        def one_step():
            ops.step_begin(B, plan.temb_table, plan.cur_add, row_of_step=row_of_step, counter=plan.counter,
                           coef_table=coef_table, cur_coef=plan.cur_coef, step_word=plan.step_word, err_word=plan.err_word,
                           clear=plan.den_stats)
            plan.denoiser_body(zero_stats=False)
            plan.tail(nv.MODE_DDPM)

        assert args.warmup + args.steps + 2 <= T
        if args.no_graph:
            one_step()
            torch.cuda.synchronize()

            def run(times):
                for _ in range(times):
                    one_step()
        else:
            run = plan.capture_step(one_step).replay
        run(args.warmup)
        D.barrier()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = D.max_over_ranks(time.perf_counter() - t0)
        D.barrier()
        finite = bool(torch.isfinite(plan.x_state).all())
        roof = None
        if rank == 0 and not args.no_roofline:
            # the 3x3x3 convolutions are the largest share of this step too (20 launches, ~45 %): the same HIP-event timing as
            # config 2 (every launch replayed alone, back to back); useful FLOPs = 2 * 27 * Cin * Cout * voxels, so the 48-wide
            # layers are charged for their padding to 32-channel chunks and 64-output tiles
            avg_ms, per_step, by_launch = time_conv_launches(plan, 20)
            # (a folded call's arguments are (cat, skip channels, ...): the algorithmic figure is conv1 over the whole 2 x cout concat)
            fl = [2.0 * 27 * (2 * a[1] if k.get("__upconv__") else a[1]) * (a[10] if k.get("__upconv__") else a[5]) *
                  a[0].shape[0] * a[0].shape[1] * a[0].shape[2] * a[0].shape[3] for a, k in _LAST_CONV_CALLS]
            peak = PEAK_F16_TFLOPS if args.dtype == "f16" else PEAK_F32_TFLOPS
            roof = roofline_by_kernel(fl, by_launch, [conv_kernel_of(c) for c in _LAST_CONV_CALLS], peak,
                                      "decoder1 conv1 96->48 @96^3")
            roof["note"] = "side-stream launches are timed as they are launched in the step (one workgroup per CU)"
    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {
            "metric": "denoised voxel-steps/sec on 96^3 16-class AMOS patches", "value": world * B * VOX * args.steps / dt,
            "unit": "voxel-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "DiffSwinUNETR (feature_size 48) 96^3 patch, 16 classes, DDPM p_sample steps of the 1000-step "
                                   "process (BASELINE.json configs[4]); one patch per GPU, replicas only",
                       "patch": [96, 96, 96], "classes": CLASSES, "batch_per_gpu": B, "graph_replay": not args.no_graph,
                       "noise": "in-kernel Philox4x32-10", "weights": "torch.manual_seed(0) default init"},
            "finite": finite}
        if roof is not None:
            if roof.get("kernel") == CONV_KERNEL:
                roof["traffic"] = getattr(args, "traffic", None)
            else:
                roof["other_kernels"].get(CONV_KERNEL, {})["traffic"] = getattr(args, "traffic", None)
            line["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(state, host_threads(), swin=True)
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-full-loop", action="store_true", help="config 2: skip the 1000-step p_sample_loop pass (full_loop object)")
    ap.add_argument("--no-f32", action="store_true", help="config 2: skip the short fp32 pass (f32 object)")
    ap.add_argument("--batch", type=int, default=None, help="configs 2 / 5: patches per GPU (default 1, BASELINE config 2); config 4: samples per GPU (default 2)")
    ap.add_argument("--volume", type=int, nargs=3, default=[256, 256, 192], help="config 3: volume extents (BASELINE config 3: 256 256 192)")
    ap.add_argument("--sw-batch", type=int, default=4, help="config 3: windows per sampler pass (cfg/btcv/test.yaml:4 of the reference: 4)")
    ap.add_argument("--gather-fp16", action="store_true", help="config 3: all-gather the window sums in fp16")
    ap.add_argument("--flat-allreduce", action="store_true", help="config 4: one flat all-reduce instead of DDP buckets")
    ap.add_argument("--train-graph", action="store_true", help="config 4: whole step as one HIP graph")
    ap.add_argument("--fold-min-tiles", type=int, default=None, help="diagnostics: engine.Plan.UPCONV_MIN_TILES for this run (0 = fold every level that can; a huge value = never)")
    ap.add_argument("--no-fold", action="store_true", help="diagnostics: keep every transposed convolution + convolution on two launches (net.fold_upconv = False; A/B)")
    ap.add_argument("--no-fold-residual", action="store_true", help="diagnostics (config 5): decoder1's 1x1x1 residual branch as transposed convolution + token GEMM (net.fold_residual = False; A/B)")
    ap.add_argument("--conv-variant", type=int, default=0, help="diagnostics: dua_conv3_desc.policy of every convolution launch (same-box A/B of launch forms)")
    args = ap.parse_args()
    defaults = {2: (200, 20), 3: (1, 0), 4: (5, 2), 5: (100, 10)}[args.config]
    args.steps = defaults[0] if args.steps is None else args.steps
    args.warmup = defaults[1] if args.warmup is None else args.warmup

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    if args.gpus == 1 and not (args.no_roofline or args.no_traffic) and args.config in (2, 3, 4, 5):
        # two short child runs of this script under rocprofv3 --pmc, started before this process initialises the GPU; the kernel
        # is the one the roofline object of that config describes
        if args.config == 2:
            args.traffic = measure_traffic(args.dtype, WIDE_KERNEL if (args.dtype == "f16" and not args.conv_variant) else CONV_KERNEL)
        elif args.config == 3:
            # one sampler pass of sw_batch windows (a 168 x 168 x 96 volume = 2 x 2 x 1 windows at sw_batch_size 4): the same launches
            # as the benchmarked volume's passes
            v = ["168", "168", "96"] if args.sw_batch == 4 else [str(96 + 72 * (args.sw_batch - 1)), "96", "96"]
            args.traffic = measure_traffic("f16", WIDE_KERNEL, 3, ["--volume"] + v + ["--sw-batch", str(args.sw_batch), "--steps", "1", "--warmup", "0"])
        elif args.config == 4:
            args.traffic = measure_traffic("f16", "conv3d_k3_wgrad_fo_kernel", 4, ["--flat-allreduce"] + (["--batch", str(args.batch)] if args.batch else []))
        else:
            args.traffic = measure_traffic(args.dtype, CONV_KERNEL, 5)
    D = Dist(args)
    if args.conv_variant:
        from diff_unet_amos_amd import ops
        global _CONV_VARIANT
        _CONV_VARIANT = args.conv_variant
        ops.CONV_POLICY = args.conv_variant
    {2: run_config2, 3: run_config3, 4: run_config4, 5: run_config5}[args.config](args, D)
    D.finish()


if __name__ == "__main__":
    main()
