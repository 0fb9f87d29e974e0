#!/usr/bin/env python3
"""Same-process A/B of SwinPlan switches on the config-5 step (96^3, 16 classes, one patch): each setting is captured into its
own HIP graph, rounds are interleaved.  usage: bench_swin_ab.py [rounds] [steps per round]
Settings: two_streams, fused_tail, background_convs, fused_mlp, fused_linear, fused_max_c, fused_reduction (swin_engine.SwinPlan attributes)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import _native as nv, ops                     # noqa: E402
from diff_unet_amos_amd.diff_swin_unetr import DiffSwinUNETR          # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffSwinUNETR(in_channels=1, out_channels=16, feature_size=48).to(dev).eval()
    image = torch.rand(1, 1, 96, 96, 96, device=dev)
    plan = net._rt.plan(1, (96, 96, 96), dev)
    settings = {"default": {}, "fused kernels at stage 1 too": {"fused_max_c": 96}, "fused reduction": {"fused_reduction": True},
                "one-kernel MLP off": {"fused_mlp": False}, "one stream": {"two_streams": False}, "decoder1 output materialised for the tail": {"fused_tail": False}, "side-stream convolutions at two workgroups per CU": {"background_convs": False}, "library GEMMs": {"fused_linear": False},
                "library qkv": {"tl_qkv": False}, "library proj + scatter kernel": {"tl_proj": False},
                "library conv3 + stats kernel": {"tl_conv3": False},
                "side-stream tails at full width": {"background_tails": False},
                "side-stream blocks: conv3 ahead of the 3x3x3 convolutions": {"conv3_first": True},
                "side-stream conv3 one workgroup per CU": {"background_conv3": 1},
                "side-stream conv3 two workgroups per CU": {"background_conv3": 2},
                "encoder1 with stage 0": {"side_plan": {0: (0,), 2: (3, 2, 1)}},
                "encoder1 after stage 0": {"side_plan": {1: (0,), 2: (3, 2, 1)}},
                "encoder1, 2 after stage 0": {"side_plan": {1: (0, 1), 2: (3, 2)}},
                "encoder1 first after stage 1": {"side_plan": {2: (0, 3, 2, 1)}},
                "encoder1 after stage 2": {"side_plan": {2: (3, 2, 1), 3: (0,)}}}
    if len(sys.argv) > 3:
        settings = {k: v for k, v in settings.items() if k == "default" or any(w in k for w in sys.argv[3].split(","))}
    base = {k: getattr(plan, k) for k in ("two_streams", "fused_mlp", "fused_linear", "fused_max_c", "fused_reduction", "tl_qkv",
                                          "tl_proj", "tl_conv3", "background_convs", "fused_tail", "side_plan", "background_tails", "background_conv3", "conv3_first")}
    graphs = {}
    with torch.no_grad():
        net.embed_model(image)
        T = net.diffusion.num_timesteps
        order = list(range(T))[::-1]
        coef_table = net.diffusion.ddpm_coef(torch.tensor(order)).to(dev).contiguous()
        row_of_step = torch.tensor(order, dtype=torch.int32, device=dev)
        plan.new_seed(3)

        def one_step():
            ops.step_begin(1, plan.temb_table, plan.cur_add, row_of_step=row_of_step, counter=plan.counter, coef_table=coef_table,
                           cur_coef=plan.cur_coef, step_word=plan.step_word, err_word=plan.err_word)
            plan.denoiser_body()
            plan.tail(nv.MODE_DDPM)

        for name, kv in settings.items():
            for k, v in {**base, **kv}.items():
                setattr(plan, k, v)
            plan.counter.zero_()
            graphs[name] = plan.capture_step(one_step)
        res = {n: [] for n in graphs}
        for _ in range(rounds):
            for name, g in graphs.items():
                plan.counter.zero_()
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g.replay(steps)
                torch.cuda.synchronize()
                res[name].append((time.perf_counter() - t0) / steps * 1e3)
    for name, v in res.items():
        v = sorted(v)
        print(f"{name:22s} median {v[len(v) // 2]:.3f} ms  best {v[0]:.3f}  ({' '.join(f'{x:.3f}' for x in v)})")


if __name__ == "__main__":
    main()
