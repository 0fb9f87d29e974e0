#!/usr/bin/env python3
"""Same-process A/B of the two-part decoder convolution (engine.Plan.split_levels) on the config-2 step: one HIP graph per
setting, interleaved rounds.  usage: bench_split_ab.py [rounds] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import _native as nv, ops                     # noqa: E402
from diff_unet_amos_amd.diff_unet import DiffUNet                     # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=16, features=(64, 64, 128, 256, 512, 64)).to(dev).eval()
    image = torch.rand(1, 1, 96, 96, 96, device=dev)
    plan = net._rt.plan(1, (96, 96, 96), dev)
    hi = torch.cuda.Stream(device=dev, priority=-1)
    settings = {"one launch (no split)": ((), None), "split level 0": ((0,), None), "split level 0, main stream high priority": ((0,), hi)}
    graphs = {}
    with torch.no_grad():
        net.embed_model(image)
        T = net.diffusion.num_timesteps
        order = list(range(T))[::-1]
        coef_table = net.diffusion.ddpm_coef(torch.tensor(order)).to(dev).contiguous()
        row_of_step = torch.tensor(order, dtype=torch.int32, device=dev)
        plan.new_seed(3)
        all_levels = (0,)
        plan.split_levels = all_levels
        plan.refresh_weights()             # packs the two halves, allocates the partial buffer and the side stream

        def one_step():
            plan.native_step(nv.MODE_DDPM, row_of_step=row_of_step, coef_table=coef_table, use_sum=False)

        for name, (levels, stream) in settings.items():
            plan.split_levels = tuple(l for l in levels if l in all_levels)
            plan._step_ops_version = None              # re-record the op list for this setting; the weights (both forms were
                                                       # packed above) stay where the earlier graphs expect them
            plan.counter.zero_()
            one_step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if stream is None:
                with torch.cuda.graph(g):
                    one_step()
            else:
                with torch.cuda.graph(g, stream=stream):
                    one_step()
            graphs[name] = (g, plan._step_keep)
        res = {n: [] for n in graphs}
        for _ in range(rounds):
            for name, (g, _) in graphs.items():
                plan.counter.zero_()
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    g.replay()
                torch.cuda.synchronize()
                res[name].append((time.perf_counter() - t0) / steps * 1e3)
    # ---- eager (no graph): one C call per step; a CU-masked side stream can only be tried here ----
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def masked_stream(words):
        st = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * len(words))(*words)
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value, device=dev)

    eager = {"eager, one launch": ((), None), "eager, split": ((0,), "default"),
             "eager, split, side stream on every other CU": ((0,), masked_stream([0x55555555] * 8)),
             "eager, split, side stream on every fourth CU": ((0,), masked_stream([0x11111111] * 8))}
    with torch.no_grad():
        default_side = plan.side_stream
        for name, (levels, st) in eager.items():
            plan.split_levels = tuple(l for l in levels if l in all_levels)
            plan.side_stream = default_side if st in (None, "default") else st
            plan._step_ops_version = None              # re-record (levels / side stream changed); no re-packing
            res[name] = []
            for _ in range(rounds):
                plan.counter.zero_()
                one_step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    one_step()
                torch.cuda.synchronize()
                res[name].append((time.perf_counter() - t0) / steps * 1e3)
        plan.side_stream = default_side
    for name, v in res.items():
        v = sorted(v)
        print(f"{name:44s} median {v[len(v) // 2]:.3f} ms  best {v[0]:.3f}  ({' '.join(f'{x:.3f}' for x in v)})")


if __name__ == "__main__":
    main()
