"""Config 4 (training step, batch 2, 96^3, 16 classes) on the HIP training path (training.NativeConvTrainer): eager or
as one replayed HIP graph.  One JSON line.  Usage: python tools/bench_train.py [--steps K] [--graph]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd.diff_unet import DiffUNet          # noqa: E402
from diff_unet_amos_amd.training import NativeConvTrainer         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--classes", type=int, default=16)
ap.add_argument("--size", type=int, default=96)
ap.add_argument("--dtype", default="float16")
ap.add_argument("--graph", action="store_true", help="replay the whole training step as one HIP graph")
ap.add_argument("--no-gc", action="store_true", help="disable the Python cyclic GC during the timed loop (diagnosis)")
ap.add_argument("--ab-wgrad", default="", help="comma list of weight-gradient launch forms (dua_conv3_desc.policy, ops.WGRAD_POLICY): time the loop once per value, same process")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DiffUNet(in_channels=1, out_channels=a.classes).to(dev)
tr = NativeConvTrainer(net, dtype=getattr(torch, a.dtype), graph=a.graph)
image = torch.rand(a.batch, 1, a.size, a.size, a.size, device=dev)
labels = (torch.rand(a.batch, a.classes, a.size, a.size, a.size, device=dev) > 0.8).float()
for _ in range(a.warmup):
    tr.step(image, labels)
torch.cuda.synchronize()
if a.no_gc:
    import gc
    gc.collect()
    gc.disable()
per = []
for _ in range(a.steps):
    t0 = time.perf_counter()
    loss = tr.step(image, labels)
    torch.cuda.synchronize()
    per.append(time.perf_counter() - t0)
dt = sum(per) / len(per)
ab = {}
if a.ab_wgrad:
    from diff_unet_amos_amd import ops
    for v in [int(x) for x in a.ab_wgrad.split(",")]:
        ops.WGRAD_POLICY = v
        tr.step(image, labels)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.steps):
            t0 = time.perf_counter()
            tr.step(image, labels)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ab.setdefault(str(v), []).append(round(sorted(ts)[len(ts) // 2] * 1e3, 2))
    ops.WGRAD_POLICY = 0
print(json.dumps({"metric": "train_step_time", "value": dt * 1e3, "unit": "ms", "median_ms": sorted(per)[len(per) // 2] * 1e3, "per_step_ms": [round(x * 1e3, 2) for x in per], "ab_wgrad_median_ms": ab, "native": True,
                  "path": "HIP conv fwd/dgrad/wgrad + fused InstanceNorm/LeakyReLU/add fwd/bwd under autograd; fused loss, pooling, 1x1-head and transposed-conv (in-place concat) kernels; AdamW = torch; " + a.dtype,
                  "graph": a.graph, "batch": a.batch,
                  "size": a.size, "classes": a.classes, "loss": float(loss),
                  "max_mem_GiB": torch.cuda.max_memory_allocated() / 2**30}))
