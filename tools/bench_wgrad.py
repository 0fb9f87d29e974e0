"""Time dua_conv3d_k3_wgrad on the layer shapes of config 4.  Usage: python tools/bench_wgrad.py [--batch 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--dtype", default="float16")
a = ap.parse_args()
dt = getattr(torch, a.dtype)
dev = torch.device("cuda:0")
shapes = [(96, 64, 64), (96, 128, 64), (96, 32, 64), (48, 64, 64), (48, 128, 64), (24, 128, 128), (24, 256, 128),
          (12, 256, 256), (12, 512, 256), (6, 512, 512)]
for S, Cin, Cout in shapes:
    x = torch.randn(a.batch, S, S, S, Cin, device=dev, dtype=dt)
    dy = torch.randn(a.batch, S, S, S, Cout, device=dev, dtype=dt)
    dw = torch.zeros(Cout, Cin, 3, 3, 3, device=dev)
    for _ in range(2):
        ops.conv3d_k3_wgrad(x, Cin, 0, dy, Cout, 0, dw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv3d_k3_wgrad(x, Cin, 0, dy, Cout, 0, dw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * a.batch * S ** 3 * Cin * Cout * 27
    print(f"{S:3d}^3 {Cin:4d}->{Cout:4d}  {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
