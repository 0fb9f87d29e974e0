#!/usr/bin/env python3
"""Weight gradient of the four transposed convolutions of the 96^3 denoiser at batch 2 (config 4): time per layer, HIP events,
the gradient read in place from its half of the concat buffer's gradient as the training step does.
usage: bench_deconv_bwd.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops                     # noqa: E402

SHAPES = [(6, 512, 256), (12, 256, 128), (24, 128, 64), (48, 64, 64)]      # input extent, Cin, Cout


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev, dt, N = "cuda", torch.float16, 2
    for S, cin, cout in SHAPES:
        x = torch.randn(N, S, S, S, cin, device=dev).to(dt)
        dcat = torch.randn(N, 2 * S, 2 * S, 2 * S, 2 * cout, device=dev).to(dt)
        w = torch.randn(cin, cout, 2, 2, 2, device=dev)
        ts = []
        for r in range(rounds + 2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.deconv_k2s2_bwd(x, cin, 0, dcat, cout, cout, w, need_dx=False, need_dw=True)
            b.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        mb = (x.numel() + dcat.numel() // 2) * 2 / 1e6
        print(f"{S}^3 -> {2 * S}^3  {cin:3d} -> {cout:3d}: median {ts[len(ts) // 2]:7.1f} us (with the partition reduce and the zero fill of dW), "
              f"operands {mb:6.1f} MB")


if __name__ == "__main__":
    main()
