#!/usr/bin/env python3
"""Soak of BASELINE config 2 on the shipped kernels: the same 1000-step p_sample_loop (same weights, image, x_T and Philox key)
N times -- every run must return the same bits (order-independent statistics, no race in any kernel of the step).
usage: soak_config2.py [runs]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd.diff_unet import DiffUNet


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=16).to(dev).eval()
    image = torch.rand(1, 1, 96, 96, 96, device=dev)
    xT = torch.randn(1, 16, 96, 96, 96, device=dev)
    plan = net._rt.plan(1, (96, 96, 96), dev)
    first = None
    with torch.no_grad():
        net.embed_model(image)
        for i in range(runs):
            out = plan.sample_loop(net.diffusion, "ddpm", noise=xT, seed=1234)["sample"]
            assert bool(torch.isfinite(out).all()), i
            if first is None:
                first = out.clone()
            else:
                assert torch.equal(out, first), f"run {i} differs from run 0: max |d| = {float((out - first).abs().max())}"
            print(f"run {i}: identical, mean {float(out.mean()):+.6f}", flush=True)
    print(f"{runs} x 1000 steps: bit-identical")


if __name__ == "__main__":
    main()
