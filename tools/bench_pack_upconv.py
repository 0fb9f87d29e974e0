#!/usr/bin/env python3
"""Time of ops.pack_upconv_weights (skip-half pack + composed weights + bias table: what the training step pays every step for the
folded level-0 convolution), launches back to back, HIP events.  usage: bench_pack_upconv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diff_unet_amos_amd import ops

dev = "cuda"
wc = torch.randn(64, 128, 3, 3, 3, device=dev)
wd = torch.randn(64, 64, 2, 2, 2, device=dev)
z = torch.zeros(64, device=dev)
for _ in range(3):
    ops.pack_upconv_weights(wc, z, wd, z, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.pack_upconv_weights(wc, z, wd, z, 64)
e1.record()
torch.cuda.synchronize()
print(f"pack_upconv_weights, 64 + 64 -> 64 (all three launches): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
