#!/usr/bin/env python3
"""Which aten operators of one eager training step (BASELINE config 4, batch 2) launch the small torch-side kernels (fills, copies,
elementwise): torch.profiler over one step, operators grouped with their two innermost Python frames."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

from diff_unet_amos_amd.diff_unet import DiffUNet
from diff_unet_amos_amd.training import NativeConvTrainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DiffUNet(in_channels=1, out_channels=16, features=(64, 64, 128, 256, 512, 64)).to(dev)
tr = NativeConvTrainer(net, dtype=torch.float16, overlap=False, graph=False)
g = torch.Generator(device=dev).manual_seed(10)
image = torch.rand(2, 1, 96, 96, 96, device=dev, generator=g)
labels = (torch.rand(2, 16, 96, 96, 96, device=dev, generator=g) > 0.8).float()
for _ in range(2):
    tr.step(image, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(image, labels)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=3)
rows = [e for e in ka if e.key.startswith("aten::") and e.device_time_total > 0 and not e.key.startswith("aten::_")]
rows.sort(key=lambda e: -e.count)
for e in rows[:45]:
    st = " <- ".join(s.split("/")[-1] for s in e.stack[:3])
    print(f"{e.count:4d} x {e.key:28s} device {e.device_time_total:9.1f} us   {st}")
