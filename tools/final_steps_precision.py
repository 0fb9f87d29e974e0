#!/usr/bin/env python3
"""What the LAST steps of the 1000-step DDPM loop (BASELINE config 2, gaussian_diffusion.py:441-535) contribute to the fp16
deviation of the final sample: the fused fp16 path for the first T - k steps, the exact-fp32 path for the last k, against the
oracle with the same noise.  (As t -> 0 the posterior mean turns into the clamped prediction itself -- coef1 -> 1, coef2 -> 0 --
so the final sample carries the error of the last few evaluations, not an accumulated drift.)
usage: final_steps_precision.py [classes] [size]        (default 2 classes, 32^3, tiny widths; 16 classes uses default widths)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diff_unet_amos_amd.diff_unet import DiffUNet
from oracle.unet_ref import RefDiffUNet, binarise


def dice(a, b):
    out = []
    for c in range(a.shape[1]):
        x, y = a[:, c].bool(), b[:, c].bool()
        den = float(x.sum() + y.sum())
        out.append(2.0 * float((x & y).sum()) / den if den else 1.0)
    return out


def main():
    classes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    kw = dict(in_channels=1, out_channels=classes)
    if classes == 2:
        kw["features"] = (8, 8, 16, 32, 64, 8)
    torch.manual_seed(0)
    ref = RefDiffUNet(**kw).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if ".adn.N." in n:
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if n.endswith("weight") else 0.0))
    nets = {}
    for dt in (torch.float16, torch.float32):
        net = DiffUNet(compute_dtype=dt, **kw)
        net.load_state_dict(ref.state_dict())
        nets[dt] = net.cuda().eval()
    g = torch.Generator().manual_seed(13)
    shape = (1, classes, S, S, S)
    image = torch.rand(1, 1, S, S, S, generator=g)
    xT = torch.randn(*shape, generator=g)
    T = 1000
    draws = [torch.randn(*shape, generator=g) for _ in range(T)]
    ks = (0, 1, 2, 5, 10, 20, 50, 100)
    with torch.no_grad():
        emb_r = ref.embed_model(image)
        img = xT
        for k, i in enumerate(reversed(range(T))):
            img = ref.diffusion.p_sample(ref.model, img, torch.tensor([i]), draws[k], model_kwargs={"image": image, "embeddings": emb_r})["sample"]
        want = img
        n16, n32 = nets[torch.float16], nets[torch.float32]
        n16.embed_model(image.cuda())
        plan = n16._rt.plan(1, (S, S, S), torch.device("cuda", 0))
        snaps = {T - k: None for k in ks if k}
        out = plan.sample_loop(n16.diffusion, "ddpm", noise=xT.cuda(), step_noise=draws, snapshots=snaps)
        kw32 = {"image": image.cuda(), "embeddings": n32.embed_model(image.cuda())}
        print(f"{classes} classes, {S}^3: final-sample deviation from the oracle (max / mean |dx|, worst-class Dice of the thresholded sample)")
        for k in ks:
            if k == 0:
                x = out["sample"]
            else:
                x = snaps[T - k]
                for j in range(T - k, T):
                    i = T - 1 - j
                    x = n32.diffusion.p_sample(n32.model, x, torch.tensor([i], device="cuda"), eps=draws[j].cuda(), model_kwargs=kw32)["sample"]
            d = (x.cpu() - want).abs()
            dc = dice(binarise(x.cpu()), binarise(want))
            print(f"  last {k:4d} steps in fp32: max {float(d.max()):.2e} mean {float(d.mean()):.2e}  1 - min Dice {1 - min(dc):.2e}")


main()
