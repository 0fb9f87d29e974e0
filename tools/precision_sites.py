"""Which fp16 rounding sites of one denoiser evaluation account for the logit error?  (CPU experiment on the oracle
network: rounding emulated by hooks, everything else fp32.)  Sites: XT = the state x_t at the network input,
W = convolution / transposed-convolution weights, IN = every convolution's input activation, RAW = every convolution's
stored output.  Used to decide where extra precision buys parity."""
import copy
import sys

import torch
import torch.nn as nn

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle.unet_ref import RefDiffUNet  # noqa: E402


def q(t):
    return t.half().float()


def run(net, image, x, step, sites):
    net = copy.deepcopy(net)
    hooks = []
    convs = [m for m in net.modules() if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d))]
    if "W" in sites:
        with torch.no_grad():
            for m in convs:
                m.weight.copy_(q(m.weight))
    for m in convs:
        if "IN" in sites:
            hooks.append(m.register_forward_pre_hook(lambda mod, a: (q(a[0]),)))
        if "RAW" in sites and m.kernel_size[0] == 3:
            hooks.append(m.register_forward_hook(lambda mod, a, o: q(o)))
    xx = q(x) if "XT" in sites else x
    with torch.no_grad():
        out = net(image=image, x=xx, step=step, pred_type="denoise")
    for h in hooks:
        h.remove()
    return out


def main():
    torch.manual_seed(0)
    S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 16
    net = RefDiffUNet(3, 1, C).eval()
    image = torch.rand(1, 1, S, S, S, generator=torch.Generator().manual_seed(1))
    x = torch.randn(1, C, S, S, S, generator=torch.Generator().manual_seed(3))
    step = torch.tensor([500])
    ref = run(net, image, x, step, ())
    print(f"logit std {ref.std():.4f}")
    for sites in (("XT",), ("W",), ("IN",), ("RAW",), ("W", "IN"), ("W", "IN", "RAW"), ("XT", "W", "IN", "RAW")):
        d = (run(net, image, x, step, sites) - ref).abs()
        flips = ((run(net, image, x, step, sites) > 0) != (ref > 0)).float().mean()
        print(f"{'+'.join(sites):14s} max {d.max():.2e} mean {d.mean():.2e} rms {d.pow(2).mean().sqrt():.2e} sign flips {flips:.2e}")


main()
