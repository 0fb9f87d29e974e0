"""How far is the native fp32 training path's gradient from the oracle's, tensor by tensor, and is the oracle itself that far
from an fp64 evaluation?  For each data seed: one process, both samples of the two-rank test's batch (tests/test_training_harness.py)
through NativeConvTrainer (fp32, eager) -> p.grad; the oracle network on the CPU in fp32 and in fp64.  Prints, per seed, the
worst per-tensor relative L2 (convolution biases in front of InstanceNorm excluded: their true gradient is zero) of
native-vs-fp64 and oracle-fp32-vs-fp64, and for the worst tensors of the chosen seed the full rows.
  python tools/gradient_seed_scan.py [first_seed] [count]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def oracle_grads(T, seed, dtype):
    from oracle.train_ref import RefLoss, ref_training_step
    from oracle.unet_ref import RefDiffUNet
    torch.manual_seed(0)
    init = T.DiffUNet(**T.KW).state_dict()
    ref = RefDiffUNet(**T.KW)
    ref.load_state_dict(init)
    ref = ref.to(dtype)
    # the sinusoid table of the timestep is built in fp32: cast it at the first Linear layer
    ref.model.temb.dense[0].register_forward_pre_hook(lambda m, a: (a[0].to(m.weight.dtype),))
    crit = RefLoss()
    image, labels, noise, t = T._data(2, seed)
    total = 0
    for i in range(2):
        total = total + ref_training_step(ref, image[i:i + 1].to(dtype), labels[i:i + 1].to(dtype), crit, noise[i:i + 1].to(dtype),
                                          t[i:i + 1]) / 2
    total.backward()
    return {k: p.grad.detach().double() for k, p in ref.named_parameters()}


def native_grads(T, seed):
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = T.DiffUNet(**T.KW).to(dev)
    tr = NativeConvTrainer(net, lr=1e-3, dtype=torch.float32)
    image, labels, noise, t = T._data(2, seed)
    tr.step(image.to(dev), labels.to(dev), noise=noise.to(dev), t=t.to(dev))
    return {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()}


def rel(a, b):
    return float(((a - b) ** 2).sum().sqrt() / b.pow(2).sum().sqrt().clamp(min=1e-300))


def main():
    import test_training_harness as T
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for seed in range(first, first + count):
        g64 = oracle_grads(T, seed, torch.float64)
        g32 = oracle_grads(T, seed, torch.float32)
        gn = native_grads(T, seed)
        keys = [k for k in g64 if not k.endswith("conv.bias")]
        rn = sorted(((rel(gn[k], g64[k]), k) for k in keys), reverse=True)
        ro = sorted(((rel(g32[k], g64[k]), k) for k in keys), reverse=True)
        bias_abs = max(float(gn[k].abs().max()) for k in g64 if k.endswith("conv.bias"))
        print(f"seed {seed:3d}: native vs fp64 worst {rn[0][0]:.2e} ({rn[0][1]}), median {rn[len(rn) // 2][0]:.2e}; "
              f"oracle fp32 vs fp64 worst {ro[0][0]:.2e} ({ro[0][1]}); max |conv.bias grad| native {bias_abs:.1e}", flush=True)
        if seed == first:
            for r_, k in rn[:12]:
                print(f"      {r_:.3e}  {k}   (oracle fp32 vs fp64: {rel(g32[k], g64[k]):.3e}; rms |g| {float(g64[k].pow(2).mean().sqrt()):.2e})")


if __name__ == "__main__":
    main()
