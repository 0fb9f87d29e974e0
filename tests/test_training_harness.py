"""Training side of the hot path (training.py): the reference's training step (train.py:258-268) runs through the
drop-in boundary ``DiffUNet.forward`` on the HIP forward + backward kernels; its logits, loss and every parameter
gradient match the CPU oracle under torch autograd; the fused loss covers the reference's loss configurations; the
one-process-per-rank gradient averaging equals single-process training on the union batch."""
import os
import re

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diff_unet_amos_amd.diff_unet import DiffUNet
from oracle.train_ref import RefLoss as Loss
from oracle.train_ref import ref_training_step
from oracle.unet_ref import RefDiffUNet

KW = dict(in_channels=1, out_channels=2, features=(8, 8, 16, 32, 64, 8))


def _data(n, seed):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(n, 1, 32, 32, 32, generator=g)
    labels = (torch.rand(n, 2, 32, 32, 32, generator=g) > 0.7).float()
    noise = torch.randn(n, 2, 32, 32, 32, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    return image, labels, noise, t


def _union_batch_reference(seed_data, lr=1e-3):
    """Single process, both samples through the ORACLE network (torch autograd on the CPU), mean of the two per-sample
    losses, one AdamW step: what two ranks with averaged gradients must reproduce.  Returns (state_dict, gradients)."""
    torch.manual_seed(0)
    init = DiffUNet(**KW).state_dict()
    ref = RefDiffUNet(**KW)
    ref.load_state_dict(init)
    opt = torch.optim.AdamW(ref.parameters(), lr=lr, weight_decay=1e-4)
    crit = Loss()
    image, labels, noise, t = _data(2, seed_data)
    opt.zero_grad()
    total = 0
    for i in range(2):
        total = total + ref_training_step(ref, image[i:i + 1], labels[i:i + 1], crit, noise[i:i + 1], t[i:i + 1]) / 2
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    opt.step()
    return ref.state_dict(), grads


def test_package_has_no_torch_convolution_path():
    """The product must not carry a second (torch / MIOpen) backend for the network: no conv / norm / pool calls of
    torch.nn.functional anywhere in the package (the oracle and the tests may use them as checkers)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diff_unet_amos_amd")
    pat = re.compile(r"F\.(conv3d|conv_transpose3d|instance_norm|max_pool3d|leaky_relu)\b|torch\.nn\.functional\.conv")
    hits = []
    for fn in sorted(os.listdir(root)):
        if fn.endswith(".py"):
            for i, line in enumerate(open(os.path.join(root, fn)), 1):
                if pat.search(line):
                    hits.append(f"{fn}:{i}: {line.strip()}")
    assert not hits, hits
    import diff_unet_amos_amd.training as tr
    for name in ("autograd_denoise", "DDPTrainer", "training_step"):
        assert not hasattr(tr, name), name
    assert not hasattr(DiffUNet, "enable_autograd_fallback")
    assert tr.uses_native_kernels is True


def test_oracle_loss_formulas():
    g = torch.Generator().manual_seed(3)
    p = torch.randn(2, 3, 4, 4, 4, generator=g)
    y = (torch.rand(2, 3, 4, 4, 4, generator=g) > 0.5).float()
    s = torch.sigmoid(p)
    mse = ((s - y) ** 2).mean()
    bce = -(y * torch.log(s) + (1 - y) * torch.log(1 - s)).mean()
    inter = (s * y).flatten(2).sum(-1)
    den = s.flatten(2).sum(-1) + y.flatten(2).sum(-1)
    dice = (1 - (2 * inter + 1e-5) / (den + 1e-5)).mean()
    assert torch.allclose(Loss("mse,bce,dice", "sum")(p, y), mse + bce + dice, rtol=1e-5)
    assert torch.allclose(Loss("mse,bce,dice", "mean")(p, y), (mse + bce + dice) / 3, rtol=1e-5)
    assert torch.allclose(Loss("mse,dice", "log")(p, y), torch.log(1 + mse + dice), rtol=1e-5)
    assert torch.allclose(Loss("dice", "sum")(p, y), dice, rtol=1e-6)
    assert torch.allclose(Loss("bce", "log")(p, y), bce, rtol=1e-6)          # a single loss is returned as it is
    with pytest.raises(NotImplementedError):
        Loss("focal")


def test_parse_losses_mirrors_reference_errors():
    from diff_unet_amos_amd.training import parse_losses
    assert parse_losses("mse,bce,dice", "sum") == (("mse", "bce", "dice"), "sum")
    assert parse_losses("dice", "log") == (("dice",), "log")
    with pytest.raises(NotImplementedError, match=r"Loss \(focal\) is not listed yet"):
        parse_losses("mse,focal", "sum")
    with pytest.raises(NotImplementedError, match="loss_combine"):
        parse_losses("mse,bce", "max")


def _allreduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diff_unet_amos_amd.training import allreduce_mean_
        g = torch.Generator().manual_seed(100 + rank)
        ts = [torch.randn(3, 5, generator=g), torch.randn(7, generator=g), torch.randn(2, 2, 2, generator=g)]
        allreduce_mean_(ts)
        q.put((rank, [t.numpy() for t in ts]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks_gloo():
    """The N > 1 gradient path of the trainer on the CPU: one flat all-reduce over gloo (world_size 2) leaves every
    rank with the mean of the ranks' tensors, shapes preserved."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 2000
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = []
    for k in range(3):
        a = [[torch.randn(3, 5, generator=g), torch.randn(7, generator=g), torch.randn(2, 2, 2, generator=g)][k]
             for g in (torch.Generator().manual_seed(100), torch.Generator().manual_seed(101))]
        want.append(((a[0] + a[1]) / 2).numpy())
    for r in range(2):
        for k in range(3):
            assert np.allclose(outs[r][k], want[k], rtol=1e-6, atol=1e-7), (r, k)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_training_step_through_the_forward_boundary(dtype):
    """train.py:258-268 verbatim against the drop-in module: q_sample and denoise through ``forward`` with grad
    enabled, the caller's own loss on the returned logits, backward, AdamW.  Logits and gradients are checked against
    the oracle; a few steps must reduce the loss; the no-grad forward then agrees with the training forward."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ref = RefDiffUNet(**KW)
    net = DiffUNet(compute_dtype=dtype, **KW)
    net.load_state_dict(ref.state_dict())
    net = net.to(dev).train()
    image, labels, _, _ = _data(2, 11)
    images_d, labels_d = image.to(dev), labels.to(dev)
    crit = Loss()
    np.random.seed(5)
    x_start = labels_d * 2 - 1
    x_t, t, noise = net(x=x_start, pred_type="q_sample")
    preds = net(x=x_t, step=t, image=images_d, pred_type="denoise")
    assert preds.requires_grad and preds.shape == labels_d.shape and preds.dtype == torch.float32
    scale = 1024.0 if dtype == torch.float16 else 1.0
    (crit(preds, labels_d) * scale).backward()
    want = ref(image=image, x=x_t.cpu(), step=t.cpu(), pred_type="denoise")
    crit(want, labels).backward()
    ftol = 2e-4 if dtype == torch.float32 else 3e-2
    assert (preds.detach().cpu() - want.detach()).abs().max().item() < ftol
    gp = dict(net.named_parameters())
    num = den = 0.0
    for k, p in ref.named_parameters():
        if k.endswith("conv.bias"):
            continue
        g = gp[k].grad.detach().cpu().double() / scale
        num += float(((g - p.grad.double()) ** 2).sum()); den += float((p.grad.double() ** 2).sum())
    rel = (num / den) ** 0.5
    print(f"[{dtype}] boundary training step: whole-gradient relative L2 error vs oracle {rel:.2e}")
    assert rel < (1e-4 if dtype == torch.float32 else 5e-2), rel
    if dtype == torch.float16:
        return
    opt = torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=1e-4)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = crit(net(x=x_t, step=t, image=images_d, pred_type="denoise"), labels_d)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    with torch.enable_grad():
        a = net(x=x_t, step=t, image=images_d, pred_type="denoise").detach()
    with torch.no_grad():
        b = net.eval()(x=x_t, step=t, image=images_d, pred_type="denoise")
    assert (a - b).abs().max() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("names,combine", [("mse,bce,dice", "sum"), ("mse,dice", "mean"), ("bce,dice", "log"), ("dice", "sum")])
def test_fused_loss_covers_the_reference_configurations(names, combine):
    from diff_unet_amos_amd.training import _SegLoss
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(17)
    logits = (torch.randn(2, 8, 8, 8, 5, generator=g, device=dev) * 2).requires_grad_(True)
    labels = (torch.rand(2, 5, 8, 8, 8, generator=g, device=dev) > 0.6).float()
    L = _SegLoss.apply(logits, labels, tuple(names.split(",")), combine)
    (L * 3.0).backward()
    p = logits.detach().double().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    want = Loss(names, combine)(p, labels.double())
    (want * 3.0).backward()
    assert abs(float(L) - float(want)) < 1e-5 * max(1.0, abs(float(want)))
    wg = p.grad.permute(0, 2, 3, 4, 1)
    assert (logits.grad.double() - wg).abs().max().item() <= 1e-5 * wg.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_native_conv_training_path_matches_oracle_autograd(dtype):
    """Forward logits and every parameter gradient of the HIP-convolution training path against the oracle network
    under torch autograd (CPU fp32), same weights, same inputs."""
    from diff_unet_amos_amd.training import native_conv_denoise
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ref = RefDiffUNet(**KW)
    net = DiffUNet(**KW)
    net.load_state_dict(ref.state_dict())
    net = net.to(dev)
    image, labels, noise, t = _data(2, 21)
    x_t = ref.diffusion.q_sample(labels * 2 - 1, t, noise)
    crit = Loss()
    want = ref(image=image, x=x_t, step=t, pred_type="denoise")
    crit(want, labels).backward()
    got = native_conv_denoise(net, image.to(dev), x_t.to(dev), t.to(dev), dtype)
    scale = 4096.0 if dtype == torch.float16 else 1.0
    (crit(got, labels.to(dev)) * scale).backward()
    ftol = 2e-4 if dtype == torch.float32 else 3e-2
    assert (got.detach().cpu() - want.detach()).abs().max().item() < ftol
    gp = dict(net.named_parameters())
    errs, coss, num, den = [], [], 0.0, 0.0
    for k, p in ref.named_parameters():
        g = gp[k].grad.detach().cpu().double() / scale
        ref_g = p.grad.double()
        if k.endswith("conv.bias"):          # bias before InstanceNorm: true gradient is zero, both sides hold rounding noise
            continue
        errs.append(((g - ref_g).abs().max().item() / (ref_g.abs().max().item() + 1e-6), k))
        num += float(((g - ref_g) ** 2).sum()); den += float((ref_g ** 2).sum())
        if ref_g.numel() >= 64:
            coss.append((float((g * ref_g).sum() / (g.norm() * ref_g.norm() + 1e-30)), k))
    errs.sort(reverse=True); coss.sort()
    rel_l2 = (num / den) ** 0.5
    print(f"[{dtype}] whole-gradient relative L2 error {rel_l2:.2e}; worst max-relative: "
          + ", ".join(f"{k} {e:.2e}" for e, k in errs[:3]) + "; lowest cosine: " + ", ".join(f"{k} {c:.4f}" for c, k in coss[:3]))
    if dtype == torch.float32:
        assert errs[0][0] < 2e-3, errs[0]
    else:
        # fp16 activations: the 2^3-voxel bottom level normalises over 8 samples, which amplifies rounding; judge the
        # gradient as a direction
        assert rel_l2 < 5e-2 and coss[0][0] > 0.9, (rel_l2, coss[0])


@pytest.mark.gpu
def test_native_conv_trainer_learns():
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffUNet(**KW).to(dev)
    tr = NativeConvTrainer(net, lr=2e-3)
    image, labels, noise, t = _data(2, 11)
    image, labels, noise, t = image.to(dev), labels.to(dev), noise.to(dev), t.to(dev)
    losses = [float(tr.step(image, labels, noise=noise, t=t)) for _ in range(10)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


# Data seed of the two-rank test.  At 32^3 with eight channels a LeakyReLU input lands within fp32 rounding of the kink (slope
# 0.1 | 1) in about every other seed, on the native path or on the CPU oracle's own fp32 path, and a flipped activation at a
# 4^3 or 2^3 level moves every gradient that flows through it by 1e-3 .. 4e-2 of its scale: tools/gradient_seed_scan.py
# compares both fp32 paths with an fp64 evaluation of the oracle, seed by seed (seeds 7 .. 16, profiles/r5_gradient_seed_scan.txt):
# seed 7, which this test used before, has such a crossing on the native side (worst tensor 3.6e-2, the oracle's fp32 path
# 1e-5), seeds 9 and 11 have one on the ORACLE's fp32 side (2e-2, native 4e-6 / 4e-3); with seeds 8, 10 and 15 both paths agree with
# fp64 to 1e-5 on every tensor.  The test needs an arbiter-free comparison of two fp32 paths, so it uses a seed without a crossing.
DDP_DATA_SEED = 8


def _native_ddp_worker(rank, world, port, q, overlap=True, graph=False, seed=DDP_DATA_SEED):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share cuda:0; gloo moves CUDA tensors via the host
    try:
        from diff_unet_amos_amd.training import NativeConvTrainer
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        net = DiffUNet(**KW).to(dev)
        tr = NativeConvTrainer(net, lr=1e-3, dtype=torch.float32, overlap=overlap, graph=graph)
        image, labels, noise, t = _data(2, seed)
        sl = slice(rank, rank + 1)
        loss = tr.step(image[sl].to(dev), labels[sl].to(dev), noise=noise[sl].to(dev), t=t[sl].to(dev))
        grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}          # averaged over the ranks
        moments = {k: tr.optimizer.state[p]["exp_avg"].detach().cpu().numpy() for k, p in net.named_parameters()}
        q.put((rank, float(loss), {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}, grads, moments))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _zero_gradient_bias(k):
    """The bias of a convolution that InstanceNorm follows: its true gradient is zero (the norm subtracts the mean), what either
    path computes for it is rounding noise.  (`final_conv.bias` and the transposed convolutions' biases are NOT of that kind.)"""
    return k.endswith(".conv.bias")


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,graph", [(True, False), (False, False), (False, True)], ids=["ddp-buckets", "flat", "flat-graph"])
def test_native_trainer_two_ranks_equal_union_batch(overlap, graph):
    """Two processes (one GPU, gloo) each train on their own sample; the gradient averaging (DDP reducer buckets overlapped with
    backward, one flat all-reduce after it, or the flat all-reduce between the two HIP graphs of graph mode) must hand the
    optimizer the gradient of single-process training on the two-sample batch (oracle autograd, CPU fp32; train.py:258-268).
    Every check is LINEAR in the gradient -- the parameters after a first Adam step are lr * g / (|g| + eps), sign-like wherever
    |g| is of the order of eps = 1e-8, which is where the deep levels' gradients of this small network live (the 7 463 elements
    the round-4 form of this test had to tolerate: tools/two_rank_gradient_report.py, profiles/r5_two_rank_gradient_report.txt)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 177) % 2000
    procs = [ctx.Process(target=_native_ddp_worker, args=(r, 2, port + int(overlap) + 2 * int(graph), q, overlap, graph)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {r: (l, sd, g, m) for r, l, sd, g, m in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # (0) both ranks hold bit-identical parameters, gradients and moments
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
    for k in outs[0][2]:
        assert np.array_equal(outs[0][2][k], outs[1][2][k]) and np.array_equal(outs[0][3][k], outs[1][3][k]), k
    _, want_g = _union_batch_reference(DDP_DATA_SEED)
    got_g, got_m = outs[0][2], outs[0][3]
    worst = ("", 0.0)
    for k, v in want_g.items():
        ref = v.double().numpy()
        if _zero_gradient_bias(k):
            # (1a) rounding noise on both sides: small against the same convolution's weight gradient
            wk = k[:-len("bias")] + "weight"
            assert float(np.abs(got_g[k]).max()) <= 1e-3 * float(np.abs(want_g[wk].numpy()).max()), k
            continue
        # (1b) per tensor: the averaged gradient IS the union batch's gradient (a missing or doubled average is an error of order 1,
        # a wrong kernel in one layer an error of order 1 in that tensor; measured <= 1e-5 for every tensor in all three modes)
        scale = max(float(np.sqrt((ref ** 2).sum())), 1e-30)
        rel = float(np.sqrt(((got_g[k].astype(np.float64) - ref) ** 2).sum())) / scale
        worst = max(worst, (k, rel), key=lambda t: t[1])
        assert rel < 1e-4, (k, rel)
        # (2) the optimizer consumed THAT gradient: Adam's first moment after one step is (1 - beta1) g
        relm = float(np.sqrt(((got_m[k].astype(np.float64) - 0.1 * ref) ** 2).sum())) / (0.1 * scale)
        assert relm < 1e-4, (k, relm)
    print(f"worst per-tensor relative L2 of the averaged gradient: {worst[1]:.2e} ({worst[0]})")


@pytest.mark.gpu
def test_graph_replayed_training_step_equals_eager():
    """The whole training step captured as one HIP graph (the library's AdamW, device-side overflow check and loss
    scale) must walk the same parameters as the eager trainer on the same batches, timesteps and noise."""
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    image, labels, noise, t = _data(2, 31)
    image, labels, noise = image.to(dev), labels.to(dev), noise.to(dev)
    ts = [torch.tensor([100 + 37 * k, 900 - 53 * k], device=dev) for k in range(4)]
    torch.manual_seed(0)
    init = {k: v.detach().cpu().clone() for k, v in DiffUNet(**KW).state_dict().items()}
    runs = []
    # eager with torch.optim.AdamW (foreach), eager with the library's AdamW (NativeAdamW), graph (always NativeAdamW)
    for mode, fused in ((False, False), (False, True), (True, True)):
        torch.manual_seed(0)
        net = DiffUNet(**KW).to(dev)
        tr = NativeConvTrainer(net, lr=1e-3, dtype=torch.float32, graph=mode, fused_optimizer=fused)
        # a captured step must be a single-stream graph: a multi-branch one can fault inside hipGraphLaunch
        assert (tr.wgrad_stream is None) == mode
        losses = [float(tr.step(image, labels, noise=noise, t=tk)) for tk in ts]
        runs.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))

    def rel_update_diff(a, b):
        """|| (a - init) - (b - init) || / || a - init || over all parameters: how different the two 4-step walks are."""
        num = sum(float(((a[k] - b[k]).double() ** 2).sum()) for k in a)
        den = sum(float(((a[k] - init[k]).double() ** 2).sum()) for k in a)
        return (num / den) ** 0.5

    opt_gap = rel_update_diff(runs[0][1], runs[1][1])      # torch.optim.AdamW vs NativeAdamW, both eager
    graph_gap = rel_update_diff(runs[1][1], runs[2][1])    # same optimizer kernel: eager vs graph replay
    print(f"losses eager {runs[1][0]} graph {runs[2][0]}; relative 4-step update difference: torch-vs-library AdamW (eager) "
          f"{opt_gap:.3e}, eager-vs-graph with the same AdamW {graph_gap:.3e}")
    assert np.allclose(runs[1][0], runs[2][0], rtol=1e-6), (runs[1][0], runs[2][0])
    # same kernels, same order, same optimizer: the replayed step must reproduce the eager one (this also proves the
    # graph's warm-up iterations left weights, Adam state and the loss scale untouched)
    assert graph_gap < 1e-6, graph_gap


@pytest.mark.gpu
@pytest.mark.parametrize("when", ["before-capture", "after-capture"])
def test_graph_trainer_resumes_from_a_checkpoint(when):
    """NativeConvTrainer(graph=True) + load_state_dict: three captured steps, a checkpoint (model + optimizer), then a NEW
    graph trainer that loads it -- before its first step (the capture's warm-up updates must not wipe the loaded moments
    and update counter) or after it has already captured and stepped (the loaded moments must land in the tensors the
    graph updates) -- takes step four exactly like the uninterrupted run."""
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    image, labels, noise, t = _data(2, 43)
    image, labels, noise = image.to(dev), labels.to(dev), noise.to(dev)
    ts = [torch.tensor([150 + 41 * k, 820 - 67 * k], device=dev) for k in range(4)]
    torch.manual_seed(0)
    net = DiffUNet(**KW).to(dev)
    tr = NativeConvTrainer(net, lr=1e-3, dtype=torch.float32, graph=True)
    for tk in ts[:3]:
        tr.step(image, labels, noise=noise, t=tk)
    ckpt = {"model": {k: v.detach().clone() for k, v in net.state_dict().items()},
            "optimizer": {"state": {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                                    for i, st in tr.optimizer.state_dict()["state"].items()},
                          "param_groups": tr.optimizer.state_dict()["param_groups"]}}
    assert float(ckpt["optimizer"]["state"][0]["step"]) == 3.0
    want_loss = float(tr.step(image, labels, noise=noise, t=ts[3]))
    want = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}

    torch.manual_seed(1)                                     # different initial weights: everything must come from the checkpoint
    net2 = DiffUNet(**KW).to(dev)
    tr2 = NativeConvTrainer(net2, lr=1e-3, dtype=torch.float32, graph=True)
    if when == "after-capture":
        tr2.step(image, labels, noise=noise, t=ts[0])        # captures; moments now live at the addresses the graph updates
        held = {id(p): tr2.optimizer.state[p]["exp_avg"].data_ptr() for p in tr2.params}
    net2.load_state_dict(ckpt["model"])
    tr2.optimizer.load_state_dict(ckpt["optimizer"])
    if when == "after-capture":
        assert all(tr2.optimizer.state[p]["exp_avg"].data_ptr() == held[id(p)] for p in tr2.params)
    got_loss = float(tr2.step(image, labels, noise=noise, t=ts[3]))
    assert float(tr2.optimizer.state_dict()["state"][0]["step"]) == 4.0
    assert abs(got_loss - want_loss) <= 1e-6 * abs(want_loss), (got_loss, want_loss)
    # lost moments or a restarted update counter would move every element by O(lr) = 1e-3 (the first Adam step is sign-like)
    for k, v in net2.state_dict().items():
        assert float((v.detach().cpu() - want[k]).abs().max()) < 1e-6, k
    m_a = tr.optimizer.state_dict()["state"]
    m_b = tr2.optimizer.state_dict()["state"]
    for i in m_a:
        assert torch.allclose(m_a[i]["exp_avg"], m_b[i]["exp_avg"], rtol=1e-4, atol=1e-9), i
        assert torch.allclose(m_a[i]["exp_avg_sq"], m_b[i]["exp_avg_sq"], rtol=1e-4, atol=1e-12), i


@pytest.mark.gpu
def test_training_forward_fold_equals_the_unfolded_forward():
    """ops.TRAIN_FOLD_UPCONV: UpCat's first convolution runs as the folded launch in the FORWARD of the training step (composed
    weights, dua_upconv_k3_fwd) where a level has enough tiles; backward is the unfolded one either way.  At 32^3 with the tile
    threshold lowered levels 0 and 1 fold: logits and every parameter gradient against the same network unfolded."""
    from diff_unet_amos_amd import ops
    from diff_unet_amos_amd.training import native_logits_cl, _SegLoss
    kw = dict(in_channels=1, out_channels=16)
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    nets = [DiffUNet(**kw).to(dev)]
    nets.append(DiffUNet(**kw).to(dev))
    nets[1].load_state_dict(nets[0].state_dict())
    g = torch.Generator().manual_seed(43)
    image = torch.rand(2, 1, 32, 32, 32, generator=g).to(dev)
    labels = (torch.rand(2, 16, 32, 32, 32, generator=g) > 0.8).float().to(dev)
    x_t = torch.randn(2, 16, 32, 32, 32, generator=g).to(dev)
    t = torch.tensor([417, 83], device=dev)
    seen = []
    real = ops.upconv_k3
    keep = (ops.TRAIN_FOLD_UPCONV, ops.TRAIN_FOLD_MIN_TILES)
    try:
        ops.TRAIN_FOLD_MIN_TILES = 1
        ops.upconv_k3 = lambda *a, **k: (seen.append(tuple(a[0].shape)), real(*a, **k))[1]
        outs = []
        for net, fold in zip(nets, (True, False)):
            ops.TRAIN_FOLD_UPCONV = fold
            logits = native_logits_cl(net, image, x_t, t, torch.float16)
            (_SegLoss.apply(logits, labels) * 4096.0).backward()
            outs.append(logits.detach().float())
    finally:
        ops.upconv_k3 = real
        ops.TRAIN_FOLD_UPCONV, ops.TRAIN_FOLD_MIN_TILES = keep
    assert [s[1] for s in seen] == [16, 32], seen            # levels 1 and 0, in the order the decoder runs; never with the fold off
    assert (outs[0] - outs[1]).abs().max().item() < 3e-2
    num = den = 0.0
    for (k, a), (_, b) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        if k.endswith("conv.bias") or a.grad is None:
            continue
        num += float(((a.grad.double() - b.grad.double()) ** 2).sum()); den += float((b.grad.double() ** 2).sum())
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5


@pytest.mark.gpu
def test_full_size_gradients_fp16_path_vs_oracle():
    """Config-4 geometry (96^3 patch, 16 classes, full feature widths; batch 1 to bound the CPU oracle's time): parameter
    gradients of the fp16 HIP training path against the oracle network under torch autograd in fp32 on the host."""
    from diff_unet_amos_amd.training import native_logits_cl, _SegLoss
    kw = dict(in_channels=1, out_channels=16)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ref = RefDiffUNet(**kw)
    net = DiffUNet(**kw)
    net.load_state_dict(ref.state_dict())
    net = net.to(dev)
    g = torch.Generator().manual_seed(41)
    image = torch.rand(1, 1, 96, 96, 96, generator=g)
    labels = (torch.rand(1, 16, 96, 96, 96, generator=g) > 0.8).float()
    noise = torch.randn(1, 16, 96, 96, 96, generator=g)
    t = torch.tensor([417])
    x_t = ref.diffusion.q_sample(labels * 2 - 1, t, noise)
    crit = Loss()
    want = ref(image=image, x=x_t, step=t, pred_type="denoise")
    lw = crit(want, labels)
    lw.backward()
    scale = 4096.0
    logits = native_logits_cl(net, image.to(dev), x_t.to(dev), t.to(dev), torch.float16)
    lg = _SegLoss.apply(logits, labels.to(dev))
    (lg * scale).backward()
    assert abs(float(lg) - float(lw)) < 2e-3 * float(lw)
    gp = dict(net.named_parameters())
    num = den = 0.0
    coss = []
    for k, p in ref.named_parameters():
        if k.endswith("conv.bias"):
            continue
        a = gp[k].grad.detach().cpu().double() / scale
        b = p.grad.double()
        num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
        if b.numel() >= 64:
            coss.append((float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), k))
    coss.sort()
    rel = (num / den) ** 0.5
    print(f"[96^3 x 16, fp16 path] loss {float(lg):.6f} vs {float(lw):.6f}; whole-gradient relative L2 error {rel:.2e}; "
          f"lowest cosine: " + ", ".join(f"{k} {c:.4f}" for c, k in coss[:3]))
    assert rel < 3e-2 and coss[0][0] > 0.98, (rel, coss[0])


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False], ids=["graph", "eager"])
def test_trainer_fp16_dynamic_loss_scale_stays_on_the_device(graph):
    """fp16, captured or eager: an absurd initial loss scale must overflow, be halved on the device step by step (AdamW skipping
    those updates), and training must then proceed -- without any host decision inside the step (the eager step used to read the
    overflow flag back every step)."""
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffUNet(**KW).to(dev)
    w0 = net.model.conv_0.conv_0.conv.weight.detach().clone()
    tr = NativeConvTrainer(net, lr=2e-3, dtype=torch.float16, graph=graph, init_scale=2.0 ** 40)
    image, labels, noise, t = _data(2, 11)
    image, labels, noise, t = image.to(dev), labels.to(dev), noise.to(dev), t.to(dev)
    first = float(tr.step(image, labels, noise=noise, t=t))
    assert tr.last_step_overflowed() and tr.loss_scale() == 2.0 ** 39                      # overflow seen, scale halved
    assert torch.equal(net.model.conv_0.conv_0.conv.weight.detach(), w0)                   # and the update was skipped
    losses = [first] + [float(tr.step(image, labels, noise=noise, t=t)) for _ in range(40)]
    assert tr.loss_scale() < 2.0 ** 30 and all(np.isfinite(losses)) and not tr.last_step_overflowed()
    assert not torch.equal(net.model.conv_0.conv_0.conv.weight.detach(), w0)
    assert losses[-1] < losses[0], (losses[0], losses[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False], ids=["graph", "eager"])
def test_scheduler_learning_rate_reaches_the_captured_step(graph):
    """The reference steps LinearWarmupCosineAnnealingLR once per epoch (train.py:249): a learning rate changed on the optimizer's
    param_groups must be the one the NEXT step uses -- also when the step is a replayed HIP graph, where a host float would have been
    baked in at capture (NativeAdamW reads it from a device scalar there).  lr = 0 must leave every parameter untouched."""
    from diff_unet_amos_amd.schedule import LinearWarmupCosineAnnealingLR
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffUNet(**KW).to(dev)
    tr = NativeConvTrainer(net, lr=1e-3, dtype=torch.float32, graph=graph)
    sch = LinearWarmupCosineAnnealingLR(tr.optimizer, warmup_epochs=2, max_epochs=4)       # epoch 0: warm-up start lr = 0
    assert tr.optimizer.param_groups[0]["lr"] == 0.0
    image, labels, noise, t = _data(2, 5)
    image, labels, noise, t = image.to(dev), labels.to(dev), noise.to(dev), t.to(dev)
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    tr.step(image, labels, noise=noise, t=t)
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items())              # lr = 0: nothing moves
    sch.step()
    assert tr.optimizer.param_groups[0]["lr"] > 0
    tr.step(image, labels, noise=noise, t=t)
    moved = sum(int(not torch.equal(v, before[k])) for k, v in net.state_dict().items())
    assert moved > 0.9 * len(before), moved
    sd = tr.optimizer.state_dict()                                                          # two steps counted, AdamW's state layout
    assert float(sd["state"][0]["step"]) == 2.0 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.gpu
def test_native_trainer_other_loss_configuration():
    """A loss configuration other than the configs' default ("mse,dice" combined by "mean") runs on the same fused
    kernels (term weights + the combine's derivative through the gradient scale): one step must reproduce the oracle's
    loss, and training must learn."""
    from diff_unet_amos_amd.training import NativeConvTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = DiffUNet(**KW).to(dev)
    tr = NativeConvTrainer(net, lr=2e-3, dtype=torch.float32, losses="mse,dice", loss_combine="mean")
    image, labels, noise, t = _data(2, 13)
    image, labels, noise, t = image.to(dev), labels.to(dev), noise.to(dev), t.to(dev)
    losses = [float(tr.step(image, labels, noise=noise, t=t)) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    ref = RefDiffUNet(**KW)
    torch.manual_seed(0)
    net2 = DiffUNet(**KW)
    ref.load_state_dict(net2.state_dict())
    want = ref_training_step(ref, image.cpu(), labels.cpu(), Loss("mse,dice", "mean"), noise.cpu(), t.cpu())
    torch.manual_seed(0)
    net3 = DiffUNet(**KW).to(dev)
    tr3 = NativeConvTrainer(net3, lr=0.0, dtype=torch.float32, losses="mse,dice", loss_combine="mean")
    got = float(tr3.step(image, labels, noise=noise, t=t))
    assert abs(got - float(want)) < 1e-5 * max(1.0, abs(float(want))), (got, float(want))
    with pytest.raises(NotImplementedError, match="not listed yet"):
        NativeConvTrainer(net3, losses="mse,hausdorff_er")
