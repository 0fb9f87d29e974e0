"""The training step around the network (csrc/train_glue.hip) against the torch code of the reference it replaces:
x_start = label * 2 - 1 + q_sample (train.py:258-262), TimeStepEmbedder + temb_proj under autograd (models/diffusion/utils.py:5-54,
models/basic_unet/denoiser.py:51-52,65), the loss tail (losses/loss.py:64-86), torch.optim.AdamW with torch.cuda.amp's overflow
check and loss-scale rule (train.py:121-126,264-268)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from diff_unet_amos_amd import ops
    return ops


def test_stats_channel_sums_equals_decoded_sum():
    """One launch instead of stats_decode(...)[:, :c, 0].sum(0).float(): same words, same double sums, same rounding."""
    ops = _ops()
    x = (torch.randn(3, 5, 6, 7, 24, device=DEV) * 3).half()
    st = ops.stats_buffer(3, 24, x.device)
    ops.instnorm_stats(x, 24, st)
    want = ops.stats_decode(st)[:, :24, 0].sum(0).float()
    got = ops.stats_channel_sums(st, 24)
    assert torch.equal(got, want)
    assert torch.allclose(got.cpu(), x.float().sum((0, 1, 2, 3)).cpu(), rtol=1e-5, atol=1e-3)


def test_q_sample_affine_is_bit_equal_to_the_two_passes():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(3)
    labels = (torch.rand(3, 4, 6, 10, 9, device=DEV, generator=g) > 0.7).float()         # per-sample size not a multiple of 4 * 256
    noise = torch.randn(labels.shape, device=DEV, generator=g)
    T = 1000
    betas = torch.linspace(1e-4, 2e-2, T, dtype=torch.float64)
    acp = torch.cumprod(1 - betas, 0)
    sched = torch.stack([acp.sqrt(), (1 - acp).sqrt()], 1).float().to(DEV).contiguous()
    t = torch.tensor([0, 999, 417], device=DEV)
    want = ops.q_sample((labels * 2 - 1).contiguous(), noise, sched[t].contiguous())
    got = ops.q_sample_affine(labels, 2.0, -1.0, noise, sched, t)
    assert torch.equal(got, want)
    ref = sched[t, 0].view(-1, 1, 1, 1, 1) * (labels * 2 - 1) + sched[t, 1].view(-1, 1, 1, 1, 1) * noise
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)
    odd = labels[:, :, :, :, :7].contiguous()                                            # unaligned rows: the scalar path
    got = ops.q_sample_affine(odd, 2.0, -1.0, noise[..., :7].contiguous(), sched, t)
    assert torch.equal(got, ops.q_sample((odd * 2 - 1).contiguous(), noise[..., :7].contiguous(), sched[t].contiguous()))


@pytest.mark.parametrize("names,combine", [(("mse", "bce", "dice"), "sum"), (("mse", "dice"), "mean"), (("bce", "dice"), "log"),
                                           (("dice",), "log")])
def test_seg_loss_finish_matches_the_formulas(names, combine):
    """L and dL/d(total) from the reduce kernel's sums, against losses/loss.py:64-86 evaluated in torch (double)."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    N, C_, D = 2, 8, 12
    logits = torch.randn(N, D, D, D, C_, device=DEV, generator=g)
    labels = (torch.rand(N, C_, D, D, D, device=DEV, generator=g) > 0.6).float()
    L, sums, dcomb = ops.seg_loss_reduce(logits, labels, names, combine)
    p = logits.permute(0, 4, 1, 2, 3).double()
    y = labels.double()
    s = torch.sigmoid(p)
    terms = {"mse": ((s - y) ** 2).mean(), "bce": F.binary_cross_entropy_with_logits(p, y),
             "dice": (1 - (2 * (s * y).sum((2, 3, 4)) + 1e-5) / (s.sum((2, 3, 4)) + y.sum((2, 3, 4)) + 1e-5)).mean()}
    total = sum(terms[n] for n in names)
    if len(names) == 1 or combine == "sum":
        want, dwant = total, 1.0
    elif combine == "mean":
        want, dwant = total / len(names), 1.0 / len(names)
    else:
        want, dwant = torch.log(1 + total), float(1 / (1 + total))
    assert L.dim() == 0 and dcomb.dim() == 0
    assert abs(float(L) - float(want)) < 2e-6 * max(1.0, abs(float(want)))
    assert abs(float(dcomb) - dwant) < 1e-6


def _temb_reference(t, half, params):
    """models/diffusion/utils.py:5-54 + denoiser.py:51-52,65 in torch, block adds as a list."""
    w0, b0, w1, b1 = params[:4]
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(t.device)
    arg = t.float()[:, None] * freq[None, :]
    e = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    h = F.linear(e, w0, b0)
    h = h * torch.sigmoid(h)
    temb = F.linear(h, w1, b1)
    s = temb * torch.sigmoid(temb)
    return [F.linear(s, params[4 + 2 * i], params[5 + 2 * i]) for i in range((len(params) - 4) // 2)]


def test_timestep_embedding_forward_and_backward_match_torch_autograd():
    from diff_unet_amos_amd.training import _TembAdds, _TembState
    torch.manual_seed(0)
    half, hid = 64, 512
    couts = [64, 64, 128, 256, 512, 256, 128, 64, 72]
    N = 3
    params = [torch.randn(hid, 2 * half) * 0.09, torch.randn(hid) * 0.1, torch.randn(hid, hid) * 0.045, torch.randn(hid) * 0.1]
    for c in couts:
        params += [torch.randn(c, hid) * 0.045, torch.randn(c) * 0.1]
    params = [p.to(DEV).requires_grad_() for p in params]
    t = torch.tensor([0, 999, 123], device=DEV)
    state = _TembState(N, couts)
    add = _TembAdds.apply(t, half, state, *params)
    want = _temb_reference(t, half, params)
    for i, w in enumerate(want):
        assert torch.allclose(state.rows(add, i), w, rtol=2e-5, atol=2e-5), i
    # an arbitrary cotangent per block; the native side gets it in the block-major layout
    gens = [torch.randn(N, c, device=DEV) for c in couts]
    gw = torch.autograd.grad(want, params, gens)
    flat = torch.cat([g.reshape(-1) for g in gens])
    gg = torch.autograd.grad(add, params, flat)
    for i, (a, b) in enumerate(zip(gg, gw)):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) < 3e-5 * max(scale, 1.0), (i, float((a - b).abs().max()), scale)
    # deterministic: a second evaluation gives the same bits
    gg2 = torch.autograd.grad(_TembAdds.apply(t, half, _TembState(N, couts), *params), params, flat)
    assert all(torch.equal(a, b) for a, b in zip(gg, gg2))


def test_native_adamw_walks_like_torch_adamw():
    """Five steps on 70 tensors of awkward sizes (two by-value lists; unaligned views; a tensor larger than one workgroup's chunk)
    against torch.optim.AdamW, with a loss scale applied in the kernel and one skipped (overflow) step."""
    from diff_unet_amos_amd.training import NativeAdamW
    ops = _ops()
    torch.manual_seed(1)
    sizes = [1, 3, 5, 64, 130, 4096, 4097, 70001] + [17 + 13 * i for i in range(62)]
    base = [torch.randn(n) for n in sizes]
    a = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    b_ = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    opt_a = NativeAdamW(a, lr=3e-3, weight_decay=1e-2)
    opt_b = torch.optim.AdamW(b_, lr=3e-3, weight_decay=1e-2)
    scale = torch.full((), 1024.0, device=DEV)
    found = torch.zeros((), device=DEV)
    growth = torch.zeros((), dtype=torch.int32, device=DEV)
    flatbuf = torch.zeros(sum(sizes) + 8 * len(sizes), device=DEV)
    for k in range(5):
        gs = [torch.randn(n, device=DEV) for n in sizes]
        overflow = k == 2
        mult = float(scale)                        # the loss scale the backward pass of this step would have carried
        off = 1                                    # gradients as 4-byte-aligned views (the kernel's scalar path) on even steps
        for p, q, g in zip(a, b_, gs):
            gv = g * mult
            if overflow and p.numel() == 130:
                gv[7] = float("inf")
            if k % 2 == 0:
                view = flatbuf[off:off + g.numel()]
                view.copy_(gv)
                p.grad = view
                off += g.numel() + 5
            else:
                p.grad = gv.clone()
            q.grad = g.clone()
        ops.grads_nonfinite([p.grad for p in a], found)
        assert bool(found.item()) == overflow
        opt_a.step(grad_scale=scale, found_inf=found, advance=False)
        ops.adamw_advance(opt_a._count, found, scale, growth, 2.0, 0.5, 3)
        if not overflow:
            opt_b.step()
        assert float(found) == 0.0
    assert int(opt_a._count) == 4
    # loss-scale rule of torch._amp_update_scale_: two good steps, a halving, two good steps (interval 3 not reached again)
    assert float(scale) == 512.0 and int(growth) == 2
    for p, q in zip(a, b_):
        assert torch.allclose(p.detach(), q.detach(), rtol=2e-6, atol=2e-7), float((p - q).abs().max())
    # checkpoints move between the two optimizers (engine.py:118-135 stores optimizer.state_dict())
    sd = opt_a.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    opt_c = torch.optim.AdamW([torch.nn.Parameter(x.detach().clone()) for x in a], lr=3e-3, weight_decay=1e-2)
    opt_c.load_state_dict(sd)
    opt_d = NativeAdamW([torch.nn.Parameter(x.detach().clone()) for x in a], lr=3e-3, weight_decay=1e-2)
    opt_d.load_state_dict(opt_b.state_dict())
    assert int(opt_d._count) == 4
    gs = [torch.randn(n, device=DEV) for n in sizes]
    for opt in (opt_c, opt_d):
        for p, g in zip(opt.param_groups[0]["params"], gs):
            p.grad = g.clone()
        opt.step()
    for p, q in zip(opt_c.param_groups[0]["params"], opt_d.param_groups[0]["params"]):
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-7)


def test_adamw_growth_interval_doubles_the_scale():
    ops = _ops()
    step = torch.zeros((), dtype=torch.int32, device=DEV)
    scale = torch.full((), 8.0, device=DEV)
    growth = torch.zeros((), dtype=torch.int32, device=DEV)
    found = torch.zeros((), device=DEV)
    ref_scale, ref_growth = torch.full((1,), 8.0, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    for k in range(7):
        bad = k == 4
        found.fill_(1.0 if bad else 0.0)
        torch._amp_update_scale_(ref_scale, ref_growth, found.reshape(1).clone(), 2.0, 0.5, 2)
        ops.adamw_advance(step, found, scale, growth, 2.0, 0.5, 2)
        assert float(scale) == float(ref_scale) and int(growth) == int(ref_growth), k
    assert int(step) == 6
