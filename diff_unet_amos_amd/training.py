"""Training-side harness for BASELINE config 4 (row 8(f)-2 of the scope table) -- NOT yet native.

The forward kernels of this package are inference-only this round: no backward (dgrad/wgrad, InstanceNorm /
LeakyReLU / MaxPool / deconv backward) is written yet.  So that ``Trainer.training_step`` (train.py:258-268)
and a one-process-per-GPU DDP loop over RCCL can already be exercised end to end, this module provides an
**explicitly labelled PyTorch-autograd fallback** for the *training* branch only: the same parameters, run
through torch.nn.functional ops on the device (MIOpen/rocBLAS kernels, torch autograd).  Nothing on the
sampling/inference path ever routes through it, it is opt-in (``DiffUNet.enable_autograd_fallback()``), and
``uses_native_kernels`` below says False so a harness can report what it measured.  q_sample in the training
step IS the HIP kernel.

Also here: the loss the reference's configs use (losses/loss.py:25-86 with ``mse,bce,dice`` / ``sum``), restated
with MONAI DiceLoss(sigmoid=True) defaults, and a DDP step (gradient all-reduce through
torch.distributed: backend "nccl" is RCCL on ROCm).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

uses_native_kernels = False


def _two_conv(block, x, temb):
    """TwoConv.forward (denoiser.py:63-67 / pretrained/basic_unet.py:28-65) on torch ops."""
    for i, cb in enumerate((block.conv_0, block.conv_1)):
        x = F.conv3d(x, cb.conv.weight, cb.conv.bias, padding=1)
        x = F.instance_norm(x, weight=cb.adn.N.weight, bias=cb.adn.N.bias, eps=1e-5)
        x = F.leaky_relu(x, 0.1)
        if i == 0 and temb is not None:
            s = temb * torch.sigmoid(temb)
            x = x + F.linear(s, block.temb_proj.weight, block.temb_proj.bias)[:, :, None, None, None]
    return x


def _time_embedding(temb_mod, t):
    half = temb_mod.embedding_dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * -(math.log(10000) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    e = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    h = F.linear(e, temb_mod.dense[0].weight, temb_mod.dense[0].bias)
    h = h * torch.sigmoid(h)
    return F.linear(h, temb_mod.dense[1].weight, temb_mod.dense[1].bias)


def autograd_denoise(net, image, x, step):
    """Diffusion.denoise (diffusion.py:71-84) with torch autograd: encoder + denoiser on torch ops."""
    enc, den = net.embed_model, net.model
    emb = [_two_conv(enc.conv_0, image, None)]
    for d in enc.down:
        emb.append(_two_conv(d.convs, F.max_pool3d(emb[-1], 2), None))
    temb = _time_embedding(den.temb, step)
    h = torch.cat([image, x], dim=1)
    x0 = _two_conv(den.conv_0, h, temb) + emb[0]
    x1 = _two_conv(den.down_1.convs, F.max_pool3d(x0, 2), temb) + emb[1]
    x2 = _two_conv(den.down_2.convs, F.max_pool3d(x1, 2), temb) + emb[2]
    x3 = _two_conv(den.down_3.convs, F.max_pool3d(x2, 2), temb) + emb[3]
    x4 = _two_conv(den.down_4.convs, F.max_pool3d(x3, 2), temb) + emb[4]

    def up(block, lo, skip):
        u = F.conv_transpose3d(lo, block.upsample.deconv.weight, block.upsample.deconv.bias, stride=2)
        return _two_conv(block.convs, torch.cat([skip, u], dim=1), temb)

    u4 = up(den.upcat_4, x4, x3)
    u3 = up(den.upcat_3, u4, x2)
    u2 = up(den.upcat_2, u3, x1)
    u1 = up(den.upcat_1, u2, x0)
    return F.conv3d(u1, den.final_conv.weight, den.final_conv.bias)


class Loss:
    """losses/loss.py:25-86 for the names the diffusion configs use: ``mse`` (on sigmoid(pred), :68-69), ``bce``
    (BCEWithLogits), ``dice`` (MONAI DiceLoss(sigmoid=True): include_background, smooth_nr = smooth_dr = 1e-5,
    sums over the spatial axes, mean over batch x class); combined by ``sum`` / ``mean`` / ``log``."""

    def __init__(self, losses="mse,bce,dice", loss_combine="sum"):
        self.names = losses.split(",")
        for n in self.names:
            if n not in ("mse", "bce", "dice"):
                raise NotImplementedError(f"Loss ({n}) is not listed yet")
        self.loss_combine = loss_combine

    @staticmethod
    def _dice(pred, target):
        p = torch.sigmoid(pred)
        dims = tuple(range(2, pred.dim()))
        inter = (p * target).sum(dims)
        denom = p.sum(dims) + target.sum(dims)
        return (1.0 - (2.0 * inter + 1e-5) / (denom + 1e-5)).mean()

    def __call__(self, preds, labels):
        out = []
        for n in self.names:
            if n == "mse":
                out.append(F.mse_loss(torch.sigmoid(preds), labels))
            elif n == "bce":
                out.append(F.binary_cross_entropy_with_logits(preds, labels))
            else:
                out.append(self._dice(preds, labels))
        if len(out) == 1:
            return out[0]
        st = torch.stack(out)
        if self.loss_combine == "sum":
            return st.sum()
        if self.loss_combine == "mean":
            return st.mean()
        if self.loss_combine == "log":
            return torch.log(1 + st.sum())
        raise NotImplementedError("Unsupported value for loss_combine. Please choose from 'sum', 'mean', or 'log'.")


def training_step(net, images, labels, criterion, noise=None, t=None):
    """Trainer.training_step (train.py:258-268): x_start = 2*labels - 1 -> q_sample (HIP kernel on a GPU) -> denoise
    (autograd fallback) -> loss.  ``noise`` / ``t`` can be injected for tests."""
    x_start = labels * 2 - 1
    if noise is None and t is None and images.is_cuda:
        x_t, t, _ = net(x=x_start, pred_type="q_sample")
    else:
        noise = torch.randn_like(x_start) if noise is None else noise
        if t is None:
            t, _ = net.sampler.sample(x_start.shape[0], x_start.device)
        d = net.diffusion
        q = d.q_coef(t).to(x_start.device)
        x_t = q[:, 0].view(-1, 1, 1, 1, 1) * x_start + q[:, 1].view(-1, 1, 1, 1, 1) * noise
    preds = autograd_denoise(net, images, x_t, t)
    return criterion(preds, labels)


class DDPTrainer:
    """One process per GPU; gradients all-reduced by torch DistributedDataParallel (RCCL on a GPU node, gloo in the
    CPU tests).  InstanceNorm needs no cross-rank statistics (per-sample)."""

    def __init__(self, net, lr=2e-4, weight_decay=1e-4, losses="mse,bce,dice", loss_combine="sum", device_ids=None):
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel

        class _Step(torch.nn.Module):
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, images, x_t, t):
                return autograd_denoise(self.inner, images, x_t, t)

        self.net = net
        self.criterion = Loss(losses, loss_combine)
        self.wrapped = DistributedDataParallel(_Step(net), device_ids=device_ids) if dist.is_initialized() else _Step(net)
        self.optimizer = torch.optim.AdamW(net.parameters(), lr=lr, weight_decay=weight_decay)     # train.py:121-126

    def step(self, images, labels, noise=None, t=None):
        x_start = labels * 2 - 1
        if t is None:
            t, _ = self.net.sampler.sample(x_start.shape[0], x_start.device)
        noise = torch.randn_like(x_start) if noise is None else noise
        if x_start.is_cuda:
            x_t = self.net.diffusion.q_sample(x_start, t, noise)
        else:
            q = self.net.diffusion.q_coef(t)
            x_t = q[:, 0].view(-1, 1, 1, 1, 1) * x_start + q[:, 1].view(-1, 1, 1, 1, 1) * noise
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.criterion(self.wrapped(images, x_t, t), labels)
        loss.backward()
        self.optimizer.step()
        return loss.detach()
