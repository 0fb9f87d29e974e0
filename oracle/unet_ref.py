"""Oracle (test infrastructure, never shipped): plain ``torch.nn`` CPU restatement
of the two networks DiffUNet wires together, with the reference's state-dict
key names.

Follows, in the reference tree:
  models/basic_unet/denoiser.py:23-67 (TwoConv + temb_proj), :70-108 (Down),
      :110-194 (UpCat), :196-312 (BasicUNetRDenoiser)
  models/basic_unet/pretrained/basic_unet.py:28-65, 68-102, 419-512 (encoder)
  models/diffusion/utils.py:6-54 (sinusoid, swish, TimeStepEmbedder)
  models/diff_unet.py:9-35, models/diffusion/diffusion.py:11-102 (DiffUNet API)

The reference builds its layers through MONAI factories; MONAI is not in this
image, so the wiring below is read from source + MONAI's documented behaviour:
Convolution(...) = Sequential(conv=Conv3d(k3,s1,p1,bias), adn=ADN("NDA":
InstanceNorm3d(affine, eps 1e-5) -> Dropout(p) -> LeakyReLU(0.1))),
Pool["MAX",3](2) = MaxPool3d(2), UpSample(mode="deconv") =
Sequential(deconv=ConvTranspose3d(k2,s2,bias)).  PARITY UNPINNED for that
wiring; the time embedding is pinned by goldens from the reference file.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .diffusion_ref import RefDiffusion, uniform_timesteps


def sinusoid_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """models/diffusion/utils.py:6-24."""
    assert t.dim() == 1
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def swish(x):
    """models/diffusion/utils.py:27-29."""
    return x * torch.sigmoid(x)


class RefTimeStepEmbedder(nn.Module):
    """models/diffusion/utils.py:31-54."""

    def __init__(self, embedding_dim=128, out_features=512):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.dense = nn.ModuleList([nn.Linear(embedding_dim, out_features),
                                    nn.Linear(out_features, out_features)])

    def forward(self, t):
        h = self.dense[0](sinusoid_embedding(t, self.embedding_dim))
        return self.dense[1](swish(h))


class _ADN(nn.Module):
    """MONAI ADN with ordering "NDA": keys adn.N / adn.D / adn.A."""

    def __init__(self, ch, dropout, slope):
        super().__init__()
        self.N = nn.InstanceNorm3d(ch, affine=True)
        self.D = nn.Dropout(dropout)
        self.A = nn.LeakyReLU(negative_slope=slope)

    def forward(self, x):
        return self.A(self.D(self.N(x)))


class _ConvBlock(nn.Module):
    """MONAI Convolution(spatial_dims=3, ..., padding=1): keys conv / adn."""

    def __init__(self, cin, cout, dropout, slope):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
        self.adn = _ADN(cout, dropout, slope)

    def forward(self, x):
        return self.adn(self.conv(x))


class RefTwoConv(nn.Module):
    """denoiser.py:23-67 (with_temb) / pretrained/basic_unet.py:28-65 (without)."""

    def __init__(self, cin, cout, with_temb, dropout=0.0, slope=0.1):
        super().__init__()
        if with_temb:
            self.temb_proj = nn.Linear(512, cout)
        self.conv_0 = _ConvBlock(cin, cout, dropout, slope)
        self.conv_1 = _ConvBlock(cout, cout, dropout, slope)
        self.with_temb = with_temb

    def forward(self, x, temb=None):
        x = self.conv_0(x)
        if self.with_temb:
            x = x + self.temb_proj(swish(temb))[:, :, None, None, None]
        return self.conv_1(x)


class RefDown(nn.Module):
    """denoiser.py:70-108 / pretrained/basic_unet.py:68-102."""

    def __init__(self, cin, cout, with_temb):
        super().__init__()
        self.max_pooling = nn.MaxPool3d(kernel_size=2)
        self.convs = RefTwoConv(cin, cout, with_temb)

    def forward(self, x, temb=None):
        return self.convs(self.max_pooling(x), temb)


class _Deconv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.deconv = nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2, bias=True)

    def forward(self, x):
        return self.deconv(x)


class RefUpCat(nn.Module):
    """denoiser.py:110-194."""

    def __init__(self, cin, cat, cout, halves=True):
        super().__init__()
        up = cin // 2 if halves else cin
        self.upsample = _Deconv(cin, up)
        self.convs = RefTwoConv(cat + up, cout, with_temb=True)

    def forward(self, x, x_e, temb):
        x0 = self.upsample(x)
        pad = [0] * 6
        for i in range(3):
            if x_e.shape[-i - 1] != x0.shape[-i - 1]:
                pad[i * 2 + 1] = 1
        x0 = F.pad(x0, pad, "replicate")
        return self.convs(torch.cat([x_e, x0], dim=1), temb)


class RefDenoiser(nn.Module):
    """denoiser.py:196-312 (BasicUNetRDenoiser)."""

    def __init__(self, in_channels, out_channels, features: Sequence[int]):
        super().__init__()
        f = tuple(features)
        assert len(f) == 6
        self.temb = RefTimeStepEmbedder()
        self.conv_0 = RefTwoConv(in_channels, f[0], True)
        self.down_1 = RefDown(f[0], f[1], True)
        self.down_2 = RefDown(f[1], f[2], True)
        self.down_3 = RefDown(f[2], f[3], True)
        self.down_4 = RefDown(f[3], f[4], True)
        self.upcat_4 = RefUpCat(f[4], f[3], f[3])
        self.upcat_3 = RefUpCat(f[3], f[2], f[2])
        self.upcat_2 = RefUpCat(f[2], f[1], f[1])
        self.upcat_1 = RefUpCat(f[1], f[0], f[5], halves=False)
        self.final_conv = nn.Conv3d(f[5], out_channels, kernel_size=1)

    def forward(self, x, t, image=None, embeddings=None):
        temb = self.temb(t)
        x = torch.cat([image, x], dim=1)
        x0 = self.conv_0(x, temb) + embeddings[0]
        x1 = self.down_1(x0, temb) + embeddings[1]
        x2 = self.down_2(x1, temb) + embeddings[2]
        x3 = self.down_3(x2, temb) + embeddings[3]
        x4 = self.down_4(x3, temb) + embeddings[4]
        u4 = self.upcat_4(x4, x3, temb)
        u3 = self.upcat_3(u4, x2, temb)
        u2 = self.upcat_2(u3, x1, temb)
        u1 = self.upcat_1(u2, x0, temb)
        return self.final_conv(u1)


class RefEncoder(nn.Module):
    """pretrained/basic_unet.py:419-512 (BasicUNetEncoder; ModuleList ``down``)."""

    def __init__(self, in_channels, features: Sequence[int]):
        super().__init__()
        f = tuple(features)
        self.conv_0 = RefTwoConv(in_channels, f[0], False)
        self.down = nn.ModuleList([RefDown(f[d], f[d + 1], False) for d in range(4)])

    def forward(self, x):
        outs = [self.conv_0(x)]
        for d in self.down:
            outs.append(d(outs[-1]))
        return outs


class RefDiffUNet(nn.Module):
    """models/diff_unet.py:9-35 on top of models/diffusion/diffusion.py:11-102."""

    def __init__(self, spatial_dims=3, in_channels=3, out_channels=1, image_size=96, spatial_size=96,
                 features=(64, 64, 128, 256, 512, 64), dropout=0.2, timesteps=1000, mode="train",
                 sample_steps=10):
        super().__init__()
        self.num_classes = out_channels
        self.mode = mode
        self.diffusion = RefDiffusion(timesteps, [timesteps])
        self.sample_diffusion = RefDiffusion(timesteps, [sample_steps])
        self.timesteps = timesteps
        self.embed_model = RefEncoder(in_channels, features)
        self.model = RefDenoiser(out_channels + 1, out_channels, features)

    def forward(self, image=None, x=None, step=None, pred_type=None, **inject):
        if pred_type == "q_sample":
            return self.q_sample(x, **inject)
        if pred_type == "denoise":
            return self.denoise(image, x, step)
        if pred_type == "ddim_sample":
            return self.ddim_sample(image, **inject)
        raise NotImplementedError(f"No such prediction type : {pred_type}")

    def q_sample(self, x, noise=None, t=None, rng=None):
        """diffusion.py:65-69; noise/t injectable because RNG streams differ per device."""
        if noise is None:
            noise = torch.randn_like(x)
        if t is None:
            t, _ = uniform_timesteps(self.timesteps, x.shape[0], rng or np.random)
        return self.diffusion.q_sample(x, t, noise), t, noise

    def denoise(self, image, x, step):
        """diffusion.py:71-84."""
        assert image.size(0) == x.size(0) == step.size(0)
        return self.model(x=x, t=step, embeddings=self.embed_model(image), image=image)

    def ddim_sample(self, image, x_T=None, step_noise=None):
        """diffusion.py:86-102: per-sample loop, encoder once, sum of clamped x0 predictions.
        ``x_T[i]`` / ``step_noise[i][k]`` are the injected draws for batch item i."""
        res = []
        T = self.sample_diffusion.num_timesteps
        for i in range(len(image)):
            b = image[i:i + 1]
            emb = self.embed_model(b)
            shape = (1, self.num_classes, *image.shape[2:])
            xt = x_T[i] if x_T is not None else torch.randn(*shape)
            sn = step_noise[i] if step_noise is not None else [torch.randn(*shape) for _ in range(T)]
            out = self.sample_diffusion.ddim_sample_loop(self.model, xt, sn,
                                                         model_kwargs={"image": b, "embeddings": emb})
            acc = torch.zeros(shape)
            for s in out["all_samples"]:
                acc += s
            res.append(acc)
        return torch.cat(res, dim=0)


# ---- caller-side formulas used by the harness -------

def dice_coeff(result: torch.Tensor, reference: torch.Tensor) -> float:
    """metric.py:37-49 on binary masks."""
    inter = torch.sum(result.bool() & reference.bool()).item()
    s = torch.sum(result).item() + torch.sum(reference).item()
    return 0.0 if s == 0 else 2.0 * inter / float(s)


def binarise(logit_sum: torch.Tensor) -> torch.Tensor:
    """engine.py:179-180."""
    return (torch.sigmoid(logit_sum) > 0.5).float()
