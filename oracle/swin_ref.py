"""Oracle (test infrastructure, never shipped): CPU restatement of the Swin pieces of the diff_swin_unetr variant
(BASELINE config 5).

Follows, in the reference tree:
  models/swin_unetr/attention.py:14-120    WindowAttention: relative-position index / bias, scaled QK^T + bias (+ mask),
                                           softmax, PV, proj
  models/swin_unetr/attention.py:123-160   compute_mask (region ids 0..26 per shifted window, -100 between regions)
  models/swin_unetr/attention.py:163-222   window_partition / window_reverse (3-D branch)
  models/swin_unetr/attention.py:225-251   get_window_size
  models/swin_unetr/patch.py:19-93         PatchMergingV2 and the legacy PatchMerging (its 3-D gather lists x2 and x3
                                           twice -- x5 == x2, x6 == x3 -- and never reads the (1,1,0) / (0,1,1) corners;
                                           reproduced here as it is)

PARITY UNPINNED: attention.py and patch.py import MONAI at module top (trunc_normal_, optional_import, LayerNorm
typing), MONAI is absent from this image and from the reference tree (ordinary ModuleNotFoundError), and the reference
holds no fixtures for these functions.  The arithmetic below is plain torch.
"""
from __future__ import annotations

import itertools

import torch
import torch.nn as nn
import torch.nn.functional as F


def relative_position_index(window_size):
    """attention.py:56-73: index into the (2wd-1)(2wh-1)(2ww-1) bias table for every (query, key) pair of a window."""
    wd, wh, ww = window_size
    coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wd - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= 2 * ww - 1
    return rel.sum(-1)


def window_partition(x, window_size):
    """attention.py:163-192, 5-D branch: [b, d, h, w, c] -> [b * windows, wd*wh*ww, c]."""
    b, d, h, w, c = x.shape
    wd, wh, ww = window_size
    x = x.view(b, d // wd, wd, h // wh, wh, w // ww, ww, c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, wd * wh * ww, c)


def window_reverse(windows, window_size, dims):
    """attention.py:195-222, 4-entry dims branch."""
    b, d, h, w = dims
    wd, wh, ww = window_size
    x = windows.view(b, d // wd, h // wh, w // ww, wd, wh, ww, -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def get_window_size(x_size, window_size, shift_size=None):
    """attention.py:225-251: a window never exceeds the feature map; such an axis is not shifted."""
    ws = list(window_size)
    ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            if ss is not None:
                ss[i] = 0
    return tuple(ws) if ss is None else (tuple(ws), tuple(ss))


def compute_mask(dims, window_size, shift_size):
    """attention.py:123-160 (3-D): [windows, n, n] with 0 inside a region and -100 across regions."""
    d, h, w = dims
    img = torch.zeros((1, d, h, w, 1))
    cnt = 0
    for sd in (slice(-window_size[0]), slice(-window_size[0], -shift_size[0]), slice(-shift_size[0], None)):
        for sh in (slice(-window_size[1]), slice(-window_size[1], -shift_size[1]), slice(-shift_size[1], None)):
            for sw in (slice(-window_size[2]), slice(-window_size[2], -shift_size[2]), slice(-shift_size[2], None)):
                img[:, sd, sh, sw, :] = cnt
                cnt += 1
    mw = window_partition(img, window_size).squeeze(-1)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))


class RefWindowAttention(nn.Module):
    """attention.py:14-120 (3-D windows; dropout rates 0)."""

    def __init__(self, dim, num_heads, window_size, qkv_bias=False):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.scale = (dim // num_heads) ** -0.5
        wd, wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(self.window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def bias(self, n):
        """attention.py:103-106: [heads, n, n]."""
        idx = self.relative_position_index[:n, :n].reshape(-1)
        return self.relative_position_bias_table[idx].reshape(n, n, -1).permute(2, 0, 1).contiguous()

    def attention_core(self, qkv, mask):
        """Everything between the two Linear layers (attention.py:99-117): [b, n, 3c] -> [b, n, c]."""
        b, n, c3 = qkv.shape
        c = c3 // 3
        qkv = qkv.reshape(b, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1) + self.bias(n).unsqueeze(0)
        if mask is not None:
            nw = mask.shape[0]
            attn = attn.view(b // nw, nw, self.num_heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, n, n)
        attn = torch.softmax(attn, dim=-1)
        return (attn @ v).transpose(1, 2).reshape(b, n, c)

    def forward(self, x, mask):
        return self.proj(self.attention_core(self.qkv(x), mask))


def patch_merging_gather(x, legacy=True):
    """patch.py:44-61 (V2) / :70-91 (legacy): [b, d, h, w, c] -> [b, d/2, h/2, w/2, 8c] before norm + reduction."""
    b, d, h, w, c = x.shape
    if (h % 2 == 1) or (w % 2 == 1) or (d % 2 == 1):
        x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2, 0, d % 2))
    if not legacy:
        return torch.cat([x[:, i::2, j::2, k::2, :] for i, j, k in itertools.product(range(2), range(2), range(2))], -1)
    corners = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1)]    # x5 == x2, x6 == x3
    return torch.cat([x[:, i::2, j::2, k::2, :] for i, j, k in corners], -1)


class RefPatchMerging(nn.Module):
    """patch.py:67-93: gather -> LayerNorm(8c) -> Linear(8c, 2c, bias=False)."""

    def __init__(self, dim, legacy=True):
        super().__init__()
        self.dim, self.legacy = dim, legacy
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim)

    def forward(self, x):
        return self.reduction(self.norm(patch_merging_gather(x, self.legacy)))


def nonlinearity(x):
    """models/diffusion/utils.py:27-29 (swish), applied to the time embedding before every t_proj."""
    return x * torch.sigmoid(x)


class RefUnetResBlock(nn.Module):
    """models/swin_unetr/blocks.py:219-316 for 3-D, kernel 3, stride 1, instance norm:
        conv1 -> norm1 -> LeakyReLU(0.01) -> + t_proj(swish(t)) -> conv2 -> norm2 -> (+ norm3(conv3(inp)) | + inp) -> LeakyReLU
    MONAI's get_conv_layer(..., act=None, norm=None, conv_only=False) is restated as a bias-free Conv3d with "same"
    padding (its documented default bias=False), get_norm_layer(("instance", {"affine": True})) as
    InstanceNorm3d(affine=True) -- the norm_name the reference passes (swin_unetr/denoiser.py) -- PARITY UNPINNED."""

    def __init__(self, in_channels, out_channels, embedding_size=512, affine=True):
        super().__init__()
        self.conv1 = nn.Conv3d(in_channels, out_channels, 3, 1, 1, bias=False)
        self.t_proj = nn.Linear(embedding_size, out_channels)
        self.conv2 = nn.Conv3d(out_channels, out_channels, 3, 1, 1, bias=False)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01)
        self.norm1 = nn.InstanceNorm3d(out_channels, affine=affine)
        self.norm2 = nn.InstanceNorm3d(out_channels, affine=affine)
        self.downsample = in_channels != out_channels
        if self.downsample:
            self.conv3 = nn.Conv3d(in_channels, out_channels, 1, 1, 0, bias=False)
            self.norm3 = nn.InstanceNorm3d(out_channels, affine=affine)

    def forward(self, inp, t):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = out + self.t_proj(nonlinearity(t))[:, :, None, None, None]
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        return self.lrelu(out + residual)
