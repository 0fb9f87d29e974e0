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

  models/swin_unetr/transformer.py:24-121  BasicLayer (two blocks, the second shifted by window // 2, then PatchMerging)
  models/swin_unetr/transformer.py:124-316 SwinTransformer with the t_proj adds between stages (denoiser side)
  models/swin_unetr/transformer.py:319-481 SwinTransformerBlock (norm1 -> pad -> roll -> windows -> attention -> reverse
                                           -> roll back -> crop; + shortcut; + mlp(norm2))
  models/swin_unetr/blocks.py:26-142       UnetrUpBlock / UnetrBasicBlock, :219-316 UnetResBlock with t_proj, :319-337 out
  models/swin_unetr/denoiser.py:36-408     SwinUNETRDenoiser (forward :353-403, reverse_attention :405-408)
  models/swin_unetr/encoder.py:19-219      SwinUNETREncoder -- built on MONAI'S OWN SwinTransformer / UnetrBasicBlock
  models/diff_swin_unetr.py:7-47           DiffSwinUNETR

PARITY UNPINNED: every file above imports MONAI at module top (trunc_normal_, optional_import, MLPBlock, PatchEmbed,
get_conv_layer, get_norm_layer, DropPath ...), MONAI is absent from this image and from the reference tree (ordinary
ModuleNotFoundError), and the reference holds no fixtures for these functions.  The MONAI pieces are restated from
their published behaviour (MONAI 1.x): PatchEmbed = Conv3d(k = s = patch) on an input padded to a patch multiple;
MLPBlock(act="GELU", dropout_mode="swin") = linear1 -> GELU(erf) -> drop -> linear2 -> drop; get_conv_layer(act=None,
norm=None) = Convolution whose only child is ``conv`` (bias=False unless asked; "same" padding; transposed k2 s2:
padding 0, output_padding 0); get_norm_layer("instance") = InstanceNorm3d(channels) -- affine=False, no parameters;
MONAI SwinTransformer.forward(x, normalize) = the reference's transformer.py forward without the t_proj adds; MONAI
UnetResBlock = blocks.py:298-316 without the t_proj add.  Dropout / DropPath rates only act in train mode; the oracle
and the product run the eval-mode arithmetic.  The arithmetic below is plain torch.
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


class _Conv(nn.Module):
    """MONAI Convolution with neither norm, activation nor dropout: one child, ``conv`` (state-dict key ``*.conv.weight``)."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv

    def forward(self, x):
        return self.conv(x)


class RefUnetResBlock(nn.Module):
    """models/swin_unetr/blocks.py:219-316 for 3-D, kernel 3, stride 1, norm "instance":
        conv1 -> norm1 -> LeakyReLU(0.01) -> + t_proj(swish(t)) -> conv2 -> norm2 -> (+ norm3(conv3(inp)) | + inp) -> LeakyReLU
    ``embedding_size=None`` is MONAI's own UnetResBlock (what models/swin_unetr/encoder.py:9 imports): no t_proj.
    ``affine=True`` is kept for kernel tests that want visible gamma / beta; the reference's norm_name is "instance"."""

    def __init__(self, in_channels, out_channels, embedding_size=512, affine=False):
        super().__init__()
        self.conv1 = _Conv(nn.Conv3d(in_channels, out_channels, 3, 1, 1, bias=False))
        if embedding_size is not None:
            self.t_proj = nn.Linear(embedding_size, out_channels)
        self.conv2 = _Conv(nn.Conv3d(out_channels, out_channels, 3, 1, 1, bias=False))
        self.lrelu = nn.LeakyReLU(negative_slope=0.01)
        self.norm1 = nn.InstanceNorm3d(out_channels, affine=affine)
        self.norm2 = nn.InstanceNorm3d(out_channels, affine=affine)
        self.downsample = in_channels != out_channels
        if self.downsample:
            self.conv3 = _Conv(nn.Conv3d(in_channels, out_channels, 1, 1, 0, bias=False))
            self.norm3 = nn.InstanceNorm3d(out_channels, affine=affine)

    def forward(self, inp, t=None):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        if hasattr(self, "t_proj"):
            out = out + self.t_proj(nonlinearity(t))[:, :, None, None, None]
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        return self.lrelu(out + residual)


class RefUnetrBasicBlock(nn.Module):
    """blocks.py:95-142 with res_block=True: key ``layer``."""

    def __init__(self, in_channels, out_channels, embedding_size=512):
        super().__init__()
        self.layer = RefUnetResBlock(in_channels, out_channels, embedding_size)

    def forward(self, inp, t=None):
        return self.layer(inp, t)


class RefUnetrUpBlock(nn.Module):
    """blocks.py:26-93 with res_block=True, upsample_kernel_size 2: ConvTranspose3d(k2, s2, no bias) -> cat((up, skip)) ->
    UnetResBlock(2 * out, out)."""

    def __init__(self, in_channels, out_channels, embedding_size=512):
        super().__init__()
        self.transp_conv = _Conv(nn.ConvTranspose3d(in_channels, out_channels, 2, 2, bias=False))
        self.conv_block = RefUnetResBlock(2 * out_channels, out_channels, embedding_size)

    def forward(self, inp, skip, t):
        return self.conv_block(torch.cat((self.transp_conv(inp), skip), dim=1), t)


class RefMlp(nn.Module):
    """MONAI MLPBlock(hidden, mlp_dim, act="GELU", dropout_mode="swin") in eval mode."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Linear(dim, hidden)
        self.linear2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.linear2(F.gelu(self.linear1(x)))


class RefSwinBlock(nn.Module):
    """transformer.py:319-481 (3-D branch, drop rates 0)."""

    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = RefWindowAttention(dim, num_heads, window_size, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = RefMlp(dim, int(dim * mlp_ratio))

    def part1(self, x, mask_matrix):
        """transformer.py:378-431."""
        x = self.norm1(x)
        b, d, h, w, c = x.shape
        ws, ss = get_window_size((d, h, w), self.window_size, self.shift_size)
        pd, ph, pw = [(ws[i] - s % ws[i]) % ws[i] for i, s in enumerate((d, h, w))]
        x = F.pad(x, (0, 0, 0, pw, 0, ph, 0, pd))
        dims = [b, x.shape[1], x.shape[2], x.shape[3]]
        if any(i > 0 for i in ss):
            shifted, mask = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3)), mask_matrix
        else:
            shifted, mask = x, None
        win = self.attn(window_partition(shifted, ws), mask)
        shifted = window_reverse(win.view(-1, *(ws + (c,))), ws, dims)
        if any(i > 0 for i in ss):
            x = torch.roll(shifted, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
        else:
            x = shifted
        return x[:, :d, :h, :w, :].contiguous()

    def forward(self, x, mask_matrix):
        x = x + self.part1(x, mask_matrix)
        return x + self.mlp(self.norm2(x))


class RefBasicLayer(nn.Module):
    """transformer.py:24-121: blocks (even: unshifted, odd: shifted by window // 2), then the legacy PatchMerging."""

    def __init__(self, dim, depth, num_heads, window_size, legacy_merging=True):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.blocks = nn.ModuleList([
            RefSwinBlock(dim, num_heads, self.window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size)
            for i in range(depth)])
        self.downsample = RefPatchMerging(dim, legacy=legacy_merging)

    def forward(self, x):
        b, c, d, h, w = x.shape
        ws, ss = get_window_size((d, h, w), self.window_size, self.shift_size)
        x = x.permute(0, 2, 3, 4, 1)
        dp, hp, wp = [-(-s // ws[i]) * ws[i] for i, s in enumerate((d, h, w))]
        mask = compute_mask([dp, hp, wp], ws, ss) if any(i > 0 for i in ss) else None
        for blk in self.blocks:
            x = blk(x, mask)
        x = self.downsample(x.reshape(b, d, h, w, -1))
        return x.permute(0, 4, 1, 2, 3)


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim, patch=2):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv3d(in_chans, embed_dim, patch, patch)

    def forward(self, x):
        p = self.patch
        d, h, w = x.shape[2:]
        if d % p or h % p or w % p:
            x = F.pad(x, (0, (p - w % p) % p, 0, (p - h % p) % p, 0, (p - d % p) % p))
        return self.proj(x)


class RefSwinTransformer(nn.Module):
    """transformer.py:124-316.  ``embedding_size=None`` is MONAI's SwinTransformer (encoder side): the same walk without
    t_proj."""

    def __init__(self, in_chans, embed_dim, window_size=(7, 7, 7), depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24),
                 embedding_size=512):
        super().__init__()
        self.patch_embed = _PatchEmbed(in_chans, embed_dim)
        for i, name in enumerate(("layers1", "layers2", "layers3", "layers4")):
            setattr(self, name, nn.ModuleList([RefBasicLayer(embed_dim * 2 ** i, depths[i], num_heads[i], window_size)]))
        if embedding_size is not None:
            self.t_proj = nn.ModuleList([nn.Linear(embedding_size, embed_dim * 2 ** i) for i in range(5)])

    @staticmethod
    def proj_out(x, normalize):
        if not normalize:
            return x
        return F.layer_norm(x.permute(0, 2, 3, 4, 1), [x.shape[1]]).permute(0, 4, 1, 2, 3)

    def forward(self, x, t=None, normalize=True):
        outs = []
        x = self.patch_embed(x)
        for i, layer in enumerate((None, self.layers1, self.layers2, self.layers3, self.layers4)):
            if layer is not None:
                x = layer[0](x.contiguous())
            if hasattr(self, "t_proj"):
                x = x + self.t_proj[i](nonlinearity(t))[:, :, None, None, None]
            outs.append(self.proj_out(x, normalize))
        return outs


class _OutBlock(nn.Module):
    """blocks.py:319-337: 1x1x1 convolution with bias; keys ``conv.conv.*``."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(nn.Conv3d(cin, cout, 1, 1, 0, bias=True))

    def forward(self, x):
        return self.conv(x)


class RefSwinUNETREncoder(nn.Module):
    """encoder.py:19-219 (MONAI SwinTransformer + MONAI UnetrBasicBlock(res_block=True))."""

    def __init__(self, in_channels=1, feature_size=48):
        super().__init__()
        f = feature_size
        self.swinViT = RefSwinTransformer(in_channels, f, embedding_size=None)
        self.encoder1 = RefUnetrBasicBlock(in_channels, f, None)
        self.encoder2 = RefUnetrBasicBlock(f, f, None)
        self.encoder3 = RefUnetrBasicBlock(2 * f, 2 * f, None)
        self.encoder4 = RefUnetrBasicBlock(4 * f, 4 * f, None)

    def forward(self, x_in):
        hs = self.swinViT(x_in, None, True)
        return [hs, self.encoder1(x_in), self.encoder2(hs[0]), self.encoder3(hs[1]), self.encoder4(hs[2])]


def reverse_attention(x):
    """denoiser.py:405-408."""
    return x * (1 - torch.sigmoid(x))


class RefSwinUNETRDenoiser(nn.Module):
    """denoiser.py:36-408."""

    def __init__(self, in_channels, out_channels, feature_size=48, embedding_size=512, embedding_dim=128):
        super().__init__()
        from .unet_ref import RefTimeStepEmbedder
        f, e = feature_size, embedding_size
        self.t_embedder = RefTimeStepEmbedder(embedding_dim, e)
        self.swinViT = RefSwinTransformer(in_channels, f, embedding_size=e)
        self.encoder1 = RefUnetrBasicBlock(in_channels, f, e)
        self.encoder2 = RefUnetrBasicBlock(f, f, e)
        self.encoder3 = RefUnetrBasicBlock(2 * f, 2 * f, e)
        self.encoder4 = RefUnetrBasicBlock(4 * f, 4 * f, e)
        self.encoder10 = RefUnetrBasicBlock(16 * f, 16 * f, e)
        self.decoder5 = RefUnetrUpBlock(16 * f, 8 * f, e)
        self.decoder4 = RefUnetrUpBlock(8 * f, 4 * f, e)
        self.decoder3 = RefUnetrUpBlock(4 * f, 2 * f, e)
        self.decoder2 = RefUnetrUpBlock(2 * f, f, e)
        self.decoder1 = RefUnetrUpBlock(f, f, e)
        self.out = _OutBlock(f, out_channels)

    def forward(self, x, t, image=None, embeddings=None):
        t = self.t_embedder(t)
        x = torch.cat([image, x], dim=1)
        hs = self.swinViT(x, t, True)
        hs = [h + e for h, e in zip(hs, embeddings[0])]
        enc0 = self.encoder1(x, t) + embeddings[1]
        enc1 = self.encoder2(hs[0], t) + embeddings[2]
        enc2 = self.encoder3(hs[1], t) + embeddings[3]
        enc3 = self.encoder4(hs[2], t) + embeddings[4]
        r0, r1, r2, r3 = (reverse_attention(e) for e in (enc0, enc1, enc2, enc3))
        dec4 = self.encoder10(hs[4], t)
        dec3 = self.decoder5(dec4, hs[3], t)
        dec2 = self.decoder4(dec3, enc3, t) + r3
        dec1 = self.decoder3(dec2, enc2, t) + r2
        dec0 = self.decoder2(dec1, enc1, t) + r1
        out = self.decoder1(dec0, enc0, t) + r0
        return self.out(out)


def make_ref_diff_swin_unetr(in_channels=1, out_channels=16, feature_size=48, timesteps=1000, sample_steps=10):
    """models/diff_swin_unetr.py:7-47 on the oracle's Diffusion restatement (oracle/unet_ref.py RefDiffUNet)."""
    from .unet_ref import RefDiffUNet
    net = RefDiffUNet(in_channels=in_channels, out_channels=out_channels, features=(8, 8, 8, 8, 8, 8), timesteps=timesteps,
                      sample_steps=sample_steps)
    net.embed_model = RefSwinUNETREncoder(in_channels, feature_size)
    net.model = RefSwinUNETRDenoiser(out_channels + 1, out_channels, feature_size)
    return net
