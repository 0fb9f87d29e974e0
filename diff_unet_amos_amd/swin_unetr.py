"""Parameter containers of the two Swin-UNETR networks of the diff_swin_unetr variant (BASELINE config 5), with the
reference's module tree so that state-dict keys are identical:

  SwinUNETRDenoiser  models/swin_unetr/denoiser.py:36-408   (SwinTransformer transformer.py:124-316, BasicLayer :24-121,
                                                             SwinTransformerBlock :319-481, WindowAttention
                                                             attention.py:14-120, PatchMerging patch.py:67-93,
                                                             UnetrBasicBlock / UnetrUpBlock / UnetResBlock / UnetOutBlock
                                                             blocks.py:26-337, TimeStepEmbedder models/diffusion/utils.py:31-54)
  SwinUNETREncoder   models/swin_unetr/encoder.py:19-219    (MONAI's SwinTransformer and UnetrBasicBlock: the same trees
                                                             without ``t_proj``)

The reference builds the convolution blocks through MONAI factories (get_conv_layer -> Convolution with the single
child ``conv``; get_norm_layer("instance") -> InstanceNorm3d without parameters; MLPBlock -> linear1 / linear2); here
torch.nn layers are used only to own and initialise the parameters under those names.  ``forward`` never runs them:
it dispatches to the HIP launch plan (swin_engine.py), which is the inference path (eval-mode arithmetic: the
reference's Dropout / DropPath modules act in train mode only).  Training this variant is out of scope (SURVEY.md
8(f)-3 asks for the denoiser's forward kernels); calling it with gradients requested is refused.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .basic_unet import TimeStepEmbedder, _wants_grad

WINDOW = (7, 7, 7)          # ensure_tuple_rep(7, 3), denoiser.py:101
DEPTHS = (2, 2, 2, 2)
HEADS = (3, 6, 12, 24)


def relative_position_index(window_size):
    """attention.py:56-93: index into the (2wd-1)(2wh-1)(2ww-1) bias table for every (query, key) pair of a window."""
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


class _Conv(nn.Module):
    """MONAI Convolution without norm / activation / dropout: key ``conv``."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv


class UnetResBlock(nn.Module):
    """blocks.py:219-316 (kernel 3, stride 1, norm "instance": no norm parameters)."""

    def __init__(self, cin, cout, embedding_size):
        super().__init__()
        self.conv1 = _Conv(nn.Conv3d(cin, cout, 3, 1, 1, bias=False))
        if embedding_size is not None:
            self.t_proj = nn.Linear(embedding_size, cout)
        self.conv2 = _Conv(nn.Conv3d(cout, cout, 3, 1, 1, bias=False))
        if cin != cout:
            self.conv3 = _Conv(nn.Conv3d(cin, cout, 1, 1, 0, bias=False))


class UnetrBasicBlock(nn.Module):
    def __init__(self, cin, cout, embedding_size):
        super().__init__()
        self.layer = UnetResBlock(cin, cout, embedding_size)


class UnetrUpBlock(nn.Module):
    def __init__(self, cin, cout, embedding_size):
        super().__init__()
        self.transp_conv = _Conv(nn.ConvTranspose3d(cin, cout, 2, 2, bias=False))
        self.conv_block = UnetResBlock(2 * cout, cout, embedding_size)


class UnetOutBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(nn.Conv3d(cin, cout, 1, 1, 0, bias=True))


class WindowAttention(nn.Module):
    def __init__(self, dim, heads, window):
        super().__init__()
        wd, wh, ww = window
        self.num_heads = heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), heads))
        self.register_buffer("relative_position_index", relative_position_index(window))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Linear(dim, hidden)
        self.linear2 = nn.Linear(hidden, dim)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, heads, window):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, heads, window)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * 4.0))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, heads, window):
        super().__init__()
        self.blocks = nn.ModuleList([SwinTransformerBlock(dim, heads, window) for _ in range(depth)])
        self.downsample = PatchMerging(dim)


class PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, embed_dim, 2, 2)


class SwinTransformer(nn.Module):
    def __init__(self, in_chans, embed_dim, embedding_size):
        super().__init__()
        self.patch_embed = PatchEmbed(in_chans, embed_dim)
        for i, name in enumerate(("layers1", "layers2", "layers3", "layers4")):
            setattr(self, name, nn.ModuleList([BasicLayer(embed_dim * 2 ** i, DEPTHS[i], HEADS[i], WINDOW)]))
        if embedding_size is not None:
            self.t_proj = nn.ModuleList([nn.Linear(embedding_size, embed_dim * 2 ** i) for i in range(5)])

    def stages(self):
        return [self.layers1[0], self.layers2[0], self.layers3[0], self.layers4[0]]


def _refuse_autograd(*tensors):
    if _wants_grad(*tensors):
        raise NotImplementedError(
            "the Swin-UNETR networks run the inference launch plan, which keeps no autograd tape (training the "
            "diff_swin_unetr variant is out of scope of this package); wrap the call in torch.no_grad()")


def _check_feature_size(feature_size):
    if feature_size % 12 != 0:
        raise ValueError("feature_size should be divisible by 12.")                       # denoiser.py:125-126
    assert feature_size == 48, ("the windowed-attention kernel is built for head dimension 16 "
                                "(feature_size 48, the value BASELINE config 5 names)")


class SwinUNETREncoder(nn.Module):
    """Conditioning encoder: image -> [5 normalised hidden states, enc0, enc1, enc2, enc3] (encoder.py:212-219)."""

    def __init__(self, image_size: Sequence[int] | int = 96, in_channels: int = 1, spatial_dims: int = 3,
                 feature_size: int = 48, drop_rate: float = 0.0, **_unused):
        super().__init__()
        assert spatial_dims == 3, "the MI355X path is 3-D"
        assert in_channels == 1, "Diff-UNet conditions on a single-channel CT image (models/utils/model_hub.py:16-20)"
        _check_feature_size(feature_size)
        f = feature_size
        self.normalize = True
        self.swinViT = SwinTransformer(in_channels, f, None)
        self.encoder1 = UnetrBasicBlock(in_channels, f, None)
        self.encoder2 = UnetrBasicBlock(f, f, None)
        self.encoder3 = UnetrBasicBlock(2 * f, 2 * f, None)
        self.encoder4 = UnetrBasicBlock(4 * f, 4 * f, None)
        object.__setattr__(self, "_rt", None)

    def forward(self, x_in: torch.Tensor):
        _refuse_autograd(x_in)
        rt = self._rt
        assert rt is not None, "SwinUNETREncoder must be owned by a DiffSwinUNETR (shared launch plan)"
        return rt.plan_for(x_in).run_encoder(x_in)


class SwinUNETRDenoiser(nn.Module):
    """Time-conditioned denoiser: (x_t, t, image, embeddings) -> logits (denoiser.py:353-403)."""

    def __init__(self, image_size: Sequence[int] | int = 96, in_channels: int = 17, out_channels: int = 16,
                 spatial_dims: int = 3, feature_size: int = 48, embedding_size: int = 512, embedding_dim: int = 128,
                 noise_ratio: float = 0.5, drop_rate: float = 0.0, **_unused):
        super().__init__()
        assert spatial_dims == 3, "the MI355X path is 3-D"
        _check_feature_size(feature_size)
        f, e = feature_size, embedding_size
        self.in_channels, self.num_classes, self.noise_ratio, self.normalize = in_channels, out_channels, noise_ratio, True
        self.t_embedder = TimeStepEmbedder(embedding_dim, e)
        self.swinViT = SwinTransformer(in_channels, f, e)
        self.encoder1 = UnetrBasicBlock(in_channels, f, e)
        self.encoder2 = UnetrBasicBlock(f, f, e)
        self.encoder3 = UnetrBasicBlock(2 * f, 2 * f, e)
        self.encoder4 = UnetrBasicBlock(4 * f, 4 * f, e)
        self.encoder10 = UnetrBasicBlock(16 * f, 16 * f, e)
        self.decoder5 = UnetrUpBlock(16 * f, 8 * f, e)
        self.decoder4 = UnetrUpBlock(8 * f, 4 * f, e)
        self.decoder3 = UnetrUpBlock(4 * f, 2 * f, e)
        self.decoder2 = UnetrUpBlock(2 * f, f, e)
        self.decoder1 = UnetrUpBlock(f, f, e)
        self.out = UnetOutBlock(f, out_channels)
        object.__setattr__(self, "_rt", None)

    def forward(self, x: torch.Tensor, t: torch.Tensor, image: torch.Tensor = None, embeddings=None):
        _refuse_autograd(x, image)
        rt = self._rt
        assert rt is not None, "SwinUNETRDenoiser must be owned by a DiffSwinUNETR (shared launch plan)"
        assert image is not None and embeddings is not None, "the denoiser is conditioned on image and embeddings"
        plan = rt.plan_for(x)
        plan.stage_condition(image, embeddings)
        return plan.denoise(x, t)

    def fused_engine(self, shape, model_kwargs):
        """Hook used by GaussianDiffusion loops: a launch plan for ``shape`` with the conditioning staged."""
        rt = self._rt
        if rt is None:
            return None
        image = model_kwargs["image"]
        plan = rt.plan(shape[0], tuple(shape[2:]), image.device)
        plan.stage_condition(image, model_kwargs["embeddings"])
        return rt.adapter(plan)
