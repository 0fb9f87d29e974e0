"""Parameter containers of the two BasicUNet networks, with the reference's module tree so that
state-dict keys are identical:

  BasicUNetRDenoiser  models/basic_unet/denoiser.py:196-312  (TwoConv :23-67, Down :70-108, UpCat :110-194)
  BasicUNetEncoder    models/basic_unet/pretrained/basic_unet.py:419-512
  TimeStepEmbedder    models/diffusion/utils.py:31-54

The reference builds these blocks through MONAI factories (Convolution -> conv/adn.N/adn.D/adn.A,
UpSample(mode="deconv") -> deconv); here torch.nn layers are used only to own and initialise the
parameters under those names.  ``forward`` never runs them: it dispatches to the HIP launch plan
(engine.py).  The launch plan is the inference path; training goes through ``Diffusion.forward(pred_type="denoise")``,
which runs the forward AND backward HIP kernels under autograd (training.py).  Calling one of the two sub-networks
on its own with gradients requested is refused with a pointer to that entry.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn


class ADN(nn.Module):
    def __init__(self, ch, dropout=0.0, slope=0.1):
        super().__init__()
        self.N = nn.InstanceNorm3d(ch, affine=True)
        self.D = nn.Dropout(dropout)
        self.A = nn.LeakyReLU(negative_slope=slope)


class Convolution(nn.Module):
    def __init__(self, cin, cout, dropout=0.0, slope=0.1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
        self.adn = ADN(cout, dropout, slope)


class TwoConv(nn.Module):
    def __init__(self, cin, cout, with_temb):
        super().__init__()
        if with_temb:
            self.temb_proj = nn.Linear(512, cout)
        self.conv_0 = Convolution(cin, cout)
        self.conv_1 = Convolution(cout, cout)


class Down(nn.Module):
    def __init__(self, cin, cout, with_temb):
        super().__init__()
        self.max_pooling = nn.MaxPool3d(kernel_size=2)
        self.convs = TwoConv(cin, cout, with_temb)


class UpSample(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.deconv = nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2, bias=True)


class UpCat(nn.Module):
    def __init__(self, cin, cat, cout, halves=True):
        super().__init__()
        up = cin // 2 if halves else cin
        self.upsample = UpSample(cin, up)
        self.convs = TwoConv(cat + up, cout, True)


class TimeStepEmbedder(nn.Module):
    def __init__(self, embedding_dim=128, out_features=512):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.dense = nn.ModuleList([nn.Linear(embedding_dim, out_features), nn.Linear(out_features, out_features)])


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _refuse_autograd(*tensors):
    if _wants_grad(*tensors):
        raise NotImplementedError(
            "embed_model / model called on their own run the inference launch plan, which keeps no autograd tape; "
            "for gradients call DiffUNet.forward(image=..., x=..., step=..., pred_type=\"denoise\") (HIP forward and "
            "backward kernels, training.native_conv_denoise) or wrap the call in torch.no_grad()")


class BasicUNetEncoder(nn.Module):
    """Conditioning encoder: image -> 5 feature maps added to the denoiser's encoder levels."""

    def __init__(self, spatial_dims: int = 3, in_channels: int = 1, out_channels: int = 2,
                 features: Sequence[int] = (64, 64, 128, 256, 512, 64)):
        super().__init__()
        assert spatial_dims == 3, "the MI355X path is 3-D"
        assert in_channels == 1, "Diff-UNet conditions on a single-channel CT image (models/utils/model_hub.py:16-20)"
        f = tuple(features)
        assert len(f) == 6
        self.conv_0 = TwoConv(in_channels, f[0], False)
        self.down = nn.ModuleList([Down(f[d], f[d + 1], False) for d in range(4)])
        object.__setattr__(self, "_rt", None)

    def forward(self, x: torch.Tensor):
        _refuse_autograd(x)
        rt = self._rt
        assert rt is not None, "BasicUNetEncoder must be owned by a DiffUNet (shared launch plan)"
        return rt.plan_for(x).run_encoder(x)


class BasicUNetRDenoiser(nn.Module):
    """Time-conditioned denoiser: (x_t, t, image, embeddings) -> logits."""

    def __init__(self, spatial_dims: int = 3, in_channels: int = 1, out_channels: int = 2,
                 features: Sequence[int] = (32, 32, 64, 128, 256, 32), act=None):
        super().__init__()
        assert spatial_dims == 3, "the MI355X path is 3-D"
        f = tuple(features)
        assert len(f) == 6
        self.temb = TimeStepEmbedder()
        self.conv_0 = TwoConv(in_channels, f[0], True)
        self.down_1 = Down(f[0], f[1], True)
        self.down_2 = Down(f[1], f[2], True)
        self.down_3 = Down(f[2], f[3], True)
        self.down_4 = Down(f[3], f[4], True)
        self.upcat_4 = UpCat(f[4], f[3], f[3])
        self.upcat_3 = UpCat(f[3], f[2], f[2])
        self.upcat_2 = UpCat(f[2], f[1], f[1])
        self.upcat_1 = UpCat(f[1], f[0], f[5], halves=False)
        self.final_conv = nn.Conv3d(f[5], out_channels, kernel_size=1)
        object.__setattr__(self, "_rt", None)

    def forward(self, x: torch.Tensor, t: torch.Tensor, image: torch.Tensor = None, embeddings=None):
        _refuse_autograd(x, image)
        rt = self._rt
        assert rt is not None, "BasicUNetRDenoiser must be owned by a DiffUNet (shared launch plan)"
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
