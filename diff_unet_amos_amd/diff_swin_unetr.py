"""DiffSwinUNETR (models/diff_swin_unetr.py:7-47): SwinUNETREncoder + SwinUNETRDenoiser behind Diffusion -- BASELINE
config 5.  Same constructor arguments as the reference; ``pred_type="ddim_sample"`` / ``"denoise"`` under
``torch.no_grad()`` run the HIP launch plan (swin_engine.SwinPlan); the sampling loops replay one captured HIP graph per
reverse step (SwinPlan.sample_loop, reached through the denoiser's ``fused_engine`` hook like DiffUNet's)."""
from __future__ import annotations

from typing import Sequence

import torch

from .diff_unet import _Runtime
from .diffusion import Diffusion
from .swin_engine import SwinPlan
from .swin_unetr import SwinUNETRDenoiser, SwinUNETREncoder, _refuse_autograd


class DiffSwinUNETR(Diffusion):
    def __init__(self, spatial_dims: int = 3, in_channels: int = 1, out_channels: int = 1, image_size: int = 96,
                 spatial_size: int = 96, features: Sequence[int] = (64, 64, 128, 256, 512, 64), feature_size: int = 48,
                 noise_ratio: float = 0.5, dropout: float = 0.2, timesteps: int = 1000, mode: str = "train",
                 sample_steps: int = 10, compute_dtype: torch.dtype = torch.float16):
        super().__init__(spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels,
                         image_size=image_size, spatial_size=spatial_size, features=features, dropout=dropout,
                         timesteps=timesteps, mode=mode, sample_steps=sample_steps)
        self.feature_size = feature_size
        self.compute_dtype = compute_dtype
        self.embed_model = SwinUNETREncoder(image_size, in_channels, spatial_dims=spatial_dims, feature_size=feature_size,
                                            drop_rate=dropout)
        self.model = SwinUNETRDenoiser(image_size, out_channels + 1, out_channels, spatial_dims=spatial_dims,
                                       feature_size=feature_size, noise_ratio=noise_ratio, drop_rate=dropout)
        rt = _Runtime(self, SwinPlan)
        object.__setattr__(self, "_rt", rt)
        object.__setattr__(self.embed_model, "_rt", rt)
        object.__setattr__(self.model, "_rt", rt)

    def set_compute_dtype(self, dtype: torch.dtype):
        """torch.float16: fp16 operands / fp32 accumulate and an fp32 residual token stream (production);
        torch.float32: fp32 everywhere (parity mode)."""
        assert dtype in (torch.float16, torch.float32)
        self.compute_dtype = dtype
        return self

    def denoise(self, image, x, step):
        """diffusion.py:71-84 (inference arithmetic only; see swin_unetr.py)."""
        assert image.size(0) == x.size(0) == step.size(0)
        _refuse_autograd(x, image, *self.parameters())
        embeddings = self.embed_model(image)
        return self.model(x=x, t=step, embeddings=embeddings, image=image)
