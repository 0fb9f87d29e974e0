"""DiffUNet (models/diff_unet.py:9-35): BasicUNetEncoder + BasicUNetRDenoiser behind Diffusion."""
from __future__ import annotations

from typing import Sequence

import torch

from .basic_unet import BasicUNetEncoder, BasicUNetRDenoiser
from .diffusion import Diffusion
from .engine import Plan


class _FusedAdapter:
    """What GaussianDiffusion loops talk to when the model is the HIP denoiser."""

    def __init__(self, plan):
        self.plan = plan

    def sample_loop(self, diffusion, kind, shape, noise=None, model_kwargs=None, eta=0.0, step_noise=None):
        assert tuple(shape) == (self.plan.N, self.plan.C, *self.plan.dims)
        out = self.plan.sample_loop(diffusion, kind, noise=noise, eta=eta, step_noise=step_noise)
        # One tensor that already is the sum over steps: summing ``all_samples`` as
        # models/diffusion/diffusion.py:94-98 does gives the reference's result without keeping T tensors.
        if out["sum_pred_xstart"] is not None:       # DDIM loops; a DDPM loop returns its final sample alone, like the reference's
            out["all_samples"] = [out["sum_pred_xstart"]]
        return out


class _Runtime:
    """Launch plans of one network pair, keyed by (batch, patch shape, device, dtype)."""

    def __init__(self, net, plan_cls=Plan):
        self.net = net
        self.plan_cls = plan_cls
        self.plans = {}

    def plan(self, N, dims, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"{type(self.net).__name__} runs on an MI355X (device 'cuda'); there is no CPU path in this package")
        key = (N, tuple(dims), device.index, self.net.compute_dtype)
        p = self.plans.get(key)
        if p is None:
            p = self.plan_cls(self.net, N, *dims, self.net.compute_dtype, device)
            self.plans[key] = p
        return p

    def plan_for(self, x):
        return self.plan(x.shape[0], tuple(x.shape[2:]), x.device)

    @staticmethod
    def adapter(plan):
        return _FusedAdapter(plan)


class DiffUNet(Diffusion):
    def __init__(self, spatial_dims: int = 3, in_channels: int = 3, out_channels: int = 1, image_size: int = 96,
                 spatial_size: int = 96, features: Sequence[int] = (64, 64, 128, 256, 512, 64), dropout: float = 0.2,
                 timesteps: int = 1000, mode: str = "train", sample_steps: int = 10,
                 compute_dtype: torch.dtype = torch.float16):
        super().__init__(spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels,
                         image_size=image_size, spatial_size=spatial_size, features=features, dropout=dropout,
                         timesteps=timesteps, mode=mode, sample_steps=sample_steps)
        self.features = tuple(features)
        self.compute_dtype = compute_dtype
        self.embed_model = BasicUNetEncoder(3, in_channels, 2, features)
        self.model = BasicUNetRDenoiser(3, out_channels + 1, out_channels, features)
        rt = _Runtime(self)
        object.__setattr__(self, "_rt", rt)
        object.__setattr__(self.embed_model, "_rt", rt)
        object.__setattr__(self.model, "_rt", rt)

    def set_compute_dtype(self, dtype: torch.dtype):
        """torch.float16: fp16 operands / fp32 accumulate (production, the reference's AMP envelope);
        torch.float32: exact-fp32 MFMA (parity mode)."""
        assert dtype in (torch.float16, torch.float32)
        self.compute_dtype = dtype
        return self
