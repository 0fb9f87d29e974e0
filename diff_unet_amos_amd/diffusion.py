"""``Diffusion``: the drop-in boundary of the hot path.

Same constructor, attributes and ``forward(image, x, step, pred_type)`` dispatch as the reference's
models/diffusion/diffusion.py:11-102; the three branches run on HIP kernels.  ``pred_type="denoise"`` under autograd
(what Trainer.training_step calls, train.py:258-268) runs the same kernels forward and their backward kernels through
``training.native_conv_denoise``; without grad it runs the inference launch plan (engine.py).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .basic_unet import _wants_grad
from .gaussian_diffusion import UniformSampler, make_spaced


class Diffusion(nn.Module):
    def __init__(self, spatial_dims: int = 3, in_channels: int = 3, out_channels: int = 1, image_size: int = 96,
                 spatial_size: int = 96, features: Sequence[int] = (32, 64, 128, 256, 512), dropout: float = 0.2,
                 timesteps: int = 1000, mode: str = "train", sample_steps: int = 10):
        super().__init__()
        self.num_classes = out_channels
        self.mode = mode
        self.timesteps = timesteps
        self.embed_model: nn.Module = None
        self.model: nn.Module = None
        # diffusion.py:31-45: a full process for training and a respaced one for inference.
        # ``sample_steps`` (10 in the reference) is the only added knob: BASELINE configs use 10/50/1000.
        self.diffusion = make_spaced(timesteps, [timesteps])
        self.sample_diffusion = make_spaced(timesteps, [sample_steps])
        self.sampler = UniformSampler(timesteps)

    def forward(self, image: torch.Tensor = None, x: torch.Tensor = None, step: torch.Tensor = None,
                pred_type: str = None):
        if image is not None and x is not None:
            assert image.device == x.device
        if pred_type == "q_sample":
            return self.q_sample(x)
        if pred_type == "denoise":
            return self.denoise(image, x, step)
        if pred_type == "ddim_sample":
            return self.ddim_sample(image)
        raise NotImplementedError(f"No such prediction type : {pred_type}")

    def q_sample(self, x: torch.Tensor):
        """diffusion.py:65-69: eps ~ N(0,1) (torch RNG), t ~ U{0..T-1} (numpy global RNG)."""
        noise = torch.randn_like(x)
        t, _ = self.sampler.sample(x.shape[0], x.device)
        return self.diffusion.q_sample(x, t, noise), t, noise

    def denoise(self, image: torch.Tensor, x: torch.Tensor, step: torch.Tensor) -> torch.Tensor:
        """diffusion.py:71-84."""
        assert image.size(0) == x.size(0) == step.size(0)
        if _wants_grad(x, image, *self.parameters()):
            from .training import native_conv_denoise     # HIP forward + backward kernels under autograd
            return native_conv_denoise(self, image, x, step, self.compute_dtype)
        embeddings = self.embed_model(image)
        return self.model(x=x, t=step, embeddings=embeddings, image=image)

    def ddim_sample(self, image: torch.Tensor) -> torch.Tensor:
        """diffusion.py:86-102: per window, encoder once, DDIM loop, sum of the clamped x0 predictions.

        The reference walks the batch one sample at a time; every sample is independent (own encoder pass, own
        x_T), so the HIP path runs the whole batch through one launch plan -- the small U-Net levels, which
        cannot fill 256 CUs with one 96^3 patch, get B times the workgroups.  ``batched_sampling = False``
        restores the per-sample loop."""
        shape1 = (1, self.num_classes, *image.shape[2:])
        with torch.no_grad():
            if getattr(self, "batched_sampling", True) and len(image) > 1:
                embeddings = self.embed_model(image)
                out = self.sample_diffusion.ddim_sample_loop(
                    self.model, (len(image), *shape1[1:]), model_kwargs={"image": image, "embeddings": embeddings})
                acc = torch.zeros((len(image), *shape1[1:]), device=image.device)
                for s in out["all_samples"]:
                    acc += s.to(image.device)
                return acc
            res = []
            for i in range(len(image)):
                batch = image[i, ...].unsqueeze(0)
                embeddings = self.embed_model(batch)
                out = self.sample_diffusion.ddim_sample_loop(self.model, shape1,
                                                             model_kwargs={"image": batch, "embeddings": embeddings})
                acc = torch.zeros(shape1, device=image.device)
                for s in out["all_samples"]:
                    acc += s.to(image.device)
                res.append(acc)
        return torch.cat(res, dim=0)
