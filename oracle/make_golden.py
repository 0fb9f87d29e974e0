"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build
container only; /root/reference does not exist on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What is imported from /root/reference (read-only, nothing is copied):
  * guided_diffusion.{gaussian_diffusion,respace,resample}  (needs numpy+torch only)
  * models/diffusion/utils.py, loaded by file path (its package __init__ pulls MONAI)
The fixtures are data only: inputs and the reference's outputs for them.

A second file, unet_selfcheck.npz, is produced by the oracle's own torch.nn
restatement of the MONAI-wired networks (MONAI is absent, so the reference
classes cannot be built here).  It pins the restatement against regressions;
it is NOT a reference pin and says so in its "provenance" field.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ref_modules():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion import resample, respace
    spec = importlib.util.spec_from_file_location("ref_temb_utils", os.path.join(REF, "models/diffusion/utils.py"))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    return gd, respace, resample, tu


def _spaced(gd, respace, T, sections):
    return respace.SpacedDiffusion(
        use_timesteps=respace.space_timesteps(T, sections),
        betas=gd.get_named_beta_schedule("linear", T),
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=gd.ModelVarType.FIXED_LARGE,
        loss_type=gd.LossType.RESCALED_KL,
    )


TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
          "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
          "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


def stub_half(x, t, **kw):
    return 0.5 * x


def stub_tanh(x, t, **kw):
    return torch.tanh(x) + 1e-3 * t.float().view(-1, *([1] * (x.dim() - 1)))


STUBS = {"half": stub_half, "tanh": stub_tanh}


class _Dummy(torch.nn.Module):
    """The reference loops ask the model for .parameters() to find a device."""

    def __init__(self, fn):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.fn = fn

    def forward(self, x, t, **kw):
        return self.fn(x, t, **kw)


class _InjectNoise:
    """Pin the per-step th.randn_like draws."""

    def __init__(self, draws):
        self.draws = list(draws)
        self.k = 0

    def __enter__(self):
        self._orig = torch.randn_like
        torch.randn_like = self._next
        return self

    def _next(self, x, *a, **k):
        d = self.draws[self.k]
        self.k += 1
        return d.clone()

    def __exit__(self, *a):
        torch.randn_like = self._orig


def main():
    os.makedirs(OUT, exist_ok=True)
    gd, respace, resample, tu = _ref_modules()
    g = {}

    # G1 -- schedule tables + timestep maps
    for tag, sections in (("s10", [10]), ("s50", [50]), ("s1000", [1000])):
        d = _spaced(gd, respace, 1000, sections)
        g[f"G1_{tag}_timestep_map"] = np.array(d.timestep_map, dtype=np.int64)
        for name in TABLES:
            g[f"G1_{tag}_{name}"] = np.asarray(getattr(d, name), dtype=np.float64)
    g["G1_ddim25_kept"] = np.array(sorted(respace.space_timesteps(1000, "ddim25")), dtype=np.int64)
    g["G1_sections_10_15_20_of_300"] = np.array(sorted(respace.space_timesteps(300, [10, 15, 20])), dtype=np.int64)

    d1000 = _spaced(gd, respace, 1000, [1000])
    d10 = _spaced(gd, respace, 1000, [10])
    gen = torch.Generator().manual_seed(1234)

    # G2 -- q_sample
    x0 = torch.rand(2, 2, 8, 8, 8, generator=gen) * 2 - 1
    eps = torch.randn(2, 2, 8, 8, 8, generator=gen)
    g["G2_x0"], g["G2_eps"] = x0.numpy(), eps.numpy()
    for tag, tt in (("a", [0, 999]), ("b", [500, 250])):
        t = torch.tensor(tt)
        g[f"G2_t_{tag}"] = t.numpy()
        g[f"G2_xt_{tag}"] = d1000.q_sample(x0, t, eps).numpy()

    # G3 -- single reverse steps, injected noise
    x = torch.randn(2, 3, 6, 6, 6, generator=gen)
    nz = torch.randn(2, 3, 6, 6, 6, generator=gen)
    g["G3_x"], g["G3_noise"] = x.numpy(), nz.numpy()
    for dtag, d, ts in (("s10", d10, [0, 1, 5, 9]), ("s1000", d1000, [0, 500, 999])):
        for sname, fn in STUBS.items():
            for ti in ts:
                t = torch.tensor([ti, ti])
                key = f"G3_{dtag}_{sname}_t{ti}"
                o = d.p_mean_variance(fn, x, t)
                for k in ("mean", "variance", "log_variance", "pred_xstart", "model_output"):
                    g[f"{key}_pmv_{k}"] = o[k].contiguous().numpy()
                with _InjectNoise([nz]):
                    o = d.p_sample(fn, x, t)
                g[f"{key}_psample"] = o["sample"].numpy()
                with _InjectNoise([nz]):
                    o = d.ddim_sample(fn, x, t)
                g[f"{key}_ddim"] = o["sample"].numpy()
                with _InjectNoise([nz]):
                    o = d.ddim_sample(fn, x, t, eta=0.7)
                g[f"{key}_ddim_eta07"] = o["sample"].numpy()
    # mixed per-sample timesteps in one batch
    t = torch.tensor([3, 7])
    with _InjectNoise([nz]):
        g["G3_s10_tanh_tmixed_psample"] = d10.p_sample(stub_tanh, x, t)["sample"].numpy()
    with _InjectNoise([nz]):
        g["G3_s10_tanh_tmixed_ddim"] = d10.ddim_sample(stub_tanh, x, t)["sample"].numpy()

    # G4 -- whole loops with a stub model
    shape = (1, 2, 6, 6, 6)
    xT = torch.randn(*shape, generator=gen)
    draws10 = [torch.randn(*shape, generator=gen) for _ in range(10)]
    g["G4_xT"] = xT.numpy()
    g["G4_draws10"] = torch.stack(draws10).numpy()
    for sname, fn in STUBS.items():
        m = _Dummy(fn)
        with _InjectNoise(draws10):
            out = d10.ddim_sample_loop(m, shape, noise=xT)
        acc = torch.zeros(shape)
        for s in out["all_samples"]:          # models/diffusion/diffusion.py:94-98
            acc += s
        g[f"G4_{sname}_ddim10_sum_xstart"] = acc.numpy()
        g[f"G4_{sname}_ddim10_final"] = out["sample"].numpy()
        with _InjectNoise(draws10):
            g[f"G4_{sname}_ddpm10_final"] = d10.p_sample_loop(m, shape, noise=xT).numpy()
    shape_s = (1, 1, 4, 4, 4)
    xTs = torch.randn(*shape_s, generator=gen)
    draws1000 = [torch.randn(*shape_s, generator=gen) for _ in range(1000)]
    g["G4_xT_small"] = xTs.numpy()
    g["G4_draws1000"] = torch.stack(draws1000).numpy()
    with _InjectNoise(draws1000):
        g["G4_tanh_ddpm1000_final"] = d1000.p_sample_loop(_Dummy(stub_tanh), shape_s, noise=xTs).numpy()

    # G5 -- time embedding
    tt = torch.tensor([0, 1, 111, 999])
    g["G5_t"] = tt.numpy()
    g["G5_sinusoid128"] = tu.get_timestep_embedding(tt, 128).numpy()
    g["G5_sinusoid7"] = tu.get_timestep_embedding(tt, 7).numpy()
    torch.manual_seed(7)
    te = tu.TimeStepEmbedder()
    for k, v in te.state_dict().items():
        g[f"G5_w_{k}"] = v.numpy()
    with torch.no_grad():
        g["G5_temb"] = te(tt).numpy()
        g["G5_swish"] = tu.nonlinearity(torch.linspace(-6, 6, 25)).numpy()

    # G7a -- UniformSampler under a seeded numpy global RNG
    np.random.seed(99)
    s = resample.UniformSampler(1000)
    idx, w = s.sample(8, "cpu")
    g["G7_uniform_seed99_idx"], g["G7_uniform_seed99_w"] = idx.numpy(), w.numpy()

    np.savez_compressed(os.path.join(OUT, "diffusion_golden.npz"), **g)
    print("diffusion_golden.npz:", len(g), "arrays")

    # ---- restatement self-check fixture (NOT a reference pin) ----------------
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from oracle.unet_ref import RefDiffUNet
    torch.manual_seed(0)
    net = RefDiffUNet(in_channels=1, out_channels=2, features=(8, 8, 16, 32, 64, 8)).eval()
    # non-trivial affine params so gamma/beta wiring is exercised
    with torch.no_grad():
        for n, p in net.named_parameters():
            if ".adn.N." in n:
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if n.endswith("weight") else 0.0))
    gen = torch.Generator().manual_seed(5)
    image = torch.rand(1, 1, 32, 32, 32, generator=gen)
    xt = torch.randn(1, 2, 32, 32, 32, generator=gen)
    t = torch.tensor([377])
    u = {"provenance": np.array("oracle self-check (torch.nn restatement; MONAI absent => not a reference pin)")}
    for k, v in net.state_dict().items():
        u["w/" + k] = v.numpy()
    with torch.no_grad():
        emb = net.embed_model(image)
        logits = net(image=image, x=xt, step=t, pred_type="denoise")
    u["image"], u["x_t"], u["t"] = image.numpy(), xt.numpy(), t.numpy()
    for i, e in enumerate(emb):
        u[f"emb{i}"] = e.numpy().astype(np.float16)     # coarse pin, keeps the file small
    u["logits"] = logits.numpy()
    np.savez_compressed(os.path.join(OUT, "unet_selfcheck.npz"), **u)
    print("unet_selfcheck.npz:", len(u), "arrays")


if __name__ == "__main__":
    main()
