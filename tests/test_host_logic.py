"""CPU checks of the product's host logic (no GPU): schedule tables, respacing and coefficient
rows against the goldens generated from the reference; module tree / state-dict layout; errors."""
import numpy as np
import pytest
import torch

from diff_unet_amos_amd.gaussian_diffusion import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,
                                                   UniformSampler, get_named_beta_schedule, make_spaced,
                                                   space_timesteps)

TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
          "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
          "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


@pytest.mark.parametrize("tag,sections", [("s10", [10]), ("s50", [50]), ("s1000", [1000])])
def test_tables_bit_exact_vs_reference(golden, tag, sections):
    d = make_spaced(1000, sections)
    assert d.timestep_map == golden[f"G1_{tag}_timestep_map"].tolist()
    for name in TABLES:
        assert np.array_equal(getattr(d, name), golden[f"G1_{tag}_{name}"]), name


def test_space_timesteps(golden):
    assert sorted(space_timesteps(1000, "ddim25")) == golden["G1_ddim25_kept"].tolist()
    assert sorted(space_timesteps(300, [10, 15, 20])) == golden["G1_sections_10_15_20_of_300"].tolist()
    with pytest.raises(ValueError):
        space_timesteps(10, [11])
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(NotImplementedError):
        get_named_beta_schedule("nope", 10)


def test_coefficient_rows_reproduce_reference_steps(golden):
    """Apply the coefficient rows with plain torch fp32 ops: must equal the reference's outputs bit for bit."""
    x, nz = torch.from_numpy(golden["G3_x"]), torch.from_numpy(golden["G3_noise"])
    for dtag, sections, ts in (("s10", [10], [0, 1, 5, 9]), ("s1000", [1000], [0, 500, 999])):
        d = make_spaced(1000, sections)
        for ti in ts:
            key = f"G3_{dtag}_tanh_t{ti}"
            xs = torch.from_numpy(golden[f"{key}_pmv_pred_xstart"])
            t = torch.tensor([ti, ti])
            k = d.ddpm_coef(t)[:, :, None, None, None, None]
            got = (k[:, 0] * xs + k[:, 1] * x) + k[:, 2] * nz
            assert np.array_equal(got.numpy(), golden[f"{key}_psample"]), key
            for eta, suffix in ((0.0, "ddim"), (0.7, "ddim_eta07")):
                k = d.ddim_coef(t, eta)[:, :, None, None, None, None]
                e = (k[:, 0] * x - xs) / k[:, 1]
                got = (xs * k[:, 2] + k[:, 3] * e) + k[:, 4] * nz
                assert np.array_equal(got.numpy(), golden[f"{key}_{suffix}"]), (key, suffix)
    q = make_spaced(1000, [1000]).q_coef(torch.from_numpy(golden["G2_t_a"]))[:, :, None, None, None, None]
    x0, eps = torch.from_numpy(golden["G2_x0"]), torch.from_numpy(golden["G2_eps"])
    assert np.array_equal((q[:, 0] * x0 + q[:, 1] * eps).numpy(), golden["G2_xt_a"])


def test_uniform_sampler_uses_numpy_global_rng(golden):
    np.random.seed(99)
    idx, w = UniformSampler(1000).sample(8, "cpu")
    assert np.array_equal(idx.numpy(), golden["G7_uniform_seed99_idx"])
    assert np.array_equal(w.numpy(), golden["G7_uniform_seed99_w"])


def test_unsupported_process_types_are_refused():
    b = get_named_beta_schedule("linear", 10)
    with pytest.raises(NotImplementedError):
        GaussianDiffusion(betas=b, model_mean_type=ModelMeanType.EPSILON, model_var_type=ModelVarType.FIXED_LARGE,
                          loss_type=LossType.MSE)
    with pytest.raises(NotImplementedError):
        GaussianDiffusion(betas=b, model_mean_type=ModelMeanType.START_X, model_var_type=ModelVarType.LEARNED,
                          loss_type=LossType.MSE)


def test_module_tree_matches_reference_state_dict_layout():
    from diff_unet_amos_amd.diff_unet import DiffUNet
    from oracle.unet_ref import RefDiffUNet
    kw = dict(in_channels=1, out_channels=2, features=(8, 8, 16, 32, 64, 8))
    net, ref = DiffUNet(**kw), RefDiffUNet(**kw)
    sd, rsd = net.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys()) or set(sd.keys()) == set(rsd.keys())
    for k in sd:
        assert sd[k].shape == rsd[k].shape, k
    ref.load_state_dict(sd)                     # checkpoints move both ways
    net.load_state_dict(ref.state_dict())
    assert net.num_classes == 2 and net.sample_diffusion.num_timesteps == 10 and net.diffusion.num_timesteps == 1000
    assert hasattr(net, "sampler") and hasattr(net, "embed_model") and hasattr(net, "model")
    n_full = DiffUNet(in_channels=1, out_channels=16)
    assert sum(p.numel() for p in n_full.parameters()) == 38405520


def test_forward_dispatch_errors_and_no_cpu_fallback():
    from diff_unet_amos_amd.diff_unet import DiffUNet
    net = DiffUNet(in_channels=1, out_channels=2, features=(8, 8, 16, 32, 64, 8))
    with pytest.raises(NotImplementedError, match="No such prediction type"):
        net(image=torch.zeros(1, 1, 32, 32, 32), pred_type="sample")
    with pytest.raises(AssertionError):
        net(image=torch.zeros(2, 1, 32, 32, 32), x=torch.zeros(1, 2, 32, 32, 32), step=torch.zeros(1).long(), pred_type="denoise")
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU path"):
            net(image=torch.zeros(1, 1, 32, 32, 32), x=torch.zeros(1, 2, 32, 32, 32), step=torch.zeros(1).long(), pred_type="denoise")
    # with grad enabled the same call is the training path (HIP forward + backward kernels): no CPU path there either
    x = torch.zeros(1, 2, 32, 32, 32, requires_grad=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(image=torch.zeros(1, 1, 32, 32, 32), x=x, step=torch.zeros(1).long(), pred_type="denoise")
    # the sub-networks on their own run the inference launch plan, which keeps no tape: gradients are refused there
    with pytest.raises(NotImplementedError, match="pred_type"):
        net.model(x, torch.zeros(1).long(), image=torch.zeros(1, 1, 32, 32, 32), embeddings=[None] * 5)


def test_patch_embed_weight_packing_is_the_strided_convolution():
    """ops.pack_patch_embed_weights: [E, Cin, 2, 2, 2] -> [8 taps (kd, kh, kw), cin_packed, E] with a channel permutation; a
    plain einsum over the packed form on gathered 2x2x2 patches equals Conv3d(k = s = 2) (MONAI PatchEmbed)."""
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(0)
    E, cin, cp = 48, 17, 24
    w = torch.randn(E, cin, 2, 2, 2, generator=g)
    x = torch.randn(1, cin, 4, 6, 8, generator=g)
    perm = list(range(1, cin)) + [0]                          # packed channel p holds source channel perm[p]
    wp = ops.pack_patch_embed_weights(w, cp, perm)
    assert tuple(wp.shape) == (8, cp, E) and float(wp[:, cin:].abs().max()) == 0.0
    xp = torch.zeros(1, cp, 4, 6, 8)
    xp[:, :cin] = x[:, perm]
    patches = xp.unfold(2, 2, 2).unfold(3, 2, 2).unfold(4, 2, 2)          # [1, cp, 2, 3, 4, kd, kh, kw]
    got = torch.einsum("ncdhwijk,ijkce->nedhw", patches, wp.view(2, 2, 2, cp, E))
    want = F.conv3d(x, w, stride=2)
    assert torch.allclose(got, want, atol=1e-4)
