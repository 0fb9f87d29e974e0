"""The oracle (oracle/*.py) against golden vectors produced by the reference's own
guided_diffusion package and time-embedding file (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle.diffusion_ref import RefDiffusion, kept_timesteps, uniform_timesteps
from oracle.unet_ref import RefDiffUNet, RefTimeStepEmbedder, sinusoid_embedding, swish

TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
          "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
          "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


def stub_half(x, t, **kw):
    return 0.5 * x


def stub_tanh(x, t, **kw):
    return torch.tanh(x) + 1e-3 * t.float().view(-1, *([1] * (x.dim() - 1)))


STUBS = {"half": stub_half, "tanh": stub_tanh}


@pytest.mark.parametrize("tag,sections", [("s10", [10]), ("s50", [50]), ("s1000", [1000])])
def test_schedule_tables_bit_exact(golden, tag, sections):
    d = RefDiffusion(1000, sections)
    assert np.array_equal(np.array(d.timestep_map), golden[f"G1_{tag}_timestep_map"])
    for name in TABLES:
        assert np.array_equal(getattr(d, name), golden[f"G1_{tag}_{name}"]), name


def test_space_timesteps_forms(golden):
    assert kept_timesteps(1000, "ddim25") == golden["G1_ddim25_kept"].tolist()
    assert kept_timesteps(300, [10, 15, 20]) == golden["G1_sections_10_15_20_of_300"].tolist()
    assert kept_timesteps(300, "10,15,20") == golden["G1_sections_10_15_20_of_300"].tolist()
    with pytest.raises(ValueError):
        kept_timesteps(10, [11])


def test_q_sample(golden):
    d = RefDiffusion(1000)
    x0, eps = torch.from_numpy(golden["G2_x0"]), torch.from_numpy(golden["G2_eps"])
    for tag in "ab":
        t = torch.from_numpy(golden[f"G2_t_{tag}"])
        assert np.array_equal(d.q_sample(x0, t, eps).numpy(), golden[f"G2_xt_{tag}"])


@pytest.mark.parametrize("dtag,sections,ts", [("s10", [10], [0, 1, 5, 9]), ("s1000", [1000], [0, 500, 999])])
@pytest.mark.parametrize("sname", ["half", "tanh"])
def test_single_reverse_steps(golden, dtag, sections, ts, sname):
    d = RefDiffusion(1000, sections)
    x, nz = torch.from_numpy(golden["G3_x"]), torch.from_numpy(golden["G3_noise"])
    fn = STUBS[sname]
    for ti in ts:
        t = torch.tensor([ti, ti])
        key = f"G3_{dtag}_{sname}_t{ti}"
        o = d.p_mean_variance(fn, x, t)
        for k in ("mean", "variance", "log_variance", "pred_xstart", "model_output"):
            assert np.array_equal(o[k].numpy(), golden[f"{key}_pmv_{k}"]), (key, k)
        assert np.array_equal(d.p_sample(fn, x, t, nz)["sample"].numpy(), golden[f"{key}_psample"])
        assert np.array_equal(d.ddim_sample(fn, x, t, nz)["sample"].numpy(), golden[f"{key}_ddim"])
        assert np.array_equal(d.ddim_sample(fn, x, t, nz, eta=0.7)["sample"].numpy(), golden[f"{key}_ddim_eta07"])


def test_mixed_timesteps(golden):
    d = RefDiffusion(1000, [10])
    x, nz = torch.from_numpy(golden["G3_x"]), torch.from_numpy(golden["G3_noise"])
    t = torch.tensor([3, 7])
    assert np.array_equal(d.p_sample(stub_tanh, x, t, nz)["sample"].numpy(), golden["G3_s10_tanh_tmixed_psample"])
    assert np.array_equal(d.ddim_sample(stub_tanh, x, t, nz)["sample"].numpy(), golden["G3_s10_tanh_tmixed_ddim"])


@pytest.mark.parametrize("sname", ["half", "tanh"])
def test_ten_step_loops(golden, sname):
    d = RefDiffusion(1000, [10])
    xT = torch.from_numpy(golden["G4_xT"])
    draws = list(torch.from_numpy(golden["G4_draws10"]))
    out = d.ddim_sample_loop(STUBS[sname], xT, draws)
    acc = torch.zeros_like(xT)
    for s in out["all_samples"]:
        acc += s
    assert np.array_equal(acc.numpy(), golden[f"G4_{sname}_ddim10_sum_xstart"])
    assert np.array_equal(out["sample"].numpy(), golden[f"G4_{sname}_ddim10_final"])
    assert np.array_equal(d.p_sample_loop(STUBS[sname], xT, draws).numpy(), golden[f"G4_{sname}_ddpm10_final"])


def test_thousand_step_ddpm(golden):
    d = RefDiffusion(1000)
    xT = torch.from_numpy(golden["G4_xT_small"])
    draws = list(torch.from_numpy(golden["G4_draws1000"]))
    assert np.array_equal(d.p_sample_loop(stub_tanh, xT, draws).numpy(), golden["G4_tanh_ddpm1000_final"])


def test_time_embedding(golden):
    t = torch.from_numpy(golden["G5_t"])
    assert np.array_equal(sinusoid_embedding(t, 128).numpy(), golden["G5_sinusoid128"])
    assert np.array_equal(sinusoid_embedding(t, 7).numpy(), golden["G5_sinusoid7"])
    assert np.array_equal(swish(torch.linspace(-6, 6, 25)).numpy(), golden["G5_swish"])
    te = RefTimeStepEmbedder()
    te.load_state_dict({k[len("G5_w_"):]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith("G5_w_")})
    with torch.no_grad():
        assert np.array_equal(te(t).numpy(), golden["G5_temb"])


def test_uniform_sampler(golden):
    rng = np.random.RandomState(99)
    idx, w = uniform_timesteps(1000, 8, rng)
    assert np.array_equal(idx.numpy(), golden["G7_uniform_seed99_idx"])
    assert np.array_equal(w.numpy(), golden["G7_uniform_seed99_w"])


def test_state_dict_keys_match_reference_layout():
    """the key layout reference checkpoints use."""
    net = RefDiffUNet(in_channels=1, out_channels=2, features=(8, 8, 16, 32, 64, 8))
    keys = set(net.state_dict().keys())
    want = set()
    for blk in ("embed_model.conv_0",) + tuple(f"embed_model.down.{i}.convs" for i in range(4)):
        for c in ("conv_0", "conv_1"):
            for leaf in ("conv", "adn.N"):
                want |= {f"{blk}.{c}.{leaf}.weight", f"{blk}.{c}.{leaf}.bias"}
    want |= {f"model.temb.dense.{i}.{p}" for i in (0, 1) for p in ("weight", "bias")}
    dn = ("model.conv_0",) + tuple(f"model.down_{i}.convs" for i in range(1, 5)) + \
        tuple(f"model.upcat_{i}.convs" for i in range(1, 5))
    for blk in dn:
        want |= {f"{blk}.temb_proj.weight", f"{blk}.temb_proj.bias"}
        for c in ("conv_0", "conv_1"):
            for leaf in ("conv", "adn.N"):
                want |= {f"{blk}.{c}.{leaf}.weight", f"{blk}.{c}.{leaf}.bias"}
    want |= {f"model.upcat_{i}.upsample.deconv.{p}" for i in range(1, 5) for p in ("weight", "bias")}
    want |= {"model.final_conv.weight", "model.final_conv.bias"}
    assert keys == want


def test_full_size_parameter_count():
    """24 131 280 + 14 274 240 parameters at C=16."""
    net = RefDiffUNet(in_channels=1, out_channels=16)
    n_den = sum(p.numel() for p in net.model.parameters())
    n_enc = sum(p.numel() for p in net.embed_model.parameters())
    assert (n_den, n_enc) == (24131280, 14274240)


def test_unet_restatement_selfcheck(unet_selfcheck):
    u = unet_selfcheck
    net = RefDiffUNet(in_channels=1, out_channels=2, features=(8, 8, 16, 32, 64, 8)).eval()
    net.load_state_dict({k[2:]: torch.from_numpy(u[k]) for k in u.files if k.startswith("w/")})
    with torch.no_grad():
        logits = net(image=torch.from_numpy(u["image"]), x=torch.from_numpy(u["x_t"]),
                     step=torch.from_numpy(u["t"]), pred_type="denoise")
    assert np.allclose(logits.numpy(), u["logits"], rtol=0, atol=2e-5)
    with pytest.raises(NotImplementedError):
        net(image=torch.from_numpy(u["image"]), pred_type="nope")
