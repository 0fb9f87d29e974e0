"""Parity of the individual HIP kernels (through the C ABI) against torch CPU fp32 ops --
the arithmetic the reference delegates to torch.nn."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# tolerances: fp32 path = exact-fp32 MFMA (fmaf chain) vs CPU fp32 -> summation-order noise only;
# fp16 path = fp16 operands (2^-11 relative rounding) with fp32 accumulation.
TOL = {torch.float32: dict(rtol=1e-4, atol=1e-4), torch.float16: dict(rtol=2e-2, atol=2e-2)}


def _ops():
    from diff_unet_amos_amd import ops
    return ops


def _cl(x, dtype, cstride=None):
    """NCDHW cpu fp32 -> channels-last device tensor through the HIP layout kernel."""
    ops = _ops()
    N, C, D, H, W = x.shape
    cs = cstride or -(-C // 8) * 8
    dst = torch.full((N, D, H, W, cs), 7.0, dtype=dtype, device="cuda")
    ops.to_channels_last(x.cuda().contiguous(), dst, 0, cs)
    return dst


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_layout_roundtrip(dtype):
    ops = _ops()
    x = torch.randn(2, 5, 4, 6, 8)
    cl = _cl(x, dtype)
    assert cl.shape[-1] == 8
    back = ops.from_channels_last(cl, 5).cpu()
    assert torch.allclose(back, x.to(dtype).float(), rtol=0, atol=0)
    assert float(cl[..., 5:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [
    # N, Cin, Cout, D, H, W
    (1, 16, 64, 8, 8, 8),
    (2, 40, 72, 6, 10, 12),      # ragged everything: partial tiles, cin/cout not multiples of the chunk/tile
    (1, 8, 8, 4, 8, 8),
    (1, 64, 128, 12, 12, 12),
    (1, 8, 16, 2, 2, 2),         # bottom level of the 32^3 config
])
def test_conv3_raw_and_stats(dtype, shape):
    ops = _ops()
    N, Cin, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    xq, wq = x.to(dtype).float(), w.to(dtype).float()          # operands as the kernel sees them
    ref = F.conv3d(xq, wq, b, padding=1)

    xcl = _cl(x, dtype)
    wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), dtype)
    y = torch.zeros((N, D, H, W, Cout), dtype=dtype, device="cuda")
    rows = ops.conv3_rows(D, H, W)
    cpad = -(-Cout // 64) * 64
    partials = torch.zeros(N * rows * cpad * 2, device="cuda")
    counts = torch.zeros(rows, device="cuda")
    ops.conv3d_k3(xcl, Cin, 0, wp, bp, Cout, y, 0, partials, counts)
    got = ops.from_channels_last(y, Cout).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())

    # statistics -> scale/shift, against instance_norm of the kernel's own (rounded) output
    gamma = torch.rand(Cout, generator=g) + 0.5
    beta = torch.randn(Cout, generator=g)
    scale = torch.zeros(N * Cout, device="cuda"); shift = torch.zeros(N * Cout, device="cuda")
    ops.instnorm_finalize(N, Cout, rows, cpad, partials, counts, gamma.cuda(), beta.cuda(), scale, shift)
    assert float(counts.sum()) == D * H * W
    var, mean = torch.var_mean(got.double(), dim=(2, 3, 4), unbiased=False)
    sc_ref = gamma.double()[None] / torch.sqrt(var + 1e-5)
    sh_ref = beta.double()[None] - mean * sc_ref
    assert torch.allclose(scale.cpu().view(N, Cout).double(), sc_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(shift.cpu().view(N, Cout).double(), sh_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_conv3_fused_input_transform_and_channel_slices(dtype):
    """Producer IN+LeakyReLU+temb add fused into the consumer's halo staging; input read from and
    output written to channel slices of wider buffers (the concat-in-place layout)."""
    ops = _ops()
    N, Cin, Cout, D, H, W = 2, 24, 40, 8, 8, 16
    g = torch.Generator().manual_seed(11)
    raw = torch.randn(N, Cin, D, H, W, generator=g) * 2 + 0.5
    scale = torch.rand(N, Cin, generator=g) + 0.5
    shift = torch.randn(N, Cin, generator=g)
    add = torch.randn(N, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    rq = raw.to(dtype).float()
    act = F.leaky_relu(rq * scale[:, :, None, None, None] + shift[:, :, None, None, None], 0.1) + add[:, :, None, None, None]
    ref = F.conv3d(act.to(dtype).float(), w.to(dtype).float(), b, padding=1)

    xbuf = torch.full((N, D, H, W, 48), 3.0, dtype=dtype, device="cuda")
    ops.to_channels_last(raw.cuda(), xbuf, 16, Cin)
    ybuf = torch.full((N, D, H, W, 64), -5.0, dtype=dtype, device="cuda")
    wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), dtype)
    rows = ops.conv3_rows(D, H, W)
    partials = torch.zeros(N * rows * 64 * 2, device="cuda"); counts = torch.zeros(rows, device="cuda")
    ops.conv3d_k3(xbuf, Cin, 16, wp, bp, Cout, ybuf, 8, partials, counts,
                  in_scale=scale.cuda().contiguous(), in_shift=shift.cuda().contiguous(), in_add=add.cuda().contiguous())
    got = ops.from_channels_last(ybuf, Cout, 8).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())
    assert float((ybuf[..., :8].float() + 5).abs().max()) == 0 and float((ybuf[..., 48:].float() + 5).abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_conv3_permuted_padded_input_channels(dtype):
    """First denoiser layer: torch.cat([image, x_t]) (denoiser.py:298) stored as [x_t | image | 0-pad]."""
    ops = _ops()
    C = 5
    g = torch.Generator().manual_seed(3)
    image = torch.rand(1, 1, 8, 8, 8, generator=g); xt = torch.randn(1, C, 8, 8, 8, generator=g)
    w = torch.randn(16, C + 1, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(16, generator=g)
    ref = F.conv3d(torch.cat([image, xt], 1).to(dtype).float(), w.to(dtype).float(), b, padding=1)
    buf = torch.zeros((1, 8, 8, 8, 8), dtype=dtype, device="cuda")
    ops.to_channels_last(xt.cuda(), buf, 0, C)
    bufv = buf.view(-1, 8); bufv[:, C] = image.cuda().view(-1).to(dtype)
    perm = list(range(1, C + 1)) + [0]
    wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), dtype, cin_packed=8, perm=perm + [-1] * (8 - len(perm)))
    y = torch.zeros((1, 8, 8, 8, 16), dtype=dtype, device="cuda")
    rows = ops.conv3_rows(8, 8, 8)
    partials = torch.zeros(rows * 64 * 2, device="cuda"); counts = torch.zeros(rows, device="cuda")
    ops.conv3d_k3(buf, 8, 0, wp, bp, 16, y, 0, partials, counts)
    got = ops.from_channels_last(y, 16).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("with_emb", [False, True])
def test_materialize(dtype, pool, with_emb):
    ops = _ops()
    N, C, D, H, W = 2, 24, 4, 6, 8
    g = torch.Generator().manual_seed(5)
    raw = torch.randn(N, C, D, H, W, generator=g)
    emb = torch.randn(N, C, D, H, W, generator=g)
    scale = torch.rand(N, C, generator=g) + 0.5; shift = torch.randn(N, C, generator=g)
    y = F.leaky_relu(raw.to(dtype).float() * scale[:, :, None, None, None] + shift[:, :, None, None, None], 0.1)
    if with_emb:
        y = y + emb.to(dtype).float()
    y = y.to(dtype).float()
    out = torch.zeros((N, D, H, W, 40), dtype=dtype, device="cuda")
    pooled = torch.zeros((N, D // 2, H // 2, W // 2, C), dtype=dtype, device="cuda") if pool else None
    ops.materialize(_cl(raw, dtype), C, scale.cuda().contiguous(), shift.cuda().contiguous(), out, 8,
                    emb=_cl(emb, dtype) if with_emb else None, pooled=pooled)
    got = ops.from_channels_last(out, C, 8).cpu()
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)
    assert torch.allclose(got, y, **tol)
    if pool:
        gp = ops.from_channels_last(pooled, C).cpu()
        assert torch.equal(gp, F.max_pool3d(got, 2))
