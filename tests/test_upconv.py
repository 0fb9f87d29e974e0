"""UpCat's first convolution with the transposed convolution folded in (csrc/upconv.hip, dua_upconv_k3_fwd).

Reference semantics: models/basic_unet/denoiser.py:172-194 -- x_0 = ConvTranspose3d(k2, s2)(x); y = Conv3d(k3, p1)(cat([x_e, x_0])).
CPU: the regrouping itself (8 parents x composed weights per output parity + 27 border-class biases) is checked against
torch's conv_transpose3d -> cat -> conv3d in float64, with no kernel involved.  GPU: the packer against that composition,
the kernel against torch on the fp16-rounded operands."""
import itertools

import pytest
import torch
import torch.nn.functional as F

TOL16 = dict(rtol=2e-2, atol=2e-2)


def _taps(phi, delta):
    """per dimension: the (conv tap k, deconv child a) pairs whose input voxel o + k - 1 (o = 2 m + phi) is child a of parent
    m + delta - 1 + phi"""
    return {(0, 0): [(0, 1)], (0, 1): [(1, 0), (2, 1)], (1, 0): [(0, 0), (1, 1)], (1, 1): [(2, 0)]}[(phi, delta)]


def compose(wc_up, wd):
    """wc_up [Cout, Cmid, 3,3,3], wd [Cu, Cmid, 2,2,2] -> W'[pd,ph,pw, dd,dh,dw][Cout, Cu] (float64)"""
    out = {}
    for phi in itertools.product(range(2), repeat=3):
        for delta in itertools.product(range(2), repeat=3):
            acc = 0
            for (kd, ad), (kh, ah), (kw, aw) in itertools.product(_taps(phi[0], delta[0]), _taps(phi[1], delta[1]), _taps(phi[2], delta[2])):
                acc = acc + wc_up[:, :, kd, kh, kw].double() @ wd[:, :, ad, ah, aw].double().t()
            out[phi + delta] = acc
    return out


def bias_table(wc_up, bc, bd):
    """[27][Cout]: class (cd, ch, cw), 0 = the voxel lies on the low border (tap 0 falls outside), 2 = on the high border"""
    rows = []
    for cd, ch, cw in itertools.product(range(3), repeat=3):
        v = bc.double().clone()
        for kd, kh, kw in itertools.product(range(3), repeat=3):
            if any((c == 0 and k == 0) or (c == 2 and k == 2) for c, k in ((cd, kd), (ch, kh), (cw, kw))):
                continue
            v = v + wc_up[:, :, kd, kh, kw].double() @ bd.double()
        rows.append(v)
    return torch.stack(rows)


def composed_forward(xs, u, wc, bc, wd, bd):
    """The regrouped evaluation, written with plain indexing (float64): what the kernel computes."""
    N, Cs = xs.shape[:2]
    Cu, Cmid = wd.shape[:2]
    D, H, W = xs.shape[2:]
    out = F.conv3d(xs.double(), wc[:, :Cs].double(), None, padding=1)
    Wp = compose(wc[:, Cs:], wd)
    bt = bias_table(wc[:, Cs:], bc, bd)
    up = torch.zeros_like(out)
    upad = F.pad(u.double(), (1, 1, 1, 1, 1, 1))                    # parent cells outside the volume contribute nothing
    for phi in itertools.product(range(2), repeat=3):
        for delta in itertools.product(range(2), repeat=3):
            s = [d - 1 + p for d, p in zip(delta, phi)]             # parent offset in cells
            src = upad[:, :, 1 + s[0]:1 + s[0] + D // 2, 1 + s[1]:1 + s[1] + H // 2, 1 + s[2]:1 + s[2] + W // 2]
            up[:, :, phi[0]::2, phi[1]::2, phi[2]::2] += torch.einsum("oc,ncdhw->nodhw", Wp[phi + delta], src)
    cls = lambda n: torch.tensor([0] + [1] * (n - 2) + [2])          # noqa: E731
    idx = (cls(D)[:, None, None] * 3 + cls(H)[None, :, None]) * 3 + cls(W)[None, None, :]
    return out + up + bt[idx].permute(3, 0, 1, 2)[None]


@pytest.mark.parametrize("shape", [(1, 8, 8, 8, 8, 8, 8, 8), (2, 4, 12, 6, 10, 8, 12, 16)])
def test_regrouping_equals_deconv_cat_conv(shape):
    N, Cs, Cu, Cmid, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    xs = torch.randn(N, Cs, D, H, W, generator=g)
    u = torch.randn(N, Cu, D // 2, H // 2, W // 2, generator=g)
    wc = torch.randn(Cout, Cs + Cmid, 3, 3, 3, generator=g)
    bc = torch.randn(Cout, generator=g)
    wd = torch.randn(Cu, Cmid, 2, 2, 2, generator=g)
    bd = torch.randn(Cmid, generator=g)
    want = F.conv3d(torch.cat([xs.double(), F.conv_transpose3d(u.double(), wd.double(), bd.double(), stride=2)], 1), wc.double(),
                    bc.double(), padding=1)
    got = composed_forward(xs, u, wc, bc, wd, bd)
    assert torch.allclose(got, want, rtol=1e-10, atol=1e-10), float((got - want).abs().max())


def _ops():
    from diff_unet_amos_amd import ops
    return ops


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 64, 64), (72, 32, 24, 128)])
def test_packer_writes_the_composed_weights_in_streaming_order(shape):
    Cout, Cs, Cmid, Cu = shape
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape))
    wc = torch.randn(Cout, Cs + Cmid, 3, 3, 3, generator=g) / (27 * (Cs + Cmid)) ** 0.5
    bc = torch.randn(Cout, generator=g)
    wd = torch.randn(Cu, Cmid, 2, 2, 2, generator=g) / Cmid ** 0.5
    bd = torch.randn(Cmid, generator=g)
    _, wu, btab = ops.pack_upconv_weights(wc.cuda(), bc.cuda(), wd.cuda(), bd.cuda(), Cs)
    nct, G = -(-Cout // 64), Cu // 64
    # [cout tile][wave = (pd, ph)][g][dd][dh][hcl][pw][dw][q][hh][r][e]
    t = wu.view(torch.float16).view(nct, 2, 2, G, 2, 2, 4, 2, 2, 2, 2, 32, 8).float().cpu()
    Wp = compose(wc[:, Cs:], wd)
    for (pd, ph, pw, dd, dh, dw), M in Wp.items():
        # M [Cout, Cu] -> [ct][q][r] x [g][hcl][hh][e]
        Mp = torch.zeros(nct * 64, Cu, dtype=torch.float64)
        Mp[:Cout] = M
        want = Mp.view(nct, 2, 32, G, 4, 2, 8).permute(0, 3, 4, 1, 5, 2, 6)       # [ct][g][hcl][q][hh][r][e]
        got = t[:, pd, ph, :, dd, dh, :, pw, dw]                                    # [ct][g][hcl][q][hh][r][e]
        assert torch.allclose(got.double(), want, rtol=2e-3, atol=2e-3), (pd, ph, pw, dd, dh, dw)
    want_b = torch.zeros(27, nct * 64, dtype=torch.float64)
    want_b[:, :Cout] = bias_table(wc[:, Cs:], bc, bd)
    assert torch.allclose(btab.cpu().double(), want_b, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [(False, False), (True, True)])
@pytest.mark.parametrize("shape", [
    # N, Cs, Cu, Cmid, Cout, D, H, W (output extents), skip offset / stride, coarse offset / stride, output offset / stride
    (1, 64, 64, 64, 64, 32, 32, 32, 0, 128, 0, 64, 0, 64),       # upcat_1 of the denoiser at a quarter of its size: every border class
    (2, 32, 128, 64, 72, 16, 24, 40, 32, 64, 64, 192, 16, 96),   # two samples, two coarse groups, two cout tiles (the second 8 wide), slices
    (1, 16, 64, 32, 64, 8, 8, 8, 0, 16, 0, 64, 0, 64),           # a single tile: every face is a border; one skip half chunk
])
def test_upconv_matches_deconv_cat_conv(shape, layout):
    ops = _ops()
    dt = torch.float16
    N, Cs, Cu, Cmid, Cout, D, H, W, soff, sstride, uoff, ustride, ooff, ostride = shape
    in_blk, out_blk = layout
    if in_blk and (soff % 16 or sstride % 16 or ooff % 16 or ostride % 16):
        pytest.skip("blocked buffers need 16-channel offsets")
    g = torch.Generator().manual_seed(sum(shape))
    xs = torch.randn(N, Cs, D, H, W, generator=g)
    raw_u = torch.randn(N, Cu, D // 2, H // 2, W // 2, generator=g) * 1.5 + 0.25
    wc = torch.randn(Cout, Cs + Cmid, 3, 3, 3, generator=g) / (27 * (Cs + Cmid)) ** 0.5
    bc = torch.randn(Cout, generator=g)
    wd = torch.randn(Cu, Cmid, 2, 2, 2, generator=g) / Cu ** 0.5
    bd = torch.randn(Cmid, generator=g)
    from test_kernels_gpu import _producer
    norm, act = _producer(raw_u, dt, g, add=torch.randn(N, Cu, generator=g))
    up = F.conv_transpose3d(act.to(dt).float(), wd, bd, stride=2)
    ref = F.conv3d(torch.cat([xs.to(dt).float(), up], 1), wc, bc, padding=1)
    xbuf = torch.full((N, D, H, W, sstride), 3.0, dtype=dt, device="cuda")
    ops.to_channels_last(xs.cuda(), xbuf, soff, Cs)
    if in_blk:
        xbuf = ops.to_blocked(xbuf)
    ubuf = torch.full((N, D // 2, H // 2, W // 2, ustride), 2.0, dtype=dt, device="cuda")
    ops.to_channels_last(raw_u.cuda(), ubuf, uoff, Cu)
    ybuf = torch.full((N, D, H, W, ostride), -5.0, dtype=dt, device="cuda")
    w_skip, wu, btab = ops.pack_upconv_weights(wc.cuda(), bc.cuda(), wd.cuda(), bd.cuda(), Cs)
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.upconv_k3(xbuf, Cs, soff, ubuf, Cu, uoff, norm, w_skip, wu, btab, Cout, ybuf, ooff, stats, in_blocked=in_blk, out_blocked=out_blk)
    ycl = ops.from_blocked(ybuf) if out_blk else ybuf
    got = ops.from_channels_last(ycl, Cout, ooff).cpu()
    err = (got - ref).abs()
    assert torch.allclose(got, ref, **TOL16), (float(err.max()), [int(v) for v in torch.nonzero(err == err.max())[0]])
    if ooff:
        assert float((ycl[..., :ooff].float() + 5).abs().max()) == 0
    if ooff + Cout < ostride:
        assert float((ycl[..., ooff + Cout:].float() + 5).abs().max()) == 0
    # the statistics are the sums of what the kernel computed (fp32 accumulators, stored rounded to fp16): against the sums of
    # its own output, with the rounding of n stored values as the bound (the composed weights are rounded once, AFTER the
    # composition, so the kernel's values differ from torch's by a per-channel systematic ~1e-4 that a sum over 32k voxels keeps)
    st = ops.stats_decode(stats).cpu()[:, :Cout]
    gd = got.double().flatten(2)
    assert bool(((st[..., 0] - gd.sum(-1)).abs() <= 1e-3 * gd.abs().sum(-1) + 0.5).all())
    assert bool(((st[..., 1] - (gd * gd).sum(-1)).abs() <= 2e-3 * (gd * gd).sum(-1) + 0.5).all())
    y2 = torch.full_like(ybuf, -5.0)
    st2 = ops.stats_buffer(N, Cout, "cuda")
    ops.upconv_k3(xbuf, Cs, soff, ubuf, Cu, uoff, norm, w_skip, wu, btab, Cout, y2, ooff, st2, in_blocked=in_blk, out_blocked=out_blk)
    assert torch.equal(y2, ybuf) and torch.equal(st2, stats)


@pytest.mark.gpu
def test_upconv_refuses_what_it_cannot_run():
    ops = _ops()
    assert ops.upconv_supported(torch.float16, 1, 96, 96, 96, 64, 128, 64, 64, 64, 64, True, True)
    assert not ops.upconv_supported(torch.float32, 1, 96, 96, 96, 64, 128, 64, 64, 64, 64)        # fp16 kernel only
    assert not ops.upconv_supported(torch.float16, 1, 12, 96, 96, 64, 128, 64, 64, 64, 64)        # depth not a multiple of 8
    assert not ops.upconv_supported(torch.float16, 1, 96, 96, 96, 24, 128, 64, 64, 64, 64)        # skip half not in 16-channel half chunks
    assert not ops.upconv_supported(torch.float16, 1, 96, 96, 96, 64, 128, 96, 96, 64, 64)        # coarse channels not in groups of 64


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 48, 48, 48, 48, 32, 32, 32), (2, 48, 96, 48, 48, 16, 16, 24)])
def test_upconv_swin_decoder_form(shape):
    """The Swin-UNETR decoder's shape of the same fold (MONAI UnetrUpBlock under models/swin_unetr/denoiser.py:388-397):
    torch.cat((up, skip)) -- the upsampled half FIRST --, 48-channel widths (the coarse buffer padded to a 64-channel stride with
    zeros, the skip half at channel offset 48 of the 96-channel concat buffer, 48 outputs on a 64-wide tile), a transposed
    convolution without bias, and a coarse input that already is an activation (no producer descriptor)."""
    ops = _ops()
    dt = torch.float16
    N, Cs, Cu, Cmid, Cout, D, H, W = shape
    cu_packed = -(-Cu // 64) * 64
    g = torch.Generator().manual_seed(sum(shape))
    skip = torch.randn(N, Cs, D, H, W, generator=g)
    u = torch.randn(N, Cu, D // 2, H // 2, W // 2, generator=g)
    wc = torch.randn(Cout, Cmid + Cs, 3, 3, 3, generator=g) / (27 * (Cs + Cmid)) ** 0.5
    bc = torch.randn(Cout, generator=g)
    wd = torch.randn(Cu, Cmid, 2, 2, 2, generator=g) / Cu ** 0.5
    up = F.conv_transpose3d(u.to(dt).float(), wd, None, stride=2)
    ref = F.conv3d(torch.cat([up, skip.to(dt).float()], 1), wc, bc, padding=1)
    cat = torch.full((N, D, H, W, Cmid + Cs), 3.0, dtype=dt, device="cuda")
    ops.to_channels_last(skip.cuda(), cat, Cmid, Cs)
    ubuf = torch.zeros((N, D // 2, H // 2, W // 2, cu_packed), dtype=dt, device="cuda")
    ops.to_channels_last(u.cuda(), ubuf, 0, Cu)
    y = torch.full((N, D, H, W, Cout), -5.0, dtype=dt, device="cuda")
    w_skip, wu, btab = ops.pack_upconv_weights(wc.cuda(), bc.cuda(), wd.cuda(), None, Cs, up_first=True, cu_packed=cu_packed)
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.upconv_k3(cat, Cs, Cmid, ubuf, cu_packed, 0, None, w_skip, wu, btab, Cout, y, 0, stats)
    got = ops.from_channels_last(y, Cout, 0).cpu()
    err = (got - ref).abs()
    assert torch.allclose(got, ref, **TOL16), (float(err.max()), [int(v) for v in torch.nonzero(err == err.max())[0]])
    st = ops.stats_decode(stats).cpu()[:, :Cout]
    gd = got.double().flatten(2)
    assert bool(((st[..., 0] - gd.sum(-1)).abs() <= 1e-3 * gd.abs().sum(-1) + 0.5).all())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 48, 96, 48, 48, 16, 16, 16), (2, 96, 64, 96, 96, 6, 10, 14), (1, 8, 32, 16, 24, 4, 6, 2),
                                   (1, 48, 96, 48, 48, 48, 48, 48)])
def test_deconv_res_matches_deconv_cat_pointwise(shape):
    """dua_deconv_k2s2_res_fwd: the 1x1x1 residual branch of a UnetResBlock over torch.cat((ConvTranspose3d_k2s2(lo), skip)) without
    the upsampled tensor -- against torch's three layers on the same fp16 operands; the coarse buffer padded to a 64-channel
    stride, the skip half inside the concat buffer, ragged tile counts, one and two output-channel tiles; InstanceNorm sums."""
    ops = _ops()
    dt = torch.float16
    N, Cs, Cu, Cmid, Cout, D, H, W = shape            # D, H, W: COARSE extents
    cu_packed = -(-Cu // 64) * 64
    g = torch.Generator().manual_seed(sum(shape))
    skip = torch.randn(N, Cs, 2 * D, 2 * H, 2 * W, generator=g)
    lo = torch.randn(N, Cu, D, H, W, generator=g)
    w3 = torch.randn(Cout, Cmid + Cs, generator=g) / (Cmid + Cs) ** 0.5
    wd = torch.randn(Cu, Cmid, 2, 2, 2, generator=g) / Cu ** 0.5
    up = F.conv_transpose3d(lo.to(dt).float(), wd, None, stride=2)
    ref = F.conv3d(torch.cat([up, skip.to(dt).float()], 1), w3[:, :, None, None, None])
    cat = torch.full((N, 2 * D, 2 * H, 2 * W, Cmid + Cs), 3.0, dtype=dt, device="cuda")     # the upsampled half is never read
    ops.to_channels_last(skip.cuda(), cat, Cmid, Cs)
    lbuf = torch.zeros((N, D, H, W, cu_packed), dtype=dt, device="cuda")
    ops.to_channels_last(lo.cuda(), lbuf, 0, Cu)
    y = torch.full((N, 2 * D, 2 * H, 2 * W, Cout), -5.0, dtype=dt, device="cuda")
    assert ops.deconv_res_supported(dt, N, D, H, W, Cu, cu_packed, Cout, Cout, Cs)
    wp, wsp = ops.pack_deconv_res_weights(w3.cuda(), wd.cuda(), Cs, up_first=True)
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.deconv_res(lbuf, Cu, 0, wp, cat, Cs, Cmid, wsp, Cout, y, 0, stats)
    got = ops.from_channels_last(y, Cout, 0).cpu()
    err = (got - ref).abs()
    assert torch.allclose(got, ref, **TOL16), (float(err.max()), [int(v) for v in torch.nonzero(err == err.max())[0]])
    st = ops.stats_decode(stats).cpu()[:, :Cout]
    gd = got.double().flatten(2)
    assert bool(((st[..., 0] - gd.sum(-1)).abs() <= 1e-3 * gd.abs().sum(-1) + 0.5).all())
    assert bool(((st[..., 1] - (gd * gd).sum(-1)).abs() <= 2e-3 * (gd * gd).sum(-1) + 0.5).all())
    assert not ops.deconv_res_supported(dt, N, D, H, W, 192, 192, Cout, Cout, Cs)          # more than 128 coarse channels


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_upconv_random_shapes(seed):
    """Seeded random draws over everything the descriptor lets vary: batch, ragged extents (multiples of 8), skip / coarse / middle /
    output channel counts, channel offsets and strides of all three buffers, blocked or channels-last buffers, with and without a
    producer descriptor, either channel order of the concat."""
    import random
    ops = _ops()
    dt = torch.float16
    rnd = random.Random(1000 + seed)
    N = rnd.choice([1, 1, 2])
    D, H, W = (8 * rnd.randint(1, 3) for _ in range(3))
    Cs, Cu, Cmid = 16 * rnd.randint(1, 4), 64 * rnd.randint(1, 2), 8 * rnd.randint(1, 9)
    Cout = 8 * rnd.randint(1, 17)
    up_first = rnd.random() < 0.4
    with_norm = rnd.random() < 0.6
    blocked = rnd.random() < 0.4 and not up_first
    soff = (Cmid if up_first else 16 * rnd.randint(0, 1))
    if blocked:
        soff = 16 * rnd.randint(0, 1)
    sstride = soff + Cs + 16 * rnd.randint(0, 1)
    if blocked:
        sstride = -(-sstride // 16) * 16
    uoff, ooff = 8 * rnd.randint(0, 2), 16 * rnd.randint(0, 1)
    ustride, ostride = uoff + Cu + 8 * rnd.randint(0, 1), -(-(ooff + Cout + 16 * rnd.randint(0, 1)) // 16) * 16
    g = torch.Generator().manual_seed(seed)
    skip = torch.randn(N, Cs, D, H, W, generator=g)
    raw_u = torch.randn(N, Cu, D // 2, H // 2, W // 2, generator=g) * 1.5 + 0.25
    cin = Cs + Cmid
    wc = torch.randn(Cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    bc = torch.randn(Cout, generator=g) if rnd.random() < 0.7 else None
    wd = torch.randn(Cu, Cmid, 2, 2, 2, generator=g) / Cu ** 0.5
    bd = torch.randn(Cmid, generator=g) if rnd.random() < 0.7 else None
    if with_norm:
        from test_kernels_gpu import _producer
        norm, act = _producer(raw_u, dt, g, add=torch.randn(N, Cu, generator=g))
        act = act.to(dt).float()
    else:
        norm, act = None, raw_u.to(dt).float()
    up = F.conv_transpose3d(act, wd, bd, stride=2)
    parts = [up, skip.to(dt).float()] if up_first else [skip.to(dt).float(), up]
    ref = F.conv3d(torch.cat(parts, 1), wc, bc, padding=1)
    xbuf = torch.full((N, D, H, W, sstride), 3.0, dtype=dt, device="cuda")
    ops.to_channels_last(skip.cuda(), xbuf, soff, Cs)
    if blocked:
        xbuf = ops.to_blocked(xbuf)
    ubuf = torch.full((N, D // 2, H // 2, W // 2, ustride), 2.0, dtype=dt, device="cuda")
    ops.to_channels_last(raw_u.cuda(), ubuf, uoff, Cu)
    ybuf = torch.full((N, D, H, W, ostride), -5.0, dtype=dt, device="cuda")
    w_skip, wu, btab = ops.pack_upconv_weights(wc.cuda(), None if bc is None else bc.cuda(), wd.cuda(), None if bd is None else bd.cuda(),
                                               Cs, up_first=up_first)
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.upconv_k3(xbuf, Cs, soff, ubuf, Cu, uoff, norm, w_skip, wu, btab, Cout, ybuf, ooff, stats, in_blocked=blocked, out_blocked=blocked)
    ycl = ops.from_blocked(ybuf) if blocked else ybuf
    got = ops.from_channels_last(ycl, Cout, ooff).cpu()
    desc = dict(N=N, dims=(D, H, W), Cs=Cs, Cu=Cu, Cmid=Cmid, Cout=Cout, up_first=up_first, norm=with_norm, blocked=blocked,
                soff=soff, sstride=sstride, uoff=uoff, ustride=ustride, ooff=ooff, ostride=ostride)
    assert torch.allclose(got, ref, **TOL16), (float((got - ref).abs().max()), desc)
    if ooff:
        assert float((ycl[..., :ooff].float() + 5).abs().max()) == 0, desc
    if ooff + Cout < ostride:
        assert float((ycl[..., ooff + Cout:].float() + 5).abs().max()) == 0, desc
