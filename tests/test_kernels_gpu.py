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
@pytest.mark.parametrize("c0,c1,cs", [(1, 0, 8), (1, 16, 24), (3, 2, 8), (16, 0, 16), (5, 7, 16)])
def test_whole_row_layout_kernel_equals_the_per_channel_one(dtype, c0, c1, cs):
    """dua_to_channels_last_rows (torch.cat((image, x), 1) -> channels-last rows in 16-byte stores) against the per-channel kernel."""
    ops = _ops()
    if cs * (2 if dtype == torch.float16 else 4) > 64:
        pytest.skip("rows wider than 64 bytes take the per-channel kernel")
    g = torch.Generator().manual_seed(c0 * 100 + c1)
    parts = [torch.randn(2, c0, 5, 6, 7, generator=g).cuda()] + ([torch.randn(2, c1, 5, 6, 7, generator=g).cuda()] if c1 else [])
    want = torch.full((2, 5, 6, 7, cs), 7.0, dtype=dtype, device="cuda")
    off = 0
    for i, p in enumerate(parts):
        ops.to_channels_last(p, want, off, c_fill=(cs - off) if i == len(parts) - 1 else None)
        off += p.shape[1]
    got = ops.to_channels_last_rows(parts, torch.full((2, 5, 6, 7, cs), 7.0, dtype=dtype, device="cuda"))
    assert torch.equal(got, want)


def _producer(raw, dtype, g, add=None):
    """Stats + affine of a fictitious producer layer whose raw output is ``raw`` (as stored in ``dtype``):
    returns (ops.Norm, reference activation = LeakyReLU(InstanceNorm(raw)*gamma+beta) [+ add])."""
    ops = _ops()
    N, C = raw.shape[:2]
    rq = raw.to(dtype).double()
    stats = ops.stats_buffer(N, C, "cuda")
    # spread the sums over the replica rows like 8 groups of workgroups would
    flat = rq.flatten(2)
    for r in range(8):
        part = flat[:, :, r::8]
        row = ops.stats_encode(torch.stack([part.sum(-1), (part * part).sum(-1)], -1))
        stats[:, r] = row[:, 0].cuda()
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    act = F.leaky_relu(F.instance_norm(rq.float(), weight=gamma, bias=beta, eps=1e-5), 0.1)
    kw = {}
    if add is not None:
        act = act + add[:, :, None, None, None]
        kw = dict(add=add.cuda().contiguous())
    count = raw.shape[2] * raw.shape[3] * raw.shape[4]
    return ops.Norm(stats, gamma.cuda(), beta.cuda(), count, **kw), act


@pytest.fixture
def conv_variant(request):
    """Force one of the conv3d_k3 launch shapes (0: automatic policy with split-K / 2x8x8 tiles where they pay,
    2: 4x8x8 tiles without split-K, 3: 2x8x8 tiles, 6: the policy with the kd-plane / LDS-DMA form of the small layers
    switched off)."""
    ops = _ops()
    ops.CONV_POLICY = request.param          # handed to the kernels with every call (dua_conv3_desc.policy)
    yield request.param
    ops.CONV_POLICY = 0


@pytest.mark.parametrize("conv_variant", [0, 2, 3, 6], indirect=True)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [
    (1, 64, 128, 24, 24, 24),    # a 24^3-sized layer: 108 workgroups of 4x8x8 -> the 2x8x8 forms (kd planes by default)
    (2, 40, 72, 16, 24, 24),     # the same regime, ragged channel counts, two samples
    (1, 64, 128, 24, 24, 24),    # auto picks 2x8x8 tiles (108 workgroups of 4x8x8 would half-fill the chip)
    (1, 40, 72, 22, 20, 26),     # the same path with ragged edges in every axis
    # N, Cin, Cout, D, H, W
    (1, 16, 64, 8, 8, 8),
    (2, 40, 72, 6, 10, 12),      # ragged everything: partial tiles, cin/cout not multiples of the chunk/tile
    (1, 8, 8, 4, 8, 8),
    (1, 64, 128, 12, 12, 12),
    (1, 8, 16, 2, 2, 2),         # bottom level of the 32^3 config
    (1, 72, 64, 16, 16, 24),     # several full tiles, three Cin chunks (the last one 8 channels: its second k-step is skipped)
    (2, 48, 48, 16, 16, 32),     # the Swin-UNETR width: 32 + 16 channels, 48 outputs on a 64-wide tile
    (1, 112, 48, 8, 16, 16),     # three full chunks + 16
])
def test_conv3_raw_and_stats(dtype, shape, conv_variant):
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
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.conv3d_k3(xcl, Cin, 0, wp, bp, Cout, y, 0, stats)
    got = ops.from_channels_last(y, Cout).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())

    # statistics (taken from the fp32 accumulators) -> scale/shift, against instance_norm of the exact conv
    st = ops.stats_decode(stats).cpu()[:, :Cout]
    rd = ref.double().flatten(2)
    stol = dict(rtol=1e-5, atol=1e-3) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-2)
    assert torch.allclose(st[..., 0], rd.sum(-1), **stol)
    assert torch.allclose(st[..., 1], (rd * rd).sum(-1), **stol)
    gamma = torch.rand(Cout, generator=g) + 0.5
    beta = torch.randn(Cout, generator=g)
    scale, shift = ops.instnorm_finalize(ops.Norm(stats, gamma.cuda(), beta.cuda(), D * H * W), N, Cout)
    var, mean = torch.var_mean(ref.double(), dim=(2, 3, 4), unbiased=False)
    sc_ref = gamma.double()[None] / torch.sqrt(var + 1e-5)
    sh_ref = beta.double()[None] - mean * sc_ref
    ftol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=5e-3, atol=5e-3)
    assert torch.allclose(scale.cpu().double(), sc_ref, **ftol)
    assert torch.allclose(shift.cpu().double(), sh_ref, **ftol)


@pytest.fixture
def conv_variant_any(request):
    ops = _ops()
    ops.CONV_POLICY = request.param          # handed to the kernels with every call (dua_conv3_desc.policy)
    yield request.param
    ops.CONV_POLICY = 0


@pytest.mark.parametrize("conv_variant_any,layout", [(0, (False, False)), (0, (True, False)), (0, (False, True)), (0, (True, True)),
                                                     (7, (False, False)), (8, (False, False)), (8, (True, True))], indirect=["conv_variant_any"])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shape", [
    # N, Cin, Cout, D, H, W, input channel offset / stride, output channel offset / stride: layers with >= 1024 tiles of 4x8x8
    (1, 64, 64, 64, 64, 64, 0, 64, 0, 64),        # a 64 -> 64 layer, 8 x 8 x 8 = 512 wide tiles
    (2, 32, 64, 40, 64, 64, 32, 64, 64, 128),     # depth 40 = 5 slabs of 8, two samples; reads / writes halves of concat buffers
    (2, 48, 48, 32, 64, 64, 0, 48, 0, 48),        # the Swin-UNETR width: three half chunks, 48 outputs on a 64-wide tile; two samples
    (1, 16, 72, 64, 64, 64, 0, 16, 16, 96),       # one half chunk; two output-channel tiles, the second one 8 channels wide
])
def test_conv3_wide_tile_form(shape, fused, conv_variant_any, layout):
    """The 8-accumulator form (conv3d_wide.hip) that fp16 layers with >= 1024 tiles take: against torch conv3d on the
    fp16-rounded operands (0: one tile per workgroup; 8: persistent workgroups with the accumulators in v[128:255] by name, the
    round-5 form that measured slower and stays for the A/B), like the 4x8x8 kernel it replaces (7); statistics against the exact convolution.  ``layout``:
    input / output buffer in 16-channel blocks (dua_conv3_desc.layout) instead of channels-last rows."""
    ops = _ops()
    dtype = torch.float16
    N, Cin, Cout, D, H, W, ioff, istride, ooff, ostride = shape
    g = torch.Generator().manual_seed(sum(shape) + int(fused))
    raw = torch.randn(N, Cin, D, H, W, generator=g) * 1.5 + 0.25
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    if fused:
        norm, act = _producer(raw, dtype, g, add=torch.randn(N, Cin, generator=g))
        xin = act.to(dtype).float()
    else:
        norm, xin = None, raw.to(dtype).float()
    ref = F.conv3d(xin, w.to(dtype).float(), b, padding=1)
    xbuf = torch.full((N, D, H, W, istride), 3.0, dtype=dtype, device="cuda")
    ops.to_channels_last(raw.cuda(), xbuf, ioff, Cin)
    in_blk, out_blk = layout
    if in_blk:
        xbuf = ops.to_blocked(xbuf)
    ybuf = torch.full((N, D, H, W, ostride), -5.0, dtype=dtype, device="cuda")
    wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), dtype)
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.conv3d_k3(xbuf, Cin, ioff, wp, bp, Cout, ybuf, ooff, stats, norm=norm, in_blocked=in_blk, out_blocked=out_blk)
    ycl = ops.from_blocked(ybuf) if out_blk else ybuf
    got = ops.from_channels_last(ycl, Cout, ooff).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())
    if ooff:
        assert float((ycl[..., :ooff].float() + 5).abs().max()) == 0
    if ooff + Cout < ostride:
        assert float((ycl[..., ooff + Cout:].float() + 5).abs().max()) == 0
    st = ops.stats_decode(stats).cpu()[:, :Cout]
    rd = ref.double().flatten(2)
    assert torch.allclose(st[..., 0], rd.sum(-1), rtol=2e-3, atol=0.5)
    assert torch.allclose(st[..., 1], (rd * rd).sum(-1), rtol=2e-3, atol=0.5)
    # bit-reproducible: a second launch gives the same bytes and the same statistics words
    y2 = torch.full_like(ybuf, -5.0)
    st2 = ops.stats_buffer(N, Cout, "cuda")
    ops.conv3d_k3(xbuf, Cin, ioff, wp, bp, Cout, y2, ooff, st2, norm=norm, in_blocked=in_blk, out_blocked=out_blk)
    assert torch.equal(y2, ybuf) and torch.equal(st2, stats)


def test_blocked_layout_is_refused_by_kernels_that_cannot():
    """A launch that would read or write 16-channel blocks with a kernel that only knows channels-last rows fails loudly."""
    ops = _ops()
    dtype = torch.float16
    x = torch.zeros(1, 16, 16, 16, 32, dtype=dtype, device="cuda")          # 16^3: far below the wide-tile form's 1024 tiles
    w = torch.zeros(64, 32, 3, 3, 3, device="cuda")
    wp, bp = ops.pack_conv3_weights(w, torch.zeros(64, device="cuda"), dtype)
    y = torch.zeros(1, 16, 16, 16, 64, dtype=dtype, device="cuda")
    assert ops.conv3_kernel_kind(dtype, 1, 16, 16, 16, 32, 32, 64) == ops.KIND_V2
    for kw in (dict(in_blocked=True), dict(out_blocked=True)):
        with pytest.raises(RuntimeError):
            ops.conv3d_k3(x, 32, 0, wp, bp, 64, y, 0, ops.stats_buffer(1, 64, "cuda"), **kw)
    assert ops.conv3_kernel_kind(dtype, 1, 96, 96, 96, 64, 64, 64, fused=True) == ops.KIND_WIDE
    assert ops.conv3_kernel_kind(torch.float32, 1, 96, 96, 96, 64, 64, 64) == ops.KIND_V2
    assert ops.conv3_kernel_kind(dtype, 1, 96, 96, 96, 24, 24, 64, tap_channel=16) == ops.KIND_FIRST


@pytest.mark.parametrize("conv_variant", [0, 6], indirect=True)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(1, 128, 256, 12, 12, 12), (2, 72, 136, 6, 6, 6), (1, 136, 64, 8, 16, 16)])
def test_conv3_split_k(dtype, shape, conv_variant):
    """Layers too small to fill 256 CUs split K = (Cin chunk, kd) over workgroups; fp32 partial tiles are
    summed by the finish kernel, which also adds the bias and takes the InstanceNorm sums."""
    ops = _ops()
    N, Cin, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    raw = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    norm, act = _producer(raw, dtype, g, add=torch.randn(N, Cin, generator=g))
    ref = F.conv3d(act.to(dtype).float(), w.to(dtype).float(), b, padding=1)
    nbytes = ops.conv3_workspace_bytes(dtype, N, D, H, W, Cin, Cout)
    assert nbytes > 0, "these shapes are expected to take the split-K path"
    ws = torch.empty(nbytes // 4, device="cuda")
    wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), dtype)
    y = torch.full((N, D, H, W, Cout + 8), 2.0, dtype=dtype, device="cuda")
    stats = ops.stats_buffer(N, Cout, "cuda")
    ops.conv3d_k3(_cl(raw, dtype), Cin, 0, wp, bp, Cout, y, 8, stats, norm=norm, workspace=ws)
    got = ops.from_channels_last(y, Cout, 8).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())
    assert float((y[..., :8].float() - 2).abs().max()) == 0
    st = ops.stats_decode(stats).cpu()[:, :Cout]
    gd = got.double().flatten(2)                        # the split-K finish kernel takes its sums from the stored values
    assert torch.allclose(st[..., 0], gd.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st[..., 1], (gd * gd).sum(-1), rtol=1e-5, atol=1e-3)
    # same layer without a workspace -> fused epilogue path; results must agree
    y2 = torch.zeros_like(y); stats2 = ops.stats_buffer(N, Cout, "cuda")
    ops.conv3d_k3(_cl(raw, dtype), Cin, 0, wp, bp, Cout, y2, 8, stats2, norm=norm)
    assert torch.allclose(ops.from_channels_last(y2, Cout, 8).cpu(), got, **TOL[dtype])


@pytest.mark.parametrize("conv_variant", [2, 3], indirect=True)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_conv3_fused_input_transform_and_channel_slices(dtype, conv_variant):
    """Producer IN+LeakyReLU+temb add fused into the consumer's halo staging; input read from and
    output written to channel slices of wider buffers (the concat-in-place layout)."""
    ops = _ops()
    N, Cin, Cout, D, H, W = 2, 24, 40, 8, 8, 16
    g = torch.Generator().manual_seed(11)
    raw = torch.randn(N, Cin, D, H, W, generator=g) * 2 + 0.5
    add = torch.randn(N, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    norm, act = _producer(raw, dtype, g, add=add)
    ref = F.conv3d(act.to(dtype).float(), w.to(dtype).float(), b, padding=1)

    xbuf = torch.full((N, D, H, W, 48), 3.0, dtype=dtype, device="cuda")
    ops.to_channels_last(raw.cuda(), xbuf, 16, Cin)
    ybuf = torch.full((N, D, H, W, 64), -5.0, dtype=dtype, device="cuda")
    wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), dtype)
    ops.conv3d_k3(xbuf, Cin, 16, wp, bp, Cout, ybuf, 8, ops.stats_buffer(N, Cout, "cuda"), norm=norm)
    got = ops.from_channels_last(ybuf, Cout, 8).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())
    assert float((ybuf[..., :8].float() + 5).abs().max()) == 0 and float((ybuf[..., 48:].float() + 5).abs().max()) == 0


@pytest.mark.parametrize("classes,shape,cout", [(16, (1, 16, 24, 8), 64), (16, (2, 10, 9, 13), 64), (0, (1, 8, 16, 16), 64),
                                                (16, (1, 40, 72, 72), 64),      # 810 tiles: persistent workgroups take two
                                                (16, (3, 12, 20, 16), 8), (16, (1, 9, 17, 8), 96)])
def test_conv3_single_channel_tap_form(classes, shape, cout):
    """First layers: the lone image channel behind 16 (denoiser) or 0 (encoder) ordinary channels contracted as two
    k-steps over its 27 taps; against torch conv3d on the fp16-rounded operands and against the ordinary form.  With 16
    channels the default is the resident-weight kernel (persistent workgroups); conv_variant 6 keeps the slab pipeline."""
    ops = _ops()
    from diff_unet_amos_amd import _native as nv
    N, D, H, W = shape
    cin_src, cin_p = classes + 1, classes + 8
    g = torch.Generator().manual_seed(classes + D)
    x = torch.randn(N, cin_src, D, H, W, generator=g)                    # reference channel order [image | x_t]
    w = torch.randn(cout, cin_src, 3, 3, 3, generator=g) / (27 * cin_src) ** 0.5
    b = torch.randn(cout, generator=g)
    perm = list(range(1, classes + 1)) + [0] + [-1] * 7                  # packed [x_t | image | pad]
    xp = torch.zeros(N, D, H, W, cin_p, dtype=torch.float16, device="cuda")
    xp[..., :classes] = x[:, 1:].permute(0, 2, 3, 4, 1).half()
    xp[..., classes] = x[:, 0].half()
    outs = []
    try:
        for tap, variant in ((None, 0), (classes, 0), (classes, 6)):
            ops.CONV_POLICY = variant
            wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), torch.float16, cin_packed=cin_p, perm=perm, tap_channel=tap)
            y = torch.full((N, D, H, W, cout + 8), -5.0, dtype=torch.float16, device="cuda")
            st = ops.stats_buffer(N, cout, "cuda")
            ops.conv3d_k3(xp, cin_p, 0, wp, bp, cout, y, 0, st, tap_channel=tap)
            assert float((y[..., cout:].float() + 5).abs().max()) == 0          # nothing written past Cout
            outs.append((ops.from_channels_last(y, cout).cpu(), ops.stats_decode(st).cpu()))
    finally:
        ops.CONV_POLICY = 0
    want = F.conv3d(x.half().float().cuda(), w.half().float().cuda(), b.cuda(), padding=1).cpu()
    for got, st in outs:
        assert (got - want).abs().max() < 2e-2, float((got - want).abs().max())
        gd = got.double().permute(1, 0, 2, 3, 4).reshape(cout, N, -1).permute(1, 0, 2)
        # the sums are taken from the fp32 accumulators, the stored values are their fp16 roundings (a random walk over the voxels)
        walk = 1e-3 * (D * H * W) ** 0.5
        assert torch.allclose(st[:, :cout, 0], gd.sum(-1), rtol=1e-5, atol=max(2e-2, walk))
        assert torch.allclose(st[:, :cout, 1], (gd * gd).sum(-1), rtol=2e-3, atol=max(2e-2, walk))
        assert float(st[:, cout:].abs().max() if st.shape[1] > cout else 0) == 0
    assert (outs[0][0] - outs[1][0]).abs().max() < 4e-3      # same products, different summation order
    assert (outs[1][0] - outs[2][0]).abs().max() < 4e-3
    if classes == 16 and cout % 16 == 0:
        # the resident-weight kernel writing a 16-channel-blocked buffer (the layout its consumer, the wide-tile form, reads)
        wp, bp = ops.pack_conv3_weights(w.cuda(), b.cuda(), torch.float16, cin_packed=cin_p, perm=perm, tap_channel=classes)
        yb = torch.full((N, D, H, W, cout + 16), -5.0, dtype=torch.float16, device="cuda")
        ops.conv3d_k3(xp, cin_p, 0, wp, bp, cout, yb, 16, ops.stats_buffer(N, cout, "cuda"), tap_channel=classes, out_blocked=True)
        ycl = ops.from_blocked(yb)
        y_plain = torch.full((N, D, H, W, cout + 16), -5.0, dtype=torch.float16, device="cuda")
        ops.conv3d_k3(xp, cin_p, 0, wp, bp, cout, y_plain, 16, ops.stats_buffer(N, cout, "cuda"), tap_channel=classes)
        assert torch.equal(ycl, y_plain)


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
    ops.conv3d_k3(buf, 8, 0, wp, bp, 16, y, 0, ops.stats_buffer(1, 16, "cuda"))
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
    norm, y = _producer(raw, dtype, g)
    if with_emb:
        y = y + emb.to(dtype).float()
    y = y.to(dtype).float()
    out = torch.zeros((N, D, H, W, 40), dtype=dtype, device="cuda")
    pooled = torch.zeros((N, D // 2, H // 2, W // 2, C), dtype=dtype, device="cuda") if pool else None
    ops.materialize(_cl(raw, dtype), C, norm, out, 8, emb=_cl(emb, dtype) if with_emb else None, pooled=pooled)
    got = ops.from_channels_last(out, C, 8).cpu()
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)
    assert torch.allclose(got, y, **tol), float((got - y).abs().max())
    if pool:
        gp = ops.from_channels_last(pooled, C).cpu()
        assert torch.equal(gp, F.max_pool3d(got, 2))


@pytest.mark.parametrize("pool", [False, True])
def test_materialize_into_a_blocked_buffer(pool):
    """materialize writing its slice of a concat buffer that is kept in 16-channel blocks (the level-0 skip half)."""
    ops = _ops()
    dtype = torch.float16
    N, C, D, H, W = 2, 32, 4, 6, 8
    g = torch.Generator().manual_seed(9)
    raw = torch.randn(N, C, D, H, W, generator=g)
    emb = torch.randn(N, C, D, H, W, generator=g)
    norm, y = _producer(raw, dtype, g)
    y = (y + emb.to(dtype).float()).to(dtype).float()
    out = torch.full((N, D, H, W, 64), 4.0, dtype=dtype, device="cuda")
    pooled = torch.zeros((N, D // 2, H // 2, W // 2, C), dtype=dtype, device="cuda") if pool else None
    ops.materialize(_cl(raw, dtype), C, norm, out, 16, emb=_cl(emb, dtype), pooled=pooled, out_blocked=True)
    ocl = ops.from_blocked(out)
    got = ops.from_channels_last(ocl, C, 16).cpu()
    assert torch.allclose(got, y, rtol=2e-3, atol=2e-3), float((got - y).abs().max())
    assert float((ocl[..., :16].float() - 4).abs().max()) == 0 and float((ocl[..., 48:].float() - 4).abs().max()) == 0
    if pool:
        assert torch.equal(ops.from_channels_last(pooled, C).cpu(), F.max_pool3d(got, 2))


@pytest.mark.parametrize("fused", [False, True])
def test_deconv_k2s2_into_a_blocked_buffer(fused):
    """The all-taps transposed convolution writing its half of a concat buffer kept in 16-channel blocks; kernels that
    cannot are refused."""
    ops = _ops()
    dtype = torch.float16
    N, Cin, Cout, D, H, W = 1, 64, 64, 32, 32, 32
    g = torch.Generator().manual_seed(21)
    raw = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, 2, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    xin, norm = raw.to(dtype).float(), None
    if fused:
        norm, act = _producer(raw, dtype, g)
        xin = act.to(dtype).float()
    ref = F.conv_transpose3d(xin, w.to(dtype).float(), b, stride=2)
    y = torch.full((N, 2 * D, 2 * H, 2 * W, 128), 9.0, dtype=dtype, device="cuda")
    wp, bp = ops.pack_deconv_weights(w.cuda(), b.cuda(), dtype)
    assert ops.deconv_kernel_kind(dtype, N, D, H, W, Cin, Cout) == ops.DECONV_ALLTAPS
    ops.deconv_k2s2(_cl(raw, dtype), Cin, 0, wp, bp, Cout, y, 64, norm=norm, out_blocked=True)
    ycl = ops.from_blocked(y)
    got = ops.from_channels_last(ycl, Cout, 64).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())
    assert float((ycl[..., :64].float() - 9).abs().max()) == 0
    small = torch.zeros(1, 4, 4, 4, 64, dtype=dtype, device="cuda")
    with pytest.raises(RuntimeError):
        ops.deconv_k2s2(small, 64, 0, wp, bp, Cout, torch.zeros(1, 8, 8, 8, 128, dtype=dtype, device="cuda"), 64, out_blocked=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(1, 64, 32, 3, 3, 3), (2, 40, 72, 2, 4, 6), (1, 16, 16, 6, 6, 6),
                                   (1, 64, 64, 32, 32, 32), (1, 40, 72, 30, 34, 36),      # these two: all-taps kernel
                                   # >= 8 Cin chunks: the kernel whose waves split the chunks (ragged last chunk, ragged
                                   # voxel tile, two output-channel tiles, the 6^3 x 512 layer of the denoiser)
                                   (1, 256, 128, 3, 4, 5), (2, 272, 72, 2, 3, 3), (1, 512, 256, 6, 6, 6)])
@pytest.mark.parametrize("fused", [False, True])
def test_deconv_k2s2(dtype, shape, fused):
    ops = _ops()
    N, Cin, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    raw = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, 2, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    xin = raw.to(dtype).float()
    norm = None
    if fused:
        norm, act = _producer(raw, dtype, g)
        xin = act.to(dtype).float()
    ref = F.conv_transpose3d(xin, w.to(dtype).float(), b, stride=2)
    y = torch.full((N, 2 * D, 2 * H, 2 * W, Cout + 16), 9.0, dtype=dtype, device="cuda")
    wp, bp = ops.pack_deconv_weights(w.cuda(), b.cuda(), dtype)
    ops.deconv_k2s2(_cl(raw, dtype), Cin, 0, wp, bp, Cout, y, 16, norm=norm)
    got = ops.from_channels_last(y, Cout, 16).cpu()
    assert torch.allclose(got, ref, **TOL[dtype]), float((got - ref).abs().max())
    assert float((y[..., :16].float() - 9).abs().max()) == 0


def test_temb_table_against_reference_golden(golden):
    """G5 goldens come from the reference's own models/diffusion/utils.py."""
    import math
    ops = _ops()
    t = torch.from_numpy(golden["G5_t"]).to(torch.int32).cuda()
    w = {k[len("G5_w_"):]: torch.from_numpy(golden[k]).cuda() for k in golden.files if k.startswith("G5_w_")}
    freqs = torch.exp(torch.arange(64, dtype=torch.float32) * -(math.log(10000) / 63)).cuda()
    g = torch.Generator().manual_seed(0)
    wc = torch.randn(40, 512, generator=g) * 0.05; bc = torch.randn(40, generator=g)
    tab = ops.temb_table(t, freqs, w["dense.0.weight"], w["dense.0.bias"], w["dense.1.weight"], w["dense.1.bias"],
                         wc.cuda(), bc.cuda()).cpu()
    temb = torch.from_numpy(golden["G5_temb"])
    ref = F.linear(temb * torch.sigmoid(temb), wc, bc)
    assert torch.allclose(tab, ref, rtol=1e-4, atol=1e-5), float((tab - ref).abs().max())


def test_q_sample_and_sampler_steps_against_reference_golden(golden):
    """Elementwise sampler kernels vs goldens produced by the reference's guided_diffusion."""
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd.gaussian_diffusion import make_spaced
    ops = _ops()
    d1000, d10 = make_spaced(1000, [1000]), make_spaced(1000, [10])
    x0 = torch.from_numpy(golden["G2_x0"]).cuda(); eps = torch.from_numpy(golden["G2_eps"]).cuda()
    for tag in "ab":
        t = torch.from_numpy(golden[f"G2_t_{tag}"])
        got = ops.q_sample(x0, eps, d1000.q_coef(t).cuda()).cpu().numpy()
        assert (got == golden[f"G2_xt_{tag}"]).all()
    x = torch.from_numpy(golden["G3_x"]).cuda(); nz = torch.from_numpy(golden["G3_noise"]).cuda()
    for dtag, d, ts in (("s10", d10, [0, 1, 5, 9]), ("s1000", d1000, [0, 500, 999])):
        for sname in ("half", "tanh"):
            for ti in ts:
                key = f"G3_{dtag}_{sname}_t{ti}"
                mo = torch.from_numpy(golden[f"{key}_pmv_model_output"]).cuda().contiguous()
                t = torch.tensor([ti, ti])
                xs = torch.empty_like(x)
                got = ops.sampler_step(nv.MODE_DDPM, mo, x, nz, d.ddpm_coef(t).cuda(), xstart_out=xs)
                assert (got.cpu().numpy() == golden[f"{key}_psample"]).all(), key
                assert (xs.cpu().numpy() == golden[f"{key}_pmv_pred_xstart"]).all()
                got = ops.sampler_step(nv.MODE_DDIM, mo, x, nz, d.ddim_coef(t, 0.0).cuda())
                assert torch.allclose(got.cpu(), torch.from_numpy(golden[f"{key}_ddim"]), rtol=1e-6, atol=1e-6), key
                got = ops.sampler_step(nv.MODE_DDIM, mo, x, nz, d.ddim_coef(t, 0.7).cuda())
                assert torch.allclose(got.cpu(), torch.from_numpy(golden[f"{key}_ddim_eta07"]), rtol=1e-6, atol=1e-6), key


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,K", [(2, 24), (16, 24), (16, 64), (13, 64), (9, 32)])   # K % 32 == 0 and 8 < C <= 16: MFMA tail (fp16)
@pytest.mark.parametrize("mode", ["logits", "ddpm", "ddim"])
def test_final_conv_sampler(dtype, C, K, mode):
    from diff_unet_amos_amd import _native as nv
    from diff_unet_amos_amd.gaussian_diffusion import make_spaced
    ops = _ops()
    N, D, H, W = 2, 4, 6, 10
    vox = D * H * W
    g = torch.Generator().manual_seed(C)
    raw = torch.randn(N, K, D, H, W, generator=g)
    wf = torch.randn(C, K, generator=g) / K ** 0.5; bf = torch.randn(C, generator=g)
    norm, act = _producer(raw, dtype, g)
    logits_ref = F.conv3d(act, wf.view(C, K, 1, 1, 1), bf)
    cx = ops.state_stride(C)
    d = make_spaced(1000, [10])
    t = torch.tensor([3, 7])
    xt = torch.randn(N, C, D, H, W, generator=g); nz = torch.randn(N, C, D, H, W, generator=g)
    from oracle.diffusion_ref import RefDiffusion
    rd = RefDiffusion(1000, [10])
    fn = (lambda x, tt, **k: logits_ref)
    args = dict(norm=norm, wf=wf.cuda(), bf=bf.cuda())
    rawcl = _cl(raw, dtype)
    logits = torch.zeros(N, C, D, H, W, device="cuda")
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-3, atol=5e-3)   # fp16 head operands
    if mode == "logits":
        ops.final_conv_sampler(rawcl, K, num_classes=C, mode=nv.MODE_LOGITS, logits=logits, **args)
        assert torch.allclose(logits.cpu(), logits_ref, **tol)
        return
    state = torch.zeros(N, D, H, W, cx, device="cuda")
    ops.to_channels_last(xt.cuda(), state, 0, cx)
    xin = torch.full((N, D, H, W, cx + 8), 4.0, dtype=dtype, device="cuda")
    xsum = torch.ones(N, D, H, W, cx, device="cuda")
    xstart = torch.zeros(N, C, D, H, W, device="cuda")
    if mode == "ddpm":
        ref = rd.p_sample(fn, xt, t, nz); coef = d.ddpm_coef(t); m = nv.MODE_DDPM
    else:
        ref = rd.ddim_sample(fn, xt, t, nz, eta=0.3); coef = d.ddim_coef(t, 0.3); m = nv.MODE_DDIM
    ops.final_conv_sampler(rawcl, K, num_classes=C, mode=m, coef=coef.cuda(), x_state=state, noise=nz.cuda(), xin=xin,
                           xstart_sum=xsum, logits=logits, xstart=xstart, **args)
    assert torch.allclose(logits.cpu(), logits_ref, **tol)
    assert torch.allclose(ops.from_channels_last(state, C).cpu(), ref["sample"], **tol)
    assert torch.allclose(xstart.cpu(), ref["pred_xstart"], **tol)
    assert torch.allclose(ops.from_channels_last(xsum, C).cpu(), 1 + ref["pred_xstart"], **tol)
    xin_tol = tol if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)
    assert torch.allclose(ops.from_channels_last(xin, C).cpu(), ref["sample"], **xin_tol)
    assert float((xin[..., C:].float() - 4).abs().max()) == 0     # image / pad channels untouched


@pytest.mark.parametrize("K", [8, 32])        # VALU tail / MFMA tail
def test_in_kernel_philox_noise_is_standard_normal(K):
    from diff_unet_amos_amd import _native as nv
    ops = _ops()
    N, C, D, H, W = 1, 16, 16, 32, 32
    raw = torch.zeros(N, D, H, W, K, dtype=torch.float16, device="cuda")
    z = ops.Norm(ops.stats_buffer(N, K, "cuda"), torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), D * H * W)
    state = torch.zeros(N, D, H, W, 16, device="cuda")
    coef = torch.tensor([[0, 0, 1.0, 0, 0, 0, 0, 0]], device="cuda")     # x_new = eps
    step = torch.tensor([5], dtype=torch.int32, device="cuda")
    ops.final_conv_sampler(raw, K, z, torch.zeros(C, K, device="cuda"), torch.zeros(C, device="cuda"), C, nv.MODE_DDPM,
                           coef=coef, x_state=state, step_word=step, seed=1234)
    a = state.clone()
    # known answer: Philox4x32-10 (Salmon et al., counter = (voxel lo, voxel hi, step, class quad), key = seed) + Box-Muller,
    # restated in numpy; the kernel's logarithm / sine / cosine are the fast hardware forms, hence the tolerance
    import numpy as np
    vox = D * H * W
    ctr = np.zeros((vox, 4, 4), dtype=np.uint64)
    ctr[:, :, 0] = np.arange(vox, dtype=np.uint64)[:, None]
    ctr[:, :, 2] = 5
    ctr[:, :, 3] = np.arange(4, dtype=np.uint64)[None, :]
    k0, k1 = np.uint64(1234), np.uint64(0)
    M32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * ctr[..., 0]
        p1 = np.uint64(0xCD9E8D57) * ctr[..., 2]
        n0 = ((p1 >> np.uint64(32)) ^ ctr[..., 1] ^ k0) & M32
        n2 = ((p0 >> np.uint64(32)) ^ ctr[..., 3] ^ k1) & M32
        ctr = np.stack([n0, p1 & M32, n2, p0 & M32], -1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    u = ((ctr.astype(np.float32) + np.float32(0.5)) * np.float32(2.3283064365386963e-10)).astype(np.float64)
    want = np.empty((vox, 4, 4))
    for j in (0, 2):
        rad = np.sqrt(-2.0 * np.log(u[..., j]))
        want[..., j] = rad * np.cos(2 * np.pi * u[..., j + 1])
        want[..., j + 1] = rad * np.sin(2 * np.pi * u[..., j + 1])
    got = a.view(vox, 16).double().cpu().numpy()
    assert np.abs(got - want.reshape(vox, 16)).max() < 5e-3, float(np.abs(got - want.reshape(vox, 16)).max())
    assert abs(float(a.mean())) < 1e-2 and abs(float(a.std()) - 1) < 1e-2
    assert abs(float((a ** 4).mean()) - 3) < 0.1
    flat = a.view(-1, 16)
    assert abs(float((flat[:, 0] * flat[:, 1]).mean())) < 0.03            # Box-Muller pair uncorrelated (4 sigma at n=16384)
    state.zero_()
    ops.final_conv_sampler(raw, K, z, torch.zeros(C, K, device="cuda"), torch.zeros(C, device="cuda"), C, nv.MODE_DDPM,
                           coef=coef, x_state=state, step_word=step, seed=1234)
    assert torch.equal(state, a)                                          # counter-based: reproducible
    step += 1; state.zero_()
    ops.final_conv_sampler(raw, K, z, torch.zeros(C, K, device="cuda"), torch.zeros(C, device="cuda"), C, nv.MODE_DDPM,
                           coef=coef, x_state=state, step_word=step, seed=1234)
    assert abs(float((state * a).mean())) < 1e-2                          # fresh draw per step


# ------------------------------------------------------------------------------------------------
# backward kernels (training row): weight gradient of the 3x3x3 convolution, data gradient through the
# forward kernel with flipped/transposed weights
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [
    # N, D, H, W, Cin, Cout
    (1, 8, 8, 8, 64, 64),
    (2, 6, 10, 12, 16, 8),        # ragged tiles, channel counts below one chunk
    (1, 16, 16, 16, 128, 64),     # two ci chunks (concat input)
    (2, 4, 4, 4, 72, 136),        # ragged chunk and tile in the channel axes
    (1, 3, 5, 7, 8, 8),
    (2, 6, 6, 6, 256, 256),       # 32 (co, ci) slabs: the fp16 form sums its partitions with the tap-gathering reduce kernel
    (1, 4, 6, 4, 264, 520),       # 81 slabs, ragged last slab in both channel axes
])
def test_conv3_wgrad_matches_torch(dtype, shape):
    N, D, H, W, Cin, Cout = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    cs_in, cs_out = Cin + 8, Cout + 16                 # strided buffers with an offset
    x = torch.randn(N, D, H, W, cs_in, generator=g, device=dev).to(dtype)
    dy = torch.randn(N, D, H, W, cs_out, generator=g, device=dev).to(dtype)
    dw = torch.full((Cout, Cin, 3, 3, 3), 0.5, device=dev)      # the kernels ADD into dw
    _ops().conv3d_k3_wgrad(x, Cin, 8, dy, Cout, 16, dw)
    dw -= 0.5
    xr = x[..., 8:8 + Cin].permute(0, 4, 1, 2, 3).double()
    dyr = dy[..., 16:16 + Cout].permute(0, 4, 1, 2, 3).double()
    want = torch.nn.grad.conv3d_weight(xr, (Cout, Cin, 3, 3, 3), dyr, padding=1)
    scale = want.abs().max().item()
    tol = 2e-5 if dtype == torch.float32 else 2e-3    # fp16 operands are exact products; fp32 accumulation order differs
    assert (dw.double() - want).abs().max().item() <= tol * scale


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 64), (136, 72, 80), (40, 100, 128), (512, 256, 256)])   # Cout, Cin, packed input channels
def test_fp16_weight_packing_through_lds_equals_the_elementwise_kernels(shape):
    """dua_pack_conv3_weights / _dgrad take an LDS-transposing kernel for fp16 with the identity channel map; the one-thread-per-
    element kernels (still used with a channel map and for fp32) must give the same bytes."""
    ops = _ops()
    Cout, Cin, cp = shape
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g).cuda()
    fast, _ = ops.pack_conv3_weights(w, None, torch.float16, cin_packed=cp, pad_bias=False)
    slow, _ = ops.pack_conv3_weights(w, None, torch.float16, cin_packed=cp, perm=list(range(Cin)) + [-1] * (cp - Cin), pad_bias=False)
    assert torch.equal(fast, slow)
    # data gradient: the packing of the flipped, transposed weights
    coutp = -(-Cout // 32) * 32
    dg, _ = ops.pack_conv3_weights_dgrad(w, torch.float16, cout_packed=coutp)
    wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()
    ref, _ = ops.pack_conv3_weights(wt, None, torch.float16, cin_packed=coutp, perm=list(range(Cout)) + [-1] * (coutp - Cout), pad_bias=False)
    assert torch.equal(dg, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C_,split", [(16, 16, False), (6, 64, True)], ids=["epilogue", "splitk-finish"])
def test_fp32_statistics_keep_the_variance_of_a_channel_far_from_zero(S, C_, split):
    """fp32 (parity) convolutions keep a lane's partial sum x / sum x^2 in double: a channel whose mean is hundreds of standard
    deviations (bias 100, output std ~0.2) must still normalise to 1e-4 -- with fp32 partials of x^2 its variance (sum x^2 / n -
    mean^2) came out percent-level wrong, which InstanceNorm's backward amplified into 15 % errors on single channels of weight
    gradients (found against an fp64 oracle).  Reference: InstanceNorm in double of the kernel's OWN raw output, so only
    the statistics are under test; both producers of statistics words (convolution epilogue, split-K finish kernel)."""
    ops = _ops()
    g = torch.Generator().manual_seed(S)
    x = torch.randn(1, S, S, S, C_, generator=g).cuda()
    w = (torch.randn(C_, C_, 3, 3, 3, generator=g) * 0.01).cuda()
    bias = torch.full((C_,), 100.0).cuda()
    wp, bp = ops.pack_conv3_weights(w, bias, torch.float32)
    raw = torch.empty(1, S, S, S, C_, device="cuda")
    stats = ops.stats_buffer(1, C_, "cuda")
    ws = ops.splitk_ws(torch.float32, 1, S, S, S, C_, C_, "cuda") if split else None
    ops.conv3d_k3(x, C_, 0, wp, bp, C_, raw, 0, stats, workspace=ws)
    r64 = raw.double().cpu().flatten(0, 3)                                   # [voxels, C]
    assert float((r64.mean(0).abs() / r64.std(0)).min()) > 100                # the regime under test
    gamma, beta = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g)
    want = (r64 - r64.mean(0)) / torch.sqrt(r64.var(0, unbiased=False) + 1e-5) * gamma.double() + beta.double()
    want = torch.where(want > 0, want, 0.1 * want)
    out = torch.empty_like(raw)
    ops.materialize(raw, C_, ops.Norm(stats, gamma.cuda(), beta.cuda(), S ** 3), out, 0)
    err = float((out.double().cpu().flatten(0, 3) - want).abs().max())
    assert err < 2e-4, err


@pytest.mark.gpu
def test_non_finite_contributions_poison_the_statistics_however_many_there_are():
    """An fp16 overflow upstream reaches many workgroups of a channel: each adds the poison (2^47, -2^47) to the fixed-point
    words instead of its non-finite sums, and k of them must still read as an absurd total (k * 2^47) -- the +-4e18 of round 4
    wrapped modulo 2^64 at the third contribution and could land on plausible finite words."""
    ops = _ops()
    x = torch.randn(1, 32, 32, 32, 16, device="cuda").half()
    x[..., 3] = float("inf")                                   # every output voxel of every channel sees it
    w = torch.randn(64, 16, 3, 3, 3, device="cuda") * 0.05
    wp, bp = ops.pack_conv3_weights(w, torch.zeros(64, device="cuda"), torch.float16)
    raw = torch.empty(1, 32, 32, 32, 64, device="cuda", dtype=torch.float16)
    stats = ops.stats_buffer(1, 64, "cuda")
    ops.conv3d_k3(x, 16, 0, wp, bp, 64, raw, 0, stats)
    sums = ops.stats_decode(stats)[0, :64].cpu()              # [C, 2] float64
    k = sums[:, 0] / 2.0 ** 47
    assert bool((k >= 3).all()) and bool((k == k.round()).all()), k[:8]
    assert torch.equal(sums[:, 1], -sums[:, 0])
    out = torch.empty_like(raw)
    ones = torch.ones(64, device="cuda")
    ops.materialize(raw, 64, ops.Norm(stats, ones, torch.zeros(64, device="cuda"), 32 ** 3), out, 0)
    assert not bool(torch.isfinite(out.float()).any())


@pytest.mark.gpu
def test_the_boundary_carries_the_voxel_count_as_an_integer_beyond_2_to_the_24():
    """dua_in_norm.count (ABI 8) is an integer: a volume of 264 x 256 x 256 = 17 301 504 voxels (not a power of two, above
    2^24 -- where the float 1/count of ABI <= 7 could no longer be inverted exactly) with a channel mean of 300 standard
    deviations normalises to the consumer's own fp32 rounding.  Statistics by dua_instnorm_stats, consumer dua_materialize,
    reference: InstanceNorm in double of the same tensor."""
    ops = _ops()
    D, H, W, C_ = 264, 256, 256, 8
    g = torch.Generator(device="cuda").manual_seed(5)
    raw = torch.randn(1, D, H, W, C_, generator=g, device="cuda") * 0.25 + 75.0
    stats = ops.stats_buffer(1, C_, "cuda")
    ops.instnorm_stats(raw, C_, stats)
    gamma, beta = torch.rand(C_, device="cuda") + 0.5, torch.randn(C_, device="cuda")
    norm = ops.Norm(stats, gamma, beta, D * H * W)
    assert norm.c.count == D * H * W > 2 ** 24
    out = torch.empty_like(raw)
    ops.materialize(raw, C_, norm, out, 0)
    r64 = raw.double().flatten(0, 3)
    mean, var = r64.mean(0), r64.var(0, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 250
    want = (r64 - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    want = torch.where(want > 0, want, 0.1 * want)
    err = float((out.double().flatten(0, 3) - want).abs().max())
    assert err < 2e-4, err


@pytest.mark.gpu
def test_batched_weight_packing_equals_the_per_layer_calls():
    """dua_pack_conv3_weights_batch (ops.ConvPacks): 70 tensors -- both layouts of 35 layers of mixed shapes, i.e. two by-value
    lists -- byte-equal to the per-layer packing calls; tensors the batch form does not take are refused, not mis-packed."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 64), (64, 128), (8, 8), (72, 40), (512, 256), (16, 36), (32, 32)] * 5
    ws = [torch.randn(co, ci, 3, 3, 3, generator=g).cuda() for co, ci in shapes]
    packs = ops.ConvPacks(torch.float16)
    for w in ws:
        packs.add(w, "fwd", w.shape[1] + (8 if w.shape[1] == 40 else 0))       # one layer reads a wider input buffer
        packs.add(w, "dgrad", w.shape[0])
    odd = torch.randn(64, 17, 3, 3, 3, generator=g).cuda()                         # Cin % 4 != 0: the per-layer path
    packs.add(odd, "fwd", 24)
    packs.run()
    assert packs.get(odd, "fwd", 24) is None
    for w in ws:
        cp = w.shape[1] + (8 if w.shape[1] == 40 else 0)
        want, _ = ops.pack_conv3_weights(w, None, torch.float16, cin_packed=cp, pad_bias=False)
        assert torch.equal(packs.get(w, "fwd", cp), want), tuple(w.shape)
        want, _ = ops.pack_conv3_weights_dgrad(w, torch.float16, cout_packed=w.shape[0])
        assert torch.equal(packs.get(w, "dgrad", w.shape[0]), want), tuple(w.shape)
    assert ops.ConvPacks(torch.float32).run().get(ws[0], "fwd", 64) is None


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [32, 1, 2, 64])  # plain k loop; plain block order; one workgroup per CU; 12 waves
def test_conv3_wgrad_launch_variants_agree(variant):
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(variant)
    x = torch.randn(2, 12, 16, 16, 64, generator=g, device=dev).half()
    dy = torch.randn(2, 12, 16, 16, 64, generator=g, device=dev).half()
    base = torch.zeros(64, 64, 3, 3, 3, device=dev)
    _ops().conv3d_k3_wgrad(x, 64, 0, dy, 64, 0, base)
    ops.WGRAD_POLICY = variant
    try:
        other = torch.zeros(64, 64, 3, 3, 3, device=dev)
        _ops().conv3d_k3_wgrad(x, 64, 0, dy, 64, 0, other)
    finally:
        ops.WGRAD_POLICY = 0
    assert (base - other).abs().max().item() <= 2e-4 * base.abs().max().item()     # fp32 summation order only


@pytest.mark.gpu
def test_conv3_wgrad_permuted_input_channels():
    """First denoiser conv: packed input = [x_t(C) | image | pad]; dw comes back in the reference's channel order."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    N, D, H, W, Cin_src, Cout = 1, 8, 8, 8, 5, 16
    Cp = 8
    perm = torch.full((64,), -1, dtype=torch.int32, device=dev)
    perm[:4] = torch.arange(1, 5, dtype=torch.int32)      # packed 0..3 = source 1..4 (x_t), packed 4 = source 0 (image)
    perm[4] = 0
    x = torch.randn(N, D, H, W, Cp, generator=g, device=dev)
    dy = torch.randn(N, D, H, W, Cout, generator=g, device=dev)
    dw = torch.zeros(Cout, Cin_src, 3, 3, 3, device=dev)
    _ops().conv3d_k3_wgrad(x, Cp, 0, dy, Cout, 0, dw, perm=perm)
    src = torch.stack([x[..., 4], x[..., 0], x[..., 1], x[..., 2], x[..., 3]], dim=1).double()
    want = torch.nn.grad.conv3d_weight(src, (Cout, Cin_src, 3, 3, 3), dy.permute(0, 4, 1, 2, 3).double(), padding=1)
    assert (dw.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 64), (1, 6, 10, 4, 24), (2, 4, 4, 4, 136), (1, 2, 2, 2, 512)])
def test_instnorm_lrelu_backward_matches_autograd(dtype, shape):
    """conv stats -> materialize (+ per-(n,c) add) forward, then the reduce/apply backward pair against torch autograd
    of instance_norm -> leaky_relu -> + add on the same raw tensor."""
    ops = _ops()
    N, D, H, W, Cc = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    raw = (torch.randn(N, D, H, W, Cc, generator=g, device=dev) * 1.5 + 0.3).to(dtype)
    gamma = torch.rand(Cc, generator=g, device=dev) + 0.5
    beta = torch.randn(Cc, generator=g, device=dev) * 0.2
    add = torch.randn(N, Cc, generator=g, device=dev)
    dA = torch.randn(N, D, H, W, Cc, generator=g, device=dev).to(dtype)
    V = D * H * W
    rd = raw.double()
    stats = ops.stats_encode(torch.stack([rd.sum((1, 2, 3)), (rd * rd).sum((1, 2, 3))], -1))
    norm = ops.Norm(stats, gamma, beta, V, add=add, add_stride=Cc)
    act = torch.empty_like(raw)
    ops.materialize(raw, Cc, norm, act, 0)
    dY = torch.empty_like(raw)
    dgamma, dbeta, dadd = ops.instnorm_bwd(dA, 0, raw, Cc, norm, dY)
    # reference: autograd in fp64 on NCDHW
    r = rd.permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    gm, bt, ad = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), add.double().requires_grad_(True)
    a = F.leaky_relu(F.instance_norm(r, weight=gm, bias=bt, eps=1e-5), 0.1) + ad[:, :, None, None, None]
    a.backward(dA.double().permute(0, 4, 1, 2, 3))
    tol = 1e-4 if dtype == torch.float32 else 4e-3
    assert (act.double().permute(0, 4, 1, 2, 3) - a.detach()).abs().max().item() < (1e-5 if dtype == torch.float32 else 2e-2)
    want = r.grad.permute(0, 2, 3, 4, 1)
    assert (dY.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    assert dgamma.dtype == dbeta.dtype == dadd.dtype == torch.float32            # emitted by the apply launch itself
    assert torch.allclose(dadd.double(), ad.grad, rtol=1e-4, atol=1e-4 * V ** 0.5)
    assert torch.allclose(dbeta.double(), bt.grad, rtol=1e-4, atol=1e-4 * V ** 0.5)
    assert torch.allclose(dgamma.double(), gm.grad, rtol=1e-4, atol=1e-4 * V ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 64, 64, 64, 64, 64), (2, 64, 32, 64, 128, 64)])
def test_data_gradient_launch_takes_the_norm_backward_sums_along(shape):
    """dua_conv3d_k3_dgrad_reduce (wide-tile form, training): the same dx as the plain launch, bit for bit, and the three sums of
    the owner layer's InstanceNorm backward as the separate reduce launch computes them from that dx (fp32 partial sums over
    different element subsets: 1e-5 relative); the apply pass on either agrees, parameter gradients included."""
    ops = _ops()
    N, D, H, W, Cin, Cout = shape                  # the launch: dy [.., Cin] -> dx [.., Cout]; Cout = the owner layer's channels
    dev, dt = torch.device("cuda:0"), torch.float16
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    dy = (torch.randn(N, D, H, W, Cin, generator=g, device=dev) * 0.5).to(dt)
    w = torch.randn(Cin, Cout, 3, 3, 3, generator=g, device=dev) / (27 * Cin) ** 0.5       # the forward layer's [cout_fwd = Cin, cin_fwd = Cout]
    raw = (torch.randn(N, D, H, W, Cout, generator=g, device=dev) * 1.5 + 0.3).to(dt)
    gamma = torch.rand(Cout, generator=g, device=dev) + 0.5
    beta = torch.randn(Cout, generator=g, device=dev) * 0.2
    rd = raw.double()
    stats = ops.stats_encode(torch.stack([rd.sum((1, 2, 3)), (rd * rd).sum((1, 2, 3))], -1))
    norm = ops.Norm(stats, gamma, beta, D * H * W)
    assert ops.conv3d_k3_dgrad_reduce_supported(dt, N, D, H, W, Cin, Cout)
    assert not ops.conv3d_k3_dgrad_reduce_supported(dt, 1, 16, 16, 16, Cin, Cout)          # too few tiles for the wide-tile form
    wp, bp = ops.pack_conv3_weights_dgrad(w, dt, cout_packed=Cin)
    dx0 = torch.empty(N, D, H, W, Cout, device=dev, dtype=dt)
    ops.conv3d_k3(dy, Cin, 0, wp, bp, Cout, dx0, 0, ops.stats_buffer(N, Cout, dev), workspace=ops.splitk_ws(dt, N, D, H, W, Cin, Cout, dev))
    dx1 = torch.empty_like(dx0)
    sums = ops.instnorm_bwd_sums(raw, norm)
    ops.conv3d_k3_dgrad_reduce(dy, Cin, wp, bp, Cout, dx1, raw, norm, sums)
    assert torch.equal(dx0, dx1)
    dY0, dY1 = torch.empty_like(raw), torch.empty_like(raw)
    g0 = ops.instnorm_bwd(dx0, 0, raw, Cout, norm, dY0, want_add=True)
    g1 = ops.instnorm_bwd(dx1, 0, raw, Cout, norm, dY1, want_add=True, sums=sums)
    scale = max(1.0, dY0.abs().max().item())
    assert (dY0.double() - dY1.double()).abs().max().item() <= 2e-3 * scale
    for a_, b_ in zip(g0, g1):
        assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-4 * (D * H * W) ** 0.5)
    zh = (rd - rd.mean((1, 2, 3), keepdim=True)) / (rd.var((1, 2, 3), unbiased=False, keepdim=True) + 1e-5).sqrt()
    z = zh * gamma.double() + beta.double()
    dz = torch.where(z > 0, dx1.double(), dx1.double() * 0.1)
    want = torch.stack([dx1.double().sum((1, 2, 3)), dz.sum((1, 2, 3)), (dz * zh).sum((1, 2, 3))], -1)
    got = sums.sum(1)[:, :Cout, :3]
    assert torch.allclose(got, want, rtol=2e-3, atol=2e-3 * (D * H * W) ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 16, 8, 8, 8), (1, 3, 5, 6, 7), (2, 13, 16, 8, 4), (1, 24, 9, 10, 11)])   # 16, 24: eight channels per read (fp16)
def test_seg_loss_and_gradient_match_torch(dtype, shape):
    """Fused mse+bce+dice loss and its gradient against the oracle's restatement of losses/loss.py (fp64)."""
    from oracle.train_ref import RefLoss as Loss
    ops = _ops()
    N, Cc, D, H, W = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    logits = (torch.randn(N, D, H, W, Cc, generator=g, device=dev) * 3).to(dtype)
    labels = (torch.rand(N, Cc, D, H, W, generator=g, device=dev) > 0.7).float()
    L, sums, _ = ops.seg_loss_reduce(logits, labels)
    gs = torch.tensor(3.0, device=dev)
    dl = ops.seg_loss_grad(logits, labels, sums, gs)
    p = logits.double().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    want = Loss("mse,bce,dice", "sum")(p, labels.double())
    (want * 3.0).backward()
    assert abs(float(L) - float(want)) < 1e-5 * max(1.0, abs(float(want)))
    wg = p.grad.permute(0, 2, 3, 4, 1)
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert (dl.double() - wg).abs().max().item() <= tol * wg.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_maxpool_backward_add_matches_torch(dtype):
    """Ties included (fp16 values on a coarse grid): the routed gradient must follow torch's first-maximum rule."""
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    N, D, H, W, Cc = 2, 4, 6, 8, 24
    act = (torch.randint(-3, 4, (N, D, H, W, Cc + 8), generator=g, device=dev).float() * 0.5).to(dtype)   # many ties
    dA = torch.randn(N, D, H, W, Cc + 16, generator=g, device=dev).to(dtype)
    dP = torch.randn(N, D // 2, H // 2, W // 2, Cc, generator=g, device=dev).to(dtype)
    out = ops.maxpool2_bwd_add(act, 8, Cc, dA, 16, dP)
    a = act[..., 8:8 + Cc].permute(0, 4, 1, 2, 3).double().contiguous().requires_grad_(True)
    F.max_pool3d(a, 2).backward(dP.permute(0, 4, 1, 2, 3).double())
    want = a.grad.permute(0, 2, 3, 4, 1) + dA[..., 16:16 + Cc].double()
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert (out.double() - want).abs().max().item() <= tol
    out0 = ops.maxpool2_bwd_add(act, 8, Cc, None, 0, dP)
    assert (out0.double() - a.grad.permute(0, 2, 3, 4, 1)).abs().max().item() <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 64, 16), (1, 3, 5, 7, 8, 2), (2, 6, 4, 10, 32, 13), (1, 5, 7, 9, 64, 16),
                                   (2, 4, 4, 5, 64, 11), (1, 16, 24, 20, 64, 16)])
def test_head_forward_backward_match_torch(dtype, shape):
    ops = _ops()
    N, D, H, W, Cc, K = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    u = torch.randn(N, D, H, W, Cc, generator=g, device=dev).to(dtype)
    w = torch.randn(K, Cc, generator=g, device=dev) * 0.2
    b = torch.randn(K, generator=g, device=dev)
    dl = torch.randn(N, D, H, W, K, generator=g, device=dev).to(dtype)
    out = ops.head_fwd(u, w, b)
    du, dW, db = ops.head_bwd(dl, u, w)
    ud, wd, bd = u.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    want = ud @ wd.t() + bd
    want.backward(dl.double())
    ftol = 1e-5 if dtype == torch.float32 else 4e-3
    assert (out.double() - want.detach()).abs().max().item() <= ftol * max(1.0, want.abs().max().item())
    assert (du.double() - ud.grad).abs().max().item() <= ftol * max(1.0, ud.grad.abs().max().item())
    assert torch.allclose(dW.double(), wd.grad, rtol=1e-4, atol=1e-4 * wd.grad.abs().max().item())
    assert torch.allclose(db.double(), bd.grad, rtol=1e-4, atol=1e-4 * bd.grad.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_conv3_random_shapes_forward_dgrad_wgrad(seed):
    """Seeded random geometry (ragged tiles in every axis, channel counts that are not multiples of a chunk or a 64-wide
    tile, batch 1-3, both dtypes, with and without a split-K workspace): forward, the data gradient through the forward
    kernel with flipped/transposed weights, and the weight gradient, each against torch in fp64 on the same operands."""
    import random
    ops = _ops()
    rnd = random.Random(1000 + seed)
    dtype = torch.float16 if seed % 2 else torch.float32
    N = rnd.choice([1, 1, 2, 3])
    D, H, W = (rnd.choice([1, 2, 3, 5, 6, 8, 9, 12, 17]) for _ in range(3))
    Cin, Cout = rnd.choice([8, 16, 24, 40, 64, 72, 136]), rnd.choice([8, 16, 24, 64, 72, 136])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(N, D, H, W, Cin, generator=g, device=dev).to(dtype)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g, device=dev) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g, device=dev)
    dy = torch.randn(N, D, H, W, Cout, generator=g, device=dev).to(dtype)

    def conv(inp, weight, bias, cout):
        wp, bp = ops.pack_conv3_weights(weight.contiguous(), bias, dtype, cin_packed=inp.shape[-1])
        out = torch.empty((*inp.shape[:4], cout), dtype=dtype, device=dev)
        nb = ops.conv3_workspace_bytes(dtype, N, D, H, W, inp.shape[-1], cout)
        ws = torch.empty(max(nb, 16) // 4, device=dev) if (nb > 0 and seed % 3) else None
        ops.conv3d_k3(inp, inp.shape[-1], 0, wp, bp, cout, out, 0, ops.stats_buffer(N, cout, dev), workspace=ws)
        return out

    wq = w.to(dtype).double()
    xd = x.double().permute(0, 4, 1, 2, 3)
    dyd = dy.double().permute(0, 4, 1, 2, 3)
    tol = 2e-5 if dtype == torch.float32 else 3e-3

    y = conv(x, w, b, Cout)
    want = F.conv3d(xd, wq, b.double(), padding=1).permute(0, 2, 3, 4, 1)
    assert (y.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), ("fwd", N, D, H, W, Cin, Cout)

    if seed % 4 < 2:        # weights flipped/transposed by torch, packed as a forward convolution
        dx = conv(dy, w.flip(2, 3, 4).transpose(0, 1), None, Cin)
    else:                   # packed straight from the forward weights (dua_pack_conv3_weights_dgrad)
        wp, bp = ops.pack_conv3_weights_dgrad(w.contiguous(), dtype, cout_packed=Cout)
        dx = torch.empty((N, D, H, W, Cin), dtype=dtype, device=dev)
        ops.conv3d_k3(dy, Cout, 0, wp, bp, Cin, dx, 0, ops.stats_buffer(N, Cin, dev))
    want = torch.nn.grad.conv3d_input(xd.shape, wq, dyd, padding=1).permute(0, 2, 3, 4, 1)
    assert (dx.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), ("dgrad", N, D, H, W, Cin, Cout)

    dw = torch.zeros(Cout, Cin, 3, 3, 3, device=dev)
    ops.conv3d_k3_wgrad(x, Cin, 0, dy, Cout, 0, dw)
    want = torch.nn.grad.conv3d_weight(xd, (Cout, Cin, 3, 3, 3), dyd, padding=1)
    assert (dw.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), ("wgrad", N, D, H, W, Cin, Cout)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [
    # N, D, H, W (input), Cin, Cout
    (1, 4, 4, 4, 64, 64),
    (2, 3, 5, 6, 16, 8),          # ragged tile, channels below one chunk / tile
    (1, 6, 6, 6, 136, 72),        # several chunks and tiles in both channel axes
    (2, 8, 8, 8, 64, 32),         # more than one 128-voxel tile per sample
])
def test_deconv_backward_matches_torch(dtype, shape):
    """Data and weight gradient of ConvTranspose3d(k2, s2), dy read in place from a channel slice of a wider buffer."""
    ops = _ops()
    N, D, H, W, Cin, Cout = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    x = torch.randn(N, D, H, W, Cin + 8, generator=g, device=dev).to(dtype)
    dcat = torch.randn(N, 2 * D, 2 * H, 2 * W, Cout + 24, generator=g, device=dev).to(dtype)
    w = torch.randn(Cin, Cout, 2, 2, 2, generator=g, device=dev) / (Cin ** 0.5)
    dx, dw = ops.deconv_k2s2_bwd(x, Cin, 8, dcat, Cout, 24, w)
    xr = x[..., 8:].permute(0, 4, 1, 2, 3).double().contiguous().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    y = F.conv_transpose3d(xr, wr, stride=2)
    y.backward(dcat[..., 24:].permute(0, 4, 1, 2, 3).double())
    tol = 2e-5 if dtype == torch.float32 else 3e-3
    want_dx = xr.grad.permute(0, 2, 3, 4, 1)
    assert (dx.double() - want_dx).abs().max().item() <= tol * max(1.0, want_dx.abs().max().item())
    assert (dw.double() - wr.grad).abs().max().item() <= tol * max(1.0, wr.grad.abs().max().item())


_FRESH_CAPTURE = r"""
import sys, threading
sys.path.insert(0, {root!r})
import torch
from diff_unet_amos_amd import _native as nv, ops
dev = torch.device("cuda:0")
errs = []
def prep():
    try:
        nv.prepare(dev)
    except Exception as e:            # noqa: BLE001
        errs.append(e)
ts = [threading.Thread(target=prep) for _ in range(8)]          # concurrent first use: one of them does the work
[t.start() for t in ts]; [t.join() for t in ts]
assert not errs, errs
g = torch.Generator().manual_seed(5)
N, cin, cout, S = 1, 64, 128, 24                                  # a 24^3 layer: the kd-plane form (160 KB of dynamic LDS)
dtype = {dtype}
x = torch.randn(N, S, S, S, cin, generator=g).to(dev, dtype)
w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(dev)
b = torch.randn(cout, generator=g).to(dev)
wp, bp = ops.pack_conv3_weights(w, b, dtype)
torch.cuda.synchronize()
def run(y, st, dw):
    ops.conv3d_k3(x, cin, 0, wp, bp, cout, y, 0, st)
    ops.conv3d_k3_wgrad(x, cin, 0, y, cout, 0, dw)                # y as the output gradient: any tensor of that shape
ya, sa, dwa = torch.empty(N, S, S, S, cout, device=dev, dtype=dtype), ops.stats_buffer(N, cout, dev), torch.zeros_like(w)
yb, sb, dwb = torch.empty_like(ya), ops.stats_buffer(N, cout, dev), torch.zeros_like(w)
# NO warm-up launch: the very first launch of these kernels in this process is recorded into a graph
cap = torch.cuda.Stream(dev)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=cap):
    run(ya, sa, dwa)
gr.replay()
torch.cuda.synchronize()
run(yb, sb, dwb)
torch.cuda.synchronize()
assert torch.equal(ya, yb) and torch.equal(sa, sb)
assert torch.allclose(dwa, dwb, rtol=1e-3, atol=1e-3 * float(dwb.abs().max()))
print("fresh-capture ok", float(ya.float().abs().max()))
"""


@pytest.mark.parametrize("dtype", ["torch.float16", "torch.float32"])
def test_first_launch_inside_a_capture_in_a_fresh_process(dtype):
    """Regression guard for the training-graph crash of round 3: after dua_prepare() -- called concurrently
    from eight threads here -- the FIRST launch of the 160 KB kd-plane convolution and of the weight-gradient kernel may be a
    captured one; no function attribute is set lazily inside the capture.  A process of its own, so that nothing warmed the
    kernels up."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", _FRESH_CAPTURE.format(root=root, dtype=dtype)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fresh-capture ok" in r.stdout
