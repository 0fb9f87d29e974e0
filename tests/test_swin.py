"""DiffSwinUNETR pieces (BASELINE config 5): the windowed-attention kernel against the oracle's
restatement of models/swin_unetr/attention.py (PARITY UNPINNED: MONAI absent, see oracle/swin_ref.py), and the
oracle's own invariants on the CPU."""
import pytest
import torch

from oracle.swin_ref import (RefPatchMerging, RefWindowAttention, compute_mask, get_window_size, patch_merging_gather,
                             relative_position_index, window_partition, window_reverse)


def test_oracle_window_round_trip_and_mask_structure():
    x = torch.arange(2 * 14 * 7 * 14 * 3, dtype=torch.float32).view(2, 14, 7, 14, 3)
    ws = (7, 7, 7)
    w = window_partition(x, ws)
    assert w.shape == (2 * 2 * 1 * 2, 343, 3)
    assert torch.equal(window_reverse(w, ws, (2, 14, 7, 14)), x)
    m = compute_mask((14, 14, 14), ws, (3, 3, 3))
    assert m.shape == (8, 343, 343) and set(m.unique().tolist()) == {-100.0, 0.0}
    assert torch.equal(m, m.transpose(1, 2)) and bool((m.diagonal(dim1=1, dim2=2) == 0).all())
    assert float(m[0].abs().sum()) == 0.0                       # the first window is not cut by the shift
    assert get_window_size((6, 6, 6), (7, 7, 7), (3, 3, 3)) == ((6, 6, 6), (0, 0, 0))
    idx = relative_position_index(ws)
    assert idx.shape == (343, 343) and int(idx.min()) == 0 and int(idx.max()) == 13 ** 3 - 1
    assert int(idx[0, 0]) == (13 ** 3 - 1) // 2


def test_oracle_legacy_patch_merging_duplicates():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 6, 5, 3, generator=g)                  # odd W: padded
    a, b = patch_merging_gather(x, legacy=True), patch_merging_gather(x, legacy=False)
    assert a.shape == b.shape == (1, 2, 3, 3, 24)
    assert torch.equal(a[..., 15:18], a[..., 6:9]) and torch.equal(a[..., 18:21], a[..., 9:12])   # x5 == x2, x6 == x3
    assert not torch.equal(a, b)
    assert RefPatchMerging(3)(x).shape == (1, 2, 3, 3, 6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.float16, 2e-3)])      # measured 1e-4 .. 4e-4
@pytest.mark.parametrize("heads,ws,dims,shift", [
    (3, (7, 7, 7), (14, 14, 14), (3, 3, 3)),     # stage 0 geometry in miniature: 8 shifted windows of 343 tokens
    (6, (7, 7, 7), (7, 14, 7), None),            # unshifted block, two windows
    (24, (6, 6, 6), (6, 6, 6), None),            # coarsest level: the map is smaller than the window (216 tokens)
    (3, (3, 4, 5), (6, 8, 5), (1, 2, 0)),        # odd window extents, ragged token count (60)
])
def test_window_attention_kernel_matches_oracle(dtype, tol, heads, ws, dims, shift):
    from diff_unet_amos_amd import ops
    torch.manual_seed(heads)
    dim = heads * 16
    att = RefWindowAttention(dim, heads, ws, qkv_bias=True)
    with torch.no_grad():
        att.relative_position_bias_table.normal_(0, 0.5)         # exercise the bias path with visible values
    g = torch.Generator().manual_seed(7)
    B = 2
    x = torch.randn(B, *dims, dim, generator=g)
    mask = compute_mask(dims, ws, shift) if shift is not None else None
    xw = window_partition(x, ws)                                  # [B * nw, n, c]
    n = xw.shape[1]
    with torch.no_grad():
        qkv = att.qkv(xw)
        want = att.attention_core(qkv, mask)
    bias_t = att.bias(n).detach().transpose(1, 2).contiguous().cuda()
    mask_t = mask.transpose(1, 2).contiguous().cuda() if mask is not None else None
    got = ops.window_attention(qkv.to(dtype).cuda().contiguous(), heads, bias_t, mask_t,
                               windows_per_image=mask.shape[0] if mask is not None else 1)
    assert got.shape == want.shape and got.dtype == dtype
    d = (got.float().cpu() - want).abs()
    print(f"\n[{dtype}] heads {heads} window {ws} n {n}: max |d| {d.max():.2e} mean {d.mean():.2e} (|out| max {want.abs().max():.2f})")
    assert d.max() < tol * max(1.0, float(want.abs().max())), float(d.max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 4e-3)])
@pytest.mark.parametrize("shape,legacy", [((2, 6, 8, 10, 48), True), ((1, 5, 7, 6, 96), True), ((1, 4, 4, 4, 16), False), ((1, 2, 2, 2, 384), True)])
def test_patch_merge_norm_kernel_matches_oracle(dtype, tol, shape, legacy):
    """Gather (legacy duplicates, zero padding of odd extents) + LayerNorm(8C); the reduction Linear stays a GEMM."""
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    pm = RefPatchMerging(shape[-1], legacy=legacy)
    with torch.no_grad():
        pm.norm.weight.normal_(1.0, 0.2); pm.norm.bias.normal_(0.0, 0.2)
        y = (0.1 * torch.randn(*shape, generator=g)).to(dtype)                    # the last block's MLP output, still to be added
        want = pm.norm(patch_merging_gather(x + y.float(), legacy))
        full = pm(x + y.float())
    got = ops.patch_merge_norm(x.cuda(), pm.norm.weight.detach().cuda(), pm.norm.bias.detach().cuda(), legacy=legacy,
                               y=y.cuda(), dtype=dtype)
    assert got.shape == want.shape
    assert (got.float().cpu() - want).abs().max() < tol * max(1.0, float(want.abs().max()))
    with torch.no_grad():
        red = torch.nn.functional.linear(got.float().cpu(), pm.reduction.weight)
    assert (red - full).abs().max() < 10 * tol * max(1.0, float(full.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 2e-2)])
@pytest.mark.parametrize("cin,cout", [(48, 48), (16, 48)])
def test_unet_res_block_on_hip_kernels_matches_oracle(dtype, tol, cin, cout):
    """UnetResBlock.forward (blocks.py:298-316) composed from the HIP kernels: conv3d_k3 (raw + statistics), conv3d_k3
    with the producer's InstanceNorm + LeakyReLU(0.01) + t_proj add fused into its input staging, and the residual tail
    kernel; the 1x1x1 conv3 of channel-changing blocks is a library GEMM.  Against the oracle's torch.nn restatement."""
    from diff_unet_amos_amd import ops
    from oracle.swin_ref import RefUnetResBlock, nonlinearity
    torch.manual_seed(cin + cout)
    blk = RefUnetResBlock(cin, cout, affine=True).eval()
    with torch.no_grad():
        for m in (blk.norm1, blk.norm2) + ((blk.norm3,) if blk.downsample else ()):
            m.weight.normal_(1.0, 0.3); m.bias.normal_(0.0, 0.3)
    g = torch.Generator().manual_seed(5)
    N, D, H, W = 2, 8, 16, 8
    x = torch.randn(N, cin, D, H, W, generator=g)
    t = torch.randn(N, 512, generator=g)
    with torch.no_grad():
        want = blk(x, t)
    dev = "cuda"
    V = D * H * W
    xcl = x.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(dev)
    zb = torch.zeros(cout, device=dev)
    w1, b1 = ops.pack_conv3_weights(blk.conv1.conv.weight.detach().to(dev), zb, dtype)
    w2, b2 = ops.pack_conv3_weights(blk.conv2.conv.weight.detach().to(dev), zb, dtype)
    raw1 = torch.empty(N, D, H, W, cout, dtype=dtype, device=dev); st1 = ops.stats_buffer(N, cout, dev)
    ops.conv3d_k3(xcl, cin, 0, w1, b1, cout, raw1, 0, st1)
    with torch.no_grad():
        add = blk.t_proj(nonlinearity(t)).to(dev).float().contiguous()
    n1 = ops.Norm(st1, blk.norm1.weight.detach().to(dev), blk.norm1.bias.detach().to(dev), V, add=add, add_stride=cout, slope=0.01)
    raw2 = torch.empty_like(raw1); st2 = ops.stats_buffer(N, cout, dev)
    ops.conv3d_k3(raw1, cout, 0, w2, b2, cout, raw2, 0, st2, norm=n1)
    n2 = ops.Norm(st2, blk.norm2.weight.detach().to(dev), blk.norm2.bias.detach().to(dev), V, slope=0.01)
    if blk.downsample:
        r = (xcl.float() @ blk.conv3.conv.weight.detach().reshape(cout, cin).t().to(dev)).to(dtype).contiguous()     # 1x1x1 conv = GEMM
        st3 = ops.instnorm_stats(r.view(N, D, H, W, cout), cout, ops.stats_buffer(N, cout, dev))
        rf = r.float().view(N, V, cout)
        sd = ops.stats_decode(st3)
        assert torch.allclose(sd[:, :cout, 0], rf.sum(1).double(), rtol=1e-5, atol=1e-3)
        assert torch.allclose(sd[:, :cout, 1], (rf * rf).sum(1).double(), rtol=1e-5, atol=1e-3)
        n3 = ops.Norm(st3, blk.norm3.weight.detach().to(dev), blk.norm3.bias.detach().to(dev), V, slope=0.01)
        out = ops.residual_norm_act(raw2, n2, r, n3, slope=0.01)
        assert torch.equal(out, ops.residual_norm_act(raw2, n2, r, n3, slope=0.01, background=True))   # one workgroup per CU: same values
    else:
        out = ops.residual_norm_act(raw2, n2, xcl, None, slope=0.01)
        assert torch.equal(out, ops.residual_norm_act(raw2, n2, xcl, None, slope=0.01, background=True))
    got = out.float().permute(0, 4, 1, 2, 3).cpu()
    d = (got - want).abs()
    print(f"\n[{dtype}] UnetResBlock {cin}->{cout}: max |d| {d.max():.2e} mean {d.mean():.2e}")
    assert d.max() < tol * max(1.0, float(want.abs().max()))


def test_region_ids_reproduce_compute_mask():
    """The uint8 region ids the attention kernel takes are what compute_mask (attention.py:123-160) builds its 0 / -100
    mask from; clip_window is get_window_size."""
    from diff_unet_amos_amd.swin_engine import clip_window, region_ids
    for dims, ws, ss in (((14, 14, 14), (7, 7, 7), (3, 3, 3)), ((8, 12, 10), (4, 6, 5), (2, 3, 2)), ((6, 8, 5), (3, 4, 5), (1, 2, 0))):
        reg = region_ids(dims, ws, ss).to(torch.int32)
        m = torch.where(reg[:, None, :] != reg[:, :, None], torch.tensor(-100.0), torch.tensor(0.0))
        assert torch.equal(m, compute_mask(dims, ws, ss))
    assert clip_window((6, 6, 6), (7, 7, 7), (3, 3, 3)) == get_window_size((6, 6, 6), (7, 7, 7), (3, 3, 3))
    assert clip_window((48, 48, 48), (7, 7, 7), (3, 3, 3)) == ((7, 7, 7), (3, 3, 3))


def test_diff_swin_unetr_state_dict_keys_match_the_reference_tree():
    """Same module tree as models/diff_swin_unetr.py (keys restated in the oracle from the reference's constructors)."""
    from diff_unet_amos_amd.diff_swin_unetr import DiffSwinUNETR
    from oracle.swin_ref import make_ref_diff_swin_unetr
    net = DiffSwinUNETR(in_channels=1, out_channels=16, feature_size=48)
    ref = make_ref_diff_swin_unetr(1, 16, 48)
    a, b = net.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(a[k].shape == b[k].shape for k in a)
    for k in ("model.swinViT.layers1.0.blocks.1.attn.relative_position_bias_table", "model.swinViT.t_proj.4.weight",
              "model.decoder5.transp_conv.conv.weight", "model.encoder10.layer.t_proj.bias", "model.out.conv.conv.bias",
              "embed_model.swinViT.layers4.0.downsample.reduction.weight", "embed_model.encoder1.layer.conv3.conv.weight",
              "model.t_embedder.dense.1.weight"):
        assert k in a, k
    assert not any("norm1.weight" in k and "encoder" in k for k in a)      # InstanceNorm3d("instance"): no parameters
    assert torch.equal(a["model.swinViT.layers1.0.blocks.0.attn.relative_position_index"], relative_position_index((7, 7, 7)))
    with pytest.raises(NotImplementedError):
        net(image=torch.zeros(1, 1, 64, 64, 64), x=torch.zeros(1, 16, 64, 64, 64, requires_grad=True),
            step=torch.zeros(1, dtype=torch.long), pred_type="denoise")
    with pytest.raises(RuntimeError, match="no CPU path"):
        with torch.no_grad():
            net(image=torch.zeros(1, 1, 64, 64, 64), x=torch.zeros(1, 16, 64, 64, 64), step=torch.zeros(1, dtype=torch.long),
                pred_type="denoise")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3)])
@pytest.mark.parametrize("dims,C_,ws,ss", [((9, 8, 10), 48, (7, 7, 7), (3, 3, 3)), ((6, 6, 6), 384, (6, 6, 6), (0, 0, 0)),
                                           ((8, 14, 7), 96, (7, 7, 7), (3, 3, 0)), ((4, 6, 5), 192, (4, 3, 5), (2, 1, 0))])
def test_window_gather_and_scatter_kernels_match_oracle(dtype, tol, dims, C_, ws, ss):
    """norm1 -> pad -> roll -> window_partition, and window_reverse -> roll back -> crop -> + shortcut -> norm2
    (transformer.py:378-434, 475-476) against the oracle's torch restatement."""
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(C_ + dims[0])
    B = 2
    x = torch.randn(B, *dims, C_, generator=g)
    yprev = (0.3 * torch.randn(B, *dims, C_, generator=g)).to(dtype)
    g1, b1, g2, b2 = (torch.randn(C_, generator=g) * 0.3 + (1.0 if k % 2 == 0 else 0.0) for k in range(4))
    pad = [(ws[i] - dims[i] % ws[i]) % ws[i] for i in range(3)]
    x1 = x + yprev.float()
    n1 = F.pad(F.layer_norm(x1, [C_], g1, b1), (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    shifted = torch.roll(n1, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3)) if any(ss) else n1
    want_win = window_partition(shifted, ws)
    geom = ops.window_geom(B, dims, C_, ws, ss)
    xd = x.cuda().contiguous()
    win = torch.empty(want_win.shape, dtype=dtype, device="cuda")
    ops.window_gather_norm(xd, geom, g1.cuda(), b1.cuda(), win, y=yprev.cuda())
    assert torch.allclose(xd.cpu(), x1, atol=1e-6)                                  # the stream took the MLP output
    assert (win.float().cpu() - want_win).abs().max() < tol * max(1.0, float(want_win.abs().max()))
    # the way back, with an arbitrary "attention output" per window token
    yw = torch.randn(want_win.shape, generator=g).to(dtype)
    dp = [dims[i] + pad[i] for i in range(3)]
    back = window_reverse(yw.float(), ws, [B, *dp])
    if any(ss):
        back = torch.roll(back, shifts=ss, dims=(1, 2, 3))
    x2 = x1 + back[:, :dims[0], :dims[1], :dims[2], :]
    want_ln2 = F.layer_norm(x2, [C_], g2, b2)
    ln2 = torch.empty((B, *dims, C_), dtype=dtype, device="cuda")
    ops.window_scatter_add_norm(xd, geom, yw.cuda(), g2.cuda(), b2.cuda(), ln2)
    assert torch.allclose(xd.cpu(), x2, atol=1e-5)
    assert (ln2.float().cpu() - want_ln2).abs().max() < tol * max(1.0, float(want_ln2.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float16, 6e-3)])
def test_patch_embed_stage_out_gelu_kernels_match_oracle(dtype, tol):
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    from oracle.swin_ref import RefSwinTransformer
    g = torch.Generator().manual_seed(3)
    B, D, H, W, cin, cp, E = 2, 8, 6, 10, 17, 24, 48
    x = torch.randn(B, cin, D, H, W, generator=g)
    conv = torch.nn.Conv3d(cin, E, 2, 2)
    tadd = torch.randn(B, 64, generator=g)
    emb = torch.randn(B, E, D // 2, H // 2, W // 2, generator=g).to(dtype)
    xq = x.to(dtype).float()
    with torch.no_grad():
        x0 = conv(xq) + tadd[:, 8:8 + E, None, None, None]
        want = RefSwinTransformer.proj_out(x0, True) + emb.float()
    xin = torch.zeros(B, D, H, W, 32, dtype=dtype, device="cuda")
    xin[..., :cin] = x.permute(0, 2, 3, 4, 1).to(dtype).cuda()
    wp = ops.pack_patch_embed_weights(conv.weight.detach().cuda(), cp)
    out = torch.zeros(B, D // 2, H // 2, W // 2, 2 * E, dtype=dtype, device="cuda")
    stream = torch.empty(B, D // 2, H // 2, W // 2, E, dtype=torch.float32, device="cuda")
    ops.patch_embed(xin, cp, wp, conv.bias.detach().cuda().contiguous(), out, E, tadd=tadd.cuda()[:, 8:8 + E],
                    emb=emb.permute(0, 2, 3, 4, 1).contiguous().cuda(), x=stream)
    # fp16 mode contracts on MFMA with the weights rounded to fp16 (fp32 accumulation); fp32 mode is an fmaf chain
    assert (stream.cpu().permute(0, 4, 1, 2, 3) - x0).abs().max() < (1e-4 if dtype == torch.float32 else 1e-3) * max(1.0, float(x0.abs().max()))
    got = out[..., E:].float().cpu().permute(0, 4, 1, 2, 3)
    assert float(out[..., :E].abs().max()) == 0.0
    assert (got - want).abs().max() < tol * max(1.0, float(want.abs().max()))
    # stage_out on every supported width
    for C_ in (48, 96, 192, 384, 768):
        y = torch.randn(B * 10, C_, generator=g).to(dtype)
        ta = torch.randn(B, C_ + 8, generator=g)
        e2 = torch.randn(B * 10, C_, generator=g).to(dtype)
        xs = y.float().view(B, 10, C_) + ta[:, None, 8:]
        want2 = F.layer_norm(xs, [C_]).view(-1, C_) + e2.float()
        o2 = torch.zeros(B * 10, C_ + 8, dtype=dtype, device="cuda")
        st = torch.empty(B * 10, C_, dtype=torch.float32, device="cuda")
        ops.stage_out(y.cuda(), B, C_, o2, 8, tadd=ta.cuda()[:, 8:], emb=e2.cuda(), x=st)
        assert torch.allclose(st.cpu().view(B, 10, C_), xs, atol=1e-5)
        assert (o2[:, 8:].float().cpu() - want2).abs().max() < tol * max(1.0, float(want2.abs().max())), C_
    h = torch.randn(4096, generator=g).to(dtype) * 3
    assert (ops.gelu_(h.clone().cuda()).float().cpu() - F.gelu(h.float())).abs().max() < (3e-6 if dtype == torch.float32 else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("heads,table_ws,ws,dims", [(3, (7, 7, 7), (7, 7, 7), (7, 14, 7)), (24, (7, 7, 7), (6, 6, 6), (6, 6, 6)),
                                                    (6, (7, 7, 7), (2, 2, 2), (2, 2, 2))])
def test_window_attention_bias_table_form_equals_dense_bias(heads, table_ws, ws, dims):
    """bias_table + in-kernel relative_position_index == the dense bias the oracle gathers (attention.py:103-106), also
    for clipped windows, where the reference slices the 7^3 index to [:n, :n]."""
    from diff_unet_amos_amd import ops
    torch.manual_seed(heads)
    att = RefWindowAttention(heads * 16, heads, table_ws, qkv_bias=True)
    with torch.no_grad():
        att.relative_position_bias_table.normal_(0, 0.5)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, *dims, heads * 16, generator=g)
    xw = window_partition(x, ws)
    n = xw.shape[1]
    with torch.no_grad():
        qkv = att.qkv(xw)
        want = att.attention_core(qkv, None)
    qd = qkv.half().cuda().contiguous()
    dense = ops.window_attention(qd, heads, att.bias(n).detach().transpose(1, 2).contiguous().cuda())
    table = ops.window_attention(qd, heads, None, bias_table=att.relative_position_bias_table.detach().t().contiguous().cuda(),
                                 table_grid=table_ws)
    assert torch.equal(dense, table)
    assert (table.float().cpu() - want).abs().max() < 2e-3 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_window_attention_region_ids_equal_the_dense_mask():
    from diff_unet_amos_amd import ops
    from diff_unet_amos_amd.swin_engine import region_ids
    g = torch.Generator().manual_seed(11)
    heads, ws, dims, ss = 3, (7, 7, 7), (14, 14, 14), (3, 3, 3)
    mask = compute_mask(dims, ws, ss)
    n, nw = 343, mask.shape[0]
    qkv = torch.randn(2 * nw, n, 3 * heads * 16, generator=g).half().cuda()
    bias_t = torch.randn(heads, n, n, generator=g).cuda()
    a = ops.window_attention(qkv, heads, bias_t, mask_t=mask.transpose(1, 2).contiguous().cuda(), windows_per_image=nw)
    b = ops.window_attention(qkv, heads, bias_t, region_ids=region_ids(dims, ws, ss).cuda(), windows_per_image=nw)
    assert torch.equal(a, b)


def _swin_pair(classes, dtype, seed=0):
    from diff_unet_amos_amd.diff_swin_unetr import DiffSwinUNETR
    from oracle.swin_ref import make_ref_diff_swin_unetr
    torch.manual_seed(seed)
    net = DiffSwinUNETR(in_channels=1, out_channels=classes, feature_size=48, compute_dtype=dtype).eval()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "relative_position_bias_table" in k:
                p.normal_(0, 0.3)                   # visible position bias (the default init is N(0, 0.02))
    ref = make_ref_diff_swin_unetr(1, classes, 48).eval()
    ref.load_state_dict(net.state_dict())
    return net.cuda(), ref


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol_abs,tol_emb", [(torch.float32, 1e-4, 5e-4), (torch.float16, 1e-2, 5e-3)])      # measured 7e-6 / 2e-4, 3.7e-3 / 8e-3 abs
def test_diff_swin_unetr_denoise_matches_oracle(dtype, tol_abs, tol_emb):
    """DiffSwinUNETR.forward(pred_type="denoise") -- SwinUNETREncoder + SwinUNETRDenoiser on the HIP launch plan -- against
    the oracle's torch restatement with the same weights, 64^3 patch, 3 classes, batch 2, two different timesteps."""
    net, ref = _swin_pair(3, dtype)
    g = torch.Generator().manual_seed(1)
    image = torch.randn(2, 1, 64, 64, 64, generator=g)
    x = torch.randn(2, 3, 64, 64, 64, generator=g)
    t = torch.tensor([7, 640])
    with torch.no_grad():
        emb_ref = ref.embed_model(image)
        want = ref(image=image, x=x, step=t, pred_type="denoise")
        got = net(image=image.cuda(), x=x.cuda(), step=t.cuda(), pred_type="denoise").cpu()
        emb = net.embed_model(image.cuda())
    for k in range(5):
        d = (emb[0][k].cpu() - emb_ref[0][k]).abs().max()
        print(f"[{dtype}] encoder hidden state {k}: max |d| {d:.2e} (|ref| max {emb_ref[0][k].abs().max():.2f})")
        assert d < tol_emb * max(1.0, float(emb_ref[0][k].abs().max()))
    for k in range(1, 5):
        d = (emb[k].cpu() - emb_ref[k]).abs().max()
        print(f"[{dtype}] encoder enc{k - 1}: max |d| {d:.2e} (|ref| max {emb_ref[k].abs().max():.2f})")
        assert d < tol_emb * max(1.0, float(emb_ref[k].abs().max()))
    d = (got - want).abs()
    print(f"[{dtype}] logits: max |d| {d.max():.2e} mean {d.mean():.2e} (|ref| max {want.abs().max():.2f} mean {want.abs().mean():.2f})")
    assert d.max() < tol_abs * max(1.0, float(want.abs().max())) and d.mean() < 0.1 * tol_abs
    # caller-supplied (NCDHW) embeddings take the same path as the plan's own
    with torch.no_grad():
        again = net.model(x=x.cuda(), t=t.cuda(), image=image.cuda(),
                          embeddings=[[e.cuda() for e in emb_ref[0]]] + [e.cuda() for e in emb_ref[1:]]).cpu()
    assert (again - want).abs().max() < tol_abs * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_diff_swin_unetr_ddim_sample_matches_oracle():
    """pred_type="ddim_sample" (diffusion.py:86-102) with injected x_T / step noise: 3 DDIM steps, fp32."""
    net, ref = _swin_pair(2, torch.float32, seed=4)
    from diff_unet_amos_amd.gaussian_diffusion import make_spaced
    from oracle.diffusion_ref import RefDiffusion
    net.sample_diffusion = make_spaced(1000, [3])
    ref.sample_diffusion = RefDiffusion(1000, [3])
    g = torch.Generator().manual_seed(2)
    image = torch.randn(1, 1, 64, 64, 64, generator=g)
    xT = torch.randn(1, 2, 64, 64, 64, generator=g)
    sn = [torch.randn(1, 2, 64, 64, 64, generator=g) for _ in range(3)]
    with torch.no_grad():
        want = ref.ddim_sample(image, x_T=[xT], step_noise=[sn])
        emb = net.embed_model(image.cuda())
        out = net.sample_diffusion.ddim_sample_loop(net.model, (1, 2, 64, 64, 64), noise=xT.cuda(),
                                                    model_kwargs={"image": image.cuda(), "embeddings": emb},
                                                    step_noise=[s.cuda() for s in sn])
        got = sum(s for s in out["all_samples"]).cpu()
    d = (got - want).abs()
    print(f"ddim x3 sum of x0: max |d| {d.max():.2e} mean {d.mean():.2e}")
    assert d.max() < 5e-4 and d.mean() < 2e-5             # measured 1.8e-5 / 2e-6


@pytest.mark.gpu
def test_diff_swin_unetr_graph_replay_equals_eager_and_full_size_runs():
    """The captured-graph sampling loop equals the eager launch sequence (DDIM, eta 0: no noise enters), a second call
    reuses the graph, and the BASELINE config-5 shape (96^3, 16 classes, fp16) runs to finite, bounded output."""
    net, _ = _swin_pair(2, torch.float16, seed=6)
    from diff_unet_amos_amd.gaussian_diffusion import make_spaced
    net.sample_diffusion = make_spaced(1000, [4])
    g = torch.Generator().manual_seed(8)
    image = torch.randn(1, 1, 64, 64, 64, generator=g).cuda()
    xT = torch.randn(1, 2, 64, 64, 64, generator=g).cuda()
    with torch.no_grad():
        emb = net.embed_model(image)
        plan = net._rt.plan(1, (64, 64, 64), image.device)
        a = plan.sample_loop(net.sample_diffusion, "ddim", noise=xT, use_graph=False)
        b = plan.sample_loop(net.sample_diffusion, "ddim", noise=xT, use_graph=True)
        c = plan.sample_loop(net.sample_diffusion, "ddim", noise=xT, use_graph=True)
    assert len(plan.graphs) == 1
    for k in ("sample", "sum_pred_xstart"):
        assert torch.equal(a[k], b[k]) and torch.equal(b[k], c[k]), k
    assert float(a["sum_pred_xstart"].abs().max()) <= 4.0 + 1e-6          # four clamped x0 predictions
    del net, plan, emb
    torch.cuda.empty_cache()
    from diff_unet_amos_amd.diff_swin_unetr import DiffSwinUNETR
    torch.manual_seed(0)
    big = DiffSwinUNETR(in_channels=1, out_channels=16, feature_size=48, sample_steps=3).cuda().eval()
    img = torch.rand(2, 1, 96, 96, 96, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        out = big(image=img, pred_type="ddim_sample")
    assert tuple(out.shape) == (2, 16, 96, 96, 96) and bool(torch.isfinite(out).all()) and float(out.abs().max()) <= 3.0 + 1e-6
    assert not torch.equal(out[0], out[1])


@pytest.mark.gpu
def test_full_size_config5_evaluation_matches_oracle():
    """BASELINE config 5 at its real geometry: one 96^3 patch, 16 classes, feature size 48, fp16 operands -- the shape bench.py
    --config 5 times -- against the oracle in fp32 on the host (stage grids 48/24/12/6/3: padded 7^3 windows on the first three
    stages, clipped 6^3 and 3^3 windows on the last two, which the 64^3 cases do not have)."""
    net, ref = _swin_pair(16, torch.float16, seed=11)
    g = torch.Generator().manual_seed(12)
    image = torch.rand(1, 1, 96, 96, 96, generator=g)
    x = torch.randn(1, 16, 96, 96, 96, generator=g)
    t = torch.tensor([433])
    with torch.no_grad():
        want = ref(image=image, x=x, step=t, pred_type="denoise")
        got = net(image=image.cuda(), x=x.cuda(), step=t.cuda(), pred_type="denoise").float().cpu()
    d = (got - want).abs()
    print(f"config 5 full size fp16: max |d| {d.max():.2e} mean {d.mean():.2e} (|ref| max {want.abs().max():.2f} mean {want.abs().mean():.2f})")
    assert d.max() < 1e-2 * max(1.0, float(want.abs().max())) and d.mean() < 1e-3
    agree = ((got > 0) == (want > 0)).float().mean()              # the binarisation the caller applies to sigmoid(logits)
    print(f"sign agreement {agree:.6f}")
    assert agree > 1 - 1e-3


@pytest.mark.gpu
def test_diff_swin_unetr_on_a_non_cubic_patch_matches_oracle():
    """A 64 x 96 x 32 patch: every stage has a different window clipping per axis (grids 32x48x16 ... 2x3x1), fp32."""
    net, ref = _swin_pair(2, torch.float32, seed=13)
    g = torch.Generator().manual_seed(14)
    image = torch.randn(1, 1, 64, 96, 32, generator=g)
    x = torch.randn(1, 2, 64, 96, 32, generator=g)
    t = torch.tensor([77])
    with torch.no_grad():
        want = ref(image=image, x=x, step=t, pred_type="denoise")
        got = net(image=image.cuda(), x=x.cuda(), step=t.cuda(), pred_type="denoise").cpu()
    d = (got - want).abs()
    print(f"64x96x32 fp32: max |d| {d.max():.2e} mean {d.mean():.2e}")
    assert d.max() < 1e-4 * max(1.0, float(want.abs().max())) and d.mean() < 1e-5


@pytest.mark.gpu
def test_token_linear_kernel_epilogues_match_torch():
    """dua_token_linear against torch.nn.functional on the same fp16 operands: plain / GELU (also split over 288 and 384
    outputs), conv3 + statistics, residual add on the fp32 stream, window scatter + shortcut + LayerNorm."""
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(21)
    dev = "cuda"
    for M, K, N, mode in ((1000, 48, 144, "plain"), (777, 96, 288, "plain"), (513, 384, 96, "plain"), (300, 48, 192, "gelu"),
                          (260, 96, 384, "gelu"), (129, 24, 48, "plain"), (2000, 192, 48, "plain")):
        A = torch.randn(M, K, generator=g).half().to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        b = torch.randn(N, generator=g).to(dev)
        out = torch.zeros(M, N + 8, dtype=torch.float16, device=dev)
        ops.token_linear(A, W, b, mode, out=out, out_off=8)
        want = F.linear(A.float(), W.float(), b)
        want = F.gelu(want) if mode == "gelu" else want
        assert float(out[:, :8].abs().max()) == 0.0
        assert (out[:, 8:].float() - want).abs().max() < 4e-3 * max(1.0, float(want.abs().max())), (M, K, N, mode)
    # conv3 + norm3 statistics: two samples, a channel slice of a wider buffer as input
    B, V, K, N = 2, 1500, 96, 48
    buf = torch.randn(B * V, K + 16, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    out = torch.empty(B * V, N, dtype=torch.float16, device=dev)
    st = ops.stats_buffer(B, N, dev)
    ops.token_linear(buf[:, :K], W, None, "stats", out=out, stats=st, samples=B)
    want = F.linear(buf[:, :K].float(), W.float())
    assert (out.float() - want).abs().max() < 4e-3 * max(1.0, float(want.abs().max()))
    o = out.float().view(B, V, N)
    sd = ops.stats_decode(st)
    assert torch.allclose(sd[:, :N, 0], o.sum(1).double(), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sd[:, :N, 1], (o * o).sum(1).double(), rtol=1e-4, atol=1e-2)
    # the same launch held to one workgroup per CU (dua_token_linear_desc.background): same output, same sums up to their grouping
    out_b, st_b = torch.empty_like(out), ops.stats_buffer(B, N, dev)
    ops.token_linear(buf[:, :K], W, None, "stats", out=out_b, stats=st_b, samples=B, background=1)
    assert torch.equal(out_b, out)
    assert torch.allclose(ops.stats_decode(st_b)[:, :N], sd[:, :N], rtol=1e-6, atol=1e-3)
    # residual
    for M, K, N in ((900, 192, 48), (450, 384, 96)):
        A = torch.randn(M, K, generator=g).half().to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        b = torch.randn(N, generator=g).to(dev)
        x = torch.randn(M, N, generator=g).to(dev)
        want = x + F.linear(A.float(), W.float(), b)
        ops.token_linear(A, W, b, "residual", x=x)
        assert (x - want).abs().max() < 2e-3 * max(1.0, float(want.abs().max()))
    # scatter + shortcut + LayerNorm == window_scatter_add_norm after a library GEMM
    for dims, C_, ws, ss in (((9, 8, 10), 48, (7, 7, 7), (3, 3, 3)), ((8, 14, 7), 96, (7, 7, 7), (3, 3, 0))):
        Bn = 2
        geom = ops.window_geom(Bn, dims, C_, ws, ss)
        nwin = 1
        for k in range(3):
            nwin *= -(-dims[k] // ws[k])
        Mw = Bn * nwin * ws[0] * ws[1] * ws[2]
        A = torch.randn(Mw, C_, generator=g).half().to(dev)
        W = (torch.randn(C_, C_, generator=g) / C_ ** 0.5).half().to(dev)
        b = torch.randn(C_, generator=g).to(dev)
        gm, bt = torch.randn(C_, generator=g).to(dev), torch.randn(C_, generator=g).to(dev)
        x0 = torch.randn(Bn, *dims, C_, generator=g).to(dev)
        xa, xb = x0.clone(), x0.clone()
        la, lb = torch.empty(Bn, *dims, C_, dtype=torch.float16, device=dev), torch.empty(Bn, *dims, C_, dtype=torch.float16, device=dev)
        po = F.linear(A, W, b.half())
        ops.window_scatter_add_norm(xa, geom, po, gm, bt, la)
        ops.token_linear(A, W, b, "scatter", x=xb, geom=geom, gamma=gm, beta=bt, ln_out=lb)
        assert (xa - xb).abs().max() < 4e-3 * max(1.0, float(xa.abs().max()))
        assert (la.float() - lb.float()).abs().max() < 8e-3 * max(1.0, float(la.float().abs().max()))


@pytest.mark.gpu
def test_diff_swin_unetr_under_the_sliding_window_caller():
    """Engine.infer's call (engine.py:167-182: sliding_window_inference(..., pred_type="ddim_sample")) with the swin variant
    as the model: two overlapping 64^3 windows of a 64x64x96 volume, 3 classes, 2 DDIM steps, against
    oracle.sliding_window_ref o the oracle's ddim_sample with the same per-window x_T (eta 0: no step noise)."""
    from diff_unet_amos_amd import inference
    from diff_unet_amos_amd.gaussian_diffusion import make_spaced
    from oracle.diffusion_ref import RefDiffusion
    from oracle.sliding_window_ref import sliding_window_ref
    net, ref = _swin_pair(3, torch.float32, seed=9)
    net.sample_diffusion = make_spaced(1000, [2])
    ref.sample_diffusion = RefDiffusion(1000, [2])
    g = torch.Generator().manual_seed(12)
    vol = torch.rand(1, 1, 64, 64, 96, generator=g)
    shape = (1, 3, 64, 64, 64)
    count = [0]

    def seed_of(win):
        return int(win.double().abs().sum().item() * 1e3) % (2 ** 31)

    def ref_fn(win):
        w = torch.from_numpy(win).float()
        torch.manual_seed(seed_of(w.cuda()))
        xT = torch.randn(*shape, device="cuda").cpu()
        count[0] += 1
        with torch.no_grad():
            return ref.ddim_sample(w, x_T=[xT], step_noise=[[torch.zeros(shape)] * 2]).numpy()

    def predictor(x, **kw):
        torch.manual_seed(seed_of(x))
        return net(image=x, **kw)

    want = torch.from_numpy(sliding_window_ref(vol.numpy(), (64, 64, 64), 0.5, ref_fn)).float()
    with torch.no_grad():
        got = inference.sliding_window_inference(vol.cuda(), (64, 64, 64), 1, predictor, 0.5, pred_type="ddim_sample").cpu()
    d = (got - want).abs()
    print(f"swin under the sliding window: {count[0]} windows, blended sum-x0 max |d| {d.max():.2e} mean {d.mean():.2e}")
    assert count[0] == 2 and got.shape == (1, 3, 64, 64, 96)
    assert d.max() < 5e-4 and d.mean() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("C_,M", [(48, 1000), (96, 777), (48, 128), (96, 13824)])
def test_fused_swin_mlp_kernel_matches_torch(C_, M):
    """dua_swin_mlp (linear1 + GELU + linear2 + residual on the fp32 stream, hidden activation in registers) against
    torch.nn.functional on the same fp16 operands, and against the two-launch token_linear form."""
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(C_ + M)
    dev = "cuda"
    ln2 = torch.randn(M, C_, generator=g).half().to(dev)
    w1 = (torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5).half().to(dev)
    w2 = (torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5).half().to(dev)
    b1, b2 = torch.randn(4 * C_, generator=g).to(dev), torch.randn(C_, generator=g).to(dev)
    x0 = torch.randn(M, C_, generator=g).to(dev)
    h = F.gelu(F.linear(ln2.float(), w1.float(), b1))
    want = x0 + F.linear(h.half().float(), w2.float(), b2)           # the hidden activation feeds linear2 as fp16 in both forms
    xa = x0.clone()
    ops.swin_mlp(ln2, w1, b1, w2, b2, xa)
    assert (xa - want).abs().max() < 3e-3 * max(1.0, float(want.abs().max()))
    xb = x0.clone()
    hid = torch.empty(M, 4 * C_, dtype=torch.float16, device=dev)
    ops.token_linear(ln2, w1, b1, "gelu", out=hid)
    ops.token_linear(hid, w2, b2, "residual", x=xb)
    assert (xa - xb).abs().max() < 2e-3 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["logits", "ddpm"])
def test_tail_residual_form_equals_the_materialised_route(mode):
    """dua_final_conv_sampler_res (decoder1's output assembled inside the tail) against residual_norm_act + the ordinary tail on
    the same operands, 48 real channels in a K = 64 tail, a voxel count that is not a multiple of the 256-voxel tile, two
    samples.  The fused form feeds the head the fp32 activation (split into an fp16 value + its rounding error); the
    materialised route rounds it to fp16 once on the way: equal up to that rounding."""
    from diff_unet_amos_amd import ops, _native as nv
    dev = "cuda"
    g = torch.Generator().manual_seed(31)
    N, D, H, W, Cc, K, classes = 2, 9, 10, 11, 48, 64, 16
    vox = D * H * W
    raw = torch.randn(N, D, H, W, Cc, generator=g).half().to(dev)
    res = (torch.randn(N, D, H, W, Cc, generator=g) * 2 + 0.5).half().to(dev)
    cat = torch.randn(N, D, H, W, 2 * Cc, generator=g).half().to(dev)
    ones, zeros = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    st_a, st_b = ops.stats_buffer(N, Cc, dev), ops.stats_buffer(N, Cc, dev)
    ops.instnorm_stats(raw, Cc, st_a)
    ops.instnorm_stats(res, Cc, st_b)
    na = ops.Norm(st_a, ones, zeros, vox, slope=0.01, eps=1e-5)
    nb = ops.Norm(st_b, ones, zeros, vox, slope=0.01, eps=1e-5)
    wf = torch.zeros(classes, K, device=dev)
    wf[:, :Cc] = torch.randn(classes, Cc, generator=g).to(dev) / Cc ** 0.5
    bf = torch.randn(classes, generator=g).to(dev)
    coef = torch.rand(N, 8, generator=g).to(dev)
    x0 = torch.randn(N * vox * 16, generator=g).to(dev)
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(fused):
        state, logits = x0.clone(), torch.zeros(N, classes, D, H, W, device=dev)
        kw = dict(coef=coef, x_state=state, step_word=step, seed_dev=seed) if mode == "ddpm" else dict(logits=logits)
        m = nv.MODE_DDPM if mode == "ddpm" else nv.MODE_LOGITS
        if fused:
            ops.final_conv_sampler(raw, K, na, wf, bf, classes, m, residual=(res, nb, cat, Cc, Cc), **kw)
        else:
            dec = torch.zeros(N, D, H, W, K, dtype=torch.float16, device=dev)
            ops.residual_norm_act(raw, na, res, nb, slope=0.01, out=dec, out_off=0, ra_src=cat, ra_off=Cc)
            ops.final_conv_sampler(dec, K, None, wf, bf, classes, m, **kw)
        return state if mode == "ddpm" else logits

    a, b = run(True), run(False)
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0.1
    assert float((a - b).abs().max()) < 3e-3, float((a - b).abs().max())


@pytest.mark.gpu
def test_diff_swin_unetr_long_ddpm_run_is_finite_and_follows_the_torch_seed():
    """200 reverse DDPM steps (in-kernel Philox noise, fp16 operands, x_{t-1} fed back through the fp16 input slice) on the graph
    path: finite, x0 predictions clamped, the same torch seed gives the same trajectory and the next call another
    (gaussian_diffusion.py:430 draws a fresh randn_like per step and call).  Out-of-range timesteps are a clean error."""
    net, _ = _swin_pair(16, torch.float16, seed=17)
    from diff_unet_amos_amd.gaussian_diffusion import make_spaced
    d200 = make_spaced(1000, [200])
    g = torch.Generator().manual_seed(18)
    image = torch.rand(1, 1, 64, 64, 64, generator=g).cuda()
    shape = (1, 16, 64, 64, 64)
    xT = torch.randn(*shape, generator=g).cuda()
    with torch.no_grad():
        kw = {"image": image, "embeddings": net.embed_model(image)}
        torch.manual_seed(77)
        a = d200.p_sample_loop(net.model, shape, noise=xT, model_kwargs=kw).clone()
        b = d200.p_sample_loop(net.model, shape, noise=xT, model_kwargs=kw).clone()
        torch.manual_seed(77)
        c = d200.p_sample_loop(net.model, shape, noise=xT, model_kwargs=kw).clone()
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) < 20.0
        assert float((a - b).abs().mean()) > 1e-2
        assert torch.equal(a, c)                           # same noise field, order-independent statistics: the same bits
        with pytest.raises((RuntimeError, AssertionError, ValueError)):
            net(image=image, x=xT, step=torch.tensor([1000]), pred_type="denoise")


@pytest.mark.gpu
def test_token_gemm_matches_torch_linear():
    """dua_token_gemm (the tiled MFMA GEMM of the coarse Swin stages: qkv / proj / linear1 + GELU / linear2 + residual /
    reduction / wide conv3, attention.py:97-120, transformer.py:433-435, patch.py:89-92, blocks.py:311-314) against
    F.linear on the same fp16 operands: token counts that are not multiples of the 64-row tile, N = 96 (one and a half
    column tiles), K = 96 (one and a half K steps), K = 3072, a strided A (a channel slice of a wider buffer); the small-token
    shapes take the K-split path (partial tiles + finish launch)."""
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    dev = "cuda"
    g = torch.Generator().manual_seed(41)
    for M, K, N, mode, bias in ((343, 3072, 768, "plain", False), (27, 3072, 768, "gelu", True), (27, 3072, 768, "plain", False),
                                (2744, 192, 576, "plain", True), (1728, 768, 192, "gelu", True),
                                (21952, 96, 288, "plain", True), (1000, 96, 96, "gelu", True), (216, 1536, 384, "residual", True),
                                (13824, 384, 96, "residual", True)):
        A = torch.randn(M, K + 16, generator=g).half().to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        b = torch.randn(N, generator=g).to(dev) if bias else None
        want = F.linear(A[:, :K].float(), W.float(), b)
        scale = max(1.0, float(want.abs().max()))
        if mode == "residual":
            x = torch.randn(M, N, generator=g).to(dev)
            got = ops.token_gemm(A[:, :K], W, b, "residual", x=x.clone())
            assert (got - (x + want)).abs().max() < 2e-3 * scale, (M, K, N, mode)
            continue
        out = torch.full((M, N + 8), 7.0, dtype=torch.float16, device=dev)
        ops.token_gemm(A[:, :K], W, b, mode, out=out, out_off=8)
        if mode == "gelu":
            want = F.gelu(want)
        assert float((out[:, :8].float() - 7).abs().max()) == 0.0
        assert (out[:, 8:].float() - want).abs().max() < 4e-3 * scale, (M, K, N, mode, float((out[:, 8:].float() - want).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,gelu,strided", [(1000, 48, 144, False, False), (343 * 3, 48, 192, True, False), (216, 768, 3072, False, False),
                                                (77, 20, 50, False, False), (513, 96, 48, False, True), (64, 4, 1, True, True)])
def test_linear_f32_kernel_matches_torch(M, K, N, gelu, strided):
    """dua_linear_f32 (exact-fp32 MFMA; the Linear layers / 1x1x1 conv3 of the fp32 parity plan) against F.linear (+ nn.GELU) on
    the CPU: ragged M and N, K down to one 16-byte piece, rows taken as a channel slice of a wider buffer."""
    import torch.nn.functional as F
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(M + K + N)
    wide = torch.randn(M, K + 12 if strided else K, generator=g)
    x = wide[:, :K]
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    want = F.linear(x, w, b)
    if gelu:
        want = F.gelu(want)
    xd = wide.cuda()[:, :K]
    got = ops.linear_f32(xd, w.cuda(), b.cuda(), gelu=gelu)
    assert got.shape == (M, N)
    assert torch.allclose(got.cpu(), want, rtol=1e-5, atol=2e-5), float((got.cpu() - want).abs().max())
    nob = ops.linear_f32(xd.contiguous().view(1, M, K), w.cuda())                  # no bias, leading dimensions kept
    assert nob.shape == (1, M, N) and torch.allclose(nob.cpu()[0], F.linear(x, w), rtol=1e-5, atol=2e-5)


@pytest.mark.gpu
def test_neither_swin_plan_launches_a_library_gemm():
    """No hipBLASLt / rocBLAS kernel in a denoiser evaluation of either plan (fp16: token_linear / token_gemm; fp32 parity plan:
    dua_linear_f32): kernel names collected with torch's profiler over one evaluation of each."""
    from torch.profiler import ProfilerActivity, profile
    from diff_unet_amos_amd.diff_swin_unetr import DiffSwinUNETR
    torch.manual_seed(0)
    for dtype in (torch.float32, torch.float16):
        net = DiffSwinUNETR(in_channels=1, out_channels=3, feature_size=48, compute_dtype=dtype).cuda().eval()
        image = torch.rand(1, 1, 64, 64, 64, device="cuda")
        x = torch.randn(1, 3, 64, 64, 64, device="cuda")
        t = torch.tensor([500], device="cuda")
        with torch.no_grad():
            net(image=image, x=x, step=t, pred_type="denoise")
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                net(image=image, x=x, step=t, pred_type="denoise")
                torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages()]
        assert any("linear_f32" in n or "token_" in n for n in names), names[:10]
        assert not [n for n in names if n.startswith("Cijk") or "gemm" in n.lower() and "token_gemm" not in n], dtype
