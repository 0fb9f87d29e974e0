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
@pytest.mark.parametrize("shape,legacy", [((2, 6, 8, 10, 48), True), ((1, 5, 7, 6, 96), True), ((1, 4, 4, 4, 16), False)])
def test_patch_merge_norm_kernel_matches_oracle(dtype, tol, shape, legacy):
    """Gather (legacy duplicates, zero padding of odd extents) + LayerNorm(8C); the reduction Linear stays a GEMM."""
    from diff_unet_amos_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    pm = RefPatchMerging(shape[-1], legacy=legacy)
    with torch.no_grad():
        pm.norm.weight.normal_(1.0, 0.2); pm.norm.bias.normal_(0.0, 0.2)
        want = pm.norm(patch_merging_gather(x.to(dtype).float(), legacy))
        full = pm(x.to(dtype).float())
    got = ops.patch_merge_norm(x.to(dtype).cuda(), pm.norm.weight.detach().cuda(), pm.norm.bias.detach().cuda(), legacy=legacy)
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
    blk = RefUnetResBlock(cin, cout).eval()
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
    w1, b1 = ops.pack_conv3_weights(blk.conv1.weight.detach().to(dev), zb, dtype)
    w2, b2 = ops.pack_conv3_weights(blk.conv2.weight.detach().to(dev), zb, dtype)
    raw1 = torch.empty(N, D, H, W, cout, dtype=dtype, device=dev); st1 = ops.stats_buffer(N, cout, dev)
    ops.conv3d_k3(xcl, cin, 0, w1, b1, cout, raw1, 0, st1)
    with torch.no_grad():
        add = blk.t_proj(nonlinearity(t)).to(dev).float().contiguous()
    n1 = ops.Norm(st1, blk.norm1.weight.detach().to(dev), blk.norm1.bias.detach().to(dev), V, add=add, add_stride=cout, slope=0.01)
    raw2 = torch.empty_like(raw1); st2 = ops.stats_buffer(N, cout, dev)
    ops.conv3d_k3(raw1, cout, 0, w2, b2, cout, raw2, 0, st2, norm=n1)
    n2 = ops.Norm(st2, blk.norm2.weight.detach().to(dev), blk.norm2.bias.detach().to(dev), V, slope=0.01)
    if blk.downsample:
        r = (xcl.float() @ blk.conv3.weight.detach().reshape(cout, cin).t().to(dev)).to(dtype).contiguous()     # 1x1x1 conv = GEMM
        st3 = ops.stats_buffer(N, cout, dev)
        rf = r.float().view(N, V, cout)
        st3[:, 0, :cout, 0] = rf.sum(1).double(); st3[:, 0, :cout, 1] = (rf * rf).sum(1).double()
        n3 = ops.Norm(st3, blk.norm3.weight.detach().to(dev), blk.norm3.bias.detach().to(dev), V, slope=0.01)
        out = ops.residual_norm_act(raw2, n2, r, n3, slope=0.01)
    else:
        out = ops.residual_norm_act(raw2, n2, xcl, None, slope=0.01)
    got = out.float().permute(0, 4, 1, 2, 3).cpu()
    d = (got - want).abs()
    print(f"\n[{dtype}] UnetResBlock {cin}->{cout}: max |d| {d.max():.2e} mean {d.mean():.2e}")
    assert d.max() < tol * max(1.0, float(want.abs().max()))
