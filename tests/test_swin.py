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
