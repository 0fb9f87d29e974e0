"""Tensor-level wrappers over the C ABI (include/dua_hip.h).

Each function checks shapes on the host (a kernel that faults can reset the
whole node), hands raw device pointers + the current HIP stream to
libdua_hip.so and returns without synchronising.  Activations are
channels-last tensors of shape [N, D, H, W, Cstride].
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as nv


def _cl_check(t, name):
    assert t.is_cuda and t.is_contiguous() and t.dim() == 5, f"{name}: need a contiguous channels-last [N,D,H,W,C] device tensor"
    assert t.shape[-1] % 8 == 0, f"{name}: channel stride must be a multiple of 8"


def chunk_elems(dtype):
    return 32 if dtype == torch.float16 else 16


def to_channels_last(src, dst, c_off=0, c_fill=None):
    """NCDHW fp32 -> channel slice of channels-last ``dst``; zero-fills [C, c_fill)."""
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 5
    _cl_check(dst, "dst")
    N, Cc = src.shape[:2]
    vox = src.shape[2] * src.shape[3] * src.shape[4]
    assert tuple(dst.shape[:4]) == (N, *src.shape[2:])
    c_fill = Cc if c_fill is None else c_fill
    assert c_off + max(Cc, c_fill) <= dst.shape[-1]
    nv.check(nv.lib().dua_to_channels_last(nv.dt_code(dst.dtype), N, Cc, vox, nv.ptr(src), nv.ptr(dst), dst.shape[-1],
                                           c_off, c_fill, nv.stream_ptr()), "dua_to_channels_last")
    return dst


def from_channels_last(src, C_, c_off=0, out=None):
    _cl_check(src, "src")
    N, D, H, W, Cs = src.shape
    assert c_off + C_ <= Cs
    if out is None:
        out = torch.empty((N, C_, D, H, W), dtype=torch.float32, device=src.device)
    assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (N, C_, D, H, W)
    nv.check(nv.lib().dua_from_channels_last(nv.dt_code(src.dtype), N, C_, D * H * W, nv.ptr(src), Cs, c_off,
                                             nv.ptr(out), nv.stream_ptr()), "dua_from_channels_last")
    return out


def pack_conv3_weights(w, bias, dtype, cin_packed=None, perm=None):
    """nn.Conv3d parameters -> (packed weights as a byte tensor, bias padded to a multiple of 64)."""
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
    w = w.contiguous()
    cout, cin = w.shape[:2]
    ck = chunk_elems(dtype)
    cin_packed = cin if cin_packed is None else cin_packed
    padded = -(-cin_packed // ck) * ck
    perm_t = None
    if perm is not None:
        p = list(perm) + [-1] * (padded - len(perm))
        perm_t = torch.tensor(p, dtype=torch.int32, device=w.device)
    L = nv.lib()
    code = nv.dt_code(dtype)
    nbytes = L.dua_pack_conv3_weights(code, cout, cin, cin_packed, None, None, None, None)
    assert nbytes > 0
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    rc = L.dua_pack_conv3_weights(code, cout, cin, cin_packed, nv.ptr(w), nv.ptr(perm_t), nv.ptr(buf), nv.stream_ptr())
    if rc != nbytes:
        raise RuntimeError(f"dua_pack_conv3_weights failed ({rc})")
    cpad = -(-cout // 64) * 64
    b = torch.zeros(cpad, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[:cout] = bias.detach().float()
    return buf, b


def pack_deconv_weights(w, bias, dtype):
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 5 and tuple(w.shape[2:]) == (2, 2, 2)
    w = w.contiguous()
    cin, cout = w.shape[:2]
    L = nv.lib()
    code = nv.dt_code(dtype)
    nbytes = L.dua_pack_deconv_weights(code, cin, cout, None, None, None)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    rc = L.dua_pack_deconv_weights(code, cin, cout, nv.ptr(w), nv.ptr(buf), nv.stream_ptr())
    if rc != nbytes:
        raise RuntimeError(f"dua_pack_deconv_weights failed ({rc})")
    cpad = -(-cout // 64) * 64
    b = torch.zeros(cpad, dtype=torch.float32, device=w.device)
    b[:cout] = bias.detach().float()
    return buf, b


def conv3_rows(D, H, W):
    return (-(-D // 4)) * (-(-H // 8)) * (-(-W // 8)) * 4


def conv3d_k3(x, cin, cin_off, w_packed, bias_pad, cout, y, cout_off, partials, counts,
              in_scale=None, in_shift=None, in_add=None, slope=0.1):
    """Raw 3x3x3 convolution (+bias) with optional fused producer norm/activation on the input
    and InstanceNorm partial statistics on the output."""
    _cl_check(x, "x"); _cl_check(y, "y")
    assert x.dtype == y.dtype and x.device == y.device
    N, D, H, W, cs_in = x.shape
    assert tuple(y.shape[:4]) == (N, D, H, W)
    assert cin % 8 == 0 and cin_off % 8 == 0 and cin_off + cin <= cs_in
    assert cout % 8 == 0 and cout_off % 8 == 0 and cout_off + cout <= y.shape[-1]
    ck = chunk_elems(x.dtype)
    nch, nct = -(-cin // ck), -(-cout // 64)
    esz = x.element_size()
    assert w_packed.numel() == nct * nch * 27 * 4 * 64 * 16, "packed weights do not match (Cin, Cout, dtype)"
    assert bias_pad.numel() == nct * 64 and bias_pad.dtype == torch.float32
    rows = conv3_rows(D, H, W)
    assert partials.dtype == torch.float32 and partials.numel() >= N * rows * nct * 64 * 2
    assert counts.dtype == torch.float32 and counts.numel() >= rows
    if in_scale is not None:
        for v in (in_scale, in_shift) + ((in_add,) if in_add is not None else ()):
            assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == N * cin
    d = nv.Conv3Desc(nv.dt_code(x.dtype), N, D, H, W, cin, cs_in, cin_off, cout, y.shape[-1], cout_off, slope)
    del esz
    nv.check(nv.lib().dua_conv3d_k3_fwd(C.byref(d), nv.ptr(x), nv.ptr(w_packed), nv.ptr(bias_pad), nv.ptr(in_scale),
                                        nv.ptr(in_shift), nv.ptr(in_add), nv.ptr(y), nv.ptr(partials), nv.ptr(counts),
                                        nv.stream_ptr()), "dua_conv3d_k3_fwd")
    return rows, nct * 64


def instnorm_finalize(N, Cc, rows, c_pad, partials, counts, gamma, beta, scale, shift, eps=1e-5):
    for v in (gamma, beta):
        assert v.is_cuda and v.dtype == torch.float32 and v.numel() == Cc and v.is_contiguous()
    for v in (scale, shift):
        assert v.is_cuda and v.dtype == torch.float32 and v.numel() >= N * Cc
    assert partials.numel() >= N * rows * c_pad * 2 and counts.numel() >= rows
    nv.check(nv.lib().dua_instnorm_finalize(N, Cc, rows, c_pad, nv.ptr(partials), nv.ptr(counts), nv.ptr(gamma),
                                            nv.ptr(beta), eps, nv.ptr(scale), nv.ptr(shift), nv.stream_ptr()),
             "dua_instnorm_finalize")


def materialize(raw, Cc, scale, shift, out, out_off, emb=None, pooled=None, slope=0.1):
    _cl_check(raw, "raw"); _cl_check(out, "out")
    N, D, H, W, rs = raw.shape
    assert tuple(out.shape[:4]) == (N, D, H, W) and out.dtype == raw.dtype
    assert Cc % 8 == 0 and Cc <= rs and out_off % 8 == 0 and out_off + Cc <= out.shape[-1]
    assert scale.numel() >= N * Cc and shift.numel() >= N * Cc
    es = 0
    if emb is not None:
        _cl_check(emb, "emb")
        assert tuple(emb.shape[:4]) == (N, D, H, W) and emb.dtype == raw.dtype and emb.shape[-1] >= Cc
        es = emb.shape[-1]
    ps = 0
    if pooled is not None:
        _cl_check(pooled, "pooled")
        assert D % 2 == 0 and H % 2 == 0 and W % 2 == 0
        assert tuple(pooled.shape[:4]) == (N, D // 2, H // 2, W // 2) and pooled.dtype == raw.dtype and pooled.shape[-1] >= Cc
        ps = pooled.shape[-1]
    d = nv.MaterializeDesc(nv.dt_code(raw.dtype), N, D, H, W, Cc, rs, es, out.shape[-1], out_off, ps, slope)
    nv.check(nv.lib().dua_materialize(C.byref(d), nv.ptr(raw), nv.ptr(scale), nv.ptr(shift), nv.ptr(emb), nv.ptr(out),
                                      nv.ptr(pooled), nv.stream_ptr()), "dua_materialize")
