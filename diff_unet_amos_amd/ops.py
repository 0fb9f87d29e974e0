"""Tensor-level wrappers over the C ABI (include/dua_hip.h).

Each function checks shapes on the host (a kernel that faults can reset the
whole node), hands raw device pointers + the current HIP stream to
libdua_hip.so and returns without synchronising.  Activations are
channels-last tensors of shape [N, D, H, W, Cstride].
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _native as nv


# When a list is installed here (engine.Plan building its dua_denoiser_plan), conv3d_k3 / materialize / deconv_k2s2
# validate their arguments as usual but APPEND a filled nv.StepOp instead of launching: the launch sequence is written
# once, in Python, and executed by dua_denoiser_step.
_RECORD = None


class recording:
    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        global _RECORD
        assert _RECORD is None
        _RECORD = self.sink
        return self.sink

    def __exit__(self, *exc):
        global _RECORD
        _RECORD = None
        return False


def _norm_value(norm, N, Cc):
    """(has_norm, nv.InNorm by value) for a StepOp."""
    if norm is None:
        return 0, nv.InNorm()
    norm.ref(N, Cc)
    c = norm.c
    return 1, nv.InNorm(c.stats, c.gamma, c.beta, c.add, c.add_stride, c.c_pad, c.count, c.eps, c.slope)


def _addr(t):
    return t.data_ptr() if t is not None else None


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


def to_channels_last_rows(parts, dst):
    """torch.cat(parts, 1) (one or two NCDHW fp32 tensors) -> the whole rows of a dense channels-last ``dst`` (zero fill up to
    its width) in one launch of 16-byte stores; rows of at most 64 bytes."""
    _cl_check(dst, "dst")
    assert 1 <= len(parts) <= 2 and dst.is_contiguous()
    for p in parts:
        assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 5
        assert tuple(p.shape[2:]) == tuple(dst.shape[1:4]) and p.shape[0] == dst.shape[0]
    N, vox = dst.shape[0], dst.shape[1] * dst.shape[2] * dst.shape[3]
    c0 = parts[0].shape[1]
    c1 = parts[1].shape[1] if len(parts) == 2 else 0
    nv.check(nv.lib().dua_to_channels_last_rows(nv.dt_code(dst.dtype), N, c0, nv.ptr(parts[0]), c1,
                                                nv.ptr(parts[1]) if c1 else None, vox, nv.ptr(dst), dst.shape[-1], nv.stream_ptr()),
             "dua_to_channels_last_rows")
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


_ZERO_BIAS = {}


def zero_bias(n, device):
    """A shared, never-written fp32 zero vector of at least n entries (the bias of data-gradient convolutions)."""
    key = torch.device(device)
    z = _ZERO_BIAS.get(key)
    if z is None or z.numel() < n:
        z = torch.zeros(max(n, 4096), dtype=torch.float32, device=device)     # channel counts are <= 1024 everywhere
        _ZERO_BIAS[key] = z
    return z


def pack_conv3_weights(w, bias, dtype, cin_packed=None, perm=None, tap_channel=None, pad_bias=True):
    """nn.Conv3d parameters -> (packed weights as a byte tensor, bias padded to a multiple of 64).
    ``tap_channel``: packed index of the channel handled by conv3d_k3(tap_channel=...) (single-channel tap form).
    ``pad_bias=False``: hand back the fp32 bias itself (conv3d_k3 reads Cout entries only) -- the training step, which repacks
    every layer every step, saves a fill and a copy launch per layer that way."""
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
    if tap_channel is None:
        nbytes = L.dua_pack_conv3_weights(code, cout, cin, cin_packed, None, None, None, None)
        assert nbytes > 0
        buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        rc = L.dua_pack_conv3_weights(code, cout, cin, cin_packed, nv.ptr(w), nv.ptr(perm_t), nv.ptr(buf), nv.stream_ptr())
    else:
        src = tap_channel if perm is None else perm[tap_channel]
        assert dtype == torch.float16 and cin_packed <= 32 and 0 <= src < cin
        nbytes = L.dua_pack_conv3_weights_tap(code, cout, cin, cin_packed, tap_channel, src, None, None, None, None)
        assert nbytes > 0
        buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        rc = L.dua_pack_conv3_weights_tap(code, cout, cin, cin_packed, tap_channel, src, nv.ptr(w), nv.ptr(perm_t), nv.ptr(buf),
                                          nv.stream_ptr())
    if rc != nbytes:
        raise RuntimeError(f"dua_pack_conv3_weights failed ({rc})")
    if not pad_bias:
        return buf, (zero_bias(cout, w.device) if bias is None else bias.detach().float().contiguous())
    cpad = -(-cout // 64) * 64
    b = torch.zeros(cpad, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[:cout] = bias.detach().float()
    return buf, b


def pack_conv3_weights_dgrad(w, dtype, cout_packed=None):
    """Forward weights [Cout, Cin, 3,3,3] -> (packed weights of the data-gradient convolution dy -> dx, zero bias)."""
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.is_contiguous()
    cout, cin = w.shape[:2]
    cout_packed = cout if cout_packed is None else cout_packed
    L, code = nv.lib(), nv.dt_code(dtype)
    nbytes = L.dua_pack_conv3_weights_dgrad(code, cout, cin, cout_packed, None, None, None)
    assert nbytes > 0
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    rc = L.dua_pack_conv3_weights_dgrad(code, cout, cin, cout_packed, nv.ptr(w), nv.ptr(buf), nv.stream_ptr())
    if rc != nbytes:
        raise RuntimeError(f"dua_pack_conv3_weights_dgrad failed ({rc})")
    return buf, zero_bias(cin, w.device)          # shared, never written


class ConvPacks:
    """The fp16 packings of MANY 3x3x3 convolution weights made by one launch per 64 tensors (dua_pack_conv3_weights_batch): a
    training step packs every layer twice (forward layout for the forward pass, data-gradient layout for backward) -- 56 launches
    of 3-12 us before.  ``add(w, kind, packed)`` registers a tensor (kind "fwd": packed = channels of the input buffer; "dgrad":
    packed = channels of the dy buffer); ``run()`` packs everything registered into one buffer; ``get(w, kind, packed)`` hands
    back the packed bytes or None (caller packs that layer on its own: first layers, odd channel counts, fp32)."""

    def __init__(self, dtype):
        self.dtype, self.items, self.out = dtype, [], {}

    @staticmethod
    def _key(w, kind, packed):
        return (w.data_ptr(), kind, int(packed))

    def add(self, w, kind, packed):
        cout, cin = w.shape[:2]
        if (self.dtype != torch.float16 or not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()) or cin % 4
                or w.data_ptr() % 16 or packed < (cin if kind == "fwd" else cout) or self._key(w, kind, packed) in self.out):
            return
        L, code = nv.lib(), nv.dt_code(self.dtype)
        if kind == "fwd":
            nbytes = L.dua_pack_conv3_weights(code, cout, cin, packed, None, None, None, None)
        else:
            nbytes = L.dua_pack_conv3_weights_dgrad(code, cout, cin, packed, None, None, None)
        self.items.append((w, kind, int(packed), int(nbytes)))
        self.out[self._key(w, kind, packed)] = None

    def run(self):
        if not self.items:
            return self
        offs, total = [], 0
        for *_, nbytes in self.items:
            offs.append(total)
            total += (nbytes + 255) & ~255
        flat = torch.empty(total, dtype=torch.uint8, device=self.items[0][0].device)
        L = nv.lib()
        for i0 in range(0, len(self.items), 64):
            chunk = self.items[i0:i0 + 64]
            arr = (nv.PackItem * len(chunk))()
            for j, (w, kind, packed, nbytes) in enumerate(chunk):
                view = flat[offs[i0 + j]:offs[i0 + j] + nbytes]
                arr[j] = nv.PackItem(0 if kind == "fwd" else 1, w.shape[0], w.shape[1], packed, w.data_ptr(), view.data_ptr())
                self.out[self._key(w, kind, packed)] = view
            nv.check(L.dua_pack_conv3_weights_batch(nv.dt_code(self.dtype), len(chunk), arr, nv.stream_ptr()),
                     "dua_pack_conv3_weights_batch")
        self.items = []
        return self

    def get(self, w, kind, packed):
        return self.out.get(self._key(w, kind, packed))


def pack_deconv_weights(w, bias, dtype, alias_bias=False):
    """``alias_bias``: hand the fp32 bias tensor itself back when it already is a whole number of 64-channel tiles (the training
    step, which repacks every layer every step, saves a fill and a copy launch that way); launch plans keep their own copy, so that
    packed weights and bias always belong to the same weight version."""
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
    if alias_bias and bias is not None and cpad == cout and bias.dtype == torch.float32 and bias.is_contiguous():
        return buf, bias.detach()                 # already a whole number of output tiles: no fill + copy per call
    b = torch.zeros(cpad, dtype=torch.float32, device=w.device)
    if bias is not None:
        b[:cout] = bias.detach().float()
    return buf, b


STAT_REPLICAS = 8


class ZeroArena:
    """Bump allocator over ONE device buffer that is zeroed by ONE fill per training step.

    A training step needs ~280 small zero-initialised buffers (weight-gradient accumulators, statistics rows, reduction
    scratch); as separate torch.zeros calls they were ~280 fill kernels and >1 ms of a 20 ms step.  While an arena is
    active (``with arena:``), ``ops.zeros`` hands out views of it; ``reset()`` re-zeroes everything with one kernel.
    Views must not outlive the step (gradients are consumed by the optimizer and dropped by zero_grad(set_to_none))."""

    def __init__(self, nbytes, device):
        self.buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        self.off = 0
        self.overflow = 0

    def reset(self):
        self.buf.zero_()
        self.off = 0

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        start = (self.off + 255) & ~255
        if start + nbytes > self.buf.numel():
            self.overflow += nbytes
            return None
        self.off = start + nbytes
        return self.buf[start:start + nbytes].view(dtype).view(tuple(shape))

    def __enter__(self):
        global _ARENA
        self._prev = _ARENA
        _ARENA = self
        return self

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = self._prev
        return False


_ARENA = None


def zeros(shape, dtype, device):
    """torch.zeros, or a view of the active ZeroArena (same device) when one is installed."""
    if _ARENA is not None and torch.device(device) == _ARENA.buf.device:
        t = _ARENA.take(tuple(shape) if not isinstance(shape, int) else (shape,), dtype)
        if t is not None:
            return t
    return torch.zeros(shape, dtype=dtype, device=device)


STAT_WORDS = 4
STAT_FRAC = float(2 ** 44)


def stats_buffer(N, cout, device):
    """Zeroed int64 [N][8 replicas][4 words][ceil(cout/64)*64] accumulator a convolution adds (sum x, sum x^2) into:
    fixed-point words (integer part, fraction * 2^44) added with integer atomics, so the totals do not depend on the
    order in which workgroups arrive (include/dua_hip.h, dua_in_norm)."""
    return zeros((N, STAT_REPLICAS, STAT_WORDS, -(-cout // 64) * 64), torch.int64, device)


def stats_decode(stats):
    """int64 [N, 8, 4, c_pad] statistics words -> float64 [N, c_pad, 2] = (sum x, sum x^2), as the kernels read them."""
    w = stats.sum(1)
    S = w[:, 0].double() + w[:, 1].double() / STAT_FRAC
    Q = w[:, 2].double() + w[:, 3].double() / STAT_FRAC
    return torch.stack([S, Q], -1)


def stats_channel_sums(stats, c):
    """fp32 [c] = sum over the samples of the decoded "sum x" words (stats_decode(stats)[:, :c, 0].sum(0).float()) in one
    launch: a bias gradient from the statistics rows accumulated over a layer's output gradient."""
    assert stats.is_cuda and stats.dtype == torch.int64 and stats.is_contiguous() and stats.dim() == 4
    assert stats.shape[1] == STAT_REPLICAS and stats.shape[2] == STAT_WORDS and 0 < c <= stats.shape[3]
    out = torch.empty(c, dtype=torch.float32, device=stats.device)
    nv.check(nv.lib().dua_stats_channel_sums(stats.shape[0], c, stats.shape[3], nv.ptr(stats), nv.ptr(out), nv.stream_ptr()),
             "dua_stats_channel_sums")
    return out


def stats_encode(sums, out=None):
    """float64 [N, C, 2] (sum x, sum x^2) -> statistics words in replica row 0 of an int64 [N, 8, 4, c_pad] buffer."""
    N, Cc = sums.shape[:2]
    if out is None:
        out = torch.zeros((N, STAT_REPLICAS, STAT_WORDS, -(-Cc // 64) * 64), dtype=torch.int64, device=sums.device)
    out.zero_()
    s = sums.double()
    hi = torch.round(s)
    lo = torch.round((s - hi) * STAT_FRAC)
    out[:, 0, 0, :Cc] = hi[..., 0].long(); out[:, 0, 1, :Cc] = lo[..., 0].long()
    out[:, 0, 2, :Cc] = hi[..., 1].long(); out[:, 0, 3, :Cc] = lo[..., 1].long()
    return out


class Norm:
    """Producer-side normalisation descriptor (dua_in_norm) a consumer fuses into its input staging."""

    def __init__(self, stats, gamma, beta, count, add=None, add_stride=0, slope=0.1, eps=1e-5):
        assert stats.is_cuda and stats.dtype == torch.int64 and stats.is_contiguous() and stats.dim() == 4
        assert stats.shape[1] == STAT_REPLICAS and stats.shape[2] == STAT_WORDS
        for v in (gamma, beta):
            assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()
        assert gamma.numel() == beta.numel() <= stats.shape[3]
        if add is not None:
            assert add.is_cuda and add.dtype == torch.float32
            assert add.numel() >= (stats.shape[0] - 1) * (add_stride or gamma.numel()) + gamma.numel()
        self.keep = (stats, gamma, beta, add)
        self.N, self.C = stats.shape[0], gamma.numel()
        self.c = nv.InNorm(stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), add.data_ptr() if add is not None else None,
                           add_stride, stats.shape[3], int(count), eps, slope)

    def ref(self, N, Cc):
        assert self.N == N and self.C >= Cc, "normalisation descriptor does not match the consumer's input"
        return C.byref(self.c)


def _norm_ref(norm, N, Cc):
    return None if norm is None else norm.ref(N, Cc)


def conv3d_k3_dgrad_reduce_supported(dtype, N, D, H, W, cin, cout):
    """The data-gradient launch of this shape (dense buffers) can take the norm-backward sums of its output's owner along."""
    if dtype != torch.float16 or (CONV_POLICY & 0xff) != 0:
        return False
    d = nv.Conv3Desc(nv.dt_code(dtype), N, D, H, W, cin, cin, 0, cout, cout, 0)
    return bool(nv.lib().dua_conv3d_k3_dgrad_reduce_supported(C.byref(d)))


def conv3d_k3_dgrad_reduce(dy, cin, w_packed, bias_pad, cout, dx, raw, norm, sums):
    """dx = conv3d_k3(dy) (the data gradient: dgrad-packed weights) where dx is the dA of the layer whose convolution output is
    ``raw`` and whose forward statistics / affine parameters are ``norm``: the launch adds that layer's instnorm_bwd reduce sums
    to ``sums`` (instnorm_bwd_sums) instead of statistics of dx (dua_conv3d_k3_dgrad_reduce)."""
    _cl_check(dy, "dy"); _cl_check(dx, "dx"); _cl_check(raw, "raw")
    N, D, H, W, cs_in = dy.shape
    assert dy.dtype == dx.dtype == raw.dtype == torch.float16 and tuple(dx.shape[:4]) == tuple(raw.shape[:4]) == (N, D, H, W)
    assert cin <= cs_in and cout <= dx.shape[-1] and cout <= raw.shape[-1] and _RECORD is None
    assert sums.dtype == torch.float64 and sums.is_contiguous()
    d = nv.Conv3Desc(nv.dt_code(dy.dtype), N, D, H, W, cin, cs_in, 0, cout, dx.shape[-1], 0)
    nv.check(nv.lib().dua_conv3d_k3_dgrad_reduce(C.byref(d), nv.ptr(dy), nv.ptr(w_packed), nv.ptr(bias_pad), nv.ptr(dx), nv.ptr(raw),
                                                 raw.shape[-1], 0, norm.ref(N, cout), nv.ptr(sums), nv.stream_ptr()),
             "dua_conv3d_k3_dgrad_reduce")


def conv3_workspace_bytes(dtype, N, D, H, W, cin, cout):
    d = nv.Conv3Desc(nv.dt_code(dtype), N, D, H, W, cin, cin, 0, cout, cout, 0)
    return int(nv.lib().dua_conv3d_k3_workspace(C.byref(d)))


# Launch form handed to the kernels with every call (dua_conv3_desc.policy; 0 = the launchers' automatic choice).  The kernel
# tests and the A/B tools set these to reach forms the automatic choice would not take for their shapes; the library itself keeps
# no option state.
CONV_POLICY = 0          # conv3d_k3 / deconv_k2s2: 0, 2, 3, 6, 7 (| _native.POLICY_NO_FINISH)
WGRAD_POLICY = 0         # conv3d_k3_wgrad: bit field, see include/dua_hip.h

KIND_V2, KIND_FIRST, KIND_WIDE = 0, 1, 2          # dua_conv3d_k3_kernel_kind
DECONV_ALLTAPS = 2                                # dua_deconv_k2s2_kernel_kind


def conv3_kernel_kind(dtype, N, D, H, W, cin, cin_stride, cout, fused=False, tap_channel=None, background=False):
    """Which kernel the launcher picks for this 3x3x3 convolution (dua_conv3d_k3_kernel_kind: the launcher's own rule).  Only
    KIND_WIDE reads 16-channel-blocked input; only KIND_WIDE and KIND_FIRST write blocked output."""
    d = nv.Conv3Desc(nv.dt_code(dtype), N, D, H, W, cin, cin_stride, 0, cout, -(-cout // 8) * 8, 0,
                     0 if tap_channel is None else tap_channel + 1, 1 if background else 0, 0, CONV_POLICY & 0xff)
    return int(nv.lib().dua_conv3d_k3_kernel_kind(C.byref(d), 1 if fused else 0, 0))


def deconv_kernel_kind(dtype, N, D, H, W, cin, cout):
    """dua_deconv_k2s2_kernel_kind for an input of D x H x W voxels: DECONV_ALLTAPS is the kernel that may write blocked output."""
    d = nv.Conv3Desc(nv.dt_code(dtype), N, D, H, W, cin, cin, 0, cout, cout, 0, 0, 0, 0, 6 if (CONV_POLICY & 0xff) == 6 else 0)
    return int(nv.lib().dua_deconv_k2s2_kernel_kind(C.byref(d)))


def to_blocked(t):
    """channels-last [N, D, H, W, Cs] -> the same bytes count laid out as 16-channel blocks [N][Cs / 16][voxels][16]
    (dua_conv3_desc.layout), returned with the channels-last SHAPE (a buffer is a buffer: the kernels are told its layout)."""
    N, D, H, W, Cs = t.shape
    assert Cs % 16 == 0
    return t.reshape(N, D * H * W, Cs // 16, 16).permute(0, 2, 1, 3).contiguous().view(N, D, H, W, Cs)


def from_blocked(t):
    """inverse of to_blocked"""
    N, D, H, W, Cs = t.shape
    assert Cs % 16 == 0
    return t.reshape(N, Cs // 16, D * H * W, 16).permute(0, 2, 1, 3).contiguous().view(N, D, H, W, Cs)


def conv3d_k3(x, cin, cin_off, w_packed, bias_pad, cout, y, cout_off, out_stats, norm=None, workspace=None, tap_channel=None,
              background=False, in_blocked=False, out_blocked=False):
    """Raw 3x3x3 convolution (+bias); ``norm`` = producer descriptor of x (fused IN+LeakyReLU+add);
    accumulates this layer's InstanceNorm sums into ``out_stats`` (must be zero on entry).
    ``tap_channel`` (0 or 16, fp16, cin == tap_channel + 8): the single-channel tap form for first layers -- that packed
    channel is the last real input channel and is contracted as two k-steps over its 27 taps (weights packed with the
    same ``tap_channel``).
    ``background``: the launch runs on a second stream under a chain of small launches (one workgroup per CU, see
    dua_conv3_desc.background).  ``in_blocked`` / ``out_blocked``: x / y are laid out in 16-channel blocks
    (dua_conv3_desc.layout; the launcher rejects kernels that cannot)."""
    _cl_check(x, "x"); _cl_check(y, "y")
    assert x.dtype == y.dtype and x.device == y.device
    N, D, H, W, cs_in = x.shape
    assert tuple(y.shape[:4]) == (N, D, H, W)
    assert cin % 8 == 0 and cin_off % 8 == 0 and cin_off + cin <= cs_in
    assert cout % 8 == 0 and cout_off % 8 == 0 and cout_off + cout <= y.shape[-1]
    ck = chunk_elems(x.dtype)
    nch, nct = -(-cin // ck), -(-cout // 64)
    assert nch * ck <= 1024
    tap_bytes = 0
    if tap_channel is not None:
        assert x.dtype == torch.float16 and tap_channel in (0, 16) and cin == tap_channel + 8 and norm is None
        tap_bytes = nct * 4096
    assert w_packed.numel() == nct * nch * 27 * 4 * 64 * 16 + tap_bytes, "packed weights do not match (Cin, Cout, dtype)"
    assert bias_pad.numel() >= cout and bias_pad.dtype == torch.float32 and bias_pad.is_contiguous()     # the kernels read [0, cout)
    assert out_stats.dtype == torch.int64 and out_stats.is_contiguous() and tuple(out_stats.shape) == (N, STAT_REPLICAS, STAT_WORDS, nct * 64)
    d = nv.Conv3Desc(nv.dt_code(x.dtype), N, D, H, W, cin, cs_in, cin_off, cout, y.shape[-1], cout_off,
                     0 if tap_channel is None else tap_channel + 1, 1 if background else 0,
                     (nv.IN_BLOCKED if in_blocked else 0) | (nv.OUT_BLOCKED if out_blocked else 0), CONV_POLICY)
    if _RECORD is not None:
        has, nval = _norm_value(norm, N, cin)
        _RECORD.append(nv.StepOp(nv.OP_CONV3, has, d, nv.MaterializeDesc(),
                                 nval, _addr(x), _addr(w_packed), _addr(bias_pad), _addr(y), _addr(out_stats), None, None))
        return
    ws_bytes = 0
    if workspace is not None:
        assert workspace.is_cuda and workspace.is_contiguous()
        ws_bytes = workspace.numel() * workspace.element_size()
    nv.check(nv.lib().dua_conv3d_k3_fwd(C.byref(d), nv.ptr(x), nv.ptr(w_packed), nv.ptr(bias_pad), _norm_ref(norm, N, cin),
                                        nv.ptr(y), nv.ptr(out_stats), nv.ptr(workspace), ws_bytes, nv.stream_ptr()),
             "dua_conv3d_k3_fwd")


def conv3d_k3_wgrad(x, cin, cin_off, dy, cout, cout_off, dw, perm=None, workspace=None):
    """dw[Cout, Cin_src, 3,3,3] (fp32, reference layout) += weight gradient of the 3x3x3 convolution that mapped
    channels [cin_off, cin_off+cin) of ``x`` to channels [cout_off, cout_off+cout) of ``dy``'s buffer."""
    _cl_check(x, "x"); _cl_check(dy, "dy")
    assert x.dtype == dy.dtype and x.device == dy.device
    N, D, H, W, cs_in = x.shape
    assert tuple(dy.shape[:4]) == (N, D, H, W)
    assert cin % 8 == 0 and cin_off % 8 == 0 and cin_off + cin <= cs_in
    assert cout % 8 == 0 and cout_off % 8 == 0 and cout_off + cout <= dy.shape[-1]
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.dim() == 5 and tuple(dw.shape[2:]) == (3, 3, 3)
    assert dw.shape[0] == cout
    cin_src = dw.shape[1]
    if perm is None:
        assert cin_src <= cin
    else:
        assert perm.dtype == torch.int32 and perm.numel() >= -(-cin // 64) * 64
    d = nv.Conv3Desc(nv.dt_code(x.dtype), N, D, H, W, cin, cs_in, cin_off, cout, dy.shape[-1], cout_off, 0, 0, 0, WGRAD_POLICY)
    if workspace is None:
        need = nv.lib().dua_conv3d_k3_wgrad_workspace(C.byref(d))
        workspace = _wgrad_ws(need, x.device) if need > 0 else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    nv.check(nv.lib().dua_conv3d_k3_wgrad(C.byref(d), nv.ptr(x), nv.ptr(dy), nv.ptr(dw), cin_src, nv.ptr(perm),
                                          nv.ptr(workspace), ws_bytes, nv.stream_ptr()), "dua_conv3d_k3_wgrad")


_WGRAD_WS = {}
_WGRAD_WS_RETIRED = []


def _wgrad_ws(nbytes, device):
    """One grow-only scratch buffer per (device, stream) for the weight-gradient partial sums: reuse is ordered by the
    stream, so launches on different streams (conv wgrad on the trainer's side stream, deconv / head backward on the main
    one) must not share it.  A captured training-step graph bakes the buffer's address in, so a buffer that is outgrown is
    RETIRED, never freed: returning it to the caching allocator would let a replay scribble over whoever owns the memory
    next."""
    device = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WGRAD_WS.get(device)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WGRAD_WS_RETIRED.append(buf)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device[0])
        _WGRAD_WS[device] = buf
    return buf


_SPLITK_WS = {}
TRAIN_SPLITK = True          # tools/bench_train_ab.py switches it off for a same-process comparison
TRAIN_DGRAD_REDUCE = True    # a data-gradient launch takes the norm-backward reduce sums of the layer its output belongs to (96^3 layers)
TRAIN_FOLD_UPCONV = True     # training forward: UpCat's convolution over the concat as the folded launch (dua_upconv_k3_fwd)
TRAIN_FOLD_MIN_TILES = 1024  # ... where the launch has at least this many 8x8x8 tiles (96^3: 1728 per sample)


def splitk_ws(dtype, N, D, H, W, cin, cout, device):
    """Split-K scratch of conv3d_k3 for this layer shape (None when the launcher would not split): one grow-only buffer per
    (device, stream), retired instead of freed when outgrown, like the weight-gradient scratch above.  The training step's
    convolutions take it from here (the inference plans own theirs): without it the <= 12^3 levels run as a few dozen
    workgroups walking the whole contraction -- 100 us for the 6^3 512 -> 512 layer at batch 2 instead of ~25."""
    nbytes = conv3_workspace_bytes(dtype, N, D, H, W, cin, cout) if TRAIN_SPLITK else 0
    if nbytes <= 0:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SPLITK_WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            _WGRAD_WS_RETIRED.append(buf)
        buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _SPLITK_WS[key] = buf
    return buf


def instnorm_finalize(norm, N, Cc):
    """scale, shift = fp32 [N, C] exactly as consumers compute them in their preamble."""
    dev = norm.keep[0].device
    scale = torch.empty((N, Cc), dtype=torch.float32, device=dev)
    shift = torch.empty((N, Cc), dtype=torch.float32, device=dev)
    nv.check(nv.lib().dua_instnorm_finalize(N, Cc, norm.ref(N, Cc), nv.ptr(scale), nv.ptr(shift), nv.stream_ptr()),
             "dua_instnorm_finalize")
    return scale, shift


def instnorm_bwd_sums(raw, norm):
    """Zeroed accumulator of instnorm_bwd's reduce pass (fp64 [N, replicas, c_pad, 4])."""
    return zeros((raw.shape[0], STAT_REPLICAS, norm.keep[0].shape[3], 4), torch.float64, raw.device)


def instnorm_bwd(dA, da_off, raw, Cc, norm, dY, dy_off=0, want_add=True, dadd_out=None, sums=None):
    """Backward of LeakyReLU(IN(raw)) [+ add] for channels [0, Cc) of ``raw``: writes d raw into ``dY`` and returns the
    parameter gradients (dgamma [Cc], dbeta [Cc], dadd [N, Cc] or None), fp32, emitted by the apply launch itself.
    ``sums``: the reduce pass's result when the launch that produced dA took it along (conv3d_k3_dgrad_reduce)."""
    _cl_check(dA, "dA"); _cl_check(raw, "raw"); _cl_check(dY, "dY")
    N = raw.shape[0]
    vox = raw.shape[1] * raw.shape[2] * raw.shape[3]
    assert dA.dtype == raw.dtype == dY.dtype and tuple(dA.shape[:4]) == tuple(raw.shape[:4]) == tuple(dY.shape[:4])
    assert Cc % 8 == 0 and Cc <= raw.shape[-1] and da_off % 8 == 0 and da_off + Cc <= dA.shape[-1]
    assert dy_off % 8 == 0 and dy_off + Cc <= dY.shape[-1]
    d = nv.NormBwdDesc(nv.dt_code(raw.dtype), N, vox, Cc, dA.shape[-1], da_off, raw.shape[-1], 0, dY.shape[-1], dy_off)
    L = nv.lib()
    if sums is None:
        sums = instnorm_bwd_sums(raw, norm)
        nv.check(L.dua_instnorm_bwd_reduce(C.byref(d), nv.ptr(dA), nv.ptr(raw), norm.ref(N, Cc), nv.ptr(sums), nv.stream_ptr()),
                 "dua_instnorm_bwd_reduce")
    pg = zeros((2, Cc), torch.float32, raw.device)              # dgamma, dbeta (accumulated over the samples)
    dadd = None
    if want_add:
        dadd = dadd_out if dadd_out is not None else torch.empty((N, Cc), dtype=torch.float32, device=raw.device)
        assert dadd.dtype == torch.float32 and dadd.is_contiguous() and dadd.numel() == N * Cc
    nv.check(L.dua_instnorm_bwd_apply(C.byref(d), nv.ptr(dA), nv.ptr(raw), norm.ref(N, Cc), nv.ptr(sums), nv.ptr(dY),
                                      nv.ptr(pg[0]), nv.ptr(pg[1]), nv.ptr(dadd), nv.stream_ptr()), "dua_instnorm_bwd_apply")
    return pg[0], pg[1], dadd


def maxpool2_bwd_add(act, act_off, Cc, dA, da_off, dP):
    """dA[..., da_off:da_off+Cc] (or zeros when dA is None) + MaxPool3d(2) backward of dP routed through ``act``."""
    _cl_check(act, "act"); _cl_check(dP, "dP")
    N, D, H, W, _ = act.shape
    assert tuple(dP.shape[:4]) == (N, D // 2, H // 2, W // 2) and dP.shape[-1] >= Cc and dP.dtype == act.dtype
    if dA is not None:
        _cl_check(dA, "dA")
        assert tuple(dA.shape[:4]) == (N, D, H, W) and dA.dtype == act.dtype
    out = torch.empty((N, D, H, W, Cc), dtype=act.dtype, device=act.device)
    nv.check(nv.lib().dua_maxpool2_bwd_add(nv.dt_code(act.dtype), N, D, H, W, Cc, nv.ptr(act), act.shape[-1], act_off,
                                           nv.ptr(dA), dA.shape[-1] if dA is not None else 0, da_off, nv.ptr(dP),
                                           dP.shape[-1], nv.ptr(out), Cc, nv.stream_ptr()), "dua_maxpool2_bwd_add")
    return out


def deconv_k2s2_bwd(x, cin, cin_off, dy, cout, cout_off, w, need_dx=True, need_dw=True):
    """Backward of deconv_k2s2: x [N,D,H,W,*] (slice cin@cin_off), dy [N,2D,2H,2W,*] (slice cout@cout_off, read in place),
    w fp32 [Cin, Cout, 2,2,2].  Returns (dx [N,D,H,W,cin] or None, dw fp32 like w or None)."""
    _cl_check(x, "x"); _cl_check(dy, "dy")
    N, D, H, W, cs_in = x.shape
    assert x.dtype == dy.dtype and tuple(dy.shape[:4]) == (N, 2 * D, 2 * H, 2 * W)
    assert cin % 8 == 0 and cin_off % 8 == 0 and cin_off + cin <= cs_in
    assert cout % 8 == 0 and cout_off % 8 == 0 and cout_off + cout <= dy.shape[-1]
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (cin, cout, 2, 2, 2)
    L, code = nv.lib(), nv.dt_code(x.dtype)
    dx = dw = wp = ws = None
    ws_bytes = 0
    d_out = nv.Conv3Desc(code, N, D, H, W, cin, cin, 0, cout, dy.shape[-1], cout_off)      # dx is its own dense buffer
    if need_dx:
        nbytes = L.dua_pack_deconv_weights_dgrad(code, cin, cout, None, None, None)
        wp = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = L.dua_pack_deconv_weights_dgrad(code, cin, cout, nv.ptr(w), nv.ptr(wp), nv.stream_ptr())
        if rc != nbytes:
            raise RuntimeError(f"dua_pack_deconv_weights_dgrad failed ({rc})")
        dx = torch.empty((N, D, H, W, cin), dtype=x.dtype, device=x.device)
        nv.check(L.dua_deconv_k2s2_bwd(C.byref(d_out), None, nv.ptr(dy), nv.ptr(wp), nv.ptr(dx), None, None, 0,
                                       nv.stream_ptr()), "dua_deconv_k2s2_bwd(dx)")
    if need_dw:
        d_in = nv.Conv3Desc(code, N, D, H, W, cin, cs_in, cin_off, cout, dy.shape[-1], cout_off)
        ws_bytes = int(L.dua_deconv_k2s2_bwd_workspace(C.byref(d_in)))
        ws = _wgrad_ws(ws_bytes, x.device)
        dw = zeros(tuple(w.shape), w.dtype, w.device)
        nv.check(L.dua_deconv_k2s2_bwd(C.byref(d_in), nv.ptr(x), nv.ptr(dy), None, None, nv.ptr(dw), nv.ptr(ws), ws.numel(),
                                       nv.stream_ptr()), "dua_deconv_k2s2_bwd(dw)")
    return dx, dw


HEAD_MAX_K, HEAD_MAX_C = 16, 64


def head_fwd(u, weight, bias, out=None):
    """logits [N, D, H, W, K] = 1x1x1 convolution of the channels-last activation ``u`` (weight fp32 [K, C], bias fp32 [K])."""
    _cl_check(u, "u")
    K, Cc = weight.shape
    assert K <= HEAD_MAX_K and Cc <= HEAD_MAX_C and Cc % 8 == 0 and Cc <= u.shape[-1]
    assert weight.dtype == torch.float32 and weight.is_contiguous() and bias.dtype == torch.float32 and bias.numel() == K
    if out is None:
        out = torch.empty((*u.shape[:4], K), dtype=u.dtype, device=u.device)
    assert out.is_contiguous() and out.dtype == u.dtype and tuple(out.shape[:4]) == tuple(u.shape[:4]) and out.shape[-1] >= K
    vox = u.numel() // u.shape[-1]
    nv.check(nv.lib().dua_head_fwd(nv.dt_code(u.dtype), vox, Cc, K, nv.ptr(u), u.shape[-1], nv.ptr(weight), nv.ptr(bias),
                                   nv.ptr(out), out.shape[-1], nv.stream_ptr()), "dua_head_fwd")
    return out


def head_bwd(dlogits, u, weight):
    """(du, dW [K, C], db [K]) of head_fwd."""
    _cl_check(u, "u")
    K, Cc = weight.shape
    assert dlogits.is_cuda and dlogits.is_contiguous() and dlogits.dtype == u.dtype and dlogits.shape[-1] == K
    assert tuple(dlogits.shape[:4]) == tuple(u.shape[:4]) and u.shape[-1] == Cc
    du = torch.empty_like(u)
    dW = zeros((K, Cc), torch.float32, u.device)
    db = zeros((K,), torch.float32, u.device)
    vox = u.numel() // Cc
    ws = _wgrad_ws(int(nv.lib().dua_head_bwd_workspace(vox)), u.device)         # shared grow-only scratch (stream-ordered)
    nv.check(nv.lib().dua_head_bwd(nv.dt_code(u.dtype), vox, Cc, K, nv.ptr(dlogits), K, nv.ptr(u), Cc, nv.ptr(weight),
                                   nv.ptr(du), Cc, nv.ptr(dW), nv.ptr(db), nv.ptr(ws), ws.numel(), nv.stream_ptr()),
             "dua_head_bwd")
    return du, dW, db


LOSS_NAMES = ("mse", "bce", "dice")


def seg_loss_reduce(logits, labels, names=LOSS_NAMES, combine="sum"):
    """logits: channels-last [N, D, H, W, Cs] (first C = labels.shape[1] channels); labels fp32 [N, C, D, H, W].
    Loss of losses/loss.py:25-86 for ``names`` (a subset of mse / bce / dice) combined by "sum" / "mean" / "log".
    Returns (L as a 0-dim fp32 tensor, the fp64 sums the gradient kernel needs, d L / d (sum of terms) as a 0-dim fp32
    device tensor -- 1 for "sum", 1/len(names) for "mean", 1/(1 + sum) for "log")."""
    assert logits.is_cuda and logits.is_contiguous() and logits.dim() == 5
    N, Cc = labels.shape[:2]
    V = labels.shape[2] * labels.shape[3] * labels.shape[4]
    assert labels.is_cuda and labels.dtype == torch.float32 and labels.is_contiguous()
    assert tuple(logits.shape[:4]) == (N, *labels.shape[2:]) and logits.shape[-1] >= Cc
    assert names and all(n_ in LOSS_NAMES for n_ in names) and combine in ("sum", "mean", "log")
    sums = zeros((N * Cc * 4 + 2,), torch.float64, logits.device)
    nv.check(nv.lib().dua_seg_loss_reduce(nv.dt_code(logits.dtype), N, Cc, V, nv.ptr(logits), logits.shape[-1], nv.ptr(labels),
                                          nv.ptr(sums), nv.stream_ptr()), "dua_seg_loss_reduce")
    out = torch.empty(2, dtype=torch.float32, device=logits.device)             # (L, d L / d total)
    nv.check(nv.lib().dua_seg_loss_finish(N, Cc, V, int("mse" in names), int("bce" in names), int("dice" in names),
                                          ("sum", "mean", "log").index(combine), nv.ptr(sums), nv.ptr(out[0:]), nv.ptr(out[1:]),
                                          nv.stream_ptr()), "dua_seg_loss_finish")
    return out[0], sums, out[1]


def seg_loss_grad(logits, labels, sums, gscale, names=LOSS_NAMES):
    """dlogits = gscale * sum over ``names`` of d term / d logits (``gscale``: 0-dim device tensor or None = 1)."""
    N, Cc = labels.shape[:2]
    V = labels.shape[2] * labels.shape[3] * labels.shape[4]
    out = zeros(tuple(logits.shape), logits.dtype, logits.device) if logits.shape[-1] > Cc else torch.empty_like(logits)
    g = gscale.detach().float().reshape(1).contiguous() if gscale is not None else None
    w = [1.0 if n_ in names else 0.0 for n_ in LOSS_NAMES]
    nv.check(nv.lib().dua_seg_loss_grad(nv.dt_code(logits.dtype), N, Cc, V, nv.ptr(logits), logits.shape[-1], nv.ptr(labels),
                                        nv.ptr(sums), nv.ptr(g), w[0], w[1], w[2], nv.ptr(out), out.shape[-1],
                                        nv.stream_ptr()), "dua_seg_loss_grad")
    return out


def materialize(raw, Cc, norm, out, out_off, emb=None, pooled=None, out_blocked=False):
    _cl_check(raw, "raw"); _cl_check(out, "out")
    N, D, H, W, rs = raw.shape
    assert tuple(out.shape[:4]) == (N, D, H, W) and out.dtype == raw.dtype
    assert Cc % 8 == 0 and Cc <= rs and out_off % 8 == 0 and out_off + Cc <= out.shape[-1]
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
    d = nv.MaterializeDesc(nv.dt_code(raw.dtype), N, D, H, W, Cc, rs, es, out.shape[-1], out_off, ps, 1 if out_blocked else 0)
    if _RECORD is not None:
        has, nval = _norm_value(norm, N, Cc)
        _RECORD.append(nv.StepOp(nv.OP_MATERIALIZE, has, nv.Conv3Desc(), d, nval, _addr(raw), None, None, _addr(out), None,
                                 _addr(emb), _addr(pooled)))
        return
    nv.check(nv.lib().dua_materialize(C.byref(d), nv.ptr(raw), norm.ref(N, Cc), nv.ptr(emb), nv.ptr(out),
                                      nv.ptr(pooled), nv.stream_ptr()), "dua_materialize")


def deconv_k2s2(x, cin, cin_off, w_packed, bias_pad, cout, y, cout_off, norm=None, out_blocked=False):
    """ConvTranspose3d(k2,s2) of a channel slice of ``x`` into a channel slice of ``y`` (2x spatial)."""
    _cl_check(x, "x"); _cl_check(y, "y")
    N, D, H, W, cs_in = x.shape
    assert x.dtype == y.dtype and tuple(y.shape[:4]) == (N, 2 * D, 2 * H, 2 * W)
    assert cin % 8 == 0 and cin_off % 8 == 0 and cin_off + cin <= cs_in
    assert cout % 8 == 0 and cout_off % 8 == 0 and cout_off + cout <= y.shape[-1]
    ck = chunk_elems(x.dtype)
    nch, nct = -(-cin // ck), -(-cout // 64)
    assert nch * ck <= 1024
    assert w_packed.numel() == 8 * nct * nch * 4 * 64 * 16 and bias_pad.numel() == nct * 64
    d = nv.Conv3Desc(nv.dt_code(x.dtype), N, D, H, W, cin, cs_in, cin_off, cout, y.shape[-1], cout_off, 0, 0,
                     nv.OUT_BLOCKED if out_blocked else 0, 6 if (CONV_POLICY & 0xff) == 6 else 0)
    if _RECORD is not None:
        has, nval = _norm_value(norm, N, cin)
        _RECORD.append(nv.StepOp(nv.OP_DECONV, has, d, nv.MaterializeDesc(), nval, _addr(x), _addr(w_packed), _addr(bias_pad),
                                 _addr(y), None, None, None))
        return
    nv.check(nv.lib().dua_deconv_k2s2_fwd(C.byref(d), nv.ptr(x), nv.ptr(w_packed), nv.ptr(bias_pad), _norm_ref(norm, N, cin),
                                          nv.ptr(y), nv.stream_ptr()), "dua_deconv_k2s2_fwd")


def pack_deconv_res_weights(w3, wd, cskip, dtype=torch.float16, up_first=True, cu_packed=None):
    """Weights of deconv_res (dua_deconv_k2s2_res_fwd): ``w3`` = the 1x1x1 convolution over cat((up, skip)) as a matrix
    [Cout, Cmid + Cskip] (``up_first``: the upsampled half first, as torch.cat((up, skip)) of the Swin-UNETR decoder), ``wd`` =
    ConvTranspose3d [Cu, Cmid, 2,2,2].  Returns (packed composed weights Cu -> Cout per tap, packed skip weights).  The
    composition W3_up . Wd[child]^T is a handful of small matrix products at weight-load time (fp32)."""
    assert w3.is_cuda and w3.dim() == 2 and wd.dim() == 5 and tuple(wd.shape[2:]) == (2, 2, 2)
    cout = w3.shape[0]
    cu, cmid = wd.shape[:2]
    assert w3.shape[1] == cmid + cskip
    w3 = w3.detach().float()
    w3_up, w3_sk = (w3[:, :cmid], w3[:, cmid:]) if up_first else (w3[:, cskip:], w3[:, :cskip])
    comp = torch.einsum("umk,om->uok", wd.detach().float().reshape(cu, cmid, 8), w3_up).contiguous()
    if cu_packed is not None and cu_packed > cu:
        comp = torch.cat([comp, torch.zeros(cu_packed - cu, cout, 8, dtype=comp.dtype, device=comp.device)], 0).contiguous()
    wp, _ = pack_deconv_weights(comp.view(-1, cout, 2, 2, 2), None, dtype)
    sk8 = w3_sk.t().contiguous()[:, :, None].expand(cskip, cout, 8).contiguous().view(cskip, cout, 2, 2, 2)
    wsp, _ = pack_deconv_weights(sk8, None, dtype)
    return wp, wsp


def deconv_res_supported(dtype, N, D, H, W, cin, cin_stride, cout, cout_stride, cs):
    """dua_deconv_k2s2_res_supported for COARSE extents D x H x W."""
    d = nv.Conv3Desc(nv.dt_code(dtype), N, D, H, W, cin, cin_stride, 0, cout, cout_stride, 0)
    return bool(nv.lib().dua_deconv_k2s2_res_supported(C.byref(d), cs))


def deconv_res(lo, cin, cin_off, w_packed, xs, cs, cs_off, ws_packed, cout, y, cout_off, stats):
    """res = conv1x1x1(cat((ConvTranspose3d_k2s2(lo), skip))) without the upsampled tensor (dua_deconv_k2s2_res_fwd): ``lo`` coarse
    [N, D, H, W, *], ``xs`` / ``y`` on the 2x grid; accumulates the layer's InstanceNorm sums into ``stats``."""
    _cl_check(lo, "lo"); _cl_check(xs, "xs"); _cl_check(y, "y")
    N, D, H, W, cs_in = lo.shape
    assert lo.dtype == xs.dtype == y.dtype == torch.float16
    assert tuple(xs.shape[:4]) == tuple(y.shape[:4]) == (N, 2 * D, 2 * H, 2 * W)
    assert cin_off + cin <= cs_in and cs_off + cs <= xs.shape[-1] and cout_off + cout <= y.shape[-1]
    assert _RECORD is None, "not an op of dua_denoiser_step"
    d = nv.Conv3Desc(nv.dt_code(lo.dtype), N, D, H, W, cin, cs_in, cin_off, cout, y.shape[-1], cout_off)
    nv.check(nv.lib().dua_deconv_k2s2_res_fwd(C.byref(d), nv.ptr(lo), nv.ptr(w_packed), nv.ptr(xs), cs, xs.shape[-1], cs_off,
                                              nv.ptr(ws_packed), nv.ptr(y), nv.ptr(stats), nv.stream_ptr()), "dua_deconv_k2s2_res_fwd")


def pack_upconv_weights(wc, bc, wd, bd, cskip, dtype=torch.float16, up_first=False, cu_packed=None):
    """UpCat's first convolution with the transposed convolution folded in (dua_upconv_k3_fwd): ``wc`` / ``bc`` = Conv3d
    [Cout, Cskip + Cmid, 3,3,3] parameters, ``wd`` / ``bd`` = ConvTranspose3d [Cu, Cmid, 2,2,2] parameters.  ``up_first``: the
    upsampled half is the FIRST Cmid input channels of ``wc`` (torch.cat((up, skip)): the Swin-UNETR decoder), else the last.
    ``cu_packed`` (a multiple of 64, >= Cu): channel count of the coarse BUFFER the kernel will read (channels behind Cu are zero
    padding; their weight rows are zeros).  Returns (packed skip-half weights, composed weights of the upsampled half, bias table
    fp32 [27, ceil(Cout/64)*64])."""
    assert wc.is_cuda and wc.dtype == torch.float32 and wc.dim() == 5 and tuple(wc.shape[2:]) == (3, 3, 3)
    assert wd.is_cuda and wd.dtype == torch.float32 and wd.dim() == 5 and tuple(wd.shape[2:]) == (2, 2, 2)
    wc, wd = wc.contiguous(), wd.contiguous()
    cout, cin = wc.shape[:2]
    cu, cmid = wd.shape[:2]
    cu_packed = cu if cu_packed is None else cu_packed
    assert cin == cskip + cmid and cskip % 16 == 0 and cu_packed % 64 == 0 and cu_packed >= cu and dtype == torch.float16
    if cu_packed > cu:
        wd = torch.cat([wd, torch.zeros(cu_packed - cu, *wd.shape[1:], dtype=wd.dtype, device=wd.device)], 0).contiguous()
    L = nv.lib()
    code = nv.dt_code(dtype)
    perm = torch.arange(cmid, cmid + cskip, dtype=torch.int32, device=wc.device) if up_first else None
    nb = L.dua_pack_conv3_weights(code, cout, cin, cskip, None, None, None, None)
    w_skip = torch.empty(nb, dtype=torch.uint8, device=wc.device)
    if L.dua_pack_conv3_weights(code, cout, cin, cskip, nv.ptr(wc), nv.ptr(perm), nv.ptr(w_skip), nv.stream_ptr()) != nb:
        raise RuntimeError("dua_pack_conv3_weights failed")
    nb = L.dua_pack_upconv_weights(code, cout, cskip, cmid, cu_packed, 0, None, None, None, None, None, None, None)
    assert nb > 0
    wu = torch.empty(nb, dtype=torch.uint8, device=wc.device)
    btab = torch.empty((27, -(-cout // 64) * 64), dtype=torch.float32, device=wc.device)
    bc_ = None if bc is None else bc.detach().float().contiguous()
    bd_ = None if bd is None else bd.detach().float().contiguous()
    rc = L.dua_pack_upconv_weights(code, cout, cskip, cmid, cu_packed, 1 if up_first else 0, nv.ptr(wc), nv.ptr(bc_), nv.ptr(wd), nv.ptr(bd_),
                                   nv.ptr(wu), nv.ptr(btab), nv.stream_ptr())
    if rc != nb:
        raise RuntimeError(f"dua_pack_upconv_weights failed ({rc})")
    return w_skip, wu, btab


def _upconv_desc(xs, cskip, cskip_off, u, cu, cu_off, cout, y, cout_off, in_blocked, out_blocked):
    N, D, H, W, cs = xs.shape
    return nv.UpConvDesc(nv.dt_code(xs.dtype), N, D, H, W, cskip, cs, cskip_off, cu, u.shape[-1], cu_off, cout, y.shape[-1], cout_off,
                         (nv.IN_BLOCKED if in_blocked else 0) | (nv.OUT_BLOCKED if out_blocked else 0))


def upconv_supported(dtype, N, D, H, W, cskip, cskip_stride, cu, cu_stride, cout, cout_stride, in_blocked=False, out_blocked=False):
    """dua_upconv_k3_supported for OUTPUT extents D x H x W."""
    d = nv.UpConvDesc(nv.dt_code(dtype), N, D, H, W, cskip, cskip_stride, 0, cu, cu_stride, 0, cout, cout_stride, 0,
                      (nv.IN_BLOCKED if in_blocked else 0) | (nv.OUT_BLOCKED if out_blocked else 0))
    return bool(nv.lib().dua_upconv_k3_supported(C.byref(d)))


def upconv_k3(xs, cskip, cskip_off, u, cu, cu_off, norm, w_skip, wu, btab, cout, y, cout_off, out_stats, in_blocked=False,
              out_blocked=False):
    """conv3x3x3(cat([x_e, ConvTranspose3d_k2s2(act(norm(u)))])) in one launch (dua_upconv_k3_fwd): ``xs`` holds x_e on the fine
    grid, ``u`` is the RAW coarse tensor whose producer ``norm`` describes (``norm`` None: ``u`` already is an activation);
    accumulates this layer's InstanceNorm sums."""
    _cl_check(xs, "xs"); _cl_check(u, "u"); _cl_check(y, "y")
    N, D, H, W, _ = xs.shape
    assert xs.dtype == u.dtype == y.dtype == torch.float16 and tuple(y.shape[:4]) == (N, D, H, W)
    assert tuple(u.shape[:4]) == (N, D // 2, H // 2, W // 2) and D % 8 == 0 and H % 8 == 0 and W % 8 == 0
    assert cskip % 16 == 0 and cskip_off + cskip <= xs.shape[-1] and cu % 64 == 0 and cu_off + cu <= u.shape[-1]
    assert cout % 8 == 0 and cout_off + cout <= y.shape[-1]
    nct = -(-cout // 64)
    assert w_skip.numel() == nct * (-(-cskip // 32)) * 27 * 4 * 64 * 16 and wu.numel() == nct * 4 * (cu // 64) * 128 * 1024
    assert btab.dtype == torch.float32 and btab.is_contiguous() and tuple(btab.shape) == (27, nct * 64)
    assert out_stats.dtype == torch.int64 and out_stats.is_contiguous() and tuple(out_stats.shape) == (N, STAT_REPLICAS, STAT_WORDS, nct * 64)
    d = _upconv_desc(xs, cskip, cskip_off, u, cu, cu_off, cout, y, cout_off, in_blocked, out_blocked)
    if _RECORD is not None:
        has, nval = _norm_value(norm, N, cu)
        _RECORD.append(nv.StepOp(nv.OP_UPCONV, has, nv.Conv3Desc(), nv.MaterializeDesc(), nval, _addr(xs), _addr(w_skip), _addr(btab),
                                 _addr(y), _addr(out_stats), None, None, d, _addr(u), _addr(wu)))
        return
    nv.check(nv.lib().dua_upconv_k3_fwd(C.byref(d), nv.ptr(xs), nv.ptr(u), _norm_ref(norm, N, cu), nv.ptr(w_skip), nv.ptr(wu), nv.ptr(btab),
                                        nv.ptr(y), nv.ptr(out_stats), nv.stream_ptr()), "dua_upconv_k3_fwd")


def _f32c(t, name):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), f"{name}: contiguous fp32 device tensor"


def q_sample(x0, eps, coef, out=None):
    """coef: fp32[N,2] device."""
    _f32c(x0, "x0"); _f32c(eps, "eps"); _f32c(coef, "coef")
    assert eps.shape == x0.shape and coef.numel() == 2 * x0.shape[0]
    out = torch.empty_like(x0) if out is None else out
    _f32c(out, "out")
    nv.check(nv.lib().dua_q_sample(x0.shape[0], x0[0].numel(), nv.ptr(x0), nv.ptr(eps), nv.ptr(coef), nv.ptr(out),
                                   nv.stream_ptr()), "dua_q_sample")
    return out


def linear_f32(x, w, bias=None, gelu=False, out=None):
    """F.linear(x, w, bias) (optionally followed by the exact GELU) on fp32 rows through dua_linear_f32 (exact-fp32 MFMA): the
    nn.Linear layers and 1x1x1 convolutions of the Swin path's fp32 parity plan.  ``x``: [..., K] with unit stride along K and
    one row stride (a contiguous tensor, or a channel slice of one); ``out``: [..., N] contiguous."""
    K, Nn = x.shape[-1], w.shape[0]
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1 and K % 4 == 0
    _f32c(w, "weight")
    assert tuple(w.shape) == (Nn, K)
    rows = x.numel() // K
    if x.is_contiguous():
        lda = K
    else:
        x2 = x.reshape(-1, K) if x.dim() != 2 else x
        assert x2.data_ptr() == x.data_ptr() and x2.stride(1) == 1, "rows of x must share one stride"
        lda = x2.stride(0)
    if bias is not None:
        _f32c(bias, "bias")
    if out is None:
        out = torch.empty((*x.shape[:-1], Nn), dtype=torch.float32, device=x.device)
    _f32c(out, "out")
    assert out.numel() == rows * Nn
    nv.check(nv.lib().dua_linear_f32(rows, K, Nn, nv.ptr(x), lda, nv.ptr(w), nv.ptr(bias), nv.ptr(out), Nn, int(gelu),
                                     nv.stream_ptr()), "dua_linear_f32")
    return out


def q_sample_affine(src, a, b, eps, sched, t, out=None):
    """q_sample(a * src + b, t, eps) in one pass (train.py:258-262): sched fp32 [T, 2] = (sqrt(alphas_cumprod),
    sqrt(1 - alphas_cumprod)), t int64 [N], both on the device."""
    _f32c(src, "src"); _f32c(eps, "eps"); _f32c(sched, "sched")
    assert eps.shape == src.shape and sched.dim() == 2 and sched.shape[1] == 2
    assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == src.shape[0]
    out = torch.empty_like(src) if out is None else out
    _f32c(out, "out")
    nv.check(nv.lib().dua_q_sample_affine(src.shape[0], src[0].numel(), nv.ptr(src), float(a), float(b), nv.ptr(eps), nv.ptr(sched),
                                          sched.shape[0], nv.ptr(t), nv.ptr(out), nv.stream_ptr()), "dua_q_sample_affine")
    return out


_TEMB_FREQS = {}


def temb_freqs(half, device):
    """exp(arange(half) * -(ln 10000 / (half - 1))) as models/diffusion/utils.py:15-17 computes it (fp32), cached per device."""
    key = (half, str(device))
    if key not in _TEMB_FREQS:
        _TEMB_FREQS[key] = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(device)
    return _TEMB_FREQS[key]


def _temb_blocks(ws, bs=None, dws=None, dbs=None):
    assert 0 < len(ws) <= nv.TEMB_MAX_BLOCKS
    blk = nv.TembBlocks()
    blk.nblocks = len(ws)
    for i, w in enumerate(ws):
        _f32c(w, "temb_proj.weight")
        blk.cout[i] = w.shape[0]
        blk.w[i] = w.data_ptr()
        for arr, src in ((blk.b, bs), (blk.dw, dws), (blk.db, dbs)):
            if src is not None:
                _f32c(src[i], "temb_proj operand")
                arr[i] = src[i].data_ptr()
    return blk


def temb_train_fwd(t, half, w0, b0, w1, b1, proj_w, proj_b):
    """Timestep embedding of a training step (utils.py:5-54 + denoiser.py:51-52,65) for int64 timesteps ``t`` [N]: returns
    (add, saved): add = flat fp32, BLOCK-MAJOR (block b's [N, cout_b] rows at offset N * sum(cout[:b])); saved = the
    activations dua_temb_train_bwd needs."""
    assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()
    for v in (w0, b0, w1, b1):
        _f32c(v, "temb param")
    N, hid = t.numel(), w1.shape[0]
    assert tuple(w0.shape) == (hid, 2 * half) and tuple(w1.shape) == (hid, hid) and all(w.shape[1] == hid for w in proj_w)
    P = sum(w.shape[0] for w in proj_w)
    add = torch.empty(N * P, dtype=torch.float32, device=t.device)
    saved = torch.empty((N, 2 * half + 4 * hid), dtype=torch.float32, device=t.device)
    blk = _temb_blocks(proj_w, bs=proj_b)
    nv.check(nv.lib().dua_temb_train_fwd(N, nv.ptr(t), nv.ptr(temb_freqs(half, t.device)), half, hid, nv.ptr(w0), nv.ptr(b0),
                                         nv.ptr(w1), nv.ptr(b1), C.byref(blk), nv.ptr(add), nv.ptr(saved), nv.stream_ptr()),
             "dua_temb_train_fwd")
    return add, saved


def temb_train_bwd(dadd, saved, half, w1, proj_w):
    """Every parameter gradient of temb_train_fwd from d add (block-major like add): returns (dw0, db0, dw1, db1, [dw_b], [db_b]),
    views of one flat fp32 buffer, written (not accumulated) by three launches."""
    _f32c(dadd, "dadd"); _f32c(saved, "saved"); _f32c(w1, "w1")
    N, hid = saved.shape[0], w1.shape[0]
    ed = 2 * half
    P = sum(w.shape[0] for w in proj_w)
    assert dadd.numel() == N * P and saved.shape[1] == ed + 4 * hid
    nscratch = N * hid * (1 + -(-P // 64) + hid // 64)
    flat = torch.empty(hid * ed + hid + hid * hid + hid + P * hid + P + nscratch, dtype=torch.float32, device=saved.device)
    pos = [0]

    def take(*shape):
        n = 1
        for d_ in shape:
            n *= d_
        v = flat[pos[0]:pos[0] + n].view(*shape)
        pos[0] += n
        return v

    dw0, db0, dw1, db1 = take(hid, ed), take(hid), take(hid, hid), take(hid)
    dws = [take(w.shape[0], hid) for w in proj_w]
    dbs = [take(w.shape[0]) for w in proj_w]
    scratch = take(nscratch)
    blk = _temb_blocks(proj_w, dws=dws, dbs=dbs)
    nv.check(nv.lib().dua_temb_train_bwd(N, half, hid, nv.ptr(w1), C.byref(blk), nv.ptr(dadd), nv.ptr(saved), nv.ptr(scratch),
                                         nv.ptr(dw0), nv.ptr(db0), nv.ptr(dw1), nv.ptr(db1), nv.stream_ptr()), "dua_temb_train_bwd")
    return dw0, db0, dw1, db1, dws, dbs


def _adamw_lists(params, grads, ms, vs):
    """dua_adamw_list chunks (<= 64 tensors each, passed by value) over parallel lists of contiguous fp32 tensors."""
    lists = []
    for i0 in range(0, len(params), nv.ADAMW_MAX_TENSORS):
        l = nv.AdamWList()
        chunk = range(i0, min(len(params), i0 + nv.ADAMW_MAX_TENSORS))
        l.count = len(chunk)
        for j, i in enumerate(chunk):
            g = grads[i]
            assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous(), "gradients must be dense fp32 device tensors"
            l.numel[j] = g.numel()
            l.g[j] = g.data_ptr()
            if params is not None and params[i] is not None:
                p, m, v = params[i], ms[i], vs[i]
                assert p.numel() == m.numel() == v.numel() == g.numel() and p.dtype == torch.float32 and p.is_contiguous()
                l.p[j], l.m[j], l.v[j] = p.data_ptr(), m.data_ptr(), v.data_ptr()
        lists.append(l)
    return lists


def grads_nonfinite(grads, found_inf):
    """found_inf (fp32 device scalar, zeroed by the caller) = 1 if any element of ``grads`` is Inf / NaN -- the check of
    torch._amp_foreach_non_finite_check_and_unscale_ without the write-back (dua_adamw_step applies 1 / scale itself)."""
    assert found_inf.is_cuda and found_inf.dtype == torch.float32
    L = nv.lib()
    for l in _adamw_lists([None] * len(grads), grads, None, None):
        nv.check(L.dua_grads_nonfinite(C.byref(l), nv.ptr(found_inf), nv.stream_ptr()), "dua_grads_nonfinite")


def adamw_step(params, grads, ms, vs, step, lr, betas, eps, weight_decay, lr_dev=None, grad_scale=None, found_inf=None,
               store_grad=False):
    """One AdamW update of every tensor (torch.optim.AdamW's arithmetic, train.py:121-122) with k = step + 1; ``step``: int32
    device scalar, not advanced here (adamw_advance)."""
    assert step.is_cuda and step.dtype == torch.int32
    L = nv.lib()
    for l in _adamw_lists(params, grads, ms, vs):
        nv.check(L.dua_adamw_step(C.byref(l), float(lr), nv.ptr(lr_dev), float(betas[0]), float(betas[1]), float(eps),
                                  float(weight_decay), nv.ptr(grad_scale), nv.ptr(found_inf), nv.ptr(step), int(store_grad),
                                  nv.stream_ptr()), "dua_adamw_step")


def adamw_advance(step, found_inf=None, scale=None, growth=None, growth_factor=2.0, backoff=0.5, interval=200, seen=None):
    """End of a step: the update counter and torch._amp_update_scale_'s loss-scale rule, on the device; ``found_inf`` is cleared
    for the next step and its value left in ``seen`` (fp32 device scalar) for the host to read."""
    assert step.dtype == torch.int32 and (growth is None or growth.dtype == torch.int32)
    nv.check(nv.lib().dua_adamw_advance(nv.ptr(step), nv.ptr(found_inf), nv.ptr(scale), nv.ptr(growth), float(growth_factor),
                                        float(backoff), int(interval), nv.ptr(seen), nv.stream_ptr()), "dua_adamw_advance")


def sampler_step(mode, model_out, x, eps, coef, x_out=None, xstart_out=None, xstart_sum=None):
    for n_, t in (("model_out", model_out), ("x", x), ("eps", eps), ("coef", coef)):
        _f32c(t, n_)
    assert model_out.shape == x.shape == eps.shape and coef.numel() == 8 * x.shape[0]
    x_out = torch.empty_like(x) if x_out is None else x_out
    for t in (x_out, xstart_out, xstart_sum):
        if t is not None:
            _f32c(t, "out"); assert t.shape == x.shape
    nv.check(nv.lib().dua_sampler_step(mode, x.shape[0], x[0].numel(), nv.ptr(model_out), nv.ptr(x), nv.ptr(eps),
                                       nv.ptr(coef), nv.ptr(x_out), nv.ptr(xstart_out), nv.ptr(xstart_sum),
                                       nv.stream_ptr()), "dua_sampler_step")
    return x_out


def state_stride(num_classes):
    cx = -(-num_classes // 8) * 8
    assert cx <= 32, "at most 32 classes"
    return cx


def final_conv_sampler(raw, K, norm, wf, bf, num_classes, mode, coef=None, x_state=None, noise=None,
                       step_word=None, xin=None, xstart_sum=None, logits=None, xstart=None, seed=0, seed_dev=None,
                       residual=None):
    """``seed_dev``: optional int64[1] device tensor holding the Philox key (read by the kernel at run time, so a
    captured graph draws a fresh noise field whenever the host rewrites the word); otherwise ``seed`` is the key.
    ``residual`` = (res, res_norm, ra_src, ra_off, channels): the residual form (dua_final_conv_sampler_res) -- the 1x1x1
    convolution reads LeakyReLU(norm(raw) + res_norm(res)) + reverse_attention(ra_src[..., ra_off:ra_off + channels])
    assembled in registers; fp16, K = channels rounded up to 32, ``wf`` columns behind ``channels`` zero."""
    _cl_check(raw, "raw")
    N, D, H, W, rs = raw.shape
    vox = D * H * W
    cx = state_stride(num_classes)
    if residual is not None:
        res, res_norm, ra_src, ra_off, channels = residual
        _cl_check(res, "res")
        assert raw.dtype == torch.float16 and res.dtype == raw.dtype and tuple(res.shape[:4]) == (N, D, H, W)
        assert norm is not None and res_norm is not None and cx == 16 and K in (32, 64)
        assert channels % 8 == 0 and 0 < channels <= min(K, rs, res.shape[-1])
        if ra_src is not None:
            _cl_check(ra_src, "ra_src")
            assert ra_src.dtype == raw.dtype and tuple(ra_src.shape[:4]) == (N, D, H, W)
            assert ra_off % 8 == 0 and ra_off + channels <= ra_src.shape[-1]
    assert K % 8 == 0 and (K <= rs or residual is not None) and K <= 512
    _f32c(wf, "wf"); _f32c(bf, "bf")
    assert wf.numel() == num_classes * K and bf.numel() == num_classes
    if mode != nv.MODE_LOGITS:
        _f32c(coef, "coef"); _f32c(x_state, "x_state")
        assert coef.numel() >= 8 * N and x_state.numel() == N * vox * cx
    if noise is not None:
        _f32c(noise, "noise"); assert noise.numel() == N * num_classes * vox
    for t in (logits, xstart):
        if t is not None:
            _f32c(t, "ncdhw out"); assert t.numel() == N * num_classes * vox
    if xstart_sum is not None:
        _f32c(xstart_sum, "xstart_sum"); assert xstart_sum.numel() == N * vox * cx
    xs = 0
    if xin is not None:
        _cl_check(xin, "xin")
        assert xin.dtype == raw.dtype and tuple(xin.shape[:4]) == (N, D, H, W) and xin.shape[-1] >= num_classes
        xs = xin.shape[-1]
    if seed_dev is not None:
        assert seed_dev.is_cuda and seed_dev.dtype == torch.int64 and seed_dev.numel() >= 1
    d = nv.TailDesc(nv.dt_code(raw.dtype), N, vox, K, rs, num_classes, cx, mode, xs, seed,
                    seed_dev.data_ptr() if seed_dev is not None else None)
    if residual is not None:
        _, rn = _norm_value(res_norm, N, channels)
        r = nv.TailResidual(res.data_ptr(), res.shape[-1], rn, _addr(ra_src), ra_src.shape[-1] if ra_src is not None else 0,
                            ra_off, channels)
        nv.check(nv.lib().dua_final_conv_sampler_res(C.byref(d), nv.ptr(raw), _norm_ref(norm, N, channels), C.byref(r), nv.ptr(wf),
                                                     nv.ptr(bf), nv.ptr(coef), nv.ptr(x_state), nv.ptr(noise),
                                                     nv.ptr(step_word), nv.ptr(xin), nv.ptr(xstart_sum), nv.ptr(logits),
                                                     nv.ptr(xstart), nv.stream_ptr()), "dua_final_conv_sampler_res")
        return
    nv.check(nv.lib().dua_final_conv_sampler(C.byref(d), nv.ptr(raw), _norm_ref(norm, N, K), nv.ptr(wf),
                                             nv.ptr(bf), nv.ptr(coef), nv.ptr(x_state), nv.ptr(noise),
                                             nv.ptr(step_word), nv.ptr(xin), nv.ptr(xstart_sum), nv.ptr(logits),
                                             nv.ptr(xstart), nv.stream_ptr()), "dua_final_conv_sampler")


def temb_table(timesteps, freqs, w0, b0, w1, b1, w_cat, b_cat, out=None):
    """timesteps: int32 device [T]; returns fp32 [T, P]."""
    assert timesteps.is_cuda and timesteps.dtype == torch.int32 and timesteps.is_contiguous()
    for t in (freqs, w0, b0, w1, b1, w_cat, b_cat):
        _f32c(t, "temb param")
    half, hid, P = freqs.numel(), w1.shape[0], w_cat.shape[0]
    assert tuple(w0.shape) == (hid, 2 * half) and tuple(w1.shape) == (hid, hid) and w_cat.shape[1] == hid
    assert b0.numel() == hid and b1.numel() == hid and b_cat.numel() == P
    T = timesteps.numel()
    out = torch.empty((T, P), dtype=torch.float32, device=w0.device) if out is None else out
    nv.check(nv.lib().dua_temb_table(T, nv.ptr(timesteps), nv.ptr(freqs), half, hid, nv.ptr(w0), nv.ptr(b0), nv.ptr(w1),
                                     nv.ptr(b1), nv.ptr(w_cat), nv.ptr(b_cat), P, nv.ptr(out), nv.stream_ptr()),
             "dua_temb_table")
    return out


def step_begin(N, table, cur_add, rows_per_sample=None, row_of_step=None, counter=None, coef_table=None,
               cur_coef=None, step_word=None, err_word=None, clear=None):
    """``err_word`` (int32[1] device, optional): set to 1 by the kernel when a timestep row / step counter read from
    device memory is outside the tables (the offender is clamped, nothing faults).  ``clear``: a contiguous device tensor
    (the statistics arena of the evaluation that starts here) zeroed by the same launch."""
    T, P = table.shape
    assert table.is_cuda and table.dtype == torch.float32 and table.is_contiguous()
    assert cur_add.numel() >= N * P
    nsteps = 0
    if rows_per_sample is not None:
        assert rows_per_sample.dtype == torch.int32 and rows_per_sample.numel() == N
    else:
        assert row_of_step is not None and row_of_step.dtype == torch.int32 and counter is not None
        nsteps = row_of_step.numel()
        if coef_table is not None:
            assert coef_table.numel() >= 8 * nsteps
    cbytes = 0
    if clear is not None:
        assert clear.is_cuda and clear.is_contiguous()
        cbytes = clear.numel() * clear.element_size()
        assert cbytes % 16 == 0 and clear.data_ptr() % 16 == 0
    nv.check(nv.lib().dua_step_begin_clear(N, P, nv.ptr(table), T, nv.ptr(rows_per_sample), nv.ptr(row_of_step), nsteps,
                                           nv.ptr(coef_table), nv.ptr(counter), nv.ptr(cur_add), nv.ptr(cur_coef),
                                           nv.ptr(step_word), nv.ptr(err_word), nv.ptr(clear), cbytes, nv.stream_ptr()),
             "dua_step_begin_clear")


def denoiser_step(plan_struct):
    """dua_denoiser_step: the whole evaluation + tail described by an nv.DenoiserPlan, on the current stream."""
    nv.check(nv.lib().dua_denoiser_step(C.byref(plan_struct), nv.stream_ptr()), "dua_denoiser_step")


def window_attention(qkv, heads, bias_t, mask_t=None, windows_per_image=1, region_ids=None, out=None, bias_table=None,
                     table_grid=(7, 7, 7)):
    """Softmax attention inside windows (models/swin_unetr/attention.py:97-120 between the qkv and proj Linear layers).
    qkv: [windows, tokens, 3 * heads * 16] (fp16 or fp32, contiguous); bias_t: fp32 [heads, tokens, tokens] = bias[h].T;
    the shifted-window mask either as mask_t: fp32 [windows_per_image, tokens, tokens] = mask[w].T, or as region_ids:
    uint8 [windows_per_image, tokens] (what compute_mask derives it from).  ``bias_table`` (instead of bias_t): the
    reference's relative_position_bias_table transposed, fp32 [heads, (2gd-1)(2gh-1)(2gw-1)] for ``table_grid`` -- the
    kernel evaluates relative_position_index itself.  Returns [windows, tokens, heads * 16]."""
    assert qkv.is_cuda and qkv.is_contiguous() and qkv.dim() == 3 and qkv.dtype in (torch.float16, torch.float32)
    Wn, n, c3 = qkv.shape
    assert c3 == 3 * heads * 16 and n <= 352, "head dimension 16, at most 352 tokens per window"
    if bias_table is not None:
        _f32c(bias_table, "bias_table")
        gd, gh, gw = table_grid
        assert bias_t is None and tuple(bias_table.shape) == (heads, (2 * gd - 1) * (2 * gh - 1) * (2 * gw - 1)) and n <= gd * gh * gw
    else:
        _f32c(bias_t, "bias_t")
        assert tuple(bias_t.shape) == (heads, n, n)
    assert mask_t is None or region_ids is None
    if mask_t is not None:
        _f32c(mask_t, "mask_t")
        assert tuple(mask_t.shape) == (windows_per_image, n, n) and Wn % windows_per_image == 0
    if region_ids is not None:
        assert region_ids.is_cuda and region_ids.dtype == torch.uint8 and region_ids.is_contiguous()
        assert tuple(region_ids.shape) == (windows_per_image, n) and Wn % windows_per_image == 0
    if out is None:
        out = torch.empty((Wn, n, heads * 16), dtype=qkv.dtype, device=qkv.device)
    assert out.is_contiguous() and out.dtype == qkv.dtype and out.numel() == Wn * n * heads * 16
    nv.check(nv.lib().dua_window_attention_fwd(nv.dt_code(qkv.dtype), Wn, n, heads, windows_per_image, nv.ptr(qkv), nv.ptr(bias_t),
                                               nv.ptr(mask_t), nv.ptr(region_ids), nv.ptr(bias_table), table_grid[0], table_grid[1],
                                               table_grid[2], 16 ** -0.5, nv.ptr(out), nv.stream_ptr()),
             "dua_window_attention_fwd")
    return out


def patch_merge_norm(x, gamma, beta, legacy=True, eps=1e-5, y=None, dtype=None, out=None):
    """PatchMerging.forward up to its reduction Linear (models/swin_unetr/patch.py:44-91): the fp32 token stream
    x [B, D, H, W, C] (+ y, the last block's MLP output in ``dtype``) -> LayerNorm_8C of the gathered 2x2x2 neighbourhoods,
    [B, ceil(D/2), ceil(H/2), ceil(W/2), 8C] in ``dtype``."""
    assert x.is_cuda and x.is_contiguous() and x.dim() == 5 and x.dtype == torch.float32
    B, D, H, W, Cc = x.shape
    dtype = dtype or (y.dtype if y is not None else torch.float32)
    if y is not None:
        assert y.is_contiguous() and y.dtype == dtype and y.numel() == x.numel()
    _f32c(gamma, "gamma"); _f32c(beta, "beta")
    assert gamma.numel() == beta.numel() == 8 * Cc
    shape = (B, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, 8 * Cc)
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=x.device)
    assert out.is_contiguous() and out.dtype == dtype and out.numel() == B * shape[1] * shape[2] * shape[3] * 8 * Cc
    nv.check(nv.lib().dua_patch_merge_norm(nv.dt_code(dtype), B, D, H, W, Cc, 1 if legacy else 0, nv.ptr(x), nv.ptr(y),
                                           nv.ptr(gamma), nv.ptr(beta), eps, nv.ptr(out), nv.stream_ptr()), "dua_patch_merge_norm")
    return out


def residual_norm_act(raw, norm, res, res_norm=None, slope=0.01, out=None, out_off=0, post_add=None, post_off=0,
                      ra_src=None, ra_off=0, res_off=0, background=False):
    """out = LeakyReLU(IN(raw) + residual) [+ post_add] [+ reverse_attention(ra_src)] (UnetResBlock tail,
    models/swin_unetr/blocks.py:308-316, and the adds of swin_unetr/denoiser.py:370-399); ``res_norm`` normalises the
    residual (conv3 + norm3 of channel-changing blocks)."""
    _cl_check(raw, "raw"); _cl_check(res, "res")
    N, D, H, W, Cc = raw.shape
    assert tuple(res.shape[:4]) == (N, D, H, W) and res.shape[-1] >= res_off + Cc and res.dtype == raw.dtype and res_off % 8 == 0
    if out is None:
        out = torch.empty_like(raw)
    _cl_check(out, "out")
    assert tuple(out.shape[:4]) == (N, D, H, W) and out_off % 8 == 0 and out_off + Cc <= out.shape[-1]
    for t_, o_ in ((post_add, post_off), (ra_src, ra_off)):
        if t_ is not None:
            _cl_check(t_, "add")
            assert tuple(t_.shape[:4]) == (N, D, H, W) and t_.dtype == raw.dtype and o_ % 8 == 0 and o_ + Cc <= t_.shape[-1]
    res_ptr = C.c_void_p(res.data_ptr() + res_off * res.element_size())
    nv.check(nv.lib().dua_residual_norm_act(nv.dt_code(raw.dtype), N, D * H * W, Cc, nv.ptr(raw), raw.shape[-1], norm.ref(N, Cc),
                                            res_ptr, res.shape[-1], _norm_ref(res_norm, N, Cc), nv.ptr(out), out.shape[-1],
                                            out_off, slope, nv.ptr(post_add), post_add.shape[-1] if post_add is not None else 0,
                                            post_off, nv.ptr(ra_src), ra_src.shape[-1] if ra_src is not None else 0, ra_off,
                                            1 if background else 0, nv.stream_ptr()), "dua_residual_norm_act")
    return out


def window_geom(B, dims, C_, window, shift):
    """dua_window_geom for a [B, *dims, C] token stream; ``window`` / ``shift`` already clipped (get_window_size)."""
    return nv.WindowGeom(B, dims[0], dims[1], dims[2], C_, window[0], window[1], window[2], shift[0], shift[1], shift[2])


def window_gather_norm(x, geom, gamma, beta, out, y=None, eps=1e-5):
    """[x += y] -> norm1 -> pad -> roll(-shift) -> window_partition (transformer.py:378-417).  x: fp32 [B, D, H, W, C]
    (updated in place when y is given); out: [B * windows, tokens, C] in the compute dtype."""
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and out.is_contiguous()
    assert y is None or (y.is_contiguous() and y.dtype == out.dtype and y.numel() == x.numel())
    _f32c(gamma, "gamma"); _f32c(beta, "beta")
    nv.check(nv.lib().dua_window_gather_norm(nv.dt_code(out.dtype), C.byref(geom), nv.ptr(x), nv.ptr(y), nv.ptr(gamma), nv.ptr(beta),
                                             eps, nv.ptr(out), nv.stream_ptr()), "dua_window_gather_norm")
    return out


def window_scatter_add_norm(x, geom, yw, gamma, beta, out, eps=1e-5):
    """x += crop(roll(+shift)(window_reverse(yw))); out = norm2(x) (transformer.py:417-434, 475-476)."""
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and yw.is_contiguous() and out.is_contiguous()
    assert yw.dtype == out.dtype and out.numel() == x.numel()
    _f32c(gamma, "gamma"); _f32c(beta, "beta")
    nv.check(nv.lib().dua_window_scatter_add_norm(nv.dt_code(out.dtype), C.byref(geom), nv.ptr(x), nv.ptr(yw), nv.ptr(gamma),
                                                  nv.ptr(beta), eps, nv.ptr(out), nv.stream_ptr()), "dua_window_scatter_add_norm")
    return out


def stage_out(y, B, Cc, out, out_off=0, tadd=None, emb=None, x=None, eps=1e-5):
    """x = y + tadd[b]; out[..., out_off:out_off + C] = layer_norm(x) (no affine) + emb (transformer.py:277-312,
    swin_unetr/denoiser.py:367-368).  y: [B * tokens, C]; tadd: fp32 [B, >= C] (a column slice of a wider table is fine)."""
    assert y.is_cuda and y.is_contiguous() and y.numel() % (B * Cc) == 0 and out.dtype == y.dtype and out.is_contiguous()
    per = y.numel() // (B * Cc)
    ts = 0
    if tadd is not None:
        assert tadd.dtype == torch.float32 and tadd.dim() == 2 and tadd.shape[0] == B and tadd.stride(1) == 1
        ts = tadd.stride(0)
    if emb is not None:
        assert emb.is_contiguous() and emb.dtype == y.dtype and emb.numel() == y.numel()
    if x is not None:
        assert x.is_contiguous() and x.dtype == torch.float32 and x.numel() == y.numel()
    assert out.numel() == B * per * out.shape[-1]
    nv.check(nv.lib().dua_stage_out(nv.dt_code(y.dtype), B, per, Cc, nv.ptr(y), nv.ptr(tadd), ts, eps, nv.ptr(emb), nv.ptr(x),
                                    nv.ptr(out), out.shape[-1], out_off, nv.stream_ptr()), "dua_stage_out")
    return out


def pack_patch_embed_weights(w, cin_packed, perm=None):
    """Conv3d(k2, s2) weight fp32 [E, Cin, 2, 2, 2] -> fp32 [8 taps (kd, kh, kw), cin_packed, E]; packed channel p reads
    source channel perm[p] (default: p), channels beyond the source are zero."""
    E, Cin = w.shape[:2]
    perm = list(range(Cin)) if perm is None else list(perm)
    wp = torch.zeros(8, cin_packed, E, dtype=torch.float32, device=w.device)
    wp[:, :len(perm)] = w.detach().float()[:, perm].permute(2, 3, 4, 1, 0).reshape(8, len(perm), E)
    return wp.contiguous()


def patch_embed(xin, cin_packed, w_packed, bias, out, out_off=0, tadd=None, emb=None, x=None, eps=1e-5):
    """PatchEmbed conv + bias (+ tadd[b]) -> x (fp32 stream), layer_norm(x) (+ emb) -> out slice (transformer.py:271-275)."""
    _cl_check(xin, "xin")
    B, D, H, W, Cs = xin.shape
    E = w_packed.shape[-1]
    _f32c(w_packed, "w_packed"); _f32c(bias, "bias")
    ts = 0
    if tadd is not None:
        assert tadd.dtype == torch.float32 and tadd.dim() == 2 and tadd.shape[0] == B and tadd.stride(1) == 1
        ts = tadd.stride(0)
    nv.check(nv.lib().dua_patch_embed(nv.dt_code(xin.dtype), B, D, H, W, Cs, cin_packed, E, nv.ptr(xin), nv.ptr(w_packed),
                                      nv.ptr(bias), nv.ptr(tadd), ts, eps, nv.ptr(emb), nv.ptr(x), nv.ptr(out), out.shape[-1],
                                      out_off, nv.stream_ptr()), "dua_patch_embed")
    return out


def instnorm_stats(x, Cc, stats, c_off=0):
    """Accumulate per-(n, c) sum / sum of squares of a channels-last slice into a (zeroed) statistics buffer."""
    _cl_check(x, "x")
    N, D, H, W, Cs = x.shape
    nv.check(nv.lib().dua_instnorm_stats(nv.dt_code(x.dtype), N, D * H * W, Cc, nv.ptr(x), Cs, c_off, nv.ptr(stats),
                                         stats.shape[3], nv.stream_ptr()), "dua_instnorm_stats")
    return stats


def gelu_(x):
    """Exact GELU in place (MONAI MLPBlock act "GELU")."""
    assert x.is_cuda and x.is_contiguous() and x.numel() % 8 == 0
    nv.check(nv.lib().dua_gelu(nv.dt_code(x.dtype), x.numel(), nv.ptr(x), nv.stream_ptr()), "dua_gelu")
    return x


def token_linear(A, W, bias=None, mode="plain", out=None, out_off=0, x=None, stats=None, samples=1, geom=None, gamma=None,
                 beta=None, ln_out=None, eps=1e-5, background=False):
    """dua_token_linear: fp16 A [tokens, K] (row stride A.stride(0)) times the nn.Linear weight W [N, K] with one fused
    epilogue -- "plain" / "gelu" (-> out[:, out_off:out_off+N]), "stats" (raw out + per-(sample, channel) sums into
    ``stats``), "residual" (x += result, fp32 stream) or "scatter" (window order -> voxel order, x += result,
    ln_out = LayerNorm(x) * gamma + beta).  Layers wider than 192 outputs are split by rows of W."""
    assert A.is_cuda and A.dtype == torch.float16 and A.dim() == 2 and A.stride(1) == 1
    assert W.is_cuda and W.dtype == torch.float16 and W.is_contiguous() and W.dim() == 2 and W.shape[1] == A.shape[1]
    M, K = A.shape
    N = W.shape[0]
    code = {"plain": nv.TOKLIN_PLAIN, "gelu": nv.TOKLIN_GELU, "stats": nv.TOKLIN_STATS, "residual": nv.TOKLIN_RESIDUAL,
            "scatter": nv.TOKLIN_SCATTER}[mode]
    if bias is not None:
        _f32c(bias, "bias")
        assert bias.numel() == N
    if N > 192:
        assert mode in ("plain", "gelu")
        for n0 in range(0, N, 192):
            n1 = min(N, n0 + 192)
            token_linear(A, W[n0:n1], None if bias is None else bias[n0:n1], mode, out, out_off + n0)
        return out
    d = nv.TokenLinearDesc()
    d.A, d.lda, d.M, d.K, d.N, d.W, d.bias = A.data_ptr(), A.stride(0), M // samples, K, N, W.data_ptr(), (bias.data_ptr() if bias is not None else None)
    d.mode, d.samples = code, samples
    d.background = int(background)
    if mode in ("plain", "gelu", "stats"):
        assert out is not None and out.is_cuda and out.dtype == torch.float16 and out.is_contiguous()
        ldc = out.shape[-1]
        assert out.numel() == M * ldc and out_off + N <= ldc
        d.out, d.ldc, d.out_off = out.data_ptr(), ldc, out_off
    if mode == "stats":
        assert stats is not None and stats.dtype == torch.int64 and stats.is_contiguous() and stats.shape[0] == samples and M % samples == 0
        d.stats, d.c_pad = stats.data_ptr(), stats.shape[3]
    if mode in ("residual", "scatter"):
        assert x is not None and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        d.x = x.data_ptr()
    if mode == "residual":
        assert x.numel() == M * N
    if mode == "scatter":
        _f32c(gamma, "gamma"); _f32c(beta, "beta")
        assert ln_out is not None and ln_out.dtype == torch.float16 and ln_out.is_contiguous() and ln_out.numel() == x.numel()
        d.geom, d.gamma, d.beta, d.eps, d.ln_out = geom, gamma.data_ptr(), beta.data_ptr(), eps, ln_out.data_ptr()
    nv.check(nv.lib().dua_token_linear(C.byref(d), nv.stream_ptr()), "dua_token_linear")
    return out if mode in ("plain", "gelu", "stats") else x


_GEMM_WS = {}


def _gemm_ws(nbytes, device):
    """Grow-only fp32 scratch per (device, stream) for the K-split token GEMMs (retired, never freed: graphs bake its address)."""
    key = (torch.device(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _GEMM_WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            _WGRAD_WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 16 << 20) // 4, dtype=torch.float32, device=device)
        _GEMM_WS[key] = buf
    return buf


def token_gemm(A, W, bias=None, mode="plain", out=None, out_off=0, x=None, workspace=None):
    """dua_token_gemm: the tiled MFMA GEMM of the coarse Swin stages -- fp16 A [tokens, K] (row stride A.stride(0)) times the
    nn.Linear weight W [N, K] (any K, N that are multiples of 8), "plain" / "gelu" -> out[:, out_off:out_off+N] (fp16), or
    "residual": x += result on the fp32 stream.  ``workspace``: a callable ``nbytes -> fp32 tensor`` that owns the K-split
    scratch (a plan's own buffer: a captured graph bakes the address in, so two plans must not share one); default: the
    per-(device, stream) scratch of this module."""
    assert A.is_cuda and A.dtype == torch.float16 and A.dim() == 2 and A.stride(1) == 1
    assert W.is_cuda and W.dtype == torch.float16 and W.is_contiguous() and W.dim() == 2 and W.shape[1] == A.shape[1]
    M, K = A.shape
    N = W.shape[0]
    assert K % 8 == 0 and N % 8 == 0 and A.stride(0) % 8 == 0
    code = {"plain": nv.TOKLIN_PLAIN, "gelu": nv.TOKLIN_GELU, "residual": nv.TOKLIN_RESIDUAL}[mode]
    d = nv.TokenLinearDesc()
    d.A, d.lda, d.M, d.K, d.N, d.W = A.data_ptr(), A.stride(0), M, K, N, W.data_ptr()
    if bias is not None:
        _f32c(bias, "bias")
        assert bias.numel() == N
        d.bias = bias.data_ptr()
    d.mode, d.samples = code, 1
    if mode == "residual":
        assert x is not None and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == M * N
        d.x = x.data_ptr()
    else:
        assert out is not None and out.is_cuda and out.dtype == torch.float16 and out.is_contiguous()
        ldc = out.shape[-1]
        assert out.numel() == M * ldc and out_off + N <= ldc and ldc % 8 == 0 and out_off % 8 == 0
        d.out, d.ldc, d.out_off = out.data_ptr(), ldc, out_off
    need = int(nv.lib().dua_token_gemm_workspace(M, K, N))
    ws = (workspace(need) if workspace is not None else _gemm_ws(need, A.device)) if need > 0 else None
    assert ws is None or (ws.is_cuda and ws.dtype == torch.float32 and ws.numel() * 4 >= need)
    nv.check(nv.lib().dua_token_gemm(C.byref(d), nv.ptr(ws), ws.numel() * 4 if ws is not None else 0, nv.stream_ptr()),
             "dua_token_gemm")
    return x if mode == "residual" else out


def swin_mlp(ln2, w1, b1, w2, b2, x):
    """x += linear2(GELU(linear1(ln2))) in one launch (dua_swin_mlp): ln2 fp16 [tokens, C], w1 fp16 [4C, C], w2 fp16 [C, 4C],
    biases fp32, x the fp32 stream [tokens, C] (updated in place).  C = 48 or 96."""
    assert ln2.is_cuda and ln2.dtype == torch.float16 and ln2.is_contiguous() and ln2.dim() == 2
    M, Cc = ln2.shape
    assert Cc in (48, 96) and tuple(w1.shape) == (4 * Cc, Cc) and tuple(w2.shape) == (Cc, 4 * Cc)
    assert w1.dtype == w2.dtype == torch.float16 and w1.is_contiguous() and w2.is_contiguous()
    _f32c(b1, "b1"); _f32c(b2, "b2")
    assert b1.numel() == 4 * Cc and b2.numel() == Cc
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == M * Cc
    nv.check(nv.lib().dua_swin_mlp(M, Cc, nv.ptr(ln2), nv.ptr(w1), nv.ptr(b1), nv.ptr(w2), nv.ptr(b2), nv.ptr(x), nv.stream_ptr()),
             "dua_swin_mlp")
    return x
