# models/basic_unet/_dua.py  (new file in the reference; shipped here as include/_dua.py and executed by
# tests/test_integration_stub.py, so it cannot drift from include/dua_hip.h)
"""Kernel-level binding of libdua_hip.so for a maintainer of aarchiiive/diff-unet-amos: nn.Conv3d(k3, s1, p1) of MONAI's
Convolution block (models/basic_unet/denoiser.py:56-59) on the MI355X kernel, from channels-last fp16 tensors.
Needs nothing from the diff_unet_amos_amd Python package -- only the shared library."""
import ctypes as C
import os

import torch          # first: the library binds to the libamdhip64 torch has already mapped (one HIP runtime, shared streams)

DUA_ABI_VERSION = 8   # of the include/dua_hip.h this file was written against
_L = C.CDLL(os.environ.get("DUA_HIP_SO", "/path/to/diff_unet_amos_amd/libdua_hip.so"))
_L.dua_abi_version.restype = C.c_int
if _L.dua_abi_version() != DUA_ABI_VERSION:      # struct layouts differ between ABI versions: refuse, never reinterpret
    raise ImportError(f"libdua_hip.so has ABI {_L.dua_abi_version()}, this binding was written for {DUA_ABI_VERSION}")


class Conv3Desc(C.Structure):                    # dua_conv3_desc: 15 ints
    _fields_ = [(n, C.c_int) for n in ("dtype", "N", "D", "H", "W", "Cin", "Cin_stride", "Cin_off",
                                       "Cout", "Cout_stride", "Cout_off", "tap_channel_plus1", "background",
                                       "layout", "policy")]


class InNorm(C.Structure):                       # dua_in_norm: the producer's InstanceNorm + LeakyReLU, fused into the consumer
    _fields_ = [("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("add", C.c_void_p),
                ("add_stride", C.c_int), ("c_pad", C.c_int), ("count", C.c_longlong), ("eps", C.c_float),
                ("slope", C.c_float)]


_L.dua_prepare.restype = C.c_int
_L.dua_pack_conv3_weights.restype = C.c_long
_L.dua_pack_conv3_weights.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_L.dua_conv3d_k3_fwd.restype = C.c_int
_L.dua_conv3d_k3_fwd.argtypes = [C.POINTER(Conv3Desc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(InNorm),
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
DUA_F16 = 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pack(conv):
    """nn.Conv3d(k3) parameters -> (packed fp16 weights, fp32 bias padded to a multiple of 64); once per weight update."""
    w = conv.weight.detach().float().contiguous()
    cout, cin = w.shape[:2]
    assert cin % 8 == 0 and cout % 8 == 0, "channel counts must be multiples of 8 (pad the buffers)"
    nbytes = _L.dua_pack_conv3_weights(DUA_F16, cout, cin, cin, None, None, None, None)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    if _L.dua_pack_conv3_weights(DUA_F16, cout, cin, cin, w.data_ptr(), None, buf.data_ptr(), _stream()) != nbytes:
        raise RuntimeError("dua_pack_conv3_weights failed")
    bias = torch.zeros(-(-cout // 64) * 64, dtype=torch.float32, device=w.device)
    if conv.bias is not None:
        bias[:cout] = conv.bias.detach().float()
    return buf, bias


def new_stats(N, cout, device):
    """Zeroed InstanceNorm sums the convolution accumulates into: dua_stat_word [N][8][4][ceil(cout / 64) * 64]."""
    return torch.zeros((N, 8, 4, -(-cout // 64) * 64), dtype=torch.int64, device=device)


def producer(stats, norm, voxels, slope=0.1):
    """dua_in_norm of a raw convolution output: its sums + the nn.InstanceNorm3d(affine=True) that follows it."""
    return InNorm(stats.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), None, 0, stats.shape[3], int(voxels),
                  float(norm.eps), float(slope))


def conv3(x_cl, w_packed, bias_pad, y_cl, stats, producer=None):
    """x_cl / y_cl: channels-last fp16 [N, D, H, W, C] CUDA tensors; y_cl receives the RAW output (conv + bias) and ``stats``
    its per-(n, c) sums.  ``producer``: InNorm of x_cl when x_cl is itself a raw output (its InstanceNorm + LeakyReLU are
    applied while the kernel stages its input).  Replaces Convolution.conv at denoiser.py:56-59."""
    N, D, H, W, Cin = x_cl.shape
    assert x_cl.is_cuda and x_cl.dtype == torch.float16 and x_cl.is_contiguous() and y_cl.is_contiguous()
    assert tuple(y_cl.shape[:4]) == (N, D, H, W) and Cin % 8 == 0 and y_cl.shape[-1] % 8 == 0
    d = Conv3Desc(DUA_F16, N, D, H, W, Cin, Cin, 0, y_cl.shape[-1], y_cl.shape[-1], 0, 0, 0, 0, 0)
    rc = _L.dua_conv3d_k3_fwd(C.byref(d), x_cl.data_ptr(), w_packed.data_ptr(), bias_pad.data_ptr(),
                              C.byref(producer) if producer is not None else None, y_cl.data_ptr(), stats.data_ptr(),
                              None, 0, _stream())
    if rc:
        raise RuntimeError(f"dua_conv3d_k3_fwd: {rc}")
    return y_cl
