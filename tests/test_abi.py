"""The C-ABI boundary (no GPU needed): libdua_hip.so builds for gfx950, loads next to PyTorch's HIP
runtime, and exports exactly the entry points include/dua_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dua_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dua_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from diff_unet_amos_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "diff_unet_amos_amd", "csrc"), "-j4"], check=True)
    return _native.lib()


def test_header_and_binding_agree(lib):
    from diff_unet_amos_amd import _native
    assert _declared() == _native.exported_symbols()


def test_every_declared_symbol_is_exported(lib):
    for name in _declared():
        assert getattr(lib, name) is not None, name


def test_library_carries_gfx950_code_and_binds_to_torch_hip_runtime(lib):
    from diff_unet_amos_amd import _native
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    maps = open("/proc/self/maps").read()
    hips = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hips) == 1, f"two HIP runtimes mapped: {hips}"      # one runtime => shared device context/streams
    assert any("libdua_hip.so" in line for line in maps.splitlines())


def test_argument_validation_without_a_device(lib):
    """Entry points reject bad descriptors before touching the GPU."""
    import ctypes as C
    from diff_unet_amos_amd import _native as nv
    d = nv.Conv3Desc(nv.F16, 1, 8, 8, 8, 12, 16, 0, 64, 64, 0)        # Cin not a multiple of 8
    one = C.c_void_p(16)
    assert lib.dua_conv3d_k3_fwd(C.byref(d), one, one, one, None, one, one, None, 0, None) == nv.ERR_ARG
    assert lib.dua_conv3d_k3_fwd(None, one, one, one, None, one, one, None, 0, None) == nv.ERR_ARG
    assert lib.dua_q_sample(0, 10, one, one, one, one, None) == nv.ERR_ARG
    assert lib.dua_sampler_step(7, 1, 10, one, one, one, one, one, None, None, None) == nv.ERR_ARG
    bad_policy = nv.Conv3Desc(nv.F16, 1, 8, 8, 8, 16, 16, 0, 64, 64, 0, 0, 0, 0, 5)        # not a launch form
    assert lib.dua_conv3d_k3_fwd(C.byref(bad_policy), one, one, one, None, one, one, None, 0, None) == nv.ERR_ARG
    # LeakyReLU slope outside [0, 1]: the fp16 kernels apply the activation as max(t, slope * t)
    ok = nv.Conv3Desc(nv.F16, 1, 8, 8, 8, 16, 16, 0, 64, 64, 0)
    bad = nv.InNorm(16, 16, 16, None, 0, 64, 512, 1e-5, 1.5)
    assert lib.dua_conv3d_k3_fwd(C.byref(ok), one, one, one, C.byref(bad), one, one, None, 0, None) == nv.ERR_ARG
    assert lib.dua_deconv_k2s2_fwd(C.byref(ok), one, one, one, C.byref(bad), one, None) == nv.ERR_ARG
    no_count = nv.InNorm(16, 16, 16, None, 0, 64, 0, 1e-5, 0.1)       # the integer voxel count (ABI 8) must be positive
    assert lib.dua_conv3d_k3_fwd(C.byref(ok), one, one, one, C.byref(no_count), one, one, None, 0, None) == nv.ERR_ARG
    assert lib.dua_pack_conv3_weights(nv.F16, 64, 17, 24, None, None, None, None) == 1 * 1 * 27 * 4 * 64 * 16
    # the training-step kernels (csrc/train_glue.hip)
    assert lib.dua_stats_channel_sums(1, 80, 64, one, one, None) == nv.ERR_ARG                  # more channels than the rows hold
    assert lib.dua_seg_loss_finish(1, 4, 10, 0, 0, 0, 0, one, one, one, None) == nv.ERR_ARG     # no loss term selected
    assert lib.dua_q_sample_affine(1, 10, one, 2.0, -1.0, one, one, 0, one, one, None) == nv.ERR_ARG
    blk = nv.TembBlocks()
    blk.nblocks = 17
    assert lib.dua_temb_train_fwd(2, one, one, 64, 512, one, one, one, one, C.byref(blk), one, one, None) == nv.ERR_ARG
    blk.nblocks = 1; blk.cout[0] = 64; blk.w[0] = 16; blk.b[0] = 16
    assert lib.dua_temb_train_fwd(2, one, one, 64, 384, one, one, one, one, C.byref(blk), one, one, None) == nv.ERR_ARG   # hidden
    lst = nv.AdamWList()
    lst.count = 65
    assert lib.dua_grads_nonfinite(C.byref(lst), one, None) == nv.ERR_ARG
    lst.count = 1; lst.numel[0] = 8; lst.g[0] = 16
    assert lib.dua_adamw_step(C.byref(lst), 1e-3, None, 0.9, 0.999, 1e-8, 0.0, None, None, one, 0, None) == nv.ERR_ARG   # no p / m / v
    assert lib.dua_adamw_advance(None, None, None, None, 2.0, 0.5, 200, None, None) == nv.ERR_ARG


def test_scratch_sizes_cover_whole_tiles(lib):
    """Host-side sizing functions, no device needed: the split-K scratch is ksplit x the padded output when the launcher
    splits."""
    import ctypes as C
    from diff_unet_amos_amd import _native as nv
    d = nv.Conv3Desc(nv.F16, 1, 6, 6, 6, 512, 512, 0, 512, 512, 0)            # the 6^3 level: split
    ws = lib.dua_conv3d_k3_workspace(C.byref(d))
    assert ws > 0 and ws % (6 * 6 * 6 * 512 * 4) == 0
    d = nv.Conv3Desc(nv.F16, 1, 96, 96, 96, 64, 64, 0, 64, 64, 0)             # the 96^3 level: never split
    assert lib.dua_conv3d_k3_workspace(C.byref(d)) == 0


def test_abi_version_and_prepare_registry(lib):
    """The binding refuses a library built from another header (dua_abi_version), and every kernel that needs more than the
    default dynamic-LDS limit is registered for dua_prepare() at load time -- no device needed to count them."""
    from diff_unet_amos_amd import _native as nv
    src = open(HEADER).read()
    assert int(re.search(r"#define DUA_ABI_VERSION (\d+)", src).group(1)) == nv.ABI_VERSION == lib.dua_abi_version()
    # conv (12 + the wide-tile form + the folded up-convolution) + weight gradient (6 + the fetch-once form) + transposed conv (8) + its backward (4) + Swin token kernels (3 + 30)
    assert lib.dua_prepared_kernels() == 73


def test_no_launcher_sets_function_attributes_on_its_own():
    """hipFuncSetAttribute lives in prepare.hip only: a launcher that raised its own limit lazily could do so for the first
    time inside a stream capture, or from autograd's worker thread while the main thread does the same."""
    csrc = os.path.join(ROOT, "diff_unet_amos_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")) and name != "prepare.hip":
            text = re.sub(r"//.*", "", open(os.path.join(csrc, name)).read())
            assert "hipFuncSetAttribute" not in text, name
            assert "PerDeviceOnce" not in text, name


def test_the_shipped_library_keeps_no_option_state(lib):
    """Launch forms are a per-call field (dua_conv3_desc.policy); the shipped library exports no dua_set_option and the
    environment cannot redirect or reconfigure the package without DUA_DEBUG=1."""
    assert not hasattr(lib, "dua_set_option")
    src = open(os.path.join(ROOT, "diff_unet_amos_amd", "_native.py")).read()
    assert "DUA_CONV_VARIANT" not in src
    csrc = os.path.join(ROOT, "diff_unet_amos_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            text = re.sub(r"//.*", "", open(os.path.join(csrc, name)).read())
            globals_ = re.findall(r"^(?:static )?int (g_\w+)", text, flags=re.M)          # namespace-scope mutable ints
            assert [g for g in globals_ if g != "g_wgrad_abl"] == [], (name, globals_)       # g_wgrad_abl: -DDUA_ABLATE builds only
