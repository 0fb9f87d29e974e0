"""The kernel-level binding INTEGRATION.md hands a maintainer of the reference (include/_dua.py) is a real file: its struct
layouts are compared with the package's own binding on CPU, its text with the block INTEGRATION.md shows, and on a GPU it
runs the two convolutions of a TwoConv block (models/basic_unet/denoiser.py:56-67) against torch on CPU."""
import ctypes as C
import importlib.util
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "include", "_dua.py")


def _load_stub():
    from diff_unet_amos_amd import _native as nv
    nv.lib()                                             # builds nothing: fails loudly when the library is missing
    os.environ["DUA_HIP_SO"] = nv.LIB_PATH
    spec = importlib.util.spec_from_file_location("_dua_stub", STUB)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stub_structs_match_the_header_and_the_package_binding():
    from diff_unet_amos_amd import _native as nv
    stub = _load_stub()
    assert stub.DUA_ABI_VERSION == nv.ABI_VERSION
    for mine, theirs in ((stub.Conv3Desc, nv.Conv3Desc), (stub.InNorm, nv.InNorm)):
        assert C.sizeof(mine) == C.sizeof(theirs)
        assert [(n, t) for n, t in mine._fields_] == [(n, t) for n, t in theirs._fields_]
    # and with the header text itself: field names of dua_conv3_desc / dua_in_norm in declaration order
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dua_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct \{([^}]*)\} dua_conv3_desc;", src).group(1)
    names = [n.strip() for decl in re.findall(r"int ([^;]+);", body) for n in decl.split(",")]
    assert names == [n for n, _ in stub.Conv3Desc._fields_]
    body = re.search(r"typedef struct \{([^}]*)\} dua_in_norm;", src).group(1)
    names = re.findall(r"(\w+);", body)
    assert names == [n for n, _ in stub.InNorm._fields_]


def test_integration_md_shows_the_stub_as_it_is():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert open(STUB).read() in blocks, "INTEGRATION.md section 2 must show include/_dua.py verbatim"


@pytest.mark.gpu
def test_stub_runs_a_twoconv_block_against_torch():
    stub = _load_stub()
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(0)
    N, S, cin, cmid, cout = 1, 16, 16, 64, 64
    c0, n0, c1 = nn.Conv3d(cin, cmid, 3, padding=1), nn.InstanceNorm3d(cmid, affine=True), nn.Conv3d(cmid, cout, 3, padding=1)
    with torch.no_grad():
        n0.weight.uniform_(0.5, 1.5); n0.bias.normal_()
    x = torch.randn(N, cin, S, S, S)
    with torch.no_grad():
        raw0 = c0(x)
        want = c1(F.leaky_relu(n0(raw0), 0.1))
    dev = torch.device("cuda:0")
    for m in (c0, n0, c1):
        m.to(dev)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous().to(dev).half()
    w0, b0 = stub.pack(c0)
    w1, b1 = stub.pack(c1)
    y0 = torch.empty(N, S, S, S, cmid, dtype=torch.float16, device=dev)
    y1 = torch.empty(N, S, S, S, cout, dtype=torch.float16, device=dev)
    s0, s1 = stub.new_stats(N, cmid, dev), stub.new_stats(N, cout, dev)
    stub.conv3(x_cl, w0, b0, y0, s0)
    stub.conv3(y0, w1, b1, y1, s1, producer=stub.producer(s0, n0, S ** 3))
    got0 = y0.float().cpu().permute(0, 4, 1, 2, 3)
    got1 = y1.float().cpu().permute(0, 4, 1, 2, 3)
    assert float((got0 - raw0).abs().max()) < 2e-2 * float(raw0.abs().max())
    assert float((got1 - want).abs().max()) < 2e-2 * float(want.abs().max())
    # the statistics words the second launch left: sum x of its raw output, decoded as include/dua_hip.h documents
    w = s1.sum(1).cpu()
    sums = w[:, 0].double() + w[:, 1].double() / 2.0 ** 44
    ref = y1.double().sum((1, 2, 3)).cpu()
    assert torch.allclose(sums[:, :cout], ref, rtol=1e-3, atol=1e-1)
