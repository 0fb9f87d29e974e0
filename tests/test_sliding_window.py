"""Sliding-window scheduler / blend and its sharded (multi-process, gloo on CPU) form."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diff_unet_amos_amd.inference import (_plan, binarise, dice_per_class, sharded_sliding_window_inference,
                                          sliding_window_inference)
from oracle.sliding_window_ref import sliding_window_ref
from oracle.unet_ref import dice_coeff


def _predictor(x, pred_type=None):
    """Deterministic, position-independent function of the window contents with 3 output channels."""
    assert pred_type == "ddim_sample"
    return torch.cat([x * 2.0, x.flip(-1) + 1.0, torch.tanh(x) * x.mean(dim=(2, 3, 4), keepdim=True)], dim=1)


def _predictor_np(x):
    return _predictor(torch.from_numpy(x), pred_type="ddim_sample").numpy()


def test_window_count_of_config3():
    """256x256x192, roi 96^3, overlap 0.25 -> interval 72 -> 4 x 4 x 3 = 48 windows."""
    _, _, _, _, starts = _plan(torch.zeros(1, 1, 256, 256, 192), (96, 96, 96), 0.25)
    assert len(starts) == 48
    assert sorted({s[0] for s in starts}) == [0, 72, 144, 160] and sorted({s[2] for s in starts}) == [0, 72, 96]
    assert starts[0] == (0, 0, 0) and starts[1] == (0, 0, 72)         # last axis fastest
    with pytest.raises(ValueError):
        _plan(torch.zeros(1, 1, 8, 8, 8), (4, 4, 4), 1.0)


DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]      # the scheduler also runs with device tensors under -m gpu


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("shape,roi,overlap,swb", [
    ((1, 1, 20, 17, 13), (8, 8, 8), 0.25, 4),
    ((2, 1, 9, 16, 16), (8, 8, 8), 0.5, 3),       # batch of 2 volumes
    ((1, 1, 5, 8, 11), (8, 8, 8), 0.8, 1),        # smaller than the roi along D: symmetric zero padding, then crop
    ((1, 1, 8, 8, 8), (8, 8, 8), 0.25, 2),        # exactly one window
])
def test_matches_loop_restatement(shape, roi, overlap, swb, device):
    g = torch.Generator().manual_seed(sum(shape))
    vol = torch.randn(*shape, generator=g)
    got = sliding_window_inference(vol.to(device), roi, swb, _predictor, overlap, pred_type="ddim_sample")
    want = sliding_window_ref(vol.numpy(), roi, overlap, _predictor_np)
    assert got.shape == (shape[0], 3, *shape[2:]) and got.device.type == device
    assert np.allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


def _worker(rank, world, port, shape, roi, overlap, swb, gather_dtype, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(sum(shape))
        vol = torch.randn(*shape, generator=g)
        out = sharded_sliding_window_inference(vol, roi, swb, _predictor, overlap, gather_dtype=gather_dtype,
                                               pred_type="ddim_sample")
        q.put((rank, out.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,overlap", [(2, (1, 1, 20, 17, 13), 0.25), (3, (2, 1, 9, 16, 16), 0.5), (2, (1, 1, 8, 8, 8), 0.25)])
def test_sharded_equals_single_process(world, shape, overlap):
    """world_size 2 and 3 over gloo: every rank ends with the single-process result (odd window counts,
    a rank with no window at all)."""
    roi, swb = (8, 8, 8), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world + int(overlap * 100)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, roi, overlap, swb, None, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(sum(shape))
    vol = torch.randn(*shape, generator=g)
    want = sliding_window_inference(vol, roi, swb, _predictor, overlap, pred_type="ddim_sample").numpy()
    for r in range(world):
        assert np.allclose(outs[r], want, rtol=1e-6, atol=1e-6), r


@pytest.mark.parametrize("device", DEVICES)
def test_binarise_and_dice_match_reference_formulas(device):
    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(2, 4, 6, 6, 6, generator=g) * 3).to(device)
    labels = (torch.rand(2, 4, 6, 6, 6, generator=g) > 0.6).float().to(device)
    labels[:, 3] = 0
    logits[:, 3] = -10                                   # class 3 empty on both sides -> dice 0 (metric.py:45-49)
    pred = binarise(logits)
    assert torch.equal(pred, (torch.sigmoid(logits) > 0.5).float())
    got = dice_per_class(pred, labels)
    for c in range(4):
        assert abs(float(got[c]) - dice_coeff(pred[:, c].cpu(), labels[:, c].cpu())) < 1e-12
    assert float(got[3]) == 0.0
