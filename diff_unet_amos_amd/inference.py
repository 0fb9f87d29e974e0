"""Caller side of the sampler on full volumes: sliding-window scheduling, blending, binarisation,
Dice -- and the multi-GPU form of it (one process per GPU, windows sharded, one all-gather).

Reference call site: Engine.infer (engine.py:167-182) =
    monai.inferers.sliding_window_inference(image, (96,96,96), sw_batch_size, model, overlap,
                                            pred_type="ddim_sample") -> sigmoid -> > 0.5
MONAI is not vendored in the reference (nor installed here), so the window schedule below restates
MONAI's documented behaviour for the arguments the reference passes (mode="constant",
padding_mode="constant", cval=0): pad up to the roi, scan interval int(roi*(1-overlap)) (>=1; == roi
when the image is exactly one roi), window starts clamped to image-roi, windows enumerated with the first
spatial axis slowest, constant importance map, sum / count.  PARITY UNPINNED for this function (no
MONAI to run against); tests pin it against an independent loop-based restatement in the oracle.

Windows are independent units (each gets its own encoder pass and its own x_T,
models/diffusion/diffusion.py:88-100), so the multi-GPU form shards them with no data-path
collective until the end: rank r runs windows r, r+W, r+2W, ...; one all-gather (RCCL over xGMI on a
GPU node) of the per-window outputs; every rank then blends identically.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, Sequence

import torch
import torch.nn.functional as F


def _scan_interval(image_size, roi_size, overlap):
    out = []
    for im, roi in zip(image_size, roi_size):
        if roi == im:
            out.append(int(roi))
        else:
            iv = int(roi * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def dense_window_starts(image_size: Sequence[int], roi_size: Sequence[int], scan_interval: Sequence[int]):
    """Per-axis window starts, then their product with the first axis slowest."""
    per_axis = []
    for im, roi, iv in zip(image_size, roi_size, scan_interval):
        if iv == 0:
            n = 1
        else:
            num = int(math.ceil(float(im) / iv))
            first = next((d for d in range(num) if d * iv + roi >= im), None)
            n = first + 1 if first is not None else 1
        per_axis.append([min(d * iv, im - roi) for d in range(n)])
    return list(itertools.product(*per_axis))


def _plan(inputs, roi_size, overlap):
    spatial = tuple(inputs.shape[2:])
    roi = tuple(int(r) for r in roi_size)
    assert len(roi) == len(spatial) == 3
    if not 0 <= overlap < 1:
        raise ValueError("overlap must be >= 0 and < 1.")
    padded = tuple(max(s, r) for s, r in zip(spatial, roi))
    pad = []
    for k in range(len(spatial) - 1, -1, -1):           # F.pad wants the last axis first
        diff = max(roi[k] - spatial[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    starts = dense_window_starts(padded, roi, _scan_interval(padded, roi, overlap))
    return spatial, roi, padded, pad, starts


def _blend(outputs_by_index, batch, channels, padded, roi, starts, pad, spatial, device, dtype):
    out = torch.zeros((batch, channels, *padded), dtype=dtype, device=device)
    cnt = torch.zeros((1, 1, *padded), dtype=dtype, device=device)
    nwin = len(starts)
    for idx, o in outputs_by_index:
        b, (d, h, w) = idx // nwin, starts[idx % nwin]
        out[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += o
        if b == 0:
            cnt[:, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += 1
    out = out / cnt
    sl = [slice(None), slice(None)]
    for k in range(3):                                   # crop the padding away again
        lo = pad[2 * (2 - k)]
        sl.append(slice(lo, lo + spatial[k]))
    return out[tuple(sl)]


def _window(inputs, idx, nwin, starts, roi):
    b, (d, h, w) = idx // nwin, starts[idx % nwin]
    return inputs[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]]


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable, overlap: float = 0.25,
                             **kwargs) -> torch.Tensor:
    """Single-process form (engine.py:173-177 semantics).  ``predictor(window_batch, **kwargs)`` -> [b,C,*roi]."""
    spatial, roi, padded, pad, starts = _plan(inputs, roi_size, overlap)
    x = F.pad(inputs, pad=pad, mode="constant", value=0.0)
    nwin, total = len(starts), len(starts) * inputs.shape[0]
    results = []
    for g in range(0, total, sw_batch_size):
        idxs = list(range(g, min(g + sw_batch_size, total)))
        seg = predictor(torch.cat([_window(x, i, nwin, starts, roi) for i in idxs]), **kwargs)
        results += [(i, seg[k:k + 1]) for k, i in enumerate(idxs)]
    first = results[0][1]
    return _blend(results, inputs.shape[0], first.shape[1], padded, roi, starts, pad, spatial, first.device, first.dtype)


def balanced_batches(n: int, max_batch: int):
    """Sizes of the predictor calls for ``n`` windows of one rank: as few calls as ``max_batch`` allows, of (nearly) equal size.
    Cutting [0, n) into slices of ``max_batch`` leaves a tail call that can be a single window (BASELINE config 3 on 8 GPUs:
    6 windows per rank at sw_batch_size 4 -> 4 + 2, and the 2-window pass runs the coarse levels of the network at half the
    fill); 3 + 3 takes the same number of passes and no call is smaller than half of ``max_batch``."""
    if n <= 0:
        return []
    calls = -(-n // max(1, int(max_batch)))
    base, extra = divmod(n, calls)
    return [base + 1] * extra + [base] * (calls - extra)


def sharded_sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable,
                                     overlap: float = 0.25, group=None, gather_dtype: torch.dtype = None,
                                     timings: dict = None, **kwargs) -> torch.Tensor:
    """One process per GPU: windows dealt round-robin over the ranks of ``group``, one all-gather of the
    per-window outputs (fp32, or ``gather_dtype`` to halve the xGMI bytes), identical blend on every rank.
    ``inputs`` must be the same on all ranks.  ``timings`` (optional dict): accumulates the wall time of the
    collective under "all_gather_s" (device synchronised around it) and records "gathered_bytes"."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spatial, roi, padded, pad, starts = _plan(inputs, roi_size, overlap)
    x = F.pad(inputs, pad=pad, mode="constant", value=0.0)
    nwin, total = len(starts), len(starts) * inputs.shape[0]
    mine = list(range(rank, total, world))
    per_rank = -(-total // world)
    local = None
    g = 0
    for nb in balanced_batches(len(mine), sw_batch_size):
        idxs = mine[g:g + nb]
        seg = predictor(torch.cat([_window(x, i, nwin, starts, roi) for i in idxs]), **kwargs)
        if local is None:
            gd = gather_dtype or seg.dtype
            local = torch.zeros((per_rank, *seg.shape[1:]), dtype=gd, device=seg.device)
        local[g:g + len(idxs)] = seg.to(local.dtype)
        g += nb
    if local is None:        # more ranks than windows: still take part in the collective
        probe = predictor(_window(x, 0, nwin, starts, roi), **kwargs)
        local = torch.zeros((per_rank, *probe.shape[1:]), dtype=gather_dtype or probe.dtype, device=probe.device)
    flat = torch.empty((world * per_rank, *local.shape[1:]), dtype=local.dtype, device=local.device)
    if timings is not None:
        import time
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        t0 = time.perf_counter()
    dist.all_gather_into_tensor(flat, local, group=group)
    if timings is not None:
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        timings["all_gather_s"] = timings.get("all_gather_s", 0.0) + time.perf_counter() - t0
        timings["gathered_bytes"] = flat.numel() * flat.element_size()
    gathered = flat.view(world, per_rank, *local.shape[1:])
    results = []
    for r in range(world):
        for k, i in enumerate(range(r, total, world)):
            results.append((i, gathered[r, k:k + 1].float()))
    results.sort(key=lambda t: t[0])
    return _blend(results, inputs.shape[0], local.shape[1], padded, roi, starts, pad, spatial, local.device, torch.float32)


def binarise(outputs: torch.Tensor) -> torch.Tensor:
    """engine.py:179-180."""
    return (torch.sigmoid(outputs) > 0.5).float()


def dice_per_class(outputs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """metric.py:37-49 for every class at once, on the device, one host sync for the whole vector
    (the reference calls .item() three times per class): 2|A&B| / (|A|+|B|), 0 when both are empty."""
    a, b = outputs.bool(), labels.bool()
    dims = (0, 2, 3, 4)
    inter = (a & b).sum(dims).double()
    denom = a.sum(dims).double() + b.sum(dims).double()
    return torch.where(denom > 0, 2.0 * inter / denom.clamp(min=1), torch.zeros_like(denom))


def infer(model, image: torch.Tensor, roi_size=(96, 96, 96), sw_batch_size: int = 1, overlap: float = 0.25,
          distributed: bool = False, group=None) -> torch.Tensor:
    """Engine.infer (engine.py:167-182): sliding-window DDIM sampling -> sigmoid -> > 0.5."""
    fn = sharded_sliding_window_inference if distributed else sliding_window_inference
    kw = dict(group=group) if distributed else {}
    with torch.no_grad():
        out = fn(image, roi_size, sw_batch_size, model, overlap, pred_type="ddim_sample", **kw)
    return binarise(out)
