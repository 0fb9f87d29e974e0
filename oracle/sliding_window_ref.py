"""Oracle (test infrastructure, never shipped): loop-based restatement of the sliding-window schedule
the reference obtains from monai.inferers.sliding_window_inference at engine.py:173-177 (mode="constant",
padding_mode="constant", cval=0).  MONAI is absent from the reference tree and from this image, so this
follows MONAI's documented behaviour: PARITY UNPINNED (pinned only against the hand-worked cases in
tests/test_sliding_window.py, e.g. 256x256x192 / roi 96 / overlap 0.25 -> 4x4x3 = 48 windows)."""
import numpy as np


def window_starts_1d(size, roi, overlap):
    if size <= roi:
        return [0]
    step = int(roi * (1 - overlap))
    step = step if step > 0 else 1
    starts, pos = [], 0
    while True:
        starts.append(min(pos, size - roi))
        if pos + roi >= size:
            break
        pos += step
    return starts


def sliding_window_ref(volume: np.ndarray, roi, overlap, fn):
    """volume [B, Cin, D, H, W]; fn(window [1,Cin,*roi]) -> [1, C, *roi].  Average of overlapping windows."""
    B = volume.shape[0]
    sp = volume.shape[2:]
    padw = [(0, 0), (0, 0)]
    for s, r in zip(sp, roi):
        diff = max(r - s, 0)
        padw.append((diff // 2, diff - diff // 2))
    v = np.pad(volume, padw, mode="constant")
    P = v.shape[2:]
    out = cnt = None
    for b in range(B):
        for d in window_starts_1d(P[0], roi[0], overlap):
            for h in window_starts_1d(P[1], roi[1], overlap):
                for w in window_starts_1d(P[2], roi[2], overlap):
                    o = fn(v[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]])
                    if out is None:
                        out = np.zeros((B, o.shape[1], *P), dtype=np.float64)
                        cnt = np.zeros((B, 1, *P), dtype=np.float64)
                    out[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += o
                    cnt[b:b + 1, :, d:d + roi[0], h:h + roi[1], w:w + roi[2]] += 1
    out = out / cnt
    return out[:, :, padw[2][0]:padw[2][0] + sp[0], padw[3][0]:padw[3][0] + sp[1], padw[4][0]:padw[4][0] + sp[2]]
