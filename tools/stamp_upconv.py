#!/usr/bin/env python3
"""Where a workgroup of the folded up-convolution (csrc/upconv.hip) spends its cycles: in-kernel shader-clock stamps from the
DIAGNOSTIC build (tools/build_diag.sh stamp -DDUA_STAMP; run with DUA_DEBUG=1 DUA_HIP_LIB=tools/diag/stamp/libdua_hip.so).
usage: stamp_upconv.py [level0|level1] [--hot SECONDS]
Medians over the workgroups of one launch: prologue, every K phase of the skip half, the hand-over (coarse halo transform +
store + barrier), the upsampled half per 64-channel group, the epilogue; clock = delta s_memtime / delta s_memrealtime."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diff_unet_amos_amd import _native as nv, ops

SHAPES = {"level0": (1, 96, 64, 64, 64, 64, True), "level1": (1, 48, 64, 128, 64, 64, False)}   # N, S, Cs, Cu, Cmid, Cout, blocked


def main():
    L = nv.lib()
    if not hasattr(L, "dua_debug_stamps_upconv"):
        sys.exit("not a stamp build: DUA_DEBUG=1 DUA_HIP_LIB=tools/diag/stamp/libdua_hip.so")
    stamps = L.dua_debug_stamps_upconv
    stamps.restype, stamps.argtypes = ctypes.c_long, [ctypes.c_void_p, ctypes.c_long]
    nbytes = stamps(None, 0)
    host = np.zeros(nbytes // 8, dtype=np.uint64)
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "level0"
    hot = float(sys.argv[sys.argv.index("--hot") + 1]) if "--hot" in sys.argv else 0.0
    N, S, Cs, Cu, Cmid, Cout, blk = SHAPES[which]
    dev, dt = "cuda", torch.float16
    xs = torch.randn(N, S, S, S, Cs + Cmid, device=dev).to(dt)
    u = torch.randn(N, S // 2, S // 2, S // 2, Cu, device=dev).to(dt)
    wc = torch.randn(Cout, Cs + Cmid, 3, 3, 3, device=dev) / (27 * (Cs + Cmid)) ** 0.5
    wd = torch.randn(Cu, Cmid, 2, 2, 2, device=dev) / Cu ** 0.5
    w_skip, wu, btab = ops.pack_upconv_weights(wc, torch.zeros(Cout, device=dev), wd, torch.zeros(Cmid, device=dev), Cs)
    y = torch.empty(N, S, S, S, Cout, device=dev, dtype=dt)
    stats = ops.stats_buffer(N, Cout, dev)
    sums = torch.zeros(N, Cu, 2, dtype=torch.float64, device=dev)
    sums[..., 1] = float((S // 2) ** 3)
    norm = ops.Norm(ops.stats_encode(sums), torch.ones(Cu, device=dev), torch.zeros(Cu, device=dev), (S // 2) ** 3)
    run = lambda: ops.upconv_k3(xs, Cs, 0, u, Cu, 0, norm, w_skip, wu, btab, Cout, y, 0, stats, in_blocked=blk, out_blocked=blk)  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    if hot > 0:
        t0 = time.time()
        while time.time() - t0 < hot:
            for _ in range(50):
                run()
            torch.cuda.synchronize()
    stamps(host.ctypes.data, nbytes)          # clears
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    stamps(host.ctypes.data, nbytes)
    st = host.reshape(8192, 64).astype(np.int64)
    st = st[st[:, 0] != 0]
    rt = (st[:, 1] - st[:, 0]).astype(np.float64) * 10.0            # ns
    cyc = (st[:, 63] - st[:, 2]).astype(np.float64)
    clock = np.median(cyc / rt)
    span = (st[:, 1].max() - st[:, 0].min()) * 0.01
    U = 3 * (Cs // 16)
    G = Cu // 64
    med = lambda a: int(np.median(a))                                 # noqa: E731
    ph = [med(st[:, 4 + i] - (st[:, 3 + i] if i else st[:, 3])) for i in range(U)]
    print(f"{which}: event {us:.1f} us, stamped span {span:.1f} us, {len(st)} workgroups, clock {clock:.3f} GHz, "
          f"workgroup lifetime median {np.median(rt) * 1e-3:.2f} us = {med(cyc)} cycles (start spread {(st[:, 0].max() - st[:, 0].min()) * 0.01:.1f} us)")
    print(f"    prologue {med(st[:, 3] - st[:, 2])} | skip half: {U} phases, sum {sum(ph)} (mean {sum(ph) // U}, MFMA 2304 each) | "
          f"hand-over {med(st[:, 30] - st[:, 3 + U])} | upsampled half " +
          " + ".join(str(med(st[:, 31 + g] - (st[:, 30 + g]))) for g in range(G)) + f" (MFMA 8192 per group) | "
          f"epilogue {med(st[:, 63] - st[:, 62])}")
    print("    phases:", " ".join(str(p) for p in ph))
    print(f"    epilogue split: read + stage + store (wave 0) {med(st[:, 60] - st[:, 62])} | sums to LDS + workgroup barrier {med(st[:, 61] - st[:, 60])} | "
          f"final reduction + atomics {med(st[:, 63] - st[:, 61])}")
    t0, t1 = (st[:, 0] - st[:, 0].min()) * 0.01, (st[:, 1] - st[:, 0].min()) * 0.01       # us
    edges = np.arange(0, span + 10, 10.0)
    act = [int(((t0 < e + 5) & (t1 > e + 5)).sum()) for e in edges[:-1]]
    print("    workgroups in flight at 5, 15, 25, ... us:", " ".join(str(a) for a in act))
    order = np.argsort(t0)
    life = (t1 - t0)[order]
    q = len(life) // 8
    print("    lifetime by start order (eighths, median us):", " ".join(f"{np.median(life[i * q:(i + 1) * q]):.1f}" for i in range(8)))
    border = st[:, 3] - st[:, 2]
    print(f"    prologue: quartiles {int(np.percentile(border, 25))} / {int(np.percentile(border, 75))} / max {int(border.max())}")


if __name__ == "__main__":
    main()
