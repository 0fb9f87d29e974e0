#!/usr/bin/env python3
"""Where a conv3d_k3 workgroup spends its cycles: per-phase shader-clock stamps from the DIAGNOSTIC build
(tools/build_diag.sh stamp -DDUA_STAMP; run with DUA_DEBUG=1 DUA_HIP_LIB=tools/diag/stamp/libdua_hip.so).
usage: stamp_conv.py [shape indices of tools/bench_conv.py SHAPES, comma list] [--hot SECONDS] [--variant V]
Per shape: workgroups, in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups), and the
median cycles of prologue / each K phase / epilogue.  --hot: launch the layer back to back for that long first (the clock
a chip holds under sustained MFMA load, MI355X_MICROARCH.md 'DVFS give-back' item 6)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diff_unet_amos_amd import _native as nv, ops
from bench_conv import SHAPES


def main():
    L = nv.lib()
    if not hasattr(L, "dua_debug_stamps"):
        sys.exit("not a stamp build: DUA_DEBUG=1 DUA_HIP_LIB=tools/diag/stamp/libdua_hip.so")
    stamps = L.dua_debug_stamps_wide if "--wide" in sys.argv else L.dua_debug_stamps     # --wide: the stamps of conv3d_wide.hip
    stamps.restype, stamps.argtypes = ctypes.c_long, [ctypes.c_void_p, ctypes.c_long]
    nbytes = stamps(None, 0)
    host = np.zeros(nbytes // 8, dtype=np.uint64)
    only = [int(i) for i in sys.argv[1].split(",")] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else range(len(SHAPES))
    hot = float(sys.argv[sys.argv.index("--hot") + 1]) if "--hot" in sys.argv else 0.0
    if "--variant" in sys.argv:
        ops.CONV_POLICY = int(sys.argv[sys.argv.index("--variant") + 1])
    dev, dt = "cuda", torch.float16
    for idx in only:
        S, cin, cout, fused = SHAPES[idx]
        x = torch.randn(1, S, S, S, cin, device=dev).to(dt)
        tap = 16 if cin == 24 else None                     # the denoiser's first layer (tools/bench_conv.py)
        if tap:
            w = torch.randn(cout, 17, 3, 3, 3, device=dev) / (27 * 17) ** 0.5
            wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt, cin_packed=24,
                                            perm=list(range(1, 17)) + [0] + [-1] * 7, tap_channel=tap)
        else:
            w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
            wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt)
        y = torch.empty(1, S, S, S, cout, device=dev, dtype=dt)
        stats = ops.stats_buffer(1, cout, dev)
        nb = ops.conv3_workspace_bytes(dt, 1, S, S, S, cin, cout)
        ws = torch.empty(max(nb, 16) // 4, device=dev)
        norm = None
        if fused:
            sums = torch.zeros(1, cin, 2, dtype=torch.float64, device=dev)
            sums[..., 1] = float(S ** 3)
            norm = ops.Norm(ops.stats_encode(sums), torch.ones(cin, device=dev), torch.zeros(cin, device=dev), S ** 3,
                            add=torch.zeros(cin, device=dev))
        run = lambda: ops.conv3d_k3(x, cin, 0, wp, bp, cout, y, 0, stats, norm=norm, workspace=ws, tap_channel=tap)  # noqa: E731
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
        clock = np.median(cyc / rt)                                       # GHz
        span = (st[:, 1].max() - st[:, 0].min()) * 0.01                   # us, first start to last end
        pro = np.median(st[:, 3] - st[:, 2])
        nph = int(((st[0, 4:(30 if '--wide' in sys.argv else 59)] != 0).sum()))
        ph = np.array([np.median(st[:, 4 + i] - (st[:, 3 + i] if i else st[:, 3])) for i in range(nph)])
        last = st[:, 3 + nph] if nph else st[:, 3]
        tailc = np.median(st[:, 62] - last)
        epi = np.median(st[:, 63] - st[:, 62])
        life = np.median(cyc)
        print(f"{S}^3 {cin}->{cout}{' fused' if fused else ''}: event {us:.1f} us, stamped span {span:.1f} us, {len(st)} workgroups, "
              f"clock {clock:.3f} GHz, workgroup lifetime median {life / clock / 1e3:.2f} us (start spread "
              f"{(st[:, 0].max() - st[:, 0].min()) * 0.01:.1f} us)")
        print(f"    prologue split: issue loads {np.median(st[:, 59] - st[:, 2]):.0f} | statistics preamble {np.median(st[:, 60] - st[:, 59]):.0f} | "
              f"first slab to LDS (waits for it) {np.median(st[:, 61] - st[:, 60]):.0f} | halo transform + store + barrier {np.median(st[:, 3] - st[:, 61]):.0f}")
        print(f"    cycles: prologue {pro:.0f} | {nph} phases: mean {ph.mean() if nph else 0:.0f} min {ph.min() if nph else 0:.0f} max "
              f"{ph.max() if nph else 0:.0f} | post-loop {tailc:.0f} | epilogue {epi:.0f}")
        if nph:
            print("    phases: " + " ".join(f"{p:.0f}" for p in ph))
        t0s, t1s = (st[:, 0] - st[:, 0].min()) * 0.01, (st[:, 1] - st[:, 0].min()) * 0.01       # us
        act = [int(((t0s < e + 5) & (t1s > e + 5)).sum()) for e in np.arange(0, span, 10.0)]
        print("    workgroups in flight at 5, 15, 25, ... us: " + " ".join(str(a_) for a_ in act))
        order = np.argsort(t0s)
        lifes = (t1s - t0s)[order]
        q8 = max(1, len(lifes) // 8)
        print("    lifetime by start order (eighths, median us): " + " ".join(f"{np.median(lifes[i * q8:(i + 1) * q8]):.1f}" for i in range(8)))
        if "--wide" in sys.argv and st[0, 56] != 0:
            print(f"    prologue issue: bias + statistics requests {np.median(st[:, 56] - st[:, 2]):.0f} | halo requests "
                  f"{np.median(st[:, 57] - st[:, 56]):.0f} | weight plane {np.median(st[:, 59] - st[:, 57]):.0f}")
        if "--wide" in sys.argv:       # half-chunk boundaries: next halo transform + LDS store | barrier behind it
            bd = []
            for hc in range(7):
                if st[0, 30 + 2 * hc] != 0 and 4 + 3 * hc + 2 < 30:
                    bd.append(f"{np.median(st[:, 30 + 2 * hc] - st[:, 4 + 3 * hc + 2]):.0f}+{np.median(st[:, 31 + 2 * hc] - st[:, 30 + 2 * hc]):.0f}")
            print("    boundaries (store + barrier): " + " ".join(bd))


main()
