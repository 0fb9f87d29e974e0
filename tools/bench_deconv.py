#!/usr/bin/env python3
"""A/B timing of deconv_k2s2 (forward) on the four upsampling layers of the 96^3 denoiser: this build against another build
of libdua_hip.so loaded beside it (graph replays, interleaved rounds in one process).
usage: bench_deconv.py [lib:<path>] [rounds]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import ops, _native as nv

SHAPES = [(6, 512, 256), (12, 256, 128), (24, 128, 64), (48, 128, 64)]       # (input extent, Cin, Cout); producer norm fused


def main():
    alt = [a for a in sys.argv[1:] if a.startswith("lib:")]
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    rounds = int(nums[0]) if nums else 9
    main_lib = nv.lib()
    libs = {"this": main_lib}
    if alt:
        L = ctypes.CDLL(os.path.abspath(alt[0][4:]))
        for name in ("dua_deconv_k2s2_fwd",):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = nv._SIGS[name]
        libs["other"] = L
    dev, dt = "cuda", torch.float16
    REP = 5
    print(f"{'layer':>24} " + " ".join(f"{k + ' us':>10}" for k in libs))
    for S, cin, cout in SHAPES:
        x = torch.randn(1, S, S, S, cin, device=dev).to(dt)
        w = torch.randn(cin, cout, 2, 2, 2, device=dev) / cin ** 0.5
        wp, bp = ops.pack_deconv_weights(w, torch.zeros(cout, device=dev), dt)
        dense = "--dense" in sys.argv          # the upsampled half as a tensor of its own instead of a slice of the concat buffer
        y = torch.empty(1, 2 * S, 2 * S, 2 * S, cout if dense else 2 * cout, device=dev, dtype=dt)
        sums = torch.zeros(1, cin, 2, dtype=torch.float64, device=dev)
        sums[..., 1] = float(S ** 3)
        norm = ops.Norm(ops.stats_encode(sums), torch.ones(cin, device=dev), torch.zeros(cin, device=dev), S ** 3)
        graphs = {}
        for k, L in libs.items():
            nv._lib = L
            run = lambda: ops.deconv_k2s2(x, cin, 0, wp, bp, cout, y, 0 if dense else cout, norm=norm)  # noqa: E731
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(REP):
                    run()
            graphs[k] = g
        nv._lib = main_lib
        res = {k: [] for k in libs}
        for rd in range(rounds + 1):
            for k in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); graphs[k].replay(); e1.record()
                torch.cuda.synchronize()
                if rd:
                    res[k].append(e0.elapsed_time(e1) / REP * 1e3)
        print(f"{S:>3}^3 -> {2 * S}^3 {cin:>4}->{cout:<4} " + " ".join(f"{sorted(v)[len(v) // 2]:10.1f}" for v in res.values()), flush=True)


if __name__ == "__main__":
    main()
