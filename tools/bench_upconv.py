#!/usr/bin/env python3
"""The folded up-convolution (dua_upconv_k3_fwd) against the two launches it replaces (dua_deconv_k2s2_fwd into the concat
buffer + dua_conv3d_k3_fwd over the whole concat), same process, interleaved rounds, HIP events on the launch stream.
usage: bench_upconv.py [level0|level1] [rounds]     (level0: 96^3, 64 + 64 -> 64, blocked buffers; level1: 48^3, 64 + (128 -> 64) -> 64)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diff_unet_amos_amd import ops

SHAPES = {"level0": (1, 96, 64, 64, 64, 64, True), "level1": (1, 48, 64, 128, 64, 64, False)}   # N, S, Cs, Cu, Cmid, Cout, blocked


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "level0"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    N, S, Cs, Cu, Cmid, Cout, blk = SHAPES[which]
    dev, dt = "cuda", torch.float16
    cat = torch.randn(N, S, S, S, Cs + Cmid, device=dev).to(dt)
    u = torch.randn(N, S // 2, S // 2, S // 2, Cu, device=dev).to(dt)
    wc = torch.randn(Cout, Cs + Cmid, 3, 3, 3, device=dev) / (27 * (Cs + Cmid)) ** 0.5
    wd = torch.randn(Cu, Cmid, 2, 2, 2, device=dev) / Cu ** 0.5
    zc, zm = torch.zeros(Cout, device=dev), torch.zeros(Cmid, device=dev)
    w_skip, wu, btab = ops.pack_upconv_weights(wc, zc, wd, zm, Cs)
    wp, bp = ops.pack_conv3_weights(wc, zc, dt)
    dwp, dbp = ops.pack_deconv_weights(wd, zm, dt)
    y = torch.empty(N, S, S, S, Cout, device=dev, dtype=dt)
    stats = ops.stats_buffer(N, Cout, dev)
    sums = torch.zeros(N, Cu, 2, dtype=torch.float64, device=dev)
    sums[..., 1] = float((S // 2) ** 3)
    norm = ops.Norm(ops.stats_encode(sums), torch.ones(Cu, device=dev), torch.zeros(Cu, device=dev), (S // 2) ** 3)
    ws = ops.splitk_ws(dt, N, S, S, S, Cs + Cmid, Cout, dev)

    def folded():
        ops.upconv_k3(cat, Cs, 0, u, Cu, 0, norm, w_skip, wu, btab, Cout, y, 0, stats, in_blocked=blk, out_blocked=blk)

    def two_launches():
        ops.deconv_k2s2(u, Cu, 0, dwp, dbp, Cmid, cat, Cs, norm=norm, out_blocked=blk)
        ops.conv3d_k3(cat, Cs + Cmid, 0, wp, bp, Cout, y, 0, stats, workspace=ws, in_blocked=blk, out_blocked=blk)

    alg = 2.0 * N * S ** 3 * (27 * (Cs + Cmid) * Cout + Cu * Cmid)          # the reference's two layers
    exe = 2.0 * N * S ** 3 * (27 * Cs * Cout + 8 * Cu * Cout)                # what the regrouped form multiplies
    res = {"folded": [], "two launches": []}
    for fn in (folded, two_launches):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for name, fn in (("folded", folded), ("two launches", two_launches)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 100)                      # us per call
    f, t = min(res["folded"]), min(res["two launches"])
    print(f"{which}: folded {f:.1f} us ({alg / f / 1e6:.0f} TFLOP/s algorithmic, {exe / f / 1e6:.0f} executed: "
          f"{exe / 1e9:.1f} of {alg / 1e9:.1f} GFLOP) | transposed convolution + convolution {t:.1f} us "
          f"({alg / t / 1e6:.0f} TFLOP/s) | {100 * (f / t - 1):+.1f} %")


if __name__ == "__main__":
    main()
