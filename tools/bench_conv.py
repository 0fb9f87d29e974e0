#!/usr/bin/env python3
"""A/B timing of conv3d_k3 variants on the layer shapes of the 96^3 x 16-class denoiser
(interleaved rounds in one process, HIP events on the launch stream).
usage: bench_conv.py <variants> [shape indices] [rounds]
  variants: comma list of launch forms (dua_conv3_desc.policy, ops.CONV_POLICY) for this build, and/or "lib:<path>" = the same entry point of ANOTHER
  build of libdua_hip.so loaded beside it (same-box, same-process comparison of two kernel generations; box-to-box
  spread is +-10 %, so nothing else ranks two builds)."""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import ops, _native as nv

SHAPES = [  # (S, Cin, Cout, fused prologue)
    (96, 24, 64, False), (96, 64, 64, True), (96, 128, 64, False),
    (48, 64, 64, True), (48, 128, 64, False),
    (24, 64, 128, False), (24, 128, 128, True), (24, 256, 128, False),
    (12, 128, 256, False), (12, 256, 256, True), (12, 512, 256, False),
    (6, 256, 512, False), (6, 512, 512, True),
    # Swin-UNETR widths (BASELINE config 5): encoder1 / decoder1 at 96^3, encoder2 / decoder2 at 48^3
    (96, 24, 48, False), (96, 48, 48, True), (96, 96, 48, False), (48, 48, 48, True), (48, 96, 48, False),
]


def main():
    def parse(v):
        if v.startswith("lib:"):
            return v
        if v.endswith("b"):                 # e.g. "0b": variant 0 reading its input as 16-channel blocks (dua_conv3_desc.layout) where the wide-tile form runs
            return (int(v[:-1]), 1)
        return int(v)
    variants = [parse(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
    main_lib = nv.lib()
    alts = {}
    for v in variants:
        if isinstance(v, str):
            L = ctypes.CDLL(os.path.abspath(v[4:]))
            for name in ("dua_conv3d_k3_fwd", "dua_abi_version"):
                fn = getattr(L, name)
                fn.restype, fn.argtypes = nv._SIGS[name]
            alts[v] = L
    only = [int(i) for i in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dt = torch.float16
    dev = "cuda"
    print(f"{'shape':>22} " + " ".join(f"{(os.path.basename(v)[-12:] if isinstance(v, str) else 'v' + str(v)) + ' us':>15} {'TF':>7}" for v in variants))
    for idx, (S, cin, cout, fused) in enumerate(SHAPES):
        if only is not None and idx not in only:
            continue
        B = int(os.environ.get("BENCH_CONV_BATCH", "1"))            # samples per launch (config 3 runs 4, config 4 runs 2)
        x = torch.randn(B, S, S, S, cin, device=dev).to(dt)
        tap = 16 if cin == 24 else None                     # the denoiser's first layer: [16 x_t | image | pad], tap form
        if tap:
            w = torch.randn(cout, 17, 3, 3, 3, device=dev) / (27 * 17) ** 0.5
            wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt, cin_packed=24,
                                            perm=list(range(1, 17)) + [0] + [-1] * 7, tap_channel=tap)
        else:
            w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
            wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt)
        y = torch.empty(B, S, S, S, cout, device=dev, dtype=dt)
        stats = ops.stats_buffer(B, cout, dev)
        nb = max(ops.conv3_workspace_bytes(dt, B, S, S, S, cin, cout), 8 * B * S ** 3 * 256 * 4 if S <= 24 else 0)
        ws = torch.empty(max(nb, 16) // 4, device=dev)
        norm = None
        if fused:
            sums = torch.zeros(B, cin, 2, dtype=torch.float64, device=dev)
            sums[..., 1] = float(S ** 3)                   # mean 0, variance 1
            st = ops.stats_encode(sums)
            norm = ops.Norm(st, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), S ** 3,
                            add=torch.zeros(B, cin, device=dev))
        res = {v: [] for v in variants}
        # every variant's launch is captured once (REP launches per graph): replay timing is free of the host's per-call cost,
        # which exceeds the run time of the small layers
        REP = 5
        graphs = {}
        for v in variants:
            if isinstance(v, str):
                nv._lib = alts[v]
                ops.CONV_POLICY = 0
            else:
                nv._lib = main_lib
                ops.CONV_POLICY = v[0] if isinstance(v, tuple) else v
                blk = isinstance(v, tuple) and v[1] and ops.conv3_kernel_kind(dt, B, S, S, S, cin, cin, cout, fused=fused, tap_channel=tap) == ops.KIND_WIDE
            run = lambda blk=(False if isinstance(v, str) else blk): ops.conv3d_k3(x, cin, 0, wp, bp, cout, y, 0, stats, norm=norm, workspace=ws, tap_channel=tap, in_blocked=blk)  # noqa: E731
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(REP):
                    run()
            graphs[v] = g
        nv._lib = main_lib
        ops.CONV_POLICY = 0
        for rd in range(rounds + 1):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graphs[v].replay()
                e1.record()
                torch.cuda.synchronize()
                if rd > 0:
                    res[v].append(e0.elapsed_time(e1) / REP * 1e3)
        fl = 2.0 * (17 if tap else cin) * cout * 27 * S ** 3 * B
        out = []
        for v in variants:
            us = sorted(res[v])[len(res[v]) // 2]
            out.append(f"{us:15.1f} {fl / us / 1e6:7.1f}")
        nv._lib = main_lib
        print(f"{S:>3}^3 {cin:>4}->{cout:<4} {'fused' if fused else '     '} " + " ".join(out), flush=True)


if __name__ == "__main__":
    main()
