#!/usr/bin/env python3
"""How much does a chain of small launches on one stream lose while a 96^3 convolution runs on another?  Everything is captured
into one HIP graph (fork -> [convolution(s) on the side stream | chain of n small launches on the main stream] -> join) and replayed.
Reports: chain alone, convolution alone, both, for the convolution at two / one workgroup per CU and cut into 1, 2, 4, 8 launches
over depth slabs.  usage: ubench_two_streams.py [chain length] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops                     # noqa: E402


def main():
    n_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev, dt = torch.device("cuda:0"), torch.float16
    S, C_ = 96, 48
    side = torch.cuda.Stream(device=dev)
    small = torch.randn(1728 * 192, device=dev).to(dt)                      # a stage-2 sized activation (12^3 tokens x 192)
    small2 = torch.randn(1728 * 192, device=dev).to(dt)
    w = torch.randn(C_, C_, 3, 3, 3, device=dev) / (27 * C_) ** 0.5
    wp, bp = ops.pack_conv3_weights(w, torch.zeros(C_, device=dev), dt)

    keep = []                               # a captured graph does not own the buffers its launches point at

    def conv_parts(parts, bg):
        d = S // parts
        x = torch.randn(1, d, S, S, C_, device=dev).to(dt)
        y = torch.empty_like(x)
        st = ops.stats_buffer(1, C_, dev)
        keep.extend((x, y, st))
        return lambda: [ops.conv3d_k3(x, C_, 0, wp, bp, C_, y, 0, st, background=bg) for _ in range(parts)]

    def chain():
        for _ in range(n_chain):
            ops.gelu_(small)

    def capture(side_fn, with_chain):
        def body():
            main = torch.cuda.current_stream()
            if side_fn is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    side_fn()
            if with_chain:
                chain()
            if side_fn is not None:
                main.wait_stream(side)
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        return g

    cases = {"chain alone": capture(None, True),
             "chain + ONE small launch on the side stream (fork/join only)": capture(lambda: ops.gelu_(small2), True)}
    for bg in (False, True):
        for parts in (1, 2, 4, 8):
            fn = conv_parts(parts, bg)
            tag = f"conv {'1' if bg else '2'} wg/CU x{parts}"
            cases[tag + " alone"] = capture(fn, False)
            cases[tag + " + chain"] = capture(fn, True)
    # the same pair as two single-branch graphs replayed on two streams (fork / join by events outside the graphs)
    g_chain = cases["chain alone"]
    pairs = {}
    for bg in (False, True):
        fn = conv_parts(1, bg)
        with torch.cuda.stream(side):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                fn()
        pairs[f"two graphs: conv {'1' if bg else '2'} wg/CU x1 | chain"] = g
    res = {k: [] for k in list(cases) + list(pairs)}
    for _ in range(rounds):
        for k, g in cases.items():
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 5 * 1e3)
        for k, g in pairs.items():
            torch.cuda.synchronize()
            main = torch.cuda.current_stream()
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record()
            for _ in range(5):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    g.replay()
                g_chain.replay()
                c.record()                      # the chain's own finish (last round)
                main.wait_stream(side)
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 5 * 1e3)
    for k, v in res.items():
        v.sort()
        print(f"{k:32s} median {v[len(v) // 2]:8.1f} us  best {v[0]:8.1f}")


if __name__ == "__main__":
    main()
