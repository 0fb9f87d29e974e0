#!/usr/bin/env python3
"""Companion of ubench_two_streams.py: the chain of small launches as ONE single-branch graph on the main stream and the side
work as another single-branch graph on a second stream, fork / join by events outside the graphs.  Reports the chain's own
finish time and the pair's.  usage: ubench_two_graphs.py [chain length] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops                     # noqa: E402


def main():
    n_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev, dt = torch.device("cuda:0"), torch.float16
    S, C_ = 96, 48
    side = torch.cuda.Stream(device=dev)
    work = torch.cuda.Stream(device=dev)                                     # the chain's stream (not the legacy default stream)
    small = torch.randn(1728 * 192, device=dev).to(dt)
    small2 = torch.randn(1728 * 192, device=dev).to(dt)
    w = torch.randn(C_, C_, 3, 3, 3, device=dev) / (27 * C_) ** 0.5
    wp, bp = ops.pack_conv3_weights(w, torch.zeros(C_, device=dev), dt)
    x = torch.randn(1, S, S, S, C_, device=dev).to(dt)
    y = torch.empty_like(x)
    st = ops.stats_buffer(1, C_, dev)

    def graph_of(fn, stream):
        with torch.cuda.stream(stream):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                fn()
        return g

    def chain():
        for _ in range(n_chain):
            ops.gelu_(small)

    g_chain = graph_of(chain, work)
    sides = {"nothing": None,
             "one small launch": graph_of(lambda: ops.gelu_(small2), side),
             "conv 2 wg/CU": graph_of(lambda: ops.conv3d_k3(x, C_, 0, wp, bp, C_, y, 0, st), side),
             "conv 1 wg/CU": graph_of(lambda: ops.conv3d_k3(x, C_, 0, wp, bp, C_, y, 0, st, background=True), side),
             "4 x conv 1 wg/CU": graph_of(lambda: [ops.conv3d_k3(x, C_, 0, wp, bp, C_, y, 0, st, background=True) for _ in range(4)], side)}
    res = {k: [] for k in sides}
    for _ in range(rounds):
        for k, g in sides.items():
            torch.cuda.synchronize()
            t0, t_chain, t_all = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(work):
                t0.record()
                if g is not None:
                    side.wait_stream(work)
                    with torch.cuda.stream(side):
                        g.replay()
                g_chain.replay()
                t_chain.record()
                if g is not None:
                    work.wait_stream(side)
                t_all.record()
            torch.cuda.synchronize()
            res[k].append((t0.elapsed_time(t_chain) * 1e3, t0.elapsed_time(t_all) * 1e3))
    for k, v in res.items():
        v.sort()
        m = v[len(v) // 2]
        print(f"side stream: {k:20s} chain finishes after {m[0]:8.1f} us, both after {m[1]:8.1f} us")


if __name__ == "__main__":
    main()
