#!/usr/bin/env python3
"""Per-launch floor of a chain of dependent small launches (what the <= 24^3 levels of the denoiser are): a HIP graph of 24
launches of dua_chain_probe per mode, replayed; prints microseconds per launch.
  mode 0 empty kernel | 1 one dependent round trip (load, store) | 2 + workgroup reduction and 64 system-scope atomics at the
  end (a convolution's statistics) | 3 + a read of the previous launch's atomic words first (the statistics preamble)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import _native as nv


def main():
    L = nv.lib()
    dev = "cuda"
    N = 24
    for wgs in (64, 256, 1024):
        a = torch.zeros(wgs * 256, device=dev)
        b = torch.zeros(wgs * 256, device=dev)
        words = torch.zeros(64, dtype=torch.int64, device=dev)
        row = []
        for mode in range(4):
            def run():
                for i in range(N):
                    src, dst = (a, b) if i % 2 == 0 else (b, a)
                    nv.check(L.dua_chain_probe(mode, wgs, nv.ptr(src), nv.ptr(dst), nv.ptr(words), nv.stream_ptr()), "probe")
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            ts = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / N)
            row.append(sorted(ts)[len(ts) // 2])
        print(f"{wgs:5d} workgroups: empty {row[0]:5.2f} | load+store {row[1]:5.2f} | + reduction and atomics {row[2]:5.2f} | "
              f"+ read of the previous launch's words {row[3]:5.2f}  us per launch", flush=True)


if __name__ == "__main__":
    main()
