#!/usr/bin/env python3
"""Micro-benchmark of dua_token_linear against torch.nn.functional.linear (hipBLASLt) on the shapes of the Swin stages.

    python tools/bench_toklin.py [reps]
"""
import sys
import os
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(117649, 48, 144, "plain"), (110592, 48, 192, "gelu"), (110592, 192, 48, "residual"), (110592, 96, 48, "plain"),
              (110592, 96, 48, "stats"), (884736, 96, 48, "plain"), (884736, 96, 48, "stats"), (884736, 24, 48, "stats"),
              (13824, 384, 96, "plain"), (21952, 96, 192, "plain"), (13824, 96, 192, "gelu")]
    print(f"{'M':>8s} {'K':>4s} {'N':>4s} {'mode':>9s} {'dua us':>8s} {'lib us':>8s} {'MB':>7s} {'GB/s':>7s}")
    for M, K, N, mode in shapes:
        A = torch.randn(M, K, device=dev, generator=g).half()
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).half()
        b = torch.randn(N, device=dev, generator=g)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        x = torch.zeros(M, N, device=dev)
        st = ops.stats_buffer(1, N, dev)
        if mode == "stats":
            fn = lambda: ops.token_linear(A, W, None, "stats", out=out, stats=st, samples=1)  # noqa: E731
            lib = lambda: (torch.matmul(A, W.t(), out=out), ops.instnorm_stats(out.view(1, 1, 1, M, N), N, st))  # noqa: E731
            mb = (M * K * 2 + M * N * 2) / 1e6
        elif mode == "residual":
            fn = lambda: ops.token_linear(A, W, b, "residual", x=x)  # noqa: E731
            lib = lambda: x.add_(F.linear(A, W, b.half()))  # noqa: E731
            mb = (M * K * 2 + M * N * 8) / 1e6
        elif mode == "gelu":
            fn = lambda: ops.token_linear(A, W, b, "gelu", out=out)  # noqa: E731
            lib = lambda: ops.gelu_(F.linear(A, W, b.half()))  # noqa: E731
            mb = (M * K * 2 + M * N * 2) / 1e6
        else:
            fn = lambda: ops.token_linear(A, W, b, "plain", out=out)  # noqa: E731
            lib = lambda: F.linear(A, W, b.half())  # noqa: E731
            mb = (M * K * 2 + M * N * 2) / 1e6
        t1, t2 = timeit(fn, reps), timeit(lib, reps)
        print(f"{M:8d} {K:4d} {N:4d} {mode:>9s} {t1:8.1f} {t2:8.1f} {mb:7.1f} {mb / t1 * 1e3 / 1e3:7.0f}")


if __name__ == "__main__":
    main()
