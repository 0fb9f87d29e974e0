import sys, os
sys.path.insert(0, '/root/repo')
import torch
from diff_unet_amos_amd import ops, _native as nv
dev, dt = "cuda", torch.float16
for S, cin, cout, fused in [(96, 64, 64, True), (96, 128, 64, False), (48, 128, 64, False)]:
    x = torch.randn(1, S, S, S, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
    wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt)
    y = torch.empty(1, S, S, S, cout, device=dev, dtype=dt)
    stats = ops.stats_buffer(1, cout, dev)
    norm = None
    if fused:
        sums = torch.zeros(1, cin, 2, dtype=torch.float64, device=dev); sums[..., 1] = float(S ** 3)
        norm = ops.Norm(ops.stats_encode(sums), torch.ones(cin, device=dev), torch.zeros(cin, device=dev), S ** 3, add=torch.zeros(cin, device=dev))
    res = {}
    for bg in (False, True):
        run = lambda: ops.conv3d_k3(x, cin, 0, wp, bp, cout, y, 0, stats, norm=norm, background=bg)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(5): run()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        res[bg] = sorted(ts)[len(ts) // 2]
    print(f"{S}^3 {cin}->{cout} {'fused' if fused else ''}: two workgroups per CU {res[False]:.1f} us, one per CU {res[True]:.1f} us, ratio {res[True]/res[False]:.2f}", flush=True)
