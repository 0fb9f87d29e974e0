import sys, os
sys.path.insert(0, "/root/repo")
import torch
from diff_unet_amos_amd import ops
dev, dt = "cuda", torch.float16
S, cin, cout = 96, 64, 64
x = ops.to_blocked(torch.randn(1, S, S, S, cin, device=dev).to(dt))
w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt)
y = torch.empty(1, S, S, S, cout, device=dev, dtype=dt)
stats = ops.stats_buffer(1, cout, dev)
sums = torch.zeros(1, cin, 2, dtype=torch.float64, device=dev); sums[..., 1] = float(S ** 3)
norm = ops.Norm(ops.stats_encode(sums), torch.ones(cin, device=dev), torch.zeros(cin, device=dev), S ** 3, add=torch.zeros(cin, device=dev))
def t(n):
    f = lambda: ops.conv3d_k3(x, cin, 0, wp, bp, cout, y, 0, stats, norm=n, in_blocked=True, out_blocked=True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    return best
for rep in range(3):
    print(f"64->64 @96^3 wide: input already materialised {t(None):.1f} us | producer's norm + LeakyReLU + add fused {t(norm):.1f} us")
