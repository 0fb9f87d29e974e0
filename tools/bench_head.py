import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops
dev = "cuda"
u = torch.randn(2, 96, 96, 96, 64, device=dev).half(); w = torch.randn(16, 64, device=dev) * 0.1; b = torch.zeros(16, device=dev)
dl = torch.randn(2, 96, 96, 96, 16, device=dev).half()
for name, fn in (("fwd", lambda: ops.head_fwd(u, w, b)), ("bwd", lambda: ops.head_bwd(dl, u, w))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 10 * 1e3, 1), "us (incl. output allocation/zeroing)")
