import torch, time, sys
sys.path.insert(0, '/root/repo')
from diff_unet_amos_amd import ops
dev='cuda'
wc = torch.randn(64, 128, 3,3,3, device=dev); wd = torch.randn(64, 64, 2,2,2, device=dev)
z = torch.zeros(64, device=dev)
for _ in range(3): ops.pack_upconv_weights(wc, z, wd, z, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.pack_upconv_weights(wc, z, wd, z, 64)
e1.record(); torch.cuda.synchronize()
print("pack_upconv_weights (all launches)", e0.elapsed_time(e1)/20*1e3, "us")
