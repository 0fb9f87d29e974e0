import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import ops, _native as nv
dev="cuda"; dt=torch.float16
for variant in (210, 211):
  for S,cin,cout in ((96,128,64),(96,64,64)):
    x=torch.randn(1,S,S,S,cin,device=dev).to(dt); w=torch.randn(cout,cin,3,3,3,device=dev)/(27*cin)**0.5
    wp,bp=ops.pack_conv3_weights(w,torch.zeros(cout,device=dev),dt)
    y=torch.empty(1,S,S,S,cout,device=dev,dtype=dt); stats=ops.stats_buffer(1,cout,dev)
    ws=torch.zeros(1<<20,device=dev)
    nv.check(nv.lib().dua_set_option(1,variant),"opt")
    for _ in range(3):
        ops.conv3d_k3(x,cin,0,wp,bp,cout,y,0,stats,workspace=ws)
    torch.cuda.synchronize()
    t=ws.view(torch.int64)[:256*16].view(256,4,4).cpu().double()
    tot,bar,epi,gt=t[...,0],t[...,1],t[...,2],t[...,3]
    print(f"variant {variant} {S}^3 {cin}->{cout}: slabs/WG {gt.mean():.0f} total cyc {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); barrier wait {bar.mean():.0f} ({100*bar.mean()/tot.mean():.1f}%), epilogue {epi.mean():.0f} ({100*epi.mean()/tot.mean():.1f}%); per-slab non-barrier {(tot.mean()-bar.mean()-epi.mean())/gt.mean():.0f} cyc")
