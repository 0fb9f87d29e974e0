import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import ops, _native as nv
dev="cuda"; dt=torch.float16
for S,cin,cout in ((96,128,64),(96,64,64),(96,24,64)):
    x=torch.randn(1,S,S,S,cin,device=dev).to(dt); w=torch.randn(cout,cin,3,3,3,device=dev)/(27*cin)**0.5
    wp,bp=ops.pack_conv3_weights(w,torch.zeros(cout,device=dev),dt)
    y=torch.empty(1,S,S,S,cout,device=dev,dtype=dt); stats=ops.stats_buffer(1,cout,dev)
    ws=torch.zeros(1<<20,device=dev)
    for variant in (210, 211):
        nv.check(nv.lib().dua_set_option(1,variant),"opt")
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            ops.conv3d_k3(x,cin,0,wp,bp,cout,y,0,stats,workspace=ws)
        e0.record(); ops.conv3d_k3(x,cin,0,wp,bp,cout,y,0,stats,workspace=ws); e1.record()
        torch.cuda.synchronize()
        tt=ws.view(torch.int64)[:256*12*4].view(256,12,4).cpu().double()
        t=tt[:,:8]; pp=tt[:,8:]
        tot,bar,epi,gt=t[...,0],t[...,1],t[...,2],t[...,3]
        us=e0.elapsed_time(e1)*1e3
        print(f"v4 ABL={variant-194 if variant==210 else 19} {S}^3 {cin}->{cout}: {us:.0f} us, clock {tot.mean()/us/1e3:.2f} GHz; per consumer wave: total {tot.mean():.0f} cyc, barrier {100*bar.mean()/tot.mean():.1f}%, epilogue {100*epi.mean()/tot.mean():.1f}%, slabs {gt.mean():.0f}, compute/slab {(tot.mean()-bar.mean()-epi.mean())/gt.mean():.0f} cyc (ideal 1536 for two waves per SIMD); producer: total {pp[...,0].mean():.0f}, barrier wait {100*pp[...,1].mean()/pp[...,0].mean():.1f}%, halo load/store {100*pp[...,2].mean()/pp[...,0].mean():.1f}%")
