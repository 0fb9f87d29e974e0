import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import ops, _native as nv
dev="cuda"; dt=torch.float16
for S,cin,cout in ((96,128,64),(96,64,64),(96,24,64),(48,64,64)):
    x=torch.randn(1,S,S,S,cin,device=dev).to(dt); w=torch.randn(cout,cin,3,3,3,device=dev)/(27*cin)**0.5
    wp,bp=ops.pack_conv3_weights(w,torch.zeros(cout,device=dev),dt)
    y=torch.empty(1,S,S,S,cout,device=dev,dtype=dt); stats=ops.stats_buffer(1,cout,dev)
    ws=torch.zeros(1<<22,device=dev)
    nv.check(nv.lib().dua_set_option(1,132),"opt")
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ops.conv3d_k3(x,cin,0,wp,bp,cout,y,0,stats,workspace=ws)
    e0.record(); ops.conv3d_k3(x,cin,0,wp,bp,cout,y,0,stats,workspace=ws); e1.record()
    torch.cuda.synchronize()
    nwg=(S//4)*(S//8)*(S//8)
    t=ws.view(torch.int64)[:nwg*32].view(nwg,4,8).cpu().double()
    tot,bar,pro,epi,hb,gt=(t[...,i] for i in range(6))
    us=e0.elapsed_time(e1)*1e3
    print(f"v2 {S}^3 {cin}->{cout}: {us:.0f} us; per WG: total {tot.mean():.0f} cyc (eff clock {tot.mean()*nwg/512/us/1e3:.2f} GHz if 2 WG/CU back-to-back), prologue {100*pro.mean()/tot.mean():.1f}%, barrier+slab-store {100*bar.mean()/tot.mean():.1f}%, chunk halo store {100*hb.mean()/tot.mean():.1f}%, epilogue {100*epi.mean()/tot.mean():.1f}%, slabs {gt.mean():.0f}, compute/slab {(tot.mean()-bar.mean()-pro.mean()-epi.mean()-hb.mean())/gt.mean():.0f} cyc")
