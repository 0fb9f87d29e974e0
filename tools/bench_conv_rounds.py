import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops
dev, dt = 'cuda', torch.float16
def run(D,H,W,cin=64,cout=64,fused=True, reps=7):
    x = torch.randn(1,D,H,W,cin, device=dev).to(dt)
    w = torch.randn(cout,cin,3,3,3, device=dev)/(27*cin)**0.5
    wp,bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt)
    y = torch.empty(1,D,H,W,cout, device=dev, dtype=dt)
    st = ops.stats_buffer(1,cout,dev)
    norm=None
    if fused:
        sums = torch.zeros(1,cin,2,dtype=torch.float64,device=dev); sums[...,1]=float(D*H*W)
        norm = ops.Norm(ops.stats_encode(sums), torch.ones(cin,device=dev), torch.zeros(cin,device=dev), D*H*W, add=torch.zeros(1,cin,device=dev))
    f = lambda: ops.conv3d_k3(x,cin,0,wp,bp,cout,y,0,st,norm=norm)
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5): f()
    ts=[]
    for _ in range(reps):
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)/5*1e3)
    ts.sort()
    tiles = (D//8)*(H//8)*(W//8)
    print(f"{D}x{H}x{W} {cin}->{cout}{' fused' if fused else ''}: {tiles} tiles = {tiles/512:.3f} rounds: {ts[len(ts)//2]:.1f} us  ({ts[len(ts)//2]/tiles*512:.1f} us per round of 512)")
for shape in [(64,64,64),(64,64,128),(64,96,128),(96,96,96),(96,96,128),(128,128,64),(128,128,96),(128,128,128)]:
    run(*shape)
