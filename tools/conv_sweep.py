import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frmap_amd import ops
dev="cuda"; dt=torch.bfloat16
def run(B,H,C,reps=20):
    x=torch.randn(B,H,H,C,device=dev).to(dt)
    w=ops.pack_conv_weight(torch.randn(C,C,3,3,device=dev)*(2.0/(C*9))**0.5,dt)
    sh=torch.zeros(C,device=dev)
    for _ in range(3): ops.conv_igemm(x,w,sh,C,3,1,1,True,None)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv_igemm(x,w,sh,C,3,1,1,True,None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for (H,C) in [(14,256),(28,128),(7,512),(56,64)]:
    for B in [32,64,96,128,160,192,256,320,384,512,768,1024,2048]:
        tiles=((B*H*H+255)//256)*(C//64)
        us=run(B,H,C)
        fl=2.0*B*H*H*C*C*9
        print(f"H={H} C={C} B={B:5d} tiles={tiles:6d} rounds={tiles/512:6.2f} {us:8.1f} us {fl/us/1e6:7.1f} TF  us/round={us/max(1,-(-tiles//512)):6.1f}",flush=True)
