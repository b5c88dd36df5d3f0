import torch, numpy as np, sys
sys.path.insert(0, ".")
from squeezedet_amd import ops
shapes=[("plus fire2 e3",8,92,309,96,64),("plus fire4 e3",8,92,309,192,128),("plus fire6 e3",8,45,153,288,192),("plus fire8 e3",8,45,153,384,256),
        ("res2 2b",8,94,311,64,64),("res3 2b",8,47,156,128,128),("res4 2b",8,24,78,256,256),("res5 2b",8,24,78,512,512)]
for name,n,h,w,cin,cout in shapes:
    x=torch.randn(n,h,w,cin,device="cuda").half()
    wt=(torch.randn(3,3,cin,cout,device="cuda")*0.05)
    pk=ops.pack_conv_weights(wt, torch.float16); b=torch.zeros(cout,device="cuda")
    y=torch.empty(n,h,w,cout,device="cuda",dtype=torch.float16)
    for i in range(300): ops.conv2d_nhwc(x,pk,b,1,"SAME",True,out=y)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): ops.conv2d_nhwc(x,pk,b,1,"SAME",True,out=y)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/200*1e3
    fl=2.0*9*cin*cout*n*h*w
    print("%-16s %4d->%4d %7d px: %7.1f us  %6.0f TF/s (%.2f of 2.5 PF)" % (name,cin,cout,n*h*w,us,fl/us/1e6,fl/us/1e6/2500))
