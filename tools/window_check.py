"""LDS-window 3x3 conv kernels against each other and against the implicit-GEMM kernel through the C ABI (GPU only):
register-staged window kernel (dgmr_conv_tune window=1) vs LDS-DMA window kernel (3) must be bit-identical; both vs the generic
kernel (0) differ by summation order only.  python tools/window_check.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from skillful_nowcasting_amd import ops
from skillful_nowcasting_amd._lib import call, load
load(); ops.set_precision("bf16x3")
dev="cuda"
for (n,h,w,cin,cout,up,bn) in [(4,64,64,192,96,False,True),(2,128,128,96,96,True,True),(3,32,32,64,192,False,False),(4,16,16,96,96,False,True),(4,8,8,96,192,False,False),(3,64,64,48,48,False,True),(2,32,32,40,96,False,False),(2,32,32,24,200,True,True),(2,64,64,96,48,False,False)]:
    hin,win_=(h//2,w//2) if up else (h,w)
    x=torch.randn(n*hin*win_*cin,device=dev); wt=torch.randn(cout*9*cin,device=dev)*0.05
    bias=torch.randn(cout,device=dev); scale=torch.full((n,),0.5,device=dev)
    a=torch.rand(n*cin,device=dev)+0.5; b=torch.randn(n*cin,device=dev)*0.1
    res=torch.randn(n*h*w*cout,device=dev)
    wsp=torch.empty(2*wt.numel(),device=dev,dtype=torch.int16)
    call("dgmr_split_weights",wt.data_ptr(),wsp.data_ptr(),cout*9,cin,0,0,2,0,ops._stream())
    ys=[]
    for mode in (1,3,0):
        call("dgmr_conv_tune",-1,-1,mode,-1)
        y=torch.empty(n*h*w*cout,device=dev)
        ops._launch_conv(x,wt.data_ptr(),bias,scale,y,n,1,h,w,cin,cout,1,3,3,upsample=up,pre_a=a if bn else None,pre_b=b if bn else None,pre_group=1,scale_group=1,residual=res,w_split=wsp)
        torch.cuda.synchronize(); ys.append(y)
    print((n,h,w,cin,cout,up,bn),"register-staged vs LDS-DMA window max diff",(ys[0]-ys[1]).abs().max().item(),"vs generic",(ys[0]-ys[2]).abs().max().item(),"scale",ys[0].abs().max().item())
call("dgmr_conv_tune",-1,-1,-1,-1)
