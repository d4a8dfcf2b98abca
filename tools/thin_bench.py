"""Thin 3x3 heads (N <= 4): ffsr_conv3x3_thin_f32 against the GEMM kernels and an fp64 torch convolution, with timings."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import importlib
ops = importlib.import_module("image-super-resolution_amd.ops")
torch.manual_seed(0)
dev = "cuda"
import torch.nn.functional as F
for (B,H,W,Cin,N,act,res) in [(1,1360,2040,64,3,0,False),(1,1360,2040,128,3,0,True),(1,1360,2040,32,3,0,False),(1,1360,2040,16,1,4,False),(1,1360,2040,8,1,4,False),(2,67,131,64,3,3,True),(1,70,70,16,4,1,False),(1,64,64,128,2,2,True)]:
    w = torch.randn(N,Cin,3,3)*0.1; b = torch.randn(N)
    x = torch.randn(B,H,W,Cin,device=dev)
    r = torch.randn(B,H,W,4,device=dev)[...,:N] if res else None
    cv = ops.pack_conv(w,b,dev)
    slope = 0.2 if act==3 else 0.0
    def run(thin):
        ops.THIN3 = thin
        return ops.conv2d(x, cv, act=act, slope=slope, res=r, cscale=0.1 if res else 1.0)
    y1 = run(True); y0 = run(False)
    ref = F.conv2d(x.permute(0,3,1,2).double(), w.double().to(dev), b.double().to(dev), padding=1)
    ref = {0:lambda t:t,1:F.gelu,2:F.relu,3:lambda t:F.leaky_relu(t,0.2),4:torch.sigmoid,5:F.silu}[act](ref)
    ref = ref.permute(0,2,3,1)
    if res: ref = ref*0.1 + r.double()
    e1 = (y1.double()-ref).abs().max().item(); e0 = (y0.double()-ref).abs().max().item()
    def t(thin):
        run(thin); torch.cuda.synchronize(); t0=time.time()
        for _ in range(10): run(thin)
        torch.cuda.synchronize(); return (time.time()-t0)/10*1e6
    print(f"B{B} {H}x{W} Cin{Cin} N{N} act{act} res{res}: err thin {e1:.2e} gemm {e0:.2e}; us thin {t(True):.0f} gemm {t(False):.0f}", flush=True)
