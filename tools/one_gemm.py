"""Two conv3x3 launches in a loop and nothing else: the workload for stamp / ablation builds (ES_HIP_LIB) under rocprofv3."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
def run(N,H,Cin,Cout,k,R=20):
    x=torch.randn(N,H,H,Cin,generator=g,device=dev).half()
    pw=ops.pack_weight(torch.randn(Cout,Cin,k,k,generator=g,device=dev)*0.02, torch.randn(Cout,generator=g,device=dev)*0.1, torch.float16, dev)
    out=None
    for _ in range(R): out=ops.conv_gemm(x,pw,out=out,splitk=1)
    torch.cuda.synchronize()
run(16,64,320,320,3); run(6,64,320,320,3)
