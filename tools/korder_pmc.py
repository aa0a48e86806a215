"""Counter probe for the K-order question: ONE 3x3 convolution (default: 112 x 64^2 320->320 grouped, the batch-8 level-0 launch)
run eagerly 4 times with tap-major and 4 times with chunk-major packed weights, to be wrapped in `rocprofv3 --pmc ...`;
tools/korder_pmc_sum.py averages the counters per kernel name (the two orders are different template instantiations).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o run -- python3 tools/korder_pmc.py [bn] [N H Cin Cout]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
a = [int(v) for v in sys.argv[1:]]
bn = a[0] if a else 320
N, H, Cin, Cout = a[1:5] if len(a) >= 5 else (112, 64, 320, 320)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, H, Cin, generator=g).to("cuda", torch.float16)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
b = torch.randn(Cout, generator=g) * 0.1
ops.FORCE_BN = bn
for ko in (0, 1):
    ops.CHUNK_MAJOR = bool(ko)
    pw = ops.pack_weight(w, b, torch.float16, "cuda")
    for _ in range(4):
        out = ops.conv_gemm(x, pw)
    torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))
