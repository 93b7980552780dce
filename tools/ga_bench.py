import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpnn_amd.dense import gemm_nt
n=4_000_000
gz=torch.randn(n,128,device="cuda"); w=torch.randn(128,128,device="cuda"); cnt=torch.rand(n,128,device="cuda"); gab=torch.empty(n,256,device="cuda")
def f(): return gemm_nt(gz,w,emul=cnt,out2=gab[:, :128])
for _ in range(3): f()
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
print("ga gemm ms", a.elapsed_time(b)/10)
