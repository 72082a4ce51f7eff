import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
T=65536
x=torch.randn(T,384,device='cuda').bfloat16()
W=torch.randn(1152,384,device='cuda').bfloat16()*0.02
for _ in range(5): ops.gemm_nt(x,W)
torch.cuda.synchronize()
