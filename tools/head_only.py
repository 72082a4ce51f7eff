import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
T=65536
x=torch.randn(T,384,device='cuda').bfloat16()
E=torch.randn(30592,384,device='cuda').bfloat16()*0.02
bias=torch.zeros(30522,device='cuda'); mask=torch.ones(512,128,dtype=torch.uint8,device='cuda')
for _ in range(3): ops.sparse_head_fwd(x,E,bias,mask,512,128,30522,False)
torch.cuda.synchronize()
