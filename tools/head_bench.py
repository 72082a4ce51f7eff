import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
T=65536
x=torch.randn(T,384,device='cuda').bfloat16()
E=torch.randn(30592,384,device='cuda').bfloat16()*0.02
bias=torch.zeros(30522,device='cuda'); mask=torch.ones(512,128,dtype=torch.uint8,device='cuda')
f=lambda: ops.sparse_head_fwd(x,E,bias,mask,512,128,30522,False)
for _ in range(2): f()
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
us=e0.elapsed_time(e1)/5*1e3
print(f"DBG={os.environ.get('SM_ARES_DBG','0')} head_fwd: {us:.1f} us {2*T*384*30522/us/1e6:.0f} TF/s")
