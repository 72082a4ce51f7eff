import sys, os, time, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
T=65536
x=torch.randn(T,384,device='cuda').bfloat16(); x2=torch.randn(T,1536,device='cuda').bfloat16()
for N,K in ((1152,384),(384,384),(1536,384),(384,1536)):
    A = x if K==384 else x2
    W=torch.randn(N,K,device='cuda').bfloat16()*0.02
    us=timeit(lambda: ops.gemm_nt(A,W))
    us2=timeit(lambda: torch.matmul(A,W.t()))
    print(f"gemm_nt N={N} K={K}: {us:.1f} us  {2*T*N*K/us/1e6:.0f} TF/s   (hipBLASLt via torch: {us2:.1f} us {2*T*N*K/us2/1e6:.0f} TF/s)")
E=torch.randn(30592,384,device='cuda').bfloat16()*0.02
bias=torch.zeros(30522,device='cuda'); mask=torch.ones(512,128,dtype=torch.uint8,device='cuda')
us=timeit(lambda: ops.sparse_head_fwd(x,E,bias,mask,512,128,30522,False),5)
print(f"head_fwd: {us:.1f} us {2*T*384*30522/us/1e6:.0f} TF/s")
for N,Kc in ((384,384),(1152,384),(1536,384),(384,1536)):
    A=torch.randn(T,N,device='cuda').bfloat16(); B=torch.randn(T,Kc,device='cuda').bfloat16()
    out=torch.zeros(N,Kc,device='cuda'); cs=torch.zeros(N,device='cuda')
    us=timeit(lambda: ops.gemm_tn_acc(A,B,out,cs))
    us2=timeit(lambda: torch.matmul(A.t(),B))
    print(f"gemm_tn N={N} Kc={Kc}: {us:.1f} us  {2*T*N*Kc/us/1e6:.0f} TF/s   (hipBLASLt via torch: {us2:.1f} us {2*T*N*Kc/us2/1e6:.0f} TF/s)")
