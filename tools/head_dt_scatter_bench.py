"""Density-adaptive head backward (dt half): the fused matrix form (head_dt192_kernel + LayerNorm' + GELU' epilogue) against the scatter
chain (head_dt_scatter_kernel + fp32 -> bf16 + sm_layernorm_bwd + sm_gelu_bwd) at the bench shape, by share of live activations.
The break-even density is what HipBertMLM.dt_scatter_density is set from (profiles/r5_head_dt_scatter.txt)."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import ops
dev = torch.device("cuda", 0)
V, H, B, S = 30522, 384, 512, 128
g = torch.Generator(device=dev).manual_seed(3)
T = B * S
E = (torch.randn(V, H, device=dev, generator=g) * 0.05).to(torch.bfloat16)
x32 = torch.randn(T, H, device=dev, generator=g)                     # the transform LayerNorm's fp32 input
mean, rstd = x32.mean(1), 1.0 / x32.std(1)
gamma = torch.ones(H, device=dev)
ft = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)   # the GELU's input
dgm, dbt = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
lens = (torch.randn(B, device=dev, generator=g) * 30 + 80).clamp(16, S).long()
am = (torch.rand(B, V, device=dev, generator=g) * lens[:, None]).long().to(torch.int16).view(torch.uint16)
grad = torch.randn(B, V, device=dev, generator=g) * 1e-2


def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fused(rep):
    return ops.sparse_head_bwd_dt_ln(grad, rep, am, E, B, S, V, False, None, x32, gamma, mean, rstd, ft, dgm, dbt)


def scatter(rep):
    dtn = ops.sparse_head_bwd_dt_scatter(grad, rep, am, E, B, S, V, False, None, T)
    dgt, _ = ops.layernorm_bwd(dtn, x32, gamma, mean, rstd, dgm, dbt)
    return ops.gelu_bwd(dgt, ft)


for dens in (1.0, 0.3, 0.1, 0.06, 0.03, 0.01, 0.003):
    rep = torch.rand(B, V, device=dev, generator=g) + 0.1
    rep = torch.where(torch.rand(B, V, device=dev, generator=g) < dens, rep, torch.zeros_like(rep))
    a, b = fused(rep), scatter(rep)
    err = float((a.float() - b.float()).abs().max() / a.float().abs().max())
    only = timeit(lambda: ops.sparse_head_bwd_dt_scatter(grad, rep, am, E, B, S, V, False, None, T))
    print(f"density {dens:6.3f}: fused matrix form {timeit(lambda: fused(rep)):7.0f} us | scatter chain {timeit(lambda: scatter(rep)):7.0f} us "
          f"(scatter + zero + convert alone {only:6.0f}) | max difference {err:.2e} of the largest element", flush=True)
