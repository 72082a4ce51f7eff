"""Soak of the grouped weight-gradient kernels (hand-scheduled assembly, hand-counted waits): the same launches over and over beside a
perturbing stream, every result compared with a float64 product -- a stale LDS slot, a fragment read too early or a lost atomic shows
up as an error far above the fp32 summation noise (~1e-6 relative).  python tools/tn_soak.py [iterations]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(0)
shapes = ((384, 1536), (1536, 384), (384, 384), (1152, 384))
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
worst, bad, t0 = 0.0, 0, time.time()
cases = {}
for it in range(iters):
    T = (65536, 43904, 8208, 1000, 32, 70000)[it % 6]
    if T not in cases:
        probs, refs = [], []
        for N, Kc in shapes:
            A = torch.randn(T, N, device="cuda").bfloat16()
            B = torch.randn(T, Kc, device="cuda").bfloat16()
            probs.append((A, B, torch.zeros(N, Kc, device="cuda"), torch.zeros(N, device="cuda")))
            refs.append(((A.double().t() @ B.double()).float(), A.double().sum(0).float()))
        cases[T] = (probs, refs)
    probs, refs = cases[T]
    for _, _, o, c in probs:
        o.zero_(); c.zero_()
    with torch.cuda.stream(side):  # perturbation: a neighbour that takes CUs away at varying times
        for _ in range(it % 4):
            noise_a @ noise_a
    assert ops.gemm_tn_group(probs)
    torch.cuda.synchronize()
    for (_, _, o, c), (ro, rc) in zip(probs, refs):
        e = float((o - ro).norm() / ro.norm())
        e2 = float((c - rc).abs().max() / rc.abs().max())
        worst = max(worst, e, e2)
        if e > 2e-5 or e2 > 2e-5 or not bool(torch.isfinite(o).all()):
            bad += 1
            print(f"iteration {it} T={T}: relative error {e:.3e} / colsum {e2:.3e}", flush=True)
print(f"{iters} iterations ({time.time() - t0:.0f} s), SM_TN_SYM={os.environ.get('SM_TN_SYM', '1')}: {bad} bad results, worst relative error {worst:.2e}")
