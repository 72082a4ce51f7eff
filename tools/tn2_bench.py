"""The grouped weight-gradient kernel (csrc/gemm_tn2.hip) against the per-matrix kernel (gemm_tn_pc_kernel) on one encoder layer's
four products at the bench's token count: time per layer-set, TFLOP/s, and the relative error of each against the fp32 product."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as _L, ops


def _diag_lib():
    """the DIAGNOSTIC build of the grouped weight-gradient kernels (-DSM_TN_DIAG: timing-only switches and cycle stamps; the product
    library compiles them out), built next to this script on first use"""
    import subprocess
    root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
    csrc = os.path.join(root, "opensearch-sparse-model-tuning-sample_amd", "csrc")
    so = os.path.join(root, "tools", "_libtn_diag.so")
    if not os.path.exists(so):
        srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")] + [os.path.join(csrc, "api.cpp")]
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DSM_TN_DIAG", "-shared",
                               "-o", so, *srcs])
    return so


if os.environ.get("SM_TN2_DEBUG", "0") != "0":  # timing-only switches live in the diagnostic library
    _L._LIB_PATH = _diag_lib()


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for T in [int(t) for t in os.environ.get("T", "65536,43904").split(",")]:
    shapes = ((384, 1536), (1536, 384), (384, 384), (1152, 384))  # FFN down, FFN up, attention output, QKV (the backward's order)
    probs = []
    for N, Kc in shapes:
        A = torch.randn(T, N, device="cuda").bfloat16()
        B = torch.randn(T, Kc, device="cuda").bfloat16()
        probs.append((A, B, torch.zeros(N, Kc, device="cuda"), torch.zeros(N, device="cuda")))
    flops = sum(2.0 * T * N * Kc for N, Kc in shapes)
    us_old = timeit(lambda: [ops.gemm_tn_acc(a, b, o, colsum=c) for a, b, o, c in probs])
    us_new = timeit(lambda: ops.gemm_tn_group(probs))
    parts = []
    for i in range(len(probs)):
        us_i = timeit(lambda: ops.gemm_tn_group(probs[i:i + 1]))
        parts.append(f"{shapes[i][0]}x{shapes[i][1]} {us_i:.0f}us {2.0 * T * shapes[i][0] * shapes[i][1] / us_i / 1e6:.0f}TF")
    us_3 = timeit(lambda: ops.gemm_tn_group(probs[1:]))
    errs = []
    for a, b, o, c in probs:
        o.zero_(); c.zero_()
    ops.gemm_tn_group(probs)
    torch.cuda.synchronize()
    for a, b, o, c in probs:
        ref = a.float().t() @ b.float()
        errs.append(f"{float((o - ref).norm() / ref.norm()):.1e}/{float((c - a.float().sum(0)).abs().max() / a.float().sum(0).abs().max()):.1e}")
    print(f"T={T}: per-matrix kernel {us_old:.0f} us ({flops / us_old / 1e6:.0f} TF) | grouped {us_new:.0f} us ({flops / us_new / 1e6:.0f} TF) | "
          f"grouped without FFN-down {us_3:.0f} us | alone: " + ", ".join(parts) + " | rel err C/colsum " + " ".join(errs), flush=True)
