"""Shader-clock timeline of one workgroup of the 128 x 128 NT GEMM kernel (csrc/gemm.hip built with -DNT_STAMPS into a private library
next to this script: the product library carries no stamps).  python tools/gemm_nt_stamps.py [rows] [N]"""
import ctypes as C, os, subprocess, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
csrc = os.path.join(root, "opensearch-sparse-model-tuning-sample_amd", "csrc")
so = os.path.join(root, "tools", "_libnt_dbg.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DNT_STAMPS", "-shared", "-o", so,
                           *[os.path.join(csrc, f) for f in ("gemm.hip", "head_fwd.hip", "head_fwd_wide.hip", "api.cpp")]])
sys.path.insert(0, os.path.join(root, "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L
dbg = C.CDLL(so)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 43904
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1152
K = 384
x = torch.randn(T, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(N, device="cuda")
epi = L.SmEpilogue(L.ptr(bias), 0, None, L.dropout(), None, None, 0, 0, None, None, None, None, None, 0, None, None)
P = lambda t: C.c_void_p(L.ptr(t))
args = [C.c_int(L.SM_BF16), P(x), C.c_int(K), P(W), C.c_int(K), P(out), C.c_int(N), C.c_int(T), C.c_int(N), C.c_int(K), C.byref(epi),
        C.c_void_p(torch.cuda.current_stream().cuda_stream)]
for _ in range(3):
    assert dbg.sm_gemm_nt(*args) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dbg.sm_gemm_nt(*args)
e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
assert dbg.sm_nt_debug_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).astype(np.int64)
k = lambda i: (st[i] - st[0]) / 1000.0
print(f"[{T} x {N} x {K}] {e0.elapsed_time(e1) * 50:.1f} us per launch; kilo-cycles since the workgroup's start (block (1, 40), thread 0)")
print("  k-step barriers passed: " + " ".join(f"{k(4 + i):.2f}" for i in range(K // 32)))
print(f"  main loop done {k(1):.2f}  epilogue half 0 done {k(2):.2f}  half 1 done {k(3):.2f}")
for h in range(2):
    print(f"  pass {h}: staged {k(20 + 8 * h):.2f}  barrier passed {k(21 + 8 * h):.2f}  row groups stored " + " ".join(f"{k(22 + 8 * h + i):.2f}" for i in range(4)))
