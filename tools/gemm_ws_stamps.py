"""Shader-clock timeline of one workgroup of the weight-stationary GEMM (csrc/gemm_ws.hip built with -DWS_STAMPS into a private library).
    python tools/gemm_ws_stamps.py [rows] [N]"""
import ctypes as C, glob, os, subprocess, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
csrc = os.path.join(root, "opensearch-sparse-model-tuning-sample_amd", "csrc")
so = os.path.join(root, "tools", "_libws_dbg.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DWS_STAMPS", "-shared", "-o", so,
                           *sorted(glob.glob(os.path.join(csrc, "*.hip"))), os.path.join(csrc, "api.cpp")])
sys.path.insert(0, os.path.join(root, "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L
dbg = C.CDLL(so)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 43904
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1152
x = torch.randn(T, 384, device="cuda").bfloat16()
W = (torch.randn(N, 384, device="cuda") * 0.02).bfloat16()
out = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(N, device="cuda")
epi = L.SmEpilogue(L.ptr(bias), 0, None, L.dropout(), None, None, 0, 0, None, None, None, None, None, 0, None, None)
P = lambda t: C.c_void_p(L.ptr(t))
args = [C.c_int(L.SM_BF16), P(x), C.c_int(384), P(W), C.c_int(384), P(out), C.c_int(N), C.c_int(T), C.c_int(N), C.c_int(384), C.byref(epi),
        C.c_void_p(torch.cuda.current_stream().cuda_stream)]
for _ in range(3):
    assert dbg.sm_gemm_nt(*args) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dbg.sm_gemm_nt(*args)
e1.record(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
assert dbg.sm_ws_debug_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(4, 128).astype(np.int64)
t0 = st[0][0]
k = lambda r, i: (st[r][i] - t0) / 1000.0 if st[r][i] > 0 else float("nan")   # (steps this workgroup did not have print nan)
print(f"[{T} x {N} x 384] {e0.elapsed_time(e1) * 50:.1f} us per launch; kilo-cycles since compute wave 0 reached B_0 (workgroup 8)")
print("  compute wave 0: B_0 passed %.2f;" % k(0, 1), "arrives / passes B_s: " + " ".join(f"{k(0, 2 * s):.2f}/{k(0, 2 * s + 1):.2f}" for s in (1, 2, 3, 10, 20, 30, 40)))
print("  loader 0: landed / barrier passed / issued, steps 0 1 2 10 20 30: " + " ".join(f"{k(1, 3 * s):.2f}/{k(1, 3 * s + 1):.2f}/{k(1, 3 * s + 2):.2f}" for s in (0, 1, 2, 10, 20, 30)))
print("  storer 0: barrier passed / tile stored, t = 2 3 10 20 30: " + " ".join(f"{k(2, 2 * t):.2f}/{k(2, 2 * t + 1):.2f}" for t in (2, 3, 10, 20, 30)))
print(f"  per step (B_10 -> B_40): {(k(0, 81) - k(0, 21)) / 30:.3f} kilo-cycles")
