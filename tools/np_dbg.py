"""Per-wave cycle counters of the persistent NT GEMM (SM_NP_DBG=1)."""
import sys, os, ctypes, torch, numpy as np
os.environ["SM_NP_DBG"] = "1"
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops, lib
T = 65536
for N, K in ((1536, 384), (384, 384), (384, 1536)):
    A = torch.randn(T, K, device='cuda').bfloat16(); W = torch.randn(N, K, device='cuda').bfloat16() * 0.02
    for _ in range(3): ops.gemm_nt(A, W)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    L = lib.load()
    L.sm_debug_np_counters.argtypes = [ctypes.c_void_p]
    rc = L.sm_debug_np_counters(buf.ctypes.data)
    d = buf.reshape(256, 8, 8).astype(np.float64)
    cons, load = d[:, :4, :], d[:, 4:, :]
    print(f"N={N} K={K} rc={rc} stages/WG={cons[:, 0, 4].mean():.0f}")
    print("  consumer (cycles): total %.0f  barrier-wait %.0f  mainloop %.0f  epilogue %.0f" % tuple(cons[:, :, k].mean() for k in range(4)))
    print("  loader   (cycles): total %.0f  vmcnt-wait %.0f  barrier-wait %.0f  issue %.0f" % tuple(load[:, :, k].mean() for k in range(4)))
