"""Where a stage of gemm_tn3_kernel spends its cycles: the diagnostic build (SM_TN2_DEBUG=18: stamps, no flush) records per wave the
shader cycles of its main loop and of the two waits of every stage (own LDS-DMA landed; barrier)."""
import ctypes, os, sys
os.environ["SM_TN2_DEBUG"] = os.environ.get("SM_TN2_DEBUG", "18")
import torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops, lib


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


lib._LIB_PATH = _diag_lib()

T = int(os.environ.get("T", "65536"))
shapes = ((384, 1536), (1536, 384), (384, 384), (1152, 384))
probs = []
for N, Kc in shapes:
    probs.append((torch.randn(T, N, device="cuda").bfloat16(), torch.randn(T, Kc, device="cuda").bfloat16(),
                  torch.zeros(N, Kc, device="cuda"), torch.zeros(N, device="cuda")))
for _ in range(5):
    ops.gemm_tn_group(probs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_tn_group(probs); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
n = 240 * 9 * 4
buf = (ctypes.c_uint32 * n)()
so = lib.load()
so.sm_tn3_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert so.sm_tn3_debug_stamps(buf, n) == 0
import numpy as np
a = np.array(buf, dtype=np.int64).reshape(240, 9, 4)[:, :8, :3]
stages = -(-((T + 9) // 10) // 32)
print(f"launch {us:.0f} us, ~{stages} stages per workgroup; clock if the loop is the launch: {a[:, :, 0].mean() / us / 1e3:.2f} GHz")
for name, sl in (("waves 0-3 (LDS-DMA in the first half)", slice(0, 4)), ("waves 4-7 (LDS-DMA in the second half)", slice(4, 8))):
    x = a[:, sl, :].reshape(-1, 3)
    print(f"{name}: loop {x[:, 0].mean():.0f} cycles = {x[:, 0].mean() / stages:.0f} per stage (MFMA floor 1152 per SIMD); "
          f"vmcnt wait {x[:, 1].mean() / stages:.0f}, barrier {x[:, 2].mean() / stages:.0f} per stage; "
          f"loop min / max over waves {x[:, 0].min()} / {x[:, 0].max()}")
