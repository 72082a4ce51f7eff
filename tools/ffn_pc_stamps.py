"""Shader-clock timeline of one producer / consumer pair of the fused feed-forward kernel (csrc/ffn_pc.hip built with -DPC_STAMPS into
a private library next to this script: the product library carries no stamps).  Run on the GPU box:
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPC_STAMPS -shared -o tools/_libpc_dbg.so <csrc>/ffn_pc.hip <csrc>/api.cpp
    python tools/ffn_pc_stamps.py [rows]"""
import ctypes as C, os, subprocess, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
csrc = os.path.join(root, "opensearch-sparse-model-tuning-sample_amd", "csrc")
# PC_DEFS="-DPC_X_NOGELU ...": timing experiments (the kernel's results are wrong with any of them; only the clock is read)
defs = os.environ.get("PC_DEFS", "").split()
so = os.path.join(root, "tools", "_libpc_dbg" + ("_alt" if os.environ.get("PC_SRC") else "") + "".join(d.replace("-D", "_") for d in defs) + ".so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DPC_STAMPS", *defs, "-I", csrc, "-I", os.path.join(root, "include"), "-shared", "-o", so,
                           os.environ.get("PC_SRC", os.path.join(csrc, "ffn_pc.hip")), os.path.join(csrc, "api.cpp")])
sys.path.insert(0, os.path.join(root, "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L, ops
dbg = C.CDLL(so)
H, I = 384, 1536
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
w1, w2 = rn(I, H, sc=0.03), rn(H, I, sc=0.03)
flat = torch.cat([w1.reshape(-1), w2.reshape(-1)])
shape = (1, I // 32, 24, 64, 8)
w1f, w2f = torch.empty(shape, dtype=torch.float16, device="cuda"), torch.empty(shape, dtype=torch.float16, device="cuda")
ops.ffn_pc_stage(flat[:I * H].view(I, H), flat[I * H:].view(H, I), 0, 1, w1f, w2f, None, None)
z1 = rn(T, H) + 0.1
g1, b1, g2, b2 = 1 + rn(H, sc=0.05), rn(H, sc=0.05), 1 + rn(H, sc=0.05), rn(H, sc=0.05)
bias1, bias2 = rn(I, sc=0.05), rn(H, sc=0.05)
bf = torch.bfloat16
x1, x2 = torch.empty(T, H, dtype=bf, device="cuda"), torch.empty(T, H, dtype=bf, device="cuda")
z2 = torch.empty(T, H, device="cuda"); f1 = torch.empty((T + 31) // 32 * 32, I, dtype=bf, device="cuda")
m1, r1, m2, r2 = (torch.empty(T, device="cuda") for _ in range(4))
P = lambda t: C.c_void_p(L.ptr(t))
args = [C.c_int(1), P(z1), P(g1), P(b1), C.c_float(1e-12), P(w1f), P(bias1), P(w2f), P(bias2), P(g2), P(b2), None, P(x1), P(m1), P(r1), P(f1),
        P(z2), P(x2), P(m2), P(r2), C.c_int(T), C.c_int(H), C.c_int(I), C.c_void_p(torch.cuda.current_stream().cuda_stream)]
for _ in range(3):
    assert dbg.sm_ffn_pc_fwd(*args) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dbg.sm_ffn_pc_fwd(*args)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 20
buf = (C.c_ulonglong * 256)()
assert dbg.sm_pc_debug_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(2, 128).astype(np.int64)
t0 = min(st[0][0], st[1][0])
k = lambda r, i: (st[r][i] - t0) / 1000.0
print("defs:", defs)
print(f"[{T} rows] kilo-cycles since the workgroup's start (block 3, pair 1)")
print(f"producer: LN prologue done {k(0,1):.1f}  first GEMM 1 done / P1 {k(0,2):.1f}  end {k(0,7):.1f}")
print(f"consumer: LN prologue done {k(1,1):.1f}  P1 {k(1,2):.1f}  loop done {k(1,3):.1f}  half 0 staged {k(1,4):.1f}  half 1 staged {k(1,5):.1f}  rows done {k(1,6):.1f}  end {k(1,7):.1f}")
for s in (1, 2, 10, 20, 30, 40, 46):
    print(f"  step {s:2d}: producer stream done {k(0, 10 + 2 * s):.2f} barrier passed {k(0, 11 + 2 * s):.2f} | consumer GEMM 2 done {k(1, 8 + 2 * s):.2f} barrier passed {k(1, 9 + 2 * s):.2f}")
print(f"  kernel: {us:.1f} us by events; the stamped workgroup ran {k(1,7):.1f} kilo-ticks -> the s_memtime tick is >= {k(1,7) / us:.2f} GHz for {(T + 127) // 128} workgroups")
print(f"  per step: {(k(1, 9 + 2 * 46) - k(1, 9 + 2 * 6)) / 40:.2f} kilo-cycles")
