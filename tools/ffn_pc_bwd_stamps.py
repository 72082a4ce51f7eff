"""Shader-clock timeline of one producer / consumer pair of the fused feed-forward BACKWARD (ffn_pc_bwd_kernel built with
-DPC_STAMPS into a private library; the product library carries no stamps).   python tools/ffn_pc_bwd_stamps.py [rows]"""
import ctypes as C, os, subprocess, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
csrc = os.path.join(root, "opensearch-sparse-model-tuning-sample_amd", "csrc")
defs = os.environ.get("PC_DEFS", "").split()
so = os.path.join(root, "tools", "_libpc_bdbg" + ("_alt" if os.environ.get("PC_SRC") else "") + "".join(d.replace("-D", "_") for d in defs) + ".so")
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DPC_STAMPS", *defs, "-I", csrc, "-I", os.path.join(root, "include"), "-shared", "-o", so,
                       os.environ.get("PC_SRC", os.path.join(csrc, "ffn_pc.hip")), os.path.join(csrc, "api.cpp")])
sys.path.insert(0, os.path.join(root, "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L, ops
dbg = C.CDLL(so)
H, I = 384, 1536
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
bf = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
w1, w2 = rn(I, H, sc=0.03), rn(H, I, sc=0.03)
flat = torch.cat([w1.reshape(-1), w2.reshape(-1)])
shape = (1, I // 32, 24, 64, 8)
w2tf, w1tf = torch.empty(shape, dtype=bf, device="cuda"), torch.empty(shape, dtype=bf, device="cuda")
ops.ffn_pc_stage(flat[:I * H].view(I, H), flat[I * H:].view(H, I), 0, 1, None, None, w2tf, w1tf)
Tp = (T + 127) // 128 * 128
dy, dres = rn(T, H, sc=0.5).to(bf), rn(T, H, sc=0.5).to(bf)
f1 = rn(Tp // 32, I // 32, 64, 16).to(bf)
z1 = rn(T, H) + 0.1
gamma = 1 + rn(H, sc=0.05)
_, _, m1, r1 = ops.layernorm_fwd_res32(z1, gamma, rn(H, sc=0.05), 1e-12, bf, want_y32=False)
df1, ga = torch.empty(Tp, I, dtype=bf, device="cuda"), torch.empty(Tp, I, dtype=bf, device="cuda")
dz1, dz1d = torch.empty(T, H, dtype=bf, device="cuda"), torch.empty(T, H, dtype=bf, device="cuda")
dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
drop = L.dropout(0.1, 5, 6)
P = lambda t: C.c_void_p(L.ptr(t))
args = [P(dy), P(dres), P(f1), P(w2tf), P(w1tf), P(z1), P(gamma), P(m1), P(r1), C.byref(drop), P(df1), P(ga), P(dz1), P(dz1d), P(dg), P(db),
        C.c_int(T), C.c_int(H), C.c_int(I), C.c_void_p(torch.cuda.current_stream().cuda_stream)]
for _ in range(3):
    assert dbg.sm_ffn_pc_bwd(*args) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dbg.sm_ffn_pc_bwd(*args)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 20
buf = (C.c_ulonglong * 256)()
assert dbg.sm_pc_debug_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(2, 128).astype(np.int64)
t0 = min(st[0][0], st[1][0])
k = lambda r, i: (st[r][i] - t0) / 1000.0
print(f"[{T} rows] kilo-cycles since the workgroup's start (block 3, pair 1); kernel {us:.1f} us by events")
print(f"producer: prologue done {k(0,1):.1f}  P1 {k(0,2):.1f}  last hand-over {k(0,3):.1f}  image ready {k(0,4):.1f}  rows done {k(0,5):.1f}  end {k(0,7):.1f}")
print(f"consumer: prologue done {k(1,1):.1f}  P1 {k(1,2):.1f}  loop done {k(1,3):.1f}  image written {k(1,4):.1f}  rows done {k(1,5):.1f}  end {k(1,7):.1f}")
for s in (1, 2, 10, 20, 30, 40, 46):
    print(f"  step {s:2d}: producer stream done {k(0, 10 + 2 * s):.2f} barrier passed {k(0, 11 + 2 * s):.2f} | consumer GEMM 2 done {k(1, 8 + 2 * s):.2f} barrier passed {k(1, 9 + 2 * s):.2f}")
print(f"  per step: {(k(1, 9 + 2 * 46) - k(1, 9 + 2 * 6)) / 40:.2f} kilo-cycles; the s_memtime tick is >= {k(1,7) / us:.2f} GHz for {(T + 127) // 128} workgroups")
