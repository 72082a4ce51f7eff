"""Host-side AddressSanitizer run of the launchers (GPU ASan is not available on this pool): csrc is built with
`-fsanitize=address -fno-gpu-sanitize` (make asan) and a child process, with the ASan runtime preloaded, drives one realistic call
of every entry-point family through the ctypes binding WITHOUT a GPU -- the host code of a launcher (argument checks, split / grid
planning, the by-pointer structs sm_epilogue / sm_dropout / sm_ragged as ctypes lays them out) runs to the launch, which fails
with "no device".  A ctypes structure shorter than the C one, or a planner that indexes out of bounds, is an ASan report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")
ASAN_SO = os.path.join(ROOT, "build", "asan", "libsparse_hip_asan.so")

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
from sparse_hip import lib as L
L._LIB_PATH = sys.argv[2]
lib = L.load()
P = lambda k: C.c_void_p(0x7f0000000000 + 0x1000000 * k)   # device addresses: never dereferenced on the host
T, H, I, V, B, S, A = 65536, 384, 1536, 30522, 512, 128, 12
drop = L.dropout(0.1, 1234, 5)
nodrop = L.dropout()
rag = L.SmRagged(P(40).value, P(41).value, P(42).value, 43904)
calls = 0
def run(name, *args, expect_launch=True):
    global calls
    rc = getattr(lib, name)(*args)
    msg = lib.sm_last_error().decode()
    calls += 1
    # no GPU: either an argument check (SM_ERR_INVALID) or a HIP error out of the attribute / launch call -- never success
    assert rc != 0, (name, rc)
    print(name, rc, msg[:90])
epi = L.SmEpilogue(P(1).value, 1, P(2).value, drop, P(3).value, None, 1, 1, P(4).value, P(5).value, P(6).value, P(7).value, None, 0, None, None)
run("sm_gemm_nt", L.SM_BF16, P(10), H, P(11), H, P(12), H, T, H, H, C.byref(epi), None)
# the attention-output projection of the fp32 residual stream: weight-stationary kernel, fp32-residual epilogue (gemm_ws.hip EPI 2)
epi_o = L.SmEpilogue(P(1).value, 0, None, drop, P(3).value, None, 1, 1, P(4).value, P(5).value, P(6).value, P(7).value, None, 0, None, None)
run("sm_gemm_nt", L.SM_BF16, P(10), H, P(11), H, P(12), H, T, H, H, C.byref(epi_o), None)
epi2 = L.SmEpilogue(None, 0, None, nodrop, None, P(3).value, 0, 0, None, None, None, None, P(4).value, 1, None, None)
run("sm_gemm_nt", L.SM_BF16, P(10), H, P(11), H, P(12), I, T, I, H, C.byref(epi2), None)       # weight-stationary, dF1 epilogue
run("sm_gemm_nt", L.SM_F32, P(10), 72, P(11), 72, P(12), 136, 300, 136, 64, None, None)
# fp8 operands, the epilogue emits the next GEMM's fp8 operand and no 16-bit C (ABI 6: the q8 fields at the end of sm_epilogue)
epi3 = L.SmEpilogue(P(1).value, 1, P(2).value, nodrop, None, None, 0, 0, None, None, None, None, None, 0, P(3).value, P(4).value,
                    P(5).value, P(6).value, P(7).value, P(8).value, 0, P(9).value)
assert lib.sm_gemm_nt_q8_partials(T, 3072) == (T // 128) * 24 * 4
run("sm_gemm_nt", L.SM_FP8, P(10), 768, P(11), 768, None, 3072, T, 3072, 768, C.byref(epi3), None)
run("sm_gemm_nt_ln_bwd", L.SM_BF16, P(10), I, P(11), I, T, H, I, P(12), P(13), P(14), P(15), P(16), C.byref(drop), P(17), P(18), P(19), P(20), 1, None, None)
run("sm_gemm_tn_acc", L.SM_BF16, P(10), H, P(11), I, P(12), I, T, H, I, P(13), None)
run("sm_gemm_tn_acc", L.SM_F32, P(10), 72, P(11), 136, P(12), 136, 300, 72, 136, None, None)
run("sm_gemm_tn_acc_bcm", P(10), 1, P(11), 0, P(12), H, 43904, I, H, P(13), None)
# the grouped weight gradients of a layer (both kernels: every N a multiple of 384 -> symmetric; N = 576 -> [192 x 192] tiles), a BCM operand
grp = (L.SmTnProblem * 4)(L.SmTnProblem(P(10).value, H, 0, P(11).value, 0, 1, P(12).value, I, H, I, P(13).value),
                          L.SmTnProblem(P(14).value, 0, 1, P(15).value, H, 0, P(16).value, H, I, H, P(17).value),
                          L.SmTnProblem(P(18).value, H, 0, P(19).value, H, 0, P(20).value, H, H, H, None),
                          L.SmTnProblem(P(21).value, 3 * H + 64, 0, P(22).value, H, 0, P(23).value, H, 3 * H, H, P(24).value))
run("sm_gemm_tn_group", 4, grp, T, None)
one = (L.SmTnProblem * 1)(L.SmTnProblem(P(10).value, 576, 0, P(11).value, 192, 0, P(12).value, 192, 576, 192, None))
run("sm_gemm_tn_group", 1, one, 43904 + 16, None)
run("sm_ffn_pc_stage", 1, P(10), P(11), 1000000, 6, H, I, P(12), P(13), P(14), P(15), None)
run("sm_ffn_pc_fwd", 1, P(10), P(11), P(12), 1e-12, P(13), P(14), P(15), P(16), P(17), P(18), C.byref(drop), P(19), P(20), P(21), P(22), P(23), P(24), P(25), P(26), 43904, H, I, None)
run("sm_ffn_pc_bwd", P(10), P(11), P(12), P(13), P(14), P(15), P(16), P(17), P(18), C.byref(drop), P(19), P(20), P(21), P(22), P(23), P(24), 43904, H, I, None)
run("sm_layernorm_fwd_res32", L.SM_BF16, P(10), P(11), P(12), P(13), None, P(14), P(15), T, H, 1e-12, P(16), None)
run("sm_layernorm_bwd_res32", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), P(15), P(16), C.byref(drop), P(17), P(18), T, H, None)
run("sm_embed_fwd_res32", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), P(15), P(16), P(17), P(18), P(19), P(20), 43904, 1, H, 1e-12, C.byref(drop), C.byref(rag), None)
run("sm_embed_bwd_sorted", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), 40000, P(15), P(16), P(17), H, None)
run("sm_attention_fwd", L.SM_BF16, P(10), P(11), P(12), P(13), B, S, A, 32, C.byref(drop), C.byref(rag), None)
run("sm_attention_bwd", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), P(15), B, S, A, 32, C.byref(drop), None, None)
run("sm_sparse_head_fwd", L.SM_F16, P(10), P(11), P(12), P(13), P(14), P(15), B, S, H, V, 0, C.byref(rag), None, None)
run("sm_sparse_head_fwd", L.SM_F32, P(10), P(11), P(12), P(13), P(14), P(15), B, S, H, V, 1, None, None, None)
run("sm_sparse_head_bwd", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), P(15), P(16), P(17), B, S, H, V, 0, C.byref(rag), None)
run("sm_sparse_head_bwd_dt_ln", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), B, S, H, V, 0, None, P(15), P(16), P(17), P(18), P(19), P(20), P(21), 1, None, 0, None)
# ... with the workspace: 342 row tiles -> the last 86 split along the vocabulary (the planner's arithmetic runs on the host)
run("sm_sparse_head_bwd_dt_ln", L.SM_BF16, P(10), P(11), P(12), P(13), P(14), B, S, H, V, 0, None, P(15), P(16), P(17), P(18), P(19), P(20), P(21), 1, P(22),
    lib.sm_sparse_head_bwd_dt_ws_bytes(), None)
run("sm_flops_fwd", P(10), B, 16, V, 150, P(11), P(12), P(13), None)
run("sm_scores_fwd", P(10), P(11), 32, B, V, 0, P(12), None)
run("sm_scores_fwd", P(10), P(11), 32, B, V, 2, P(12), None)   # deterministic form
run("sm_row_compact", P(10), 32, V, 32, P(11), P(12), P(13), P(14), None)
run("sm_infonce_fwd_bwd", P(10), 32, B, 16, 0, P(11), P(12), None)
run("sm_adamw", P(10), P(11), P(12), P(13), 22700000, 2e-5, 0.9, 0.999, 1e-8, 0.01, 7, 1.0, None)
run("sm_cast_weights_multi", L.SM_BF16, P(10), 26, 5000, None)
run("sm_amax", L.SM_BF16, P(10), 1000000, P(11), None)
run("sm_gelu_quantize_fp8", P(10), P(11), 1000000, 1, P(12), P(10), P(13), P(14), P(15), None)   # ABI 8, backward form, in place
run("sm_gelu_quantize_fp8", P(10), None, 1000000, 0, P(12), None, P(13), P(14), P(15), None)     # forward form without the 16-bit copy
# fp8 operands at K = 768 through the weight-stationary launcher (plain epilogue; the fp32-residual one)
epi8 = L.SmEpilogue(P(1).value, 0, None, nodrop, None, None, 0, 0, None, None, None, None, None, 0, P(3).value, P(4).value)
run("sm_gemm_nt", L.SM_FP8, P(10), 768, P(11), 768, P(12), 2304, T, 2304, 768, C.byref(epi8), None)
epi8r = L.SmEpilogue(P(1).value, 0, None, nodrop, P(2).value, None, 1, 1, None, None, None, None, None, 0, P(3).value, P(4).value)
run("sm_gemm_nt", L.SM_FP8_GRAD, P(10), 768, P(11), 768, P(12), 768, T, 768, 768, C.byref(epi8r), None)
ptrs = (C.c_void_p * 4)(P(10), P(11), P(12), P(13)); ws = (C.c_float * 4)(1, 1, 1, 1)
run("sm_loss_combine", ptrs, ws, 4, P(14), 0.05, None, 0.0, P(15), P(16), P(17), 0.01, None)
assert lib.sm_sparse_head_fwd_scratch_bytes(L.SM_F32, B, S, H, V, 1) > 0
print("ASAN_CHILD_OK", calls)
"""


def _asan_runtime():
    for cc in ("/opt/rocm/lib/llvm/bin/clang", "clang"):
        try:
            out = subprocess.run([cc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
            if os.path.isabs(out) and os.path.exists(out):
                return out
        except OSError:
            continue
    return None


def test_host_code_of_every_entry_point_family_under_address_sanitizer(tmp_path):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no ASan runtime next to the ROCm clang")
    subprocess.check_call(["make", "-j6", "-C", os.path.join(PKG, "csrc"), "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, str(script), PKG, ASAN_SO], capture_output=True, text=True, env=env, timeout=600)
    assert "AddressSanitizer" not in r.stderr and "AddressSanitizer" not in r.stdout, r.stderr[-4000:]
    assert r.returncode == 0 and "ASAN_CHILD_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
