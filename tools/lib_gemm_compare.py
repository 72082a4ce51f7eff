"""Reference point for the hand-written GEMMs: the same products through torch.matmul (hipBLASLt / rocBLAS picks the kernel), stand-alone,
at the bench's row counts.  The library calls are PLAIN products with a bf16 result; ours carry their epilogues (fp32 accumulation into the
gradient buffer + bias-gradient column sums for the weight gradients; bias for the forward).   python3 tools/lib_gemm_compare.py [T]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import ops


def timeit(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for T in ([int(sys.argv[1])] if len(sys.argv) > 1 else [65536, 43904]):
    print(f"--- T = {T} token rows, bf16 ---")
    for name, M, N in (("dW qkv   [1152 x 384]", 1152, 384), ("dW o     [ 384 x 384]", 384, 384), ("dW ffn1  [1536 x 384]", 1536, 384), ("dW ffn2  [ 384 x 1536]", 384, 1536)):
        a = torch.randn(T, M, device="cuda").bfloat16(); b = torch.randn(T, N, device="cuda").bfloat16()
        out = torch.zeros(M, N, device="cuda"); cs = torch.zeros(M, device="cuda")
        mine = timeit(lambda: ops.gemm_tn_acc(a, b, out, colsum=cs))
        at = a.t()
        lib = timeit(lambda: torch.matmul(at, b))
        fl = 2.0 * T * M * N
        print(f"{name}: ours {mine:7.1f} us ({fl/mine/1e6:6.0f} TFLOP/s, fp32 accumulate + column sums)   library {lib:7.1f} us ({fl/lib/1e6:6.0f} TFLOP/s, plain)")
    x = torch.randn(T, 384, device="cuda").bfloat16()
    for name, N in (("qkv      [T x 384] . [1152 x 384]^T", 1152), ("ffn up   [T x 384] . [1536 x 384]^T", 1536), ("attn out [T x 384] . [ 384 x 384]^T", 384)):
        W = (torch.randn(N, 384, device="cuda") * 0.02).bfloat16(); bias = torch.zeros(N, device="cuda")
        mine = timeit(lambda: ops.gemm_nt(x, W, bias=bias))
        wt = W.t()
        lib = timeit(lambda: torch.matmul(x, wt))
        fl = 2.0 * T * N * 384
        print(f"{name}: ours {mine:7.1f} us ({fl/mine/1e6:6.0f} TFLOP/s, + bias)   library {lib:7.1f} us ({fl/lib/1e6:6.0f} TFLOP/s, plain)")
    y = torch.randn(T, 1536, device="cuda").bfloat16(); W2 = (torch.randn(384, 1536, device="cuda") * 0.02).bfloat16()
    mine = timeit(lambda: ops.gemm_nt(y, W2)); w2t = W2.t(); lib = timeit(lambda: torch.matmul(y, w2t)); fl = 2.0 * T * 384 * 1536
    print(f"ffn down [T x 1536] . [384 x 1536]^T: ours {mine:7.1f} us ({fl/mine/1e6:6.0f} TFLOP/s)   library {lib:7.1f} us ({fl/lib/1e6:6.0f} TFLOP/s)")
