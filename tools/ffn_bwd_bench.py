"""Fused feed-forward BACKWARD (csrc/ffn_pc.hip ffn_pc_bwd_kernel) against the two launches it replaces -- the dF1 GEMM with the
tile-major GELU epilogue (weight-stationary kernel) and the GEMM fused with the LayerNorm-1 backward -- stand-alone, at the bench's
row counts (43 904 ragged / 65 536 dense), dropout on.   python tools/ffn_bwd_bench.py [rows ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from sparse_hip import lib, ops  # noqa: E402

H, I = 384, 1536


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    bf = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    w1 = torch.randn(I, H, device="cuda", generator=g) * 0.04
    w2 = torch.randn(H, I, device="cuda", generator=g) * 0.04
    flat = torch.cat([w1.reshape(-1), w2.reshape(-1)])
    shape = (1, I // 32, 24, 64, 8)
    w2tf, w1tf = torch.empty(shape, dtype=bf, device="cuda"), torch.empty(shape, dtype=bf, device="cuda")
    ops.ffn_pc_stage(flat[:I * H].view(I, H), flat[I * H:].view(H, I), 0, 1, None, None, w2tf, w1tf)
    w2T, w1T = w2.t().contiguous().to(bf), w1.t().contiguous().to(bf)
    for T in [int(a) for a in sys.argv[1:]] or [43904, 65536]:
        dy = (torch.randn(T, H, device="cuda", generator=g) * 0.5).to(bf)
        dres = (torch.randn(T, H, device="cuda", generator=g) * 0.5).to(bf)
        f1t = torch.randn(4 * ((T + 127) // 128), I // 32, 64, 16, device="cuda", generator=g).to(bf)
        z1 = torch.randn(T, H, device="cuda", generator=g) + 0.3
        gamma, beta = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
        _, _, m1, r1 = ops.layernorm_fwd_res32(z1, gamma, beta, 1e-12, bf, want_y32=False)
        drop = lib.dropout(0.1, 5, 6)
        dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
        ga = torch.empty(T, I, dtype=bf, device="cuda")
        fused = lambda: ops.ffn_pc_bwd(dy, dres, f1t, w2tf[0], w1tf[0], z1, gamma, m1, r1, drop, dg, db, want_drop=True)
        first = lambda: ops.gemm_nt(dy, w2T, gelu_grad_of=f1t, gelu_out=ga, gelu_grad_tiled=True)
        df1 = first()
        second = lambda: ops.gemm_nt_ln_bwd(df1, w1T, dres, z1, gamma, m1, r1, dg, db, drop, want_drop=True)
        assert fused() is not None and second() is not None
        tf, t1, t2 = timeit(fused), timeit(first), timeit(second)
        fl = 4.0 * T * H * I
        print(f"T = {T}: fused backward {tf:.1f} us ({fl / tf / 1e6:.0f} TFLOP/s); dF1 GEMM {t1:.1f} + GEMM/LayerNorm' {t2:.1f} = {t1 + t2:.1f} us "
              f"({fl / (t1 + t2) / 1e6:.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
