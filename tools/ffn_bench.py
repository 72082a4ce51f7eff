"""Fused feed-forward block (csrc/ffn_fused.hip) against the unfused kernel sequence it replaces, stand-alone, at the bench's
packed row count (43 904 rows) and at the dense one (65 536): forward = LayerNorm-1 + FFN-up (+GELU, pre-activation copy) +
FFN-down (+dropout, residual) + LayerNorm-2; backward = FFN-down input gradient (x GELU') + FFN-up input gradient fused with
the LayerNorm-1 backward.  Random operands (a zero-filled run reads 15-20 % high: the chip holds a higher clock).
    python tools/ffn_bench.py [rows ...]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L, ops

H, I = 384, 1536
bf = torch.bfloat16


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


def main():
    rows = [int(a) for a in sys.argv[1:]] or [43904, 65536]
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
    w1, w2 = rn(I, H, sc=0.03), rn(H, I, sc=0.03)
    flat = torch.cat([w1.reshape(-1), w2.reshape(-1)])
    drop = L.dropout(0.1, 7, 3)
    for f16 in (True, False):
        op = torch.float16 if f16 else bf
        w1h = torch.empty((1, I, H), dtype=op, device="cuda")
        w2p = torch.empty((1, I // 32, H, 32), dtype=op, device="cuda")
        w1tp = torch.empty((1, I // 32, H, 32), dtype=bf, device="cuda")
        ops.ffn_stage(flat[:I * H].view(I, H), flat[I * H:].view(H, I), 0, 1, w1h, w2p, w1tp)
        w1b, w2b, w1t, w2t = w1.to(bf), w2.to(bf), w1.t().contiguous().to(bf), w2.t().contiguous().to(bf)
        for T in rows:
            z1 = rn(T, H) + 0.1
            g1, b1, g2, b2 = 1 + rn(H, sc=0.05), rn(H, sc=0.05), 1 + rn(H, sc=0.05), rn(H, sc=0.05)
            bias1, bias2 = rn(I, sc=0.05), rn(H, sc=0.05)
            fused = lambda: ops.ffn_fwd(z1, g1, b1, 1e-12, w1h[0], bias1, w2p[0], bias2, g2, b2, drop, save_f1=True)
            f1buf = torch.empty(T, I, dtype=bf, device="cuda")

            def unfused():
                x1, _, m1, r1 = ops.layernorm_fwd_res32(z1, g1, b1, 1e-12, bf, want_y32=False)
                ga = ops.gemm_nt(x1, w1b, bias=bias1, act=1, preact=f1buf)
                z2 = ops.gemm_nt(ga, w2b, bias=bias2, drop=drop, residual=z1, out_f32=True, residual_ln=(m1, r1, g1, b1))
                return ops.layernorm_fwd_res32(z2, g2, b2, 1e-12, bf, want_y32=False)
            shape = (1, I // 32, 24, 64, 8)
            w1f, w2f = torch.empty(shape, dtype=op, device="cuda"), torch.empty(shape, dtype=op, device="cuda")
            ops.ffn_pc_stage(flat[:I * H].view(I, H), flat[I * H:].view(H, I), 0, 1, w1f, w2f, None, None)
            pc = lambda: ops.ffn_pc_fwd(z1, g1, b1, 1e-12, w1f[0], bias1, w2f[0], bias2, g2, b2, drop, save_f1=True)
            tf, tu, tp = timeit(fused), timeit(unfused), timeit(pc)
            flop = 4.0 * T * H * I
            print(f"forward  {'f16' if f16 else 'bf16'} operands, {T} rows: producer/consumer {tp:7.1f} us ({flop / tp / 1e6:5.0f} TFLOP/s)   "
                  f"16-token fused {tf:7.1f} us ({flop / tf / 1e6:5.0f} TFLOP/s)   "
                  f"unfused sequence {tu:7.1f} us ({flop / tu / 1e6:5.0f} TFLOP/s)", flush=True)
            if not f16:
                continue
            x1, m1, r1, f1, z2, x2, m2, r2 = fused()
            dy, dres = rn(T, H, sc=0.01).to(bf), rn(T, H, sc=0.01).to(bf)
            dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
            fb = lambda: ops.ffn_bwd(dy, dres, f1, w2t, w1tp[0], z1, g1, m1, r1, drop, dg, db, want_drop=True)

            def ub():
                df1 = ops.gemm_nt(dy, w2t, gelu_grad_of=f1)
                return ops.gemm_nt_ln_bwd(df1, w1t, dres, z1, g1, m1, r1, dg, db, drop, want_drop=True)
            tf, tu = timeit(fb), timeit(ub)
            print(f"backward bf16 operands, {T} rows: fused {tf:7.1f} us ({flop / tf / 1e6:5.0f} TFLOP/s)   "
                  f"unfused sequence {tu:7.1f} us ({flop / tu / 1e6:5.0f} TFLOP/s)  [fused also writes gelu(f1)]", flush=True)


if __name__ == "__main__":
    main()
