"""The fused feed-forward block (csrc/ffn_pc.hip, hf:334-351 + the LayerNorm in front of it) against a plain torch fp32
restatement of the same ops on the same seeded inputs, forward and backward, through the C ABI.

The reference computes in fp32 from the values the kernel multiplies (weights rounded to the operand type: that is a property of
the staged copies, checked separately by test_stage_layouts); the kernel's own rounding of x1 / gelu(f1) to the operand type and
the bf16 storage of f1 / x2 are what the tolerances cover: 1e-2 of the tensor's scale (north star, bf16), 4e-3 with fp16 operands.
Shapes: token counts that end inside a workgroup / inside a wave's 16 rows are not possible (T % 16 == 0 by construction of
the ragged layout); a partial workgroup (T = 16 .. 112 mod 128) and several workgroups are covered."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

H, I = 384, 1536


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from sparse_hip import lib, ops as _ops
    lib.load()
    return _ops


def _rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _weights(L=1, seed=0, inter=I):
    w1 = _rnd(L, inter, H, seed=seed + 1, scale=0.04)
    w2 = _rnd(L, H, inter, seed=seed + 2, scale=0.04)
    return w1, w2


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _ln(x, g, b, eps=1e-12):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g + b, mu.squeeze(-1), torch.rsqrt(var + eps).squeeze(-1)


def _reference_forward(z1, g1, b1, w1, bias1, w2, bias2, g2, b2):
    x1, m1, r1 = _ln(z1, g1, b1)
    f1 = x1 @ w1.t() + bias1
    z2 = _gelu(f1) @ w2.t() + bias2 + x1
    x2, m2, r2 = _ln(z2, g2, b2)
    return x1, m1, r1, f1, z2, x2, m2, r2


def _close(got, want, tol, what):
    got, want = got.float(), want.float()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert torch.isfinite(got).all(), what
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol} * {scale:.3e}"
    return err / scale


def test_encoder_layer_fused_matches_unfused(monkeypatch):
    """the whole encoder with the fused block against the same model on the unfused kernels (bf16 operands in both, so that
    only the fusion differs): sparse activations and parameter gradients"""
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=3000, hidden_size=384, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1536,
                         max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1000, 3000, (12, 64), generator=g)
    mask = torch.ones(12, 64, dtype=torch.long)
    mask[3, 40:] = 0
    mask[7, 17:] = 0
    up = torch.randn(12, 3000, generator=g).cuda() * 1e-2
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SM_PC_FFN", mode)
        monkeypatch.setenv("SM_FFN_F16", "0")
        bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=3)
        assert bb.pc_ffn == (mode == "1")
        bb.train()
        rep = bb.encode(ids.cuda(), mask.cuda())
        (rep * up).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (rep.detach().clone(), bb.flat_grad.clone())
    rep_f, grad_f = res["1"]
    rep_u, grad_u = res["0"]
    err = float(((rep_f - rep_u).abs() / (1 + rep_u.abs())).max())
    rel = float((grad_f - grad_u).norm() / grad_u.norm())
    print(f"[fused vs unfused encoder] worst sparse activation {err:.2e} x (1+|ref|), flat gradient rel Frobenius {rel:.2e}")
    # two bf16 paths against each other: rounding re-routes near-tied arg-max positions of the head, which moves whole rows of the
    # tied embedding gradient (the bound of the un-routed oracle comparison); the kernels themselves are held to 1e-2 above
    assert err <= 5e-3 and rel <= 1.2e-1


# ------------------------------------------------------------------ producer / consumer form (csrc/ffn_pc.hip)
def _kp(s, kg, j):
    return 16 * s + (j & 3) + 8 * (j >> 2) + 4 * kg


def _stage_pc(ops, w1, w2, op_dtype):
    L, inter, _ = w1.shape
    per = inter * H
    stride = 2 * per + 64
    flat = torch.empty(L * stride, device="cuda")
    for l in range(L):
        flat[l * stride: l * stride + per] = w1[l].reshape(-1)
        flat[l * stride + per: l * stride + 2 * per] = w2[l].reshape(-1)
    shape = (L, inter // 32, 24, 64, 8)
    w1f, w2f = torch.empty(shape, dtype=op_dtype, device="cuda"), torch.empty(shape, dtype=op_dtype, device="cuda")
    w2tf, w1tf = torch.empty(shape, dtype=torch.bfloat16, device="cuda"), torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    ops.ffn_pc_stage(flat[:per].view(inter, H), flat[per:2 * per].view(H, inter), stride, L, w1f, w2f, w2tf, w1tf)
    return w1f, w2f, w2tf, w1tf


def _f1_rows(f1t, T):
    """tile-major f1 [G][NC][64][16] -> row-major [T, I]: lane (hh, tok), register r <-> chunk row (r & 3) + 8 (r >> 2) + 4 hh"""
    G, NC = f1t.shape[:2]
    r = torch.arange(16, device=f1t.device)
    lane = torch.arange(64, device=f1t.device)
    rowidx = (r[None, :] & 3) + 8 * (r[None, :] >> 2) + 4 * (lane[:, None] >> 5)        # [64, 16]
    tok = (lane & 31)[:, None].expand(64, 16)
    out = torch.zeros(G * 32, NC * 32, dtype=f1t.dtype, device=f1t.device)
    for c in range(NC):
        for g in range(G):
            out[g * 32 + tok, c * 32 + rowidx] = f1t[g, c]
    return out[:T]


@pytest.mark.parametrize("op_dtype", [torch.float16, torch.bfloat16])
def test_pc_stage_layouts(ops, op_dtype):
    w1, w2 = _weights(L=2, seed=5, inter=128)
    w1f, w2f, w2tf, w1tf = _stage_pc(ops, w1, w2, op_dtype)
    torch.cuda.synchronize()
    bf = torch.bfloat16
    lane = torch.arange(64, device="cuda")
    r, kg = lane & 31, lane >> 5
    j = torch.arange(8, device="cuda")
    for l in range(2):
        for c in (0, 3):
            for piece in (0, 5, 23):
                k = 16 * piece + 8 * kg[:, None] + j[None, :]                              # [64, 8]
                assert torch.equal(w1f[l, c, piece], w1[l][(32 * c + r)[:, None], k].to(op_dtype))
                assert torch.equal(w2tf[l, c, piece], w2[l][k, (32 * c + r)[:, None]].to(bf))
                n, s_ = piece >> 1, piece & 1
                kp = 32 * c + 16 * s_ + (j[None, :] & 3) + 8 * (j[None, :] >> 2) + 4 * kg[:, None]
                assert torch.equal(w2f[l, c, piece], w2[l][(32 * n + r)[:, None], kp].to(op_dtype))
                assert torch.equal(w1tf[l, c, piece], w1[l][kp, (32 * n + r)[:, None]].to(bf))


@pytest.mark.parametrize("op_dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("T", [16, 112, 128, 1008, 4096 + 48])
def test_ffn_pc_forward_matches_torch(ops, op_dtype, tol, T):
    w1, w2 = _weights(seed=T)
    w1f, w2f, _, _ = _stage_pc(ops, w1, w2, op_dtype)
    z1 = _rnd(T, H, seed=11, scale=1.5) + 0.3
    g1, b1 = 1 + _rnd(H, seed=12, scale=0.1), _rnd(H, seed=13, scale=0.1)
    g2, b2 = 1 + _rnd(H, seed=14, scale=0.1), _rnd(H, seed=15, scale=0.1)
    bias1, bias2 = _rnd(I, seed=16, scale=0.1), _rnd(H, seed=17, scale=0.1)
    out = ops.ffn_pc_fwd(z1, g1, b1, 1e-12, w1f[0], bias1, w2f[0], bias2, g2, b2, None, save_f1=True)
    assert out is not None
    torch.cuda.synchronize()
    want = _reference_forward(z1, g1, b1, w1[0].to(op_dtype).float(), bias1, w2[0].to(op_dtype).float(), bias2, g2, b2)
    names = ("x1", "m1", "r1", "f1", "z2", "x2", "m2", "r2")
    out = list(out)
    out[3] = _f1_rows(out[3], T)
    errs = {n: _close(g_, w_, 1e-2 if n in ("x1", "f1", "x2") else tol, n) for n, g_, w_ in zip(names, out, want)}
    print(f"[ffn pc fwd {op_dtype} T={T}] " + " ".join(f"{n} {e:.1e}" for n, e in errs.items()))
    out2 = ops.ffn_pc_fwd(z1, g1, b1, 1e-12, w1f[0], bias1, w2f[0], bias2, g2, b2, None, save_f1=False)
    assert out2[3] is None and torch.equal(out2[4], out[4]) and torch.equal(out2[5], out[5])
    # repeated launches agree bit for bit (hand-over / ring races would show up here)
    for _ in range(3):
        again = list(ops.ffn_pc_fwd(z1, g1, b1, 1e-12, w1f[0], bias1, w2f[0], bias2, g2, b2, None, save_f1=True))
        again[3] = _f1_rows(again[3], T)
        assert all(torch.equal(x, y) for x, y in zip(out, again))


def _f1_tiles(f1, I):
    """row-major [T, I] -> the producer / consumer kernel's tile-major layout (inverse of _f1_rows), whole 128-row blocks"""
    T = f1.shape[0]
    G, NC = 4 * ((T + 127) // 128), I // 32
    pad = torch.zeros(G * 32, I, dtype=f1.dtype, device=f1.device)
    pad[:T] = f1
    r = torch.arange(16, device=f1.device)
    lane = torch.arange(64, device=f1.device)
    col = (r[None, :] & 3) + 8 * (r[None, :] >> 2) + 4 * (lane[:, None] >> 5)
    tok = (lane & 31)[:, None].expand(64, 16)
    out = torch.empty(G, NC, 64, 16, dtype=f1.dtype, device=f1.device)
    for c in range(NC):
        for g in range(G):
            out[g, c] = pad[g * 32 + tok, c * 32 + col]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("T,N,K", [(400, 256, 384), (1000, 1536, 384), (4200, 384, 1536)])
def test_gemm_reads_tile_major_gelu_grad_of(ops, T, N, K):
    """dF1 = (dy W2) * gelu'(f1) with f1 in the fused forward's tile-major layout against the same GEMM on the row-major tensor: the
    192 x 384 kernel bit for bit; the 128 x 128 kernel takes the fused forward's own sigmoid-form GELU for a tiled f1
    (value within 2.5e-5, derivative within 1.1e-4 of the erf forms: compared at bf16 rounding)"""
    g = torch.Generator(device="cuda").manual_seed(5)
    bf = torch.bfloat16
    a = (torch.randn(T, K, device="cuda", generator=g) * 0.5).to(bf)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(bf)
    f1 = torch.randn(T, N, device="cuda", generator=g).to(bf)
    ga0, ga1 = torch.empty_like(f1), torch.empty_like(f1)
    want = ops.gemm_nt(a, b, gelu_grad_of=f1, gelu_out=ga0)
    got = ops.gemm_nt(a, b, gelu_grad_of=_f1_tiles(f1, N), gelu_out=ga1, gelu_grad_tiled=True)
    if torch.equal(ga0, ga1):  # the 192 x 384 kernel (long K, many rows) keeps the erf forms: then the products agree bit for bit too
        assert torch.equal(want, got)
    else:
        assert float((ga0.float() - ga1.float()).abs().max()) <= 2e-2 and float((ga0.float() - ga1.float()).abs().mean()) < 1e-3
        scale = float(want.float().abs().max())
        assert float((want.float() - got.float()).abs().max()) <= 1e-2 * scale


@pytest.mark.gpu
def test_ffn_pc_forward_dropout_mask_is_the_unfused_kernels(ops):
    """the producer / consumer kernel's hidden dropout (hf:349) uses the GEMM epilogue's hash on the same element index: the unfused
    backward (dropout_bwd / the dz2 GEMM epilogue) then masks exactly the elements the fused forward dropped"""
    from sparse_hip import lib as L
    T = 2048 + 16
    w1, w2 = _weights(seed=3)
    w1f, w2f, _, _ = _stage_pc(ops, w1, w2, torch.float16)
    z1 = _rnd(T, H, seed=21)
    one, zero = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
    bias1, bias2 = torch.zeros(I, device="cuda"), torch.zeros(H, device="cuda")
    drop = L.dropout(0.1, 1234, 7)
    a = ops.ffn_pc_fwd(z1, one, zero, 1e-12, w1f[0], bias1, w2f[0], bias2, one, zero, drop, save_f1=True)
    plain = ops.ffn_pc_fwd(z1, one, zero, 1e-12, w1f[0], bias1, w2f[0], bias2, one, zero, None, save_f1=True)
    lin_d, lin = a[4] - _ln(z1, one, zero)[0], plain[4] - _ln(z1, one, zero)[0]
    big = lin.abs() > 1e-2
    dropped = lin_d.abs() < 1e-5
    frac = float(dropped[big].float().mean())
    assert 0.08 < frac < 0.125, frac
    kept = ~dropped & big
    assert float((lin_d[kept] / lin[kept] - 256.0 / 230.0).abs().max()) < 2e-3
    bf = torch.bfloat16
    ref = ops.gemm_nt(torch.zeros(T, 64, dtype=bf, device="cuda"), torch.zeros(H, 64, dtype=bf, device="cuda"),
                      bias=torch.ones(H, device="cuda"), drop=drop)
    assert torch.equal((ref.float() == 0)[big], dropped[big]), "fused and unfused dropout masks differ"


@pytest.mark.gpu
@pytest.mark.parametrize("inter", [128, 256, 3072])
def test_ffn_pc_forward_other_intermediate_sizes(ops, inter):
    """4, 8 and 96 chunks of 32 intermediate columns (the ring prologue covers three): same check as at I = 1536"""
    T = 400
    w1, w2 = _weights(seed=inter, inter=inter)
    w1f, w2f, _, _ = _stage_pc(ops, w1, w2, torch.float16)
    z1 = _rnd(T, H, seed=5)
    g1, b1, g2, b2 = (1 + 0.05 * _rnd(H, seed=1), 0.05 * _rnd(H, seed=2), 1 + 0.05 * _rnd(H, seed=3), 0.05 * _rnd(H, seed=4))
    bias1, bias2 = 0.05 * _rnd(inter, seed=6), 0.05 * _rnd(H, seed=7)
    out = ops.ffn_pc_fwd(z1, g1, b1, 1e-12, w1f[0], bias1, w2f[0], bias2, g2, b2, None, save_f1=True)
    assert out is not None
    x1, m1, r1 = _ln(z1, g1, b1)
    f1 = x1.to(torch.float16).float() @ w1[0].to(torch.float16).float().t() + bias1
    ga = torch.nn.functional.gelu(f1).to(torch.float16).float()
    z2 = ga @ w2[0].to(torch.float16).float().t() + bias2 + x1
    x2 = _ln(z2, g2, b2)[0]
    assert float((_f1_rows(out[3], T).float() - f1).abs().max() / (1 + f1.abs().max())) < 1e-2
    assert float((out[4] - z2).abs().max() / (1 + z2.abs().max())) < 4e-3
    assert float((out[5].float() - x2).abs().max() / (1 + x2.abs().max())) < 1e-2


# ------------------------------------------------------------------ the backward in the same form (ffn_pc_bwd_kernel)
@pytest.mark.gpu
@pytest.mark.parametrize("T", [16, 112, 128, 1008, 4096 + 48, 6160 + 16])
@pytest.mark.parametrize("with_res", [True, False])
def test_ffn_pc_backward_matches_torch_and_the_unfused_launches(ops, T, with_res):
    """dF1 = (dy W2) * gelu'(f1), ga = gelu(f1), dz1 = LayerNorm-1'(dF1 W1 + dres), its dropout-masked copy and the gamma / beta
    gradients from ONE launch, against (a) plain torch fp32 autograd on the same operands (hf:334-351 + hf:293 backward; exact-erf
    GELU: the kernel's sigmoid form differs by <= 2e-4 in the derivative) and (b) the launches it replaces (sm_gemm_nt with the
    tile-major dF1 epilogue, then sm_gemm_nt_ln_bwd / sm_gemm_nt + sm_layernorm_bwd_res32).  Token counts that end inside a
    workgroup (T % 128 != 0) included."""
    from sparse_hip import lib
    bf = torch.bfloat16
    w1, w2 = _weights(seed=20)
    _, _, w2tf, w1tf = _stage_pc(ops, w1, w2, torch.float16)
    dy = (_rnd(T, H, seed=21) * 0.5).to(bf)
    dres = (_rnd(T, H, seed=22) * 0.5).to(bf) if with_res else None
    f1 = _rnd(T, I, seed=23).to(bf)
    z1 = _rnd(T, H, seed=24, scale=1.5) + 0.3
    gamma, beta = 1.0 + 0.1 * _rnd(H, seed=25), 0.1 * _rnd(H, seed=26)
    _, _, m1, r1 = ops.layernorm_fwd_res32(z1, gamma, beta, 1e-12, bf, want_y32=False)
    drop = lib.dropout(0.1, 31, 6)
    dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    got = ops.ffn_pc_bwd(dy, dres, _f1_tiles(f1, I), w2tf[0], w1tf[0], z1, gamma, m1, r1, drop, dg, db, want_drop=True)
    assert got is not None, "the fused backward must take this shape"
    df1_b, ga_b, dz1, dz1d = got
    df1, ga = df1_b.rows(), ga_b.rows()  # block-column-major -> row-major
    torch.cuda.synchronize()
    assert all(torch.isfinite(x.float()).all() for x in (df1, ga, dz1, dz1d, dg, db))
    # (a) torch fp32 on the values the kernel multiplies (weights rounded to bf16 by the staging)
    w1b, w2b = w1[0].to(bf).float(), w2[0].to(bf).float()
    x = f1.float().requires_grad_(True)
    gelu = 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    (gp,) = torch.autograd.grad(gelu.sum(), x)
    df1_ref = (dy.float() @ w2b) * gp
    _close(df1, df1_ref, 1e-2, "dF1 vs torch")
    _close(ga, gelu.detach(), 1e-2, "gelu(f1) vs torch")
    dx1 = df1.float() @ w1b + (dres.float() if with_res else 0.0)  # (from the kernel's own bf16 dF1: isolates the second half)
    zr = z1.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(zr, (H,), gr, br, 1e-12).backward(dx1.to(bf).float())
    _close(dz1, zr.grad, 1.5e-2, "dz1 vs torch")
    fro = lambda a, b: float((a.float() - b.float()).norm() / max(1e-12, float(b.float().norm())))
    assert fro(dg, gr.grad) <= 1e-2 and fro(db, br.grad) <= 1e-2, (fro(dg, gr.grad), fro(db, br.grad))
    # the dropout-masked copy: exactly the mask dropout_bwd regenerates for the same (seed, site)
    mask = ops.dropout_bwd(torch.ones(T, H, dtype=bf, device="cuda"), drop).float()
    assert torch.equal(dz1d.float() != 0, (mask != 0) & (dz1.float() != 0) | ((dz1d.float() != 0) & (dz1.float() == 0)))
    kept = mask != 0
    assert float((dz1d.float()[kept] - (dz1.float() * mask)[kept]).abs().max()) <= 2e-2 * float(dz1.float().abs().max())
    # (b) the launches it replaces
    ga0 = torch.empty(T, I, dtype=bf, device="cuda")
    df0 = ops.gemm_nt(dy, w2[0].t().contiguous().to(bf), gelu_grad_of=_f1_tiles(f1, I), gelu_out=ga0, gelu_grad_tiled=True)
    assert fro(df1, df0) <= 4e-3 and fro(ga, ga0) <= 4e-3, (fro(df1, df0), fro(ga, ga0))
    dg0, db0 = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    dx0 = ops.gemm_nt(df0, w1[0].t().contiguous().to(bf), residual=dres)
    dz0, dzd0 = ops.layernorm_bwd(dx0, z1, gamma, m1, r1, dg0, db0, drop, want_drop=True)
    assert fro(dz1, dz0) <= 1e-2 and fro(dz1d, dzd0) <= 1e-2 and fro(dg, dg0) <= 1e-2 and fro(db, db0) <= 1e-2, \
        (fro(dz1, dz0), fro(dz1d, dzd0), fro(dg, dg0), fro(db, db0))
    # without the dropout copy
    dg2, db2 = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    got2 = ops.ffn_pc_bwd(dy, dres, _f1_tiles(f1, I), w2tf[0], w1tf[0], z1, gamma, m1, r1, None, dg2, db2, want_drop=False)
    assert got2[3] is None and torch.equal(got2[2], dz1) and torch.equal(got2[0].rows(), df1)
    # the weight-gradient GEMM on the block-column-major operands == on their row-major copies (both operand slots)
    x1 = _rnd(T, H, seed=27).to(bf)
    for A, B, Ar, Br in ((df1_b, x1, df1, x1), (dy, ga_b, dy, ga)):
        n, kc = Ar.shape[1], Br.shape[1]
        o0, o1 = torch.zeros(n, kc, device="cuda"), torch.zeros(n, kc, device="cuda")
        c0, c1 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        ops.gemm_tn_acc(Ar, Br, o0, colsum=c0)
        ops.gemm_tn_acc(A, B, o1, colsum=c1)
        assert fro(o1, o0) <= 1e-5 and fro(c1, c0) <= 1e-5, (fro(o1, o0), fro(c1, c0))


@pytest.mark.gpu
def test_encoder_with_the_fused_ffn_backward_matches_the_two_launch_backward(monkeypatch):
    """the whole encoder, dropout on: SM_PC_FFN_BWD=1 (default) against 0 -- same forward, same masks, gradients within the bf16
    bound of two differently-ordered bf16 evaluations"""
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    cfg = BertConfigLite(vocab_size=3000, hidden_size=384, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1536,
                         max_position_embeddings=128, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1000, 3000, (12, 64), generator=g)
    mask = torch.ones(12, 64, dtype=torch.long)
    mask[3, 40:] = 0
    mask[7, 17:] = 0
    up = torch.randn(12, 3000, generator=g).cuda() * 1e-2
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SM_PC_FFN_BWD", mode)
        bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=3)
        assert bb.pc_ffn and bb.pc_ffn_bwd == (mode == "1")
        bb.train()
        bb.set_dropout_seed(99)
        rep = bb.encode(ids.cuda(), mask.cuda())
        (rep * up).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (rep.detach().clone(), bb.flat_grad.clone())
    assert torch.equal(res["1"][0], res["0"][0]), "the forward is the same launch sequence"
    assert torch.isfinite(res["1"][1]).all()
    rel = float((res["1"][1] - res["0"][1]).norm() / res["0"][1].norm())
    print(f"[fused vs two-launch FFN backward] flat gradient rel Frobenius {rel:.2e}")
    assert rel <= 2e-2, rel
