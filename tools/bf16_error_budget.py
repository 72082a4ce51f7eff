"""Where does the bf16 path's elementwise error come from?  CPU experiment with the oracle's encoder: the same fp32 math with
bf16 rounding inserted at chosen storage points, at BASELINE configs[1]'s model (v2-mini shape, seq 128) on a slice of the batch.
  A  every stored activation rounded (what the HIP path stores: qkv, ctx, z1, x1, f1/ga, z2, x2, transform, tn; GEMM weights bf16)
  B  fp32 RESIDUAL STREAM: z1 / z2 and the LayerNorm outputs stay fp32 on the residual path; only GEMM operands are rounded
     (what torch autocast does: hf modeling_bert.py:289-293, 347-351)
  C  only the GEMM weights rounded (activations fp32): the floor any bf16-operand path has
Prints, against the all-fp32 oracle: worst |err| / (1 + |ref|) of rep, share of elements inside 1e-2, relative Frobenius error.
    python tools/bf16_error_budget.py [n_docs] [mini|base] [seq] [trained]   (trained: outlier dimensions, LayerNorm gains up to 5, ~1 % alive)"""
import math, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from oracle import sparse_oracle as O
from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset

r = lambda x: x.to(torch.bfloat16).float()


def encode(p, ids, mask, cfg, act_round, resid_fp32, z_fp32=None):
    """oracle.bert_mlm_logits with rounding points; act_round: round stored activations; resid_fp32: keep the residual stream fp32"""
    q = r if act_round else (lambda x: x)
    z_fp32 = resid_fp32 if z_fp32 is None else z_fp32   # pre-LayerNorm sum kept fp32 (a LayerNorm fused into the GEMM epilogue does that)
    B, S = ids.shape
    H, A = cfg.hidden_size, cfg.num_attention_heads
    dh, eps = H // A, cfg.layer_norm_eps
    pre = "bert.embeddings."
    x = O._ln(p[pre + "word_embeddings.weight"][ids] + p[pre + "token_type_embeddings.weight"][0] + p[pre + "position_embeddings.weight"][:S],
              p[pre + "LayerNorm.weight"], p[pre + "LayerNorm.bias"], eps)
    xr = x if resid_fp32 else q(x)          # residual copy
    xo = q(x)                               # GEMM operand copy
    amask = (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for l in range(cfg.num_hidden_layers):
        lp = f"bert.encoder.layer.{l}."
        lin = lambda t, n: F.linear(t, p[lp + n + ".weight"], p[lp + n + ".bias"])
        qq, kk, vv = (q(lin(xo, "attention.self." + n)).view(B, S, A, dh).transpose(1, 2) for n in ("query", "key", "value"))
        pr = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(dh) + amask, dim=-1)
        ctx = q((q(pr) @ vv).transpose(1, 2).reshape(B, S, H))
        z1 = lin(ctx, "attention.output.dense") + xr
        z1 = z1 if z_fp32 else q(z1)
        x1 = O._ln(z1, p[lp + "attention.output.LayerNorm.weight"], p[lp + "attention.output.LayerNorm.bias"], eps)
        x1r, x1o = (x1 if resid_fp32 else q(x1)), q(x1)
        ga = q(O._gelu(lin(x1o, "intermediate.dense")))
        z2 = lin(ga, "output.dense") + x1r
        z2 = z2 if z_fp32 else q(z2)
        x = O._ln(z2, p[lp + "output.LayerNorm.weight"], p[lp + "output.LayerNorm.bias"], eps)
        xr, xo = (x if resid_fp32 else q(x)), q(x)
    cp = "cls.predictions."
    t = q(O._gelu(F.linear(xo, p[cp + "transform.dense.weight"], p[cp + "transform.dense.bias"])))
    t = q(O._ln(t, p[cp + "transform.LayerNorm.weight"], p[cp + "transform.LayerNorm.bias"], eps))
    return F.linear(t, p[pre + "word_embeddings.weight"], p[cp + "bias"])


def main():
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    pos = [a for a in sys.argv[2:] if a != "trained"]
    shape = pos[0] if len(pos) > 0 else "mini"      # mini | base (bert-base, 12 layers: BASELINE configs[3] / [4])
    S = int(pos[1]) if len(pos) > 1 else 128
    cfg = O.BertShape() if shape == "mini" else O.BertShape(30522, 768, 12, 12, 3072, 512)
    p = O.init_params(cfg, seed=2)
    g = torch.Generator().manual_seed(102)
    for n in p:
        if n.endswith("bias"):
            p[n] = 0.02 * torch.randn(p[n].shape, generator=g)
        elif n.endswith("LayerNorm.weight"):
            p[n] = 1 + 0.05 * torch.randn(p[n].shape, generator=g)
    k = min(nd, 16)
    ds = SyntheticTriplesDataset(nd // k, k, S, 32, cfg.vocab_size, seed=9, len_mean=S * 0.625, len_std=S * 0.234)
    d = PreTokenizedCollator()([ds[i] for i in range(nd // k)])["docs"][0]
    if "trained" in sys.argv[2:]:  # the statistics of a fine-tuned checkpoint (oracle.make_trained_like; tests: test_c2_slice_at_trained_checkpoint_statistics)
        O.make_trained_like(p, cfg, d, g)
    pw = {n: (r(v) if v.dim() == 2 and "position" not in n and "token_type" not in n else v) for n, v in p.items()}
    with torch.no_grad():
        ref = O.sparse_activation(O.bert_mlm_logits(p, d["input_ids"], d["attention_mask"], cfg), d["attention_mask"])
        for name, (params, act, res, zf) in {"C weights only": (pw, False, True, True), "B fp32 residual stream": (pw, True, True, True),
                                             "B1 pre-LN sum fp32, LN output bf16": (pw, True, False, True),
                                             "A all activations bf16 (HIP path)": (pw, True, False, False)}.items():
            rep = O.sparse_activation(encode(params, d["input_ids"], d["attention_mask"], cfg, act, res, zf), d["attention_mask"])
            err = (rep - ref).abs() / (1 + ref.abs())
            print(f"{name:36s} worst {float(err.max()):.3e}  inside 1e-2: {100 * float((err <= 1e-2).float().mean()):.4f} %  "
                  f"rel Frobenius {float((rep - ref).norm() / ref.norm()):.3e}")


if __name__ == "__main__":
    main()
