"""What does the fp8 format cost the sparse activations?  CPU emulation with the oracle's encoder (the companion of
tools/bf16_error_budget.py): variant B of that tool (bf16 GEMM operands, fp32 residual stream = the HIP path's default) with the FOUR
ENCODER LINEARS of every layer (QKV, attention output, FFN up, FFN down) taking per-tensor-scaled e4m3 operands, exactly as
HipBertMLM(fp8=True) stages them: activation tensor and weight matrix each scaled by 448 / amax, rounded to nearest even to OCP e4m3fn
(torch.float8_e4m3fn), multiplied exactly, accumulated in fp32.  Head, attention core, LayerNorms stay as in variant B.
Prints, against the all-fp32 oracle: worst |err| / (1 + |ref|) of rep, share of elements inside 1e-2 / 5e-2, relative Frobenius error --
the figures the tolerance of tests/test_baseline_configs_gpu.py::test_c5_fp8 is taken from.
    python tools/fp8_error_budget.py [n_docs] [mini|base] [seq]"""
import math, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd"), os.path.join(ROOT, "tools")]
from oracle import sparse_oracle as O

r = lambda x: x.to(torch.bfloat16).float()


def q8(x, e5m2=False):
    """per-tensor fp8 round trip: what sm_amax + sm_quantize_fp8 + the GEMM's dequantisation scale do to a tensor"""
    fmax = 57344.0 if e5m2 else 448.0
    amax = x.abs().max().clamp_min(1e-30)
    q = (x * (fmax / amax)).clamp(-fmax, fmax).to(torch.float8_e5m2 if e5m2 else torch.float8_e4m3fn)
    return q.float() * (amax / fmax)


def encode_fp8(p, ids, mask, cfg):
    """bf16_error_budget.encode(act_round=True, resid_fp32=True) with fp8 operands in the four encoder linears"""
    B, S = ids.shape
    H, A = cfg.hidden_size, cfg.num_attention_heads
    dh, eps = H // A, cfg.layer_norm_eps
    pre = "bert.embeddings."
    x = O._ln(p[pre + "word_embeddings.weight"][ids] + p[pre + "token_type_embeddings.weight"][0] + p[pre + "position_embeddings.weight"][:S],
              p[pre + "LayerNorm.weight"], p[pre + "LayerNorm.bias"], eps)
    xr, xo = x, r(x)
    amask = (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for l in range(cfg.num_hidden_layers):
        lp = f"bert.encoder.layer.{l}."
        wq = q8(r(torch.cat([p[lp + f"attention.self.{n}.weight"] for n in ("query", "key", "value")])))  # one staged [3H, H] matrix
        bq = torch.cat([p[lp + f"attention.self.{n}.bias"] for n in ("query", "key", "value")])
        qkv = r(F.linear(q8(xo), wq, bq))
        qq, kk, vv = (t.view(B, S, A, dh).transpose(1, 2) for t in qkv.split(H, dim=-1))
        pr = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(dh) + amask, dim=-1)
        ctx = r((r(pr) @ vv).transpose(1, 2).reshape(B, S, H))
        lin8 = lambda t, n: F.linear(q8(t), q8(r(p[lp + n + ".weight"])), p[lp + n + ".bias"])
        z1 = lin8(ctx, "attention.output.dense") + xr
        x1 = O._ln(z1, p[lp + "attention.output.LayerNorm.weight"], p[lp + "attention.output.LayerNorm.bias"], eps)
        ga = r(O._gelu(lin8(r(x1), "intermediate.dense")))
        z2 = lin8(ga, "output.dense") + x1
        x = O._ln(z2, p[lp + "output.LayerNorm.weight"], p[lp + "output.LayerNorm.bias"], eps)
        xr, xo = x, r(x)
    cp = "cls.predictions."
    t = r(O._gelu(F.linear(xo, r(p[cp + "transform.dense.weight"]), p[cp + "transform.dense.bias"])))
    t = r(O._ln(t, p[cp + "transform.LayerNorm.weight"], p[cp + "transform.LayerNorm.bias"], eps))
    return F.linear(t, r(p[pre + "word_embeddings.weight"]), p[cp + "bias"])


def report(name, rep, ref):
    err = (rep - ref).abs() / (1 + ref.abs())
    print(f"{name:44s} worst {float(err.max()):.3e}  inside 1e-2: {100 * float((err <= 1e-2).float().mean()):.3f} %  inside 5e-2: "
          f"{100 * float((err <= 5e-2).float().mean()):.4f} %  rel Frobenius {float((rep - ref).norm() / ref.norm()):.3e}")
    return float(err.max())


def main():
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from bf16_error_budget import encode
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    shape = sys.argv[2] if len(sys.argv) > 2 else "base"
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    cfg = O.BertShape() if shape == "mini" else O.BertShape(30522, 768, 12, 12, 3072, 512)
    for seed in (2, 5):
        p = O.init_params(cfg, seed=seed)
        g = torch.Generator().manual_seed(100 + seed)
        for n in p:
            if n.endswith("bias"):
                p[n] = 0.02 * torch.randn(p[n].shape, generator=g)
            elif n.endswith("LayerNorm.weight"):
                p[n] = 1 + 0.05 * torch.randn(p[n].shape, generator=g)
        pw = {n: (r(v) if v.dim() == 2 and "position" not in n and "token_type" not in n else v) for n, v in p.items()}
        k = min(nd, 8)
        ds = SyntheticTriplesDataset(nd // k, k, S, 32, cfg.vocab_size, seed=seed + 7, len_mean=S * 0.625, len_std=S * 0.234)
        d = PreTokenizedCollator()([ds[i] for i in range(nd // k)])["docs"][0]
        with torch.no_grad():
            ref = O.sparse_activation(O.bert_mlm_logits(p, d["input_ids"], d["attention_mask"], cfg), d["attention_mask"])
            rep_b = O.sparse_activation(encode(pw, d["input_ids"], d["attention_mask"], cfg, True, True, True), d["attention_mask"])
            rep_8 = O.sparse_activation(encode_fp8(p, d["input_ids"], d["attention_mask"], cfg), d["attention_mask"])
        print(f"seed {seed}: {shape}, {nd} documents x seq {S}")
        report("B  bf16 operands, fp32 residual stream", rep_b, ref)
        report("F8 e4m3 operands in the encoder linears", rep_8, ref)


if __name__ == "__main__":
    main()
