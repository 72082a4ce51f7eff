"""Hunt for the training collapse of GPUTEST_r03 (loss pinned at ln 25 after one bf16 + dropout optimiser step at the bench model
size): repeats the failing test's steps with

  * every FREE block of torch's caching allocator overwritten with a NaN bit pattern before each step (0x7FC0 per 16-bit word: a
    NaN as bf16, fp16 and -- two words -- fp32), so that any kernel that READS memory it never wrote (rows past the end of a
    ragged batch, `torch.empty` scratch, recycled buffers) turns its result into NaN instead of into whatever the previous
    owner left there; the blocks are found with torch.cuda.memory_snapshot() and filled through hipMemsetD16;
  * trainer.args.check_finite on: the flat gradient buffer is checked after backward, the parameters after AdamW, and the
    first tensors that went non-finite are named.

  python tools/collapse_hunt.py --trials 20 --steps 4 [--poison nan|huge|none] [--layout ragged|dense] [--dropout 0.1]
                                [--queries 8 --docs 4] [--seed 5]
Prints one line per trial and a summary; exit code 1 when any trial failed.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

_hip = None


def _hiplib():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemsetD16.argtypes = [C.c_void_p, C.c_ushort, C.c_size_t]
        _hip.hipMemsetD16.restype = C.c_int
    return _hip


def poison_free_cache(pattern: int = 0x7FC0) -> int:
    """overwrite every inactive block of the caching allocator; returns the number of bytes poisoned"""
    torch.cuda.synchronize()
    hip = _hiplib()
    total = 0
    for seg in torch.cuda.memory_snapshot():
        addr = seg["address"]
        for blk in seg["blocks"]:
            if blk["state"] == "inactive" and blk["size"] >= 2:
                rc = hip.hipMemsetD16(C.c_void_p(addr), pattern, blk["size"] // 2)
                if rc != 0:
                    raise RuntimeError(f"hipMemsetD16 failed: {rc}")
                total += blk["size"]
            addr += blk["size"]
    torch.cuda.synchronize()
    return total


_TRACE = {"events": [], "on": False}


def _tensors(obj, prefix=""):
    if isinstance(obj, torch.Tensor):
        yield prefix, obj
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _tensors(v, f"{prefix}[{i}]")
    elif isinstance(obj, dict):
        for k, v in obj.items():
            yield from _tensors(v, f"{prefix}.{k}")


def _bad(t):
    if not t.is_cuda or not t.is_floating_point() or t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2) or t.numel() == 0:
        return None
    nb = ~torch.isfinite(t)
    c = int(nb.sum())
    if not c:
        return None
    rows = ""
    if t.dim() == 2:
        r = nb.any(1).nonzero().flatten()
        rows = f" rows {r[:6].tolist()}..{int(r[-1])} ({r.numel()} of {t.shape[0]})"
    return f"{tuple(t.shape)} {str(t.dtype)[6:]} {c} bad{rows}"


def install_op_tracing():
    """--trace-ops: every sparse_hip.ops wrapper is followed by a device sync and a finite-check of its tensor arguments (before
    and after the call: in-place outputs) and of its results; the first calls that turn a tensor non-finite are recorded"""
    import inspect
    from sparse_hip import ops

    def wrap(name, f):
        def g(*a, **k):
            if not _TRACE["on"]:
                return f(*a, **k)
            torch.cuda.synchronize()
            before = {n: _bad(t) for n, t in _tensors([a, k], "arg")}
            r = f(*a, **k)
            torch.cuda.synchronize()
            after = {n: _bad(t) for n, t in _tensors([a, k], "arg")}
            res = {n: _bad(t) for n, t in _tensors(r, "ret")}
            new = {n: v for n, v in after.items() if v and not before.get(n)}
            new.update({n: v for n, v in res.items() if v})
            if new:
                _TRACE["events"].append((name, {n: v for n, v in before.items() if v}, new))
            return r
        return g

    for name, f in list(vars(ops).items()):
        if inspect.isfunction(f) and f.__module__ == ops.__name__ and not name.startswith("_"):
            setattr(ops, name, wrap(name, f))


def make_trainer(args, seed=0):
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    from sparse_hip.encoder import BertConfigLite, HipBertMLM
    if getattr(args, "width", "mini") == "base":  # round 6: the bert-base width (configs[3] / [4]); --fp8: fp8 encoder linears -- the weight-stationary
        # fp8 GEMMs need >= 8192 token rows (e.g. --layout dense --seq 512 --queries 8 --docs 4), the fused GELU + quantise pass a second step
        cfg = BertConfigLite(hidden_size=768, num_hidden_layers=int(getattr(args, "layers", 0) or 12), num_attention_heads=12, intermediate_size=3072,
                             hidden_dropout_prob=args.dropout, attention_probs_dropout_prob=args.dropout)
    else:
        cfg = BertConfigLite(hidden_dropout_prob=args.dropout, attention_probs_dropout_prob=args.dropout)
    bb = HipBertMLM(cfg, compute_dtype=torch.bfloat16, device="cuda", init_seed=seed, fp8=bool(getattr(args, "fp8", False)))
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 99)
        bb.view("cls.predictions.bias").copy_(torch.randn(cfg.vocab_size, generator=g) * 0.5)
    bb.mark_weights_dirty()
    bb.varlen = args.layout == "ragged"
    model = SparseModel(bb, use_l0=False)
    S = int(getattr(args, "seq", 128))  # round 6: the long-document kernels (S = 256 / 512)
    ds = SyntheticTriplesDataset(args.queries, args.docs, S, 32, 30522, seed=args.seed, len_mean=80.0 * S / 128, len_std=30.0 * S / 128)
    batch = PreTokenizedCollator()([ds[i] for i in range(args.queries)])
    margs = ModelArguments(model_name_or_path="x", inf_free=True)
    dargs = DataTrainingArguments(loss_types=["infonce"], use_in_batch_negatives=True, flops_d_lambda=0.0, flops_d_T=1)
    targs = TrainingArguments(output_dir="/tmp/sm_hunt", logging_steps=10 ** 9, learning_rate=2e-4, warmup_steps=0, max_steps=100,
                              check_finite=True)
    trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                                 loss_functions=[LOSS_CLS_MAP["infonce"](use_in_batch_negatives=True, weight=1)])
    return trainer, batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--poison", default="nan", choices=["nan", "huge", "none"])
    ap.add_argument("--layout", default="ragged", choices=["ragged", "dense"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--queries", type=int, default=8)
    ap.add_argument("--docs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--width", default="mini", choices=["mini", "base"])
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--trace-ops", action="store_true")
    ap.add_argument("--fresh-gb", type=float, default=0.0,
                    help="before every trial: fill this many GiB with the poison pattern, free them and EMPTY the allocator's cache, so "
                         "the trial's allocations are fresh hipMalloc segments over recycled (unless the driver clears them) pages")
    args = ap.parse_args()
    if args.trace_ops:
        install_op_tracing()
    pattern = {"nan": 0x7FC0, "huge": 0x7F7F, "none": None}[args.poison]
    import socket
    print("box:", socket.gethostname(), torch.cuda.get_device_properties(0).name, flush=True)
    failures = 0
    for trial in range(args.trials):
        if args.fresh_gb > 0:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            junk = [torch.full((1 << 28,), pattern if pattern is not None else 0x7FC0, dtype=torch.int16, device="cuda")
                    for _ in range(int(args.fresh_gb * 2))]
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
        trainer, batch = make_trainer(args)
        inp = trainer._prepare_inputs(batch)
        losses, err, poisoned = [], None, 0
        for step in range(args.steps):
            if pattern is not None:
                poisoned = poison_free_cache(pattern)
            _TRACE["on"], _TRACE["events"] = args.trace_ops, []
            try:
                losses.append(float(trainer.training_step(inp)))
            except FloatingPointError as e:
                err = str(e)
                err = err[:200] + (" ..." if len(err) > 200 else "")
                losses.append(float(trainer._last["total"]) if "total" in trainer._last else float("nan"))
            _TRACE["on"] = False
            for name, before, new in _TRACE["events"][:12]:
                print(f"    step {step} op {name}: non-finite IN {before or '-'}  NEW {new}", flush=True)
            if err:
                break
        torch.cuda.synchronize()
        ok = err is None and all(l == l for l in losses) and (len(losses) < 3 or losses[-1] < losses[0])
        failures += 0 if ok else 1
        print(f"trial {trial}: {'ok  ' if ok else 'FAIL'} poisoned {poisoned / 2 ** 20:.0f} MiB  losses {[round(l, 4) for l in losses]}"
              + (f"  {err}" if err else ""), flush=True)
        del trainer, inp
    print(f"collapse_hunt: {failures} of {args.trials} trials failed (poison={args.poison}, layout={args.layout}, dropout={args.dropout}, "
          f"batch {args.queries} x {args.docs})")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
