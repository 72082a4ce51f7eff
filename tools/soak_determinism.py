"""Bitwise-reproducibility soak of one training step's kernels (forward + loss + backward of the GPUTEST_r03 collapse test's
workload, bf16 + dropout at the configs[1] model size).

Every kernel of the step except the fp32-atomic accumulations into the flat gradient buffer is deterministic BY DESIGN (no
atomics, fixed reduction orders, counter-based dropout): with the same parameters, inputs and dropout seed every intermediate
tensor must come out bit-identical.  The tool wraps each sparse_hip.ops call, enqueues a 64-bit checksum (integer sum of the
raw words, on the stream the op ran on: no host sync inside the step) of every tensor the call touched -- results and
arguments, the latter AFTER the call so that in-place outputs and later corruption of an input are both seen -- and compares
the list with the first iteration's.  A mismatch that shows up rarely is a race in (or just before) the named launch.
`torch.empty` is replaced by `torch.zeros` for the run so that rows a kernel legitimately leaves unwritten compare equal.

  python tools/soak_determinism.py --iters 200 [--layout ragged|dense] [--dropout 0.1] [--queries 8 --docs 4] [--perturb]
                                   [--sync-each-op]
--perturb: a second thread keeps a matrix product running on its own stream (the step's kernels then see changing numbers of
free CUs and different memory latencies).  Exit code 1 when a non-atomic tensor ever differed.
"""
import argparse
import inspect
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

REC = {"on": False, "items": [], "skip": (0, 0), "sync": False}


def _tensors(obj, prefix=""):
    if isinstance(obj, torch.Tensor):
        yield prefix, obj
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _tensors(v, f"{prefix}[{i}]")
    elif isinstance(obj, dict):
        for k, v in obj.items():
            yield from _tensors(v, f"{prefix}.{k}")
    elif hasattr(obj, "__dict__") and type(obj).__name__ in ("Ragged", "PackedDocs", "DenseHints"):
        for k, v in vars(obj).items():
            yield from _tensors(v, f"{prefix}.{k}")


def _checksum(t):
    if not t.is_cuda or t.numel() <= 1 or not t.is_contiguous():
        return None  # (single-element tensors: the loss / FLOPS scalars are summed with float atomics, their last bit depends on the order)
    lo, hi = REC["skip"]
    if lo <= t.data_ptr() < hi:  # a view of the flat gradient buffer: fp32 atomics, not reproducible by design
        return None
    n = t.element_size()
    flat = t.reshape(-1)
    if n == 2:
        w = flat.view(torch.int16)
    elif n == 4:
        w = flat.view(torch.int32)
    elif n == 8:
        w = flat.view(torch.int64)
    else:
        w = flat.view(torch.uint8)
    return w.sum(dtype=torch.int64)


def install():
    from sparse_hip import ops

    def wrap(name, f):
        def g(*a, **k):
            r = f(*a, **k)
            if REC["on"]:
                for slot, t in list(_tensors(r, "ret")) + list(_tensors([a, k], "arg")):
                    c = _checksum(t)
                    if c is not None:
                        REC["items"].append((name, slot, tuple(t.shape), c))
                if REC["sync"]:
                    torch.cuda.synchronize()
            return r
        return g

    for name, f in list(vars(ops).items()):
        if inspect.isfunction(f) and f.__module__ == ops.__name__ and not name.startswith("_"):
            setattr(ops, name, wrap(name, f))
    _empty, _empty_like = torch.empty, torch.empty_like
    torch.empty = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "memory_format"})
    torch.empty_like = lambda t, **k: torch.zeros_like(t, **{kk: vv for kk, vv in k.items() if kk != "memory_format"})


def box_identity():
    out = [f"host {socket.gethostname()}"]
    try:
        r = subprocess.run(["rocm-smi", "--showuniqueid", "--showserial"], capture_output=True, text=True, timeout=20)
        out += [ln.strip() for ln in r.stdout.splitlines() if "Unique" in ln or "Serial" in ln]
    except Exception as e:  # noqa: BLE001
        out.append(f"rocm-smi: {e}")
    return "; ".join(out)


def perturb_loop(stop):
    s = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    i = 0
    while not stop.is_set():
        with torch.cuda.stream(s):
            for _ in range(1 + i % 4):
                torch.mm(a, b)
        i += 1
        time.sleep(0.0005 * (i % 7))
        if i % 16 == 0:
            s.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--layout", default="ragged", choices=["ragged", "dense"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--queries", type=int, default=8)
    ap.add_argument("--docs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--width", default="mini", choices=["mini", "base"])
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--perturb", action="store_true")
    ap.add_argument("--sync-each-op", action="store_true")
    args = ap.parse_args()
    print("box:", box_identity(), flush=True)
    import collapse_hunt as H
    install()
    REC["sync"] = args.sync_each_op
    trainer, batch = H.make_trainer(args)
    bb = trainer.model.sparse_model.backbone
    REC["skip"] = (bb.flat_grad.data_ptr(), bb.flat_grad.data_ptr() + bb.flat_grad.numel() * 4)
    inp = trainer._prepare_inputs(batch)
    trainer.model.train()
    stop = threading.Event()
    th = None
    if args.perturb:
        th = threading.Thread(target=perturb_loop, args=(stop,), daemon=True)
        th.start()
    ref, ref_meta, noisy, rare = None, None, {}, []
    losses = set()
    t0 = time.time()
    for it in range(-1, args.iters):  # iteration -1: warm-up (staging copies are made there), not compared
        trainer.zero_grad()
        bb.set_dropout_seed(777)
        REC["items"], REC["on"] = [], True
        loss = trainer.compute_loss(trainer.model, inp)
        loss.backward()
        REC["on"] = False
        torch.cuda.synchronize()
        if it < 0:
            continue
        meta = [(n, s, sh) for n, s, sh, _ in REC["items"]]
        sums = torch.stack([c for _, _, _, c in REC["items"]]).cpu()
        finite = bool(torch.isfinite(bb.flat_grad).all())
        losses.add(float(loss.detach()))
        if ref is None:
            ref, ref_meta = sums, meta
            print(f"iteration 0: {len(meta)} checksummed tensors over {len(set(n for n, _, _ in meta))} entry points, loss {float(loss):.6f}, "
                  f"gradients finite: {finite}", flush=True)
            continue
        if meta != ref_meta:
            print(f"iteration {it}: the sequence of launches changed ({len(meta)} against {len(ref_meta)})")
            stop.set()
            return 1
        bad = (sums != ref).nonzero().flatten().tolist()
        for i in bad:
            noisy.setdefault(i, []).append(it)
        if bad or not finite:
            first = bad[0] if bad else -1
            rare.append((it, first, len(bad), finite, float(loss.detach())))
    stop.set()
    if th is not None:
        th.join(timeout=5)
    dt = time.time() - t0
    print(f"{args.iters} iterations in {dt:.1f} s; distinct loss values: {sorted(losses)}")
    fail = 0
    for i, its in sorted(noisy.items()):
        n, s, sh = ref_meta[i]
        frac = len(its) / (args.iters - 1)
        print(f"  #{i} {n} {s} {sh}: differed in {len(its)} iterations ({100 * frac:.1f} %), first {its[:5]}")
        fail = 1
    for it, first, nbad, finite, loss in rare[:20]:
        n, s, sh = ref_meta[first] if first >= 0 else ("-", "-", "-")
        print(f"  iteration {it}: first differing tensor #{first} {n} {s} {sh}, {nbad} differing, gradients finite {finite}, loss {loss:.6f}")
    spread = (max(losses) - min(losses)) / max(abs(min(losses)), 1e-30)
    fail = fail or spread > 1e-5  # the loss itself is a float-atomic sum: equal to the last bits, not bitwise
    print(f"soak_determinism: {'MISMATCH' if fail else 'reproducible'} (layout={args.layout}, dropout={args.dropout}, "
          f"perturb={args.perturb}, sync_each_op={args.sync_each_op}, loss spread {spread:.1e})")
    return 1 if fail else 0


if __name__ == "__main__":
    sys.exit(main())
