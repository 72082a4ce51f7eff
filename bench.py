#!/usr/bin/env python3
"""Headline benchmark: training samples/sec of the neural-sparse fine-tuning step
(BASELINE.json configs[1]: config_infonce.yaml recipe, v2-mini-shaped encoder, synthetic
MS-MARCO triples, bs=32 x (1 pos + 15 negs), seq 128, bf16, dropout on, fused AdamW included).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one optimisation step over one batch per rank; batches are resident in HBM
before the timed region.  `value` is measured on the DENSE layout (every padded token row
computed; the ragged rate the product's trainer runs at is reported beside it).  Rank 0 prints
ONE JSON line (contract in the task statement) with two extra objects: "roofline" (the step's
dominant kernel class, the encoder GEMMs, timed live with HIP events on the streams they run
on; the fused head forward under "roofline_head_fwd") and "cpu_baseline" (the CPU oracle timed
at the full configs[1] shape, 2 steps; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver supports dmabuf IPC only: without it RCCL's
                                                            # hipIpcGetMemHandle fails (multi-process GPU work)
# N > 1: the step keeps FIVE HIP streams busy (backward chain, weight gradients, gradient all-reduce, RCCL's own, input prefetch); with the
# runtime's default of 4 hardware queues two of them share one, and when the communication stream's wait for a layer's weight gradients
# lands in the main stream's queue the backward chain stalls for the length of that weight-gradient kernel -- 275 us behind every layer,
# 0.36 ms of a 12.4 ms step (measured with a one-rank RCCL communicator, profiles/r6_single_rank_rccl.txt).  Must be set before HIP starts.
# (Only with one process per GPU: SM_BENCH_BACKEND=gloo puts N ranks on ONE GPU for functional checks, and two processes with 8 hardware
# queues each on one device deadlocked in the first host-blocking gloo collective -- tools/r6_run15.sh.)
if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--single-rank-rccl" in sys.argv) and os.environ.get("SM_BENCH_BACKEND", "nccl") == "nccl":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK = {"bf16": 2.5e15, "f32": 157.3e12}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HEAD_KERNELS = ("sparse_head_fwd_vs_kernel",)  # the roofline kernel's name in the PMC summaries (a summary of an older kernel is not reported)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--layout", default="dense", choices=["ragged", "dense"],
                    help="layout `value` is measured on: dense = every document computed at the full padded length (SURVEY 8d "
                         "headline protocol, the default), ragged = padding tokens skipped on the device (what the product's "
                         "trainer does by default: identical outputs); the other layout is timed too (after the timed region) "
                         "and reported beside it")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--negs", type=int, default=15)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--len-scale", type=float, default=None, help="document lengths ~N(80, 30) x this factor (the seq_sweep leg: seq / 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-roofline", action="store_true", help="skip the post-run per-GEMM timing steps")
    ap.add_argument("--no-extras", action="store_true", help="skip the sparse-regime leg and the configs[4] per-GPU-shape leg (N = 1 only)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-timeout", type=float, default=240.0)
    ap.add_argument("--cpu-baseline-bounded", action="store_true",
                    help="time the oracle only on the bounded sample (2 queries per step) instead of configs[1] at FULL shape "
                         "(32 x 16 docs, 2 steps, ~36 GB of host RAM, about a minute: the default)")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--bf16-storage", action="store_true",
                    help="all-bf16 activation storage instead of the default fp32 residual stream (+4.5 %% throughput; the sparse "
                         "activations then miss the elementwise 1e-2 bound on ~0.002 %% of the elements: DESIGN 4)")
    ap.add_argument("--only-value-layout", action="store_true", help="profiling runs: do not time the other layout")
    ap.add_argument("--single-rank-rccl", action="store_true",
                    help="N = 1 only: initialise a ONE-rank process group (backend nccl = RCCL) and run the step through the N > 1 code "
                         "path (SM_DIST_SINGLE_RANK=1): every collective of the distributed step executes through RCCL on this GPU")
    return ap.parse_args()


def build_trainer(args, device, rank, layouts=("ragged",)):
    from scripts.args import DataTrainingArguments, ModelArguments, TrainingArguments
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    from scripts.model.sparse_encoders import SparseModel
    from scripts.train.loss import LOSS_CLS_MAP
    from scripts.train.trainer import SparseModelTrainer
    from sparse_hip.encoder import BertConfigLite, HipBertMLM

    p = 0.0 if args.no_dropout else 0.1
    cfg = BertConfigLite(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                         intermediate_size=1536, max_position_embeddings=512, hidden_dropout_prob=p,
                         attention_probs_dropout_prob=p)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    bb = HipBertMLM(cfg, compute_dtype=dtype, device=device, init_seed=0, residual_fp32=not args.bf16_storage)
    g = torch.Generator().manual_seed(7)
    idf = torch.exp(torch.rand(cfg.vocab_size, generator=g) * (torch.log(torch.tensor(15.6 / 0.02))) + torch.log(torch.tensor(0.02)))
    model = SparseModel(bb, idf=idf, use_l0=False)
    k = args.negs + 1
    n_batches = 4
    lens = {} if getattr(args, "len_scale", None) is None else {"len_mean": 80.0 * args.len_scale, "len_std": 30.0 * args.len_scale}
    ds = SyntheticTriplesDataset(args.bs * n_batches, k, args.seq, 32, cfg.vocab_size, seed=1234 + rank, **lens)
    coll = PreTokenizedCollator()
    margs = ModelArguments(model_name_or_path="random-init-v2-mini-shape", inf_free=True)
    dargs = DataTrainingArguments(loss_types=["infonce"], use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=200,
                                  sample_num_one_query=args.negs, max_seq_length=args.seq, data_type="posnegs")
    targs = TrainingArguments(output_dir="/tmp/sm_bench", per_device_train_batch_size=args.bs, max_steps=2000,
                              learning_rate=2e-5, weight_decay=0.01, warmup_steps=200, logging_steps=10 ** 9, bf16=True)
    trainer = SparseModelTrainer(model_args=margs, data_args=dargs, model=model, args=targs,
                                 loss_functions=[LOSS_CLS_MAP["infonce"](use_in_batch_negatives=True, weight=1)])
    batches = {}
    for layout in layouts:  # same token ids; "dense" keeps the collator's [B, S] layout, "ragged" packs on the host
        bb.varlen = layout == "ragged"
        batches[layout] = [trainer._prepare_inputs(coll([ds[b * args.bs + i] for i in range(args.bs)])) for b in range(n_batches)]
    return trainer, cfg, batches


class KernelTimer:
    """HIP events around one kernel entry point, recorded on the stream it is launched on."""

    def __init__(self, module, name):
        self.module, self.name = module, name
        self.orig = getattr(module, name)
        self.events = []
        self.rows = []
        self.enabled = False

    def __enter__(self):
        def wrapped(*a, **kw):
            if not self.enabled:
                return self.orig(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(*a, **kw)
            e1.record()
            self.events.append((e0, e1))
            self.rows.append(int(a[0].shape[0]))  # token rows the launch actually processed
            return out
        setattr(self.module, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.module, self.name, self.orig)

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events) / max(1, len(self.events))


class GemmRoofline:
    """After the timed region: a few extra steps with HIP events around every encoder GEMM launch (events on the stream
    the kernel runs on; the weight-gradient GEMMs run on the side stream and share the GPU with the backward chain,
    exactly as in the measured step).  Reported per op: launches / step, executed FLOPs, achieved TFLOP/s in the step."""

    MAX_STAMPS, STAMP_WORDS = 1024, 4096  # SM_CLOCK_STAMP_WORDS of include/sparse_hip.h: 2048 (XCD, CU) units x 2 words

    def __init__(self, ops):
        self.ops, self.rec, self.orig = ops, {}, {}
        # shader-clock stamps (sm_clock_stamp, ABI 7): per XCD a (s_memtime, s_memrealtime) pair before and after every timed launch,
        # on the launch's own stream -> the clock the chip held under that kernel (the verdict's "clock as a measured quantity")
        from sparse_hip import lib as L
        self.L = L
        self.stamps = torch.zeros(self.MAX_STAMPS * self.STAMP_WORDS, dtype=torch.int64, device="cuda")
        self.nstamp = 0
        self.extra = set()

    def _stamp(self):
        if self.nstamp >= self.MAX_STAMPS:
            return None
        slot, self.nstamp = self.nstamp, self.nstamp + 1
        self.L.call("sm_clock_stamp", self.L.ptr(self.stamps), slot, self.L.stream_ptr())
        return slot

    def _wrap(self, name, flops, label=None):
        orig = getattr(self.ops, name)
        self.orig[name] = orig

        def wrapped(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0 = self._stamp()
            e0.record()
            out = orig(*a, **kw)
            e1.record()
            s1 = self._stamp()
            if out is not None and out is not False:  # (a fused entry point that declines the shape returns None / False and its caller runs the unfused ops, which
                key = name if label is None else label(*a, **kw)                     # are counted themselves)
                self.rec.setdefault(key, []).append((e0, e1, flops(*a, **kw), s0, s1))
            return out
        setattr(self.ops, name, wrapped)

    def _clock_ghz(self, st, s0, s1):
        """median over the compute units that both stamps reached of d(s_memtime) / d(s_memrealtime) x 100 MHz (counters of
        different units are not aligned: only same-unit pairs are differenced)"""
        if s0 is None or s1 is None:
            return None
        a, b = st[s0], st[s1]  # [2048 units, 2]
        ok = (a[:, 1] > 0) & (b[:, 1] > a[:, 1]) & (b[:, 0] > a[:, 0])
        if int(ok.sum()) < 8:
            return None
        return float(((b[ok, 0] - a[ok, 0]).double() / (b[ok, 1] - a[ok, 1]).double()).median() * 0.1)

    @staticmethod
    def _nt_label(A, B, *a, n=None, **k):
        """which kernel an NT launch takes (csrc/gemm.hip launch_gemm_nt): the weight-stationary one (K = 384, plain or dF1 epilogue,
        >= 8192 rows, bf16) or a tile kernel"""
        plain = not any(k.get(x) for x in ("act", "preact", "drop", "residual", "out_f32", "residual_ln"))
        epi0 = k.get("gelu_grad_of") is None and k.get("gelu_out") is None
        epi1 = k.get("gelu_grad_of") is not None and k.get("gelu_grad_tiled") and k.get("bias") is None
        N = B.shape[0] if n is None else n
        ws = A.dtype == torch.bfloat16 and A.shape[1] == 384 and A.shape[0] >= 8192 and N % 128 == 0 and plain and (epi0 or epi1)
        return "gemm_nt (weight-stationary, K = 384)" if ws else "gemm_nt (tile kernels)"

    def __enter__(self):
        self._wrap("gemm_nt", lambda A, B, *a, n=None, **k: 2.0 * A.shape[0] * A.shape[1] * (B.shape[0] if n is None else n), self._nt_label)
        self._wrap("gemm_nt_ln_bwd", lambda A, B, *a, **k: 2.0 * A.shape[0] * A.shape[1] * B.shape[0])
        self._wrap("gemm_tn_acc", lambda A, B, *a, **k: 2.0 * A.shape[0] * A.shape[1] * B.shape[1])
        # the grouped weight-gradient launch (csrc/gemm_tn2.hip: a layer's four products in one grid)
        self._wrap("gemm_tn_group", lambda probs, *a, **k: sum(2.0 * A.shape[0] * A.shape[1] * B.shape[1] for A, B, _, _ in probs))
        # the fused feed-forward forward (LayerNorm 1 + FFN-up + GELU + FFN-down + residual + LayerNorm 2): its two GEMMs' FLOPs
        self._wrap("ffn_pc_fwd", lambda z1, g1, b1, eps, w1f, bias1, *a, **k: 4.0 * z1.shape[0] * z1.shape[1] * bias1.shape[0])
        # ... and its backward (dF1 GEMM + FFN-up input gradient, with GELU', the LayerNorm-1 backward and the residual gradient)
        self._wrap("ffn_pc_bwd", lambda dy, dres, f1, *a, **k: 4.0 * dy.shape[0] * dy.shape[1] * f1.shape[1] * 32)
        # the other matrix-pipe kernels of the step, for their CLOCK only (not part of the encoder-GEMM roofline sums)
        for extra in ("sparse_head_fwd", "sparse_head_bwd_dt_ln", "attention_fwd", "attention_bwd"):
            self.extra.add(extra)
            self._wrap(extra, lambda *a, **k: 0.0)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def summary(self, steps, peak):
        out = []
        st = self.stamps.cpu().view(self.MAX_STAMPS, self.STAMP_WORDS // 2, 2)
        self.other = []
        for name, rec in self.rec.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in rec)
            fl = sum(r[2] for r in rec)
            clk = [c for c in (self._clock_ghz(st, r[3], r[4]) for r in rec) if c is not None]
            if name in self.extra:
                self.other.append({"op": name, "launches_per_step": len(rec) / steps, "ms_per_step": ms / steps,
                                   "clock_ghz": sum(clk) / len(clk) if clk else None})
            elif ms > 0:
                out.append({"op": name, "launches_per_step": len(rec) / steps, "gflop_per_step": fl / steps / 1e9,
                            "ms_per_step": ms / steps, "achieved_tflops": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / peak,
                            # shader clock under the op's launches (stamps around each launch, on its stream); at 2.4 GHz the
                            # dense bf16 MFMA peak is 2.5 PFLOP/s, so the op's rate against what ITS clock allows is frac_at_clock
                            "clock_ghz": sum(clk) / len(clk) if clk else None,
                            "frac_at_clock": (fl / (ms * 1e-3) / (peak * (sum(clk) / len(clk)) / 2.4)) if clk else None})
        return out


def usable_cores() -> int:
    """Cores this process may really use: affinity mask and cgroup CPU quota, capped at 32 (the
    oracle's matrices are too small to scale further; oversubscribing a quota is catastrophic)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """Run the CPU leg in a child process with a hard wall-clock bound so the GPU bench line is never
    held hostage by a slow host (the child never touches the GPU).  The child prints one JSON line per finished leg (each a
    complete record): the last one that arrived counts, also when the full-shape leg ran out of time or memory."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--negs", str(args.negs), "--seq", str(args.seq),
           "--bs", str(args.bs)] + (["--cpu-baseline-bounded"] if args.cpu_baseline_bounded else [])
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    timeout = args.cpu_baseline_timeout + (0 if args.cpu_baseline_bounded else 600)
    fail = {"value": None, "unit": "samples/sec", "cores": usable_cores(), "kind": "port", "cpu_model": cpu_model()}

    def last_record(stdout):
        for line in reversed((stdout or "").splitlines()):
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    continue
        return None
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        rec = last_record(out.stdout)
        if rec is not None:
            if out.returncode != 0:
                rec["note"] = f"the CPU child exited with code {out.returncode} after this leg"
            return rec
        return dict(fail, sample="CPU leg failed: " + (out.stderr.strip().splitlines() or ["no output"])[-1][:200])
    except subprocess.TimeoutExpired as ex:
        so = ex.stdout.decode() if isinstance(ex.stdout, bytes) else ex.stdout
        rec = last_record(so)
        if rec is not None:
            rec["note"] = f"the full-shape leg did not finish within {timeout:.0f} s"
            return rec
        return dict(fail, sample=f"CPU oracle did not finish within {timeout:.0f} s")


def _oracle_steps(O, oc, nq, k, S, Sq, steps, seed, lr=2e-5):
    """`steps` optimisation steps of the CPU oracle (forward + backward + AdamW, fp32, dropout 0.1) over consecutive
    batches of nq queries x k documents x seq S of the synthetic generator; returns the per-step wall times"""
    from scripts.dataset.synthetic import PreTokenizedCollator, SyntheticTriplesDataset
    params = {n: v.requires_grad_(True) for n, v in O.init_params(oc, seed=0).items()}
    g = torch.Generator().manual_seed(7)
    idf = torch.exp(torch.rand(oc.vocab_size, generator=g) * 6.66 - 3.9)
    n_batches = min(steps, 16)
    ds = SyntheticTriplesDataset(nq * n_batches, k, S, Sq, oc.vocab_size, seed=seed)
    coll = PreTokenizedCollator()
    lc = O.LossConfig(loss_types=("infonce",), use_in_batch_negatives=True, flops_d_lambda=0.05, flops_d_T=200)
    m = {n: torch.zeros_like(v) for n, v in params.items()}
    v2 = {n: torch.zeros_like(v) for n, v in params.items()}
    times = []
    for step in range(steps):
        b = step % n_batches
        batch = coll([ds[b * nq + i] for i in range(nq)])
        q, d = batch["query"][0], batch["docs"][0]
        t0 = time.perf_counter()
        loss = O.compute_loss(params, oc, idf, [0, 100, 101, 102, 103], q["input_ids"], q["attention_mask"], d["input_ids"],
                              d["attention_mask"], None, lc, step, dropout_p=0.1, gen=g)[0]
        loss.backward()
        with torch.no_grad():
            for n, p in params.items():
                if p.grad is not None:
                    O.adamw_step(p, p.grad, m[n], v2[n], step + 1, lr)
                    p.grad = None
        times.append(time.perf_counter() - t0)
    return times


def cpu_baseline_child(args):
    """CPU oracle (the 'port'; oracle/sparse_oracle.py, validated against the reference through tests/golden) timed on
    this box's host cores:
      * `bounded`: BASELINE configs[1]'s shape (same model, seq 128, 16 docs per query) on a BOUNDED sample -- 2 queries per
        step instead of 32 (one step is a few seconds of host work), 3 steps, mean of steps 2-3;
      * `c1_full`: BASELINE configs[0] (the reference's own CPU-runnable case) IN FULL: 64 triples = 16 steps of bs 4 x
        (1 pos + 1 neg) x seq 64, mean of steps 2-16;
      * `value`: configs[1] at FULL shape, 2 steps, the second one (~36 GB of host RAM); with --cpu-baseline-bounded, or if
        the full-shape leg does not finish, the bounded sample's rate.
    One JSON line per finished leg, each a complete record (the parent takes the last)."""
    from oracle import sparse_oracle as O  # baseline leg only

    cores = usable_cores()
    torch.set_num_threads(cores)
    oc = O.BertShape()
    k = args.negs + 1
    nq = 2
    t = _oracle_steps(O, oc, nq, k, args.seq, 32, 3, seed=1234)
    mean = sum(t[1:]) / len(t[1:])
    bounded = {"value": nq / mean, "unit": "samples/sec", "s_per_step": mean,
               "sample": f"configs[1] model/seq/docs-per-query, {nq} queries x {k} docs per step (1/{args.bs // nq} of the GPU batch), "
                         f"3 steps, mean of steps 2-3"}
    out = {"value": bounded["value"], "unit": "samples/sec", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
           "sample": "CPU oracle (torch fp32, dropout 0.1, AdamW), " + bounded["sample"] + f" = {mean:.2f} s/step", "bounded": bounded}
    print(json.dumps(out), flush=True)
    t1 = _oracle_steps(O, oc, 4, 2, 64, 16, 16, seed=4321)
    m1 = sum(t1[1:]) / len(t1[1:])
    out["c1_full"] = {"value": 4 / m1, "unit": "samples/sec", "s_per_step": m1,
                      "sample": "configs[0] in full: 64 triples = 16 steps of bs 4 x (1 pos + 1 neg) x seq 64, mean of steps 2-16"}
    print(json.dumps(out), flush=True)
    if not args.cpu_baseline_bounded:
        t2 = _oracle_steps(O, oc, args.bs, k, args.seq, 32, 2, seed=1234)
        out["value"] = args.bs / t2[-1]
        out["sample"] = (f"CPU oracle (torch fp32, dropout 0.1, AdamW), configs[1] at FULL shape: {args.bs} queries x {k} docs x seq "
                         f"{args.seq} per step (the GPU batch), 2 steps, the second one = {t2[-1]:.1f} s/step")
        out["c2_full"] = {"value": out["value"], "unit": "samples/sec", "s_per_step": t2[-1]}
    return out


def measured_peaks(device):
    """On-box peaks beside the vendor ones: a bare bf16 MFMA loop (random operands in registers, one wave per
    SIMD on every CU) and a 16-byte-per-lane streaming copy of 1 GiB, both timed with HIP events on the launch stream."""
    from sparse_hip import lib as L
    sink = torch.zeros(1024, device=device)
    blocks, iters = 256 * 4, 4096
    st = L.stream_ptr()
    L.call("sm_peak_mfma_bf16", L.ptr(sink), blocks, 64, st)  # warm
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the shader clock under the bare loop (sm_clock_stamp, ABI 7): the loop issues an MFMA in every matrix-pipe slot, so its rate over
    # the vendor peak x 2.4 GHz must be the clock the stamps read -- the cross-check of the stamps themselves
    stamps = torch.zeros(2 * 4096, dtype=torch.int64, device=device)
    L.call("sm_clock_stamp", L.ptr(stamps), 0, st)
    e0.record()
    L.call("sm_peak_mfma_bf16", L.ptr(sink), blocks, iters, st)
    e1.record()
    L.call("sm_clock_stamp", L.ptr(stamps), 1, st)
    torch.cuda.synchronize()
    mfma = blocks * 4 * iters * 16 * 16384 / (e0.elapsed_time(e1) * 1e-3)
    sa, sb = stamps.cpu().view(2, 2048, 2)
    ok = (sa[:, 1] > 0) & (sb[:, 1] > sa[:, 1]) & (sb[:, 0] > sa[:, 0])
    clk = float(((sb[ok, 0] - sa[ok, 0]).double() / (sb[ok, 1] - sa[ok, 1]).double()).median() * 0.1) if int(ok.sum()) >= 8 else None
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    L.call("sm_peak_copy", L.ptr(src), L.ptr(dst), n, st)
    best = 1e9
    for _ in range(3):
        e0.record()
        L.call("sm_peak_copy", L.ptr(src), L.ptr(dst), n, st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del src, dst
    return {"mfma_bf16_tflops": mfma / 1e12, "hbm_copy_gbs": 2 * n / (best * 1e-3) / 1e9,
            "mfma_loop_clock_ghz": clk, "mfma_loop_clock_implied_by_its_rate_ghz": mfma / 2.5e15 * 2.4,
            "how": "bare v_mfma_f32_16x16x32_bf16 loop, random operands in registers, 1024 WGs x 4 waves; float4 copy of 1 GiB (read + write bytes)"}


def latest_traffic(kernel_names, layout):
    """HBM-side bytes per launch of the roofline kernel from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same command, tools/pmc_summary.py); the git revision it was measured at is reported so a
    stale figure is visible.  PMC counters cannot be read from inside this process."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get("layout", "ragged") != layout or pmc.get("csrc_sha") != csrc_sha():  # other layout / other kernels: not reported
                continue
            for name in kernel_names:
                if name in pmc["kernels"]:
                    return (pmc["kernels"][name]["bytes_per_launch"], f"{os.path.relpath(path, ROOT)}: {pmc['source']}",
                            pmc.get("git", "unknown (round 1)"), name)
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None, None


GEMM_KERNELS = ("gemm_nt_kernel", "gemm_nt192_kernel", "gemm_ws_kernel", "gemm_tn_pc_kernel", "gemm_tn_kernel", "gemm_tn2_kernel",
                "gemm_tn3_kernel", "ffn_pc_fwd_kernel", "ffn_pc_bwd_kernel")


def csrc_sha() -> str:
    """hash of the kernel sources: a PMC summary is only reported beside a bench line of the SAME kernels (tools/pmc_summary.py
    stores this value; round 3's line carried a traffic figure measured before a kernel was added)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp", ".inc")):  # (.inc: the generated main loops of the grouped weight-gradient kernels)
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def latest_gemm_traffic(layout):
    """HBM-side bytes per training step of all encoder GEMM launches, from the newest committed PMC summary that records the
    number of steps it profiled and ran on `layout` (tools/profile_round.sh): sum over the GEMM kernels of launches x bytes per launch"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if not pmc.get("steps") or pmc.get("layout") != layout or pmc.get("csrc_sha") != csrc_sha():
                continue
            ks = {n: v for n, v in pmc["kernels"].items() if any(n.startswith(g) for g in GEMM_KERNELS)}
            if ks:
                total = sum(v["launches"] * v["bytes_per_launch"] for v in ks.values()) / pmc["steps"]
                return total, f"{os.path.relpath(path, ROOT)}: {pmc['source']}", pmc.get("git"), sorted(ks)
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None, None


def latest_step_traffic(layout):
    """HBM-side bytes per training step over EVERY kernel of the step (same PMC summary as latest_gemm_traffic; the bench's own
    calibration kernels, peak_* / clock_stamp, and the one-off random-number / index kernels of the batch preparation left out)"""
    import glob
    skip = ("peak_", "clock_stamp", "void at::native::distribution", "void at::native::write_indices", "void at::native::index_elementwise",
            "void rocprim")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if not pmc.get("steps") or pmc.get("layout") != layout or pmc.get("csrc_sha") != csrc_sha():
                continue
            ks = {n: v for n, v in pmc["kernels"].items() if not any(n.startswith(x) for x in skip)}
            total = sum(v["launches"] * v["bytes_per_launch"] for v in ks.values()) / pmc["steps"]
            top = sorted(((v["launches"] * v["bytes_per_launch"] / pmc["steps"], n) for n, v in ks.items()), reverse=True)[:6]
            return total, [{"kernel": n, "gb_per_step": b / 1e9} for b, n in top], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def sparse_regime_leg(trainer, bs_, layout, density=0.01, steps=15):
    """The step in the regime a TRAINED sparse encoder lives in (config_infonce.yaml:5 fine-tunes one): the decoder bias is shifted
    down until ~1 % of the (document, vocabulary) activations are alive (the share before the shift is reported too), learning rate 0, then `steps` steps
    are timed.  Only the kernels whose work depends on the activation pattern change (head backward: rows gathered, all-zero
    G slices skipped).  The bias and the learning rate are restored afterwards."""
    bb = trainer.model.sparse_model.backbone
    bias = bb.view("cls.predictions.bias")
    # everything the leg touches is snapshotted and restored in `finally`: the weights and the AdamW moments (the moments move even
    # at learning rate 0), the step counter, the learning rate, the density probe
    keep, lr, gstep = bias.detach().clone(), trainer.args.learning_rate, trainer.state.global_step
    flat = bb.flat_param.detach().clone()
    adam = None if trainer._adam is None else {k: v.detach().clone() for k, v in trainer._adam.items()}
    mavg = trainer.ranking_loss_moving_avg
    try:
        def alive():
            with torch.no_grad():
                bb.eval()
                b = bs_[0]["docs"][0]
                rep = bb.encode(b["input_ids"][:64].to(bb.device), b["attention_mask"][:64].to(bb.device))
                bb.train()
            return float((rep > 0).float().mean())

        a0 = alive()
        lo, hi = 0.0, 10.0
        for _ in range(12):  # bisection on the shift
            mid = (lo + hi) / 2
            with torch.no_grad():
                bias.copy_(keep - mid)
            bb.mark_weights_dirty()
            lo, hi = (mid, hi) if alive() > density else (lo, mid)
        trainer.args.learning_rate = 0.0
        a1 = alive()
        bb._density, bb._density_tick, bb._density_probe = None, 0, None  # the density-adaptive head backward re-measures at once (it samples every 8th encode)
        for i in range(3):
            trainer.training_step(bs_[i % len(bs_)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            trainer.training_step(bs_[i % len(bs_)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        dens = bb._density
        return {"head_backward_form": "scatter over the live entries" if (dens is not None and dens < bb.dt_scatter_density and bb.dt_scatter) else "matrix form",
                "what": "the same step with the decoder bias shifted until ~1 % of the sparse activations are alive (a trained checkpoint's "
                        "regime; alive_fraction_random_init = the share before the shift), learning rate 0", "layout": layout, "alive_fraction_random_init": a0,
                "alive_fraction": a1, "ms_per_step": ms, "samples_per_sec": 32e3 / ms, "steps": steps}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        torch.cuda.synchronize()
        with torch.no_grad():
            bb.flat_param.copy_(flat)
            if adam is None:
                trainer._adam = None
            else:
                for k, v in adam.items():
                    trainer._adam[k].copy_(v)
        bb.mark_weights_dirty()
        trainer.args.learning_rate = lr
        trainer.state.global_step = gstep
        trainer.ranking_loss_moving_avg = mavg
        bb._density, bb._density_tick, bb._density_probe = None, 0, None


def seq_sweep_leg(args, device, base_ms, steps=12, warmup=4):
    """configs[1]'s model and recipe at the sequence lengths the reference's shipped recipes train at (max_seq_length 512 / 256:
    config_infonce.yaml:9, config_l0.yaml:9, config_kd.yaml:9) with the SAME number of token rows per step as the headline
    (65 536 padded rows: 16 queries x 16 documents x 256, 8 x 16 x 512; document lengths scaled with the sequence length), dense
    layout like `value`: tokens/s beside the S = 128 figure.  Outside every timed region of the headline; its own trainers."""
    import copy
    out = {"what": "same model / recipe / token rows per step as `value` (dense layout, dropout on, optimiser included) at longer documents; "
                   "tokens = padded token rows of the document encoder", "steps": steps, "warmup": warmup,
           "seq128": {"ms_per_step": base_ms, "tokens_per_sec": args.bs * (args.negs + 1) * args.seq / (base_ms * 1e-3)}}
    for S in (256, 512):
        try:
            a = copy.copy(args)
            a.seq, a.bs, a.len_scale = S, args.bs * args.seq // S, S / 128.0
            trainer, _, batches = build_trainer(a, device, 0, layouts=("dense", "ragged"))
            ms_of, last = {}, None
            for layout in ("dense", "ragged"):  # dense = the headline's protocol; ragged (padding tokens skipped, identical outputs) beside it
                trainer.model.sparse_model.backbone.varlen = layout == "ragged"
                bs_ = batches[layout]
                for i in range(warmup):
                    trainer.training_step(bs_[i % len(bs_)])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    last = trainer.training_step(bs_[i % len(bs_)])
                torch.cuda.synchronize()
                ms_of[layout] = (time.perf_counter() - t0) / steps * 1e3
            ms = ms_of["dense"]
            toks = a.bs * (a.negs + 1) * S
            out[f"seq{S}"] = {"queries": a.bs, "docs": a.bs * (a.negs + 1), "ms_per_step": ms, "tokens_per_sec": toks / (ms * 1e-3),
                              "vs_seq128": (toks / ms) / (args.bs * (args.negs + 1) * args.seq / base_ms), "loss": float(last),
                              "finite": bool(torch.isfinite(last).item()), "ms_per_step_ragged_layout": ms_of["ragged"],
                              "padded_tokens_per_sec_ragged_layout": toks / (ms_of["ragged"] * 1e-3)}
            del trainer, batches, bs_
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            out[f"seq{S}"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def c5_leg(timeout_s=240):
    """BASELINE configs[4] at its full PER-GPU shape (bert-base student, 64 queries x 31 documents, seq 512, kd on precomputed scores,
    fp8 operands in the encoder linears, rep-level gradient caching) in a child process: tools/c5_shape_smoke.py, 2 timed steps."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c5_shape_smoke.py"), "64", "248", "2", "fp8"], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
        line = [l for l in r.stdout.splitlines() if "ms/step" in l]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-400:]}
        import re
        m = re.search(r"(\d+) ms/step = ([\d.]+) samples/s.*peak memory ([\d.]+) GiB", line[-1])
        return {"what": "configs[4] per-GPU shape on ONE GPU (tools/c5_shape_smoke.py 64 248 2 fp8)", "ms_per_step": float(m.group(1)),
                "samples_per_sec": float(m.group(2)), "peak_memory_gib": float(m.group(3)), "line": line[-1]}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    args = parse()
    if os.environ.get("SM_FAULTHANDLER_S"):  # tests: a stalled rank prints every thread's stack and exits instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SM_FAULTHANDLER_S"]), exit=True)
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_child(args)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    ndev = torch.cuda.device_count()
    dev_index = local % max(1, ndev)  # SM_BENCH_BACKEND=gloo lets N ranks share one GPU (functional check of the N > 1 path)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    single = args.single_rank_rccl and world == 1
    if single:  # the N > 1 code path with a communicator of one rank (scripts/train/trainer.py ProcessInfo.distributed)
        os.environ["SM_DIST_SINGLE_RANK"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    dist_on = world > 1 or single
    if dist_on:
        backend = os.environ.get("SM_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        kw = {"rank": 0, "world_size": 1} if single else {}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    if os.environ.get('SM_LIB'):  # A/B two builds of the shared library in one run (development aid)
        from sparse_hip import lib as _L
        _L._LIB_PATH = os.environ['SM_LIB']
    from sparse_hip import ops
    other = "dense" if args.layout == "ragged" else "ragged"
    trainer, cfg, batches = build_trainer(args, device, rank, layouts=(args.layout,) if args.only_value_layout else (args.layout, other))

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    live = {}

    def timed(layout, kt=None):
        """W untimed + exactly K timed steps bracketed by barrier + synchronize; max over ranks.  The loss of every timed step
        stays on the device (no sync inside the region) and is read afterwards: a run whose parameters went non-finite, or whose
        loss stopped moving, is not a measurement (GPUTEST_r03: relu(NaN) = 0 pins the loss at ln(columns) without a NaN in it)"""
        bs_ = batches[layout]
        for i in range(args.warmup):
            trainer.training_step(bs_[i % len(bs_)])
        barrier()
        if kt is not None:
            kt.enabled = True
        losses, totals = [], []
        t0 = time.perf_counter()
        for i in range(args.steps):
            totals.append(trainer.training_step(bs_[i % len(bs_)]))
            losses.append(trainer._last.get("ranking", totals[-1]))  # device scalar of the step just enqueued (the FLOPS term's weight is still warming up:
        barrier()                                    # the TOTAL loss grows with it; the ranking loss is what shows the model is alive)
        el = time.perf_counter() - t0
        bb_ = trainer.model.sparse_model.backbone
        ls = [float(x) for x in torch.stack([l.reshape(()) for l in losses]).float().cpu()]
        tt = [float(x) for x in torch.stack([l.reshape(()) for l in totals]).float().cpu()]
        live[layout] = {"loss_first": ls[0], "loss_last": ls[-1], "loss_min": min(ls), "loss_max": max(ls), "total_loss_first": tt[0],
                        "total_loss_last": tt[-1], "distinct_losses": len(set(ls)),
                        "finite": bool(all(l == l and abs(l) != float("inf") for l in ls)) and bool(torch.isfinite(bb_.flat_param).all())
                        and not bb_.nonfinite_report(grad=False)}
        if kt is not None:
            kt.enabled = False
        tmax = torch.tensor([el], device=device, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    # N > 1: `value` is measured with north_star's exchange -- the reference's RCCL all-gather of the document representations
    # (SM_EXCHANGE=gather, scripts/utils.py:16-23) -- whatever the caller's environment says; the score-block exchange
    # (sparse_hip.functional.distributed_loss) is timed right after it on the same ranks and reported BESIDE it.
    env_exchange = os.environ.get("SM_EXCHANGE")
    if dist_on:
        os.environ["SM_EXCHANGE"] = "gather"
    with KernelTimer(ops, "sparse_head_fwd") as kt:
        elapsed = timed(args.layout, kt)   # <- the line's `value`
        head_ms = kt.mean_ms()
        rows = list(kt.rows)
    elapsed_scores = None
    if dist_on:
        os.environ["SM_EXCHANGE"] = "scores"
        try:
            elapsed_scores = timed(args.layout)
            live["scores_exchange"] = live[args.layout]
        finally:
            os.environ["SM_EXCHANGE"] = "gather"
        live[args.layout] = None
        elapsed_again = timed(args.layout)  # liveness record of the value layout in gather mode again (and a second gather sample)
    elapsed_other = float("nan") if args.only_value_layout else timed(other)  # the other layout, outside the headline region
    gemm_lines = gemm_lines_serial = None
    if world == 1 and not single and not args.no_gemm_roofline:  # outside the timed region: does not touch `value`
        with GemmRoofline(ops) as gr:
            bs_ = batches[args.layout]
            for i in range(3):
                trainer.training_step(bs_[i % len(bs_)])
            torch.cuda.synchronize()
        gemm_lines = gr.summary(3, MFMA_PEAK[args.dtype])
        other_clock_lines = gr.other
        # the same three steps with the weight-gradient launches on the main queue: every GEMM alone on the chip.  In the
        # overlapped step a GEMM's time includes what its neighbour on the other queue takes from it.
        wg = getattr(trainer.model.sparse_model.backbone, "_wgrad", None)
        gemm_lines_serial = None
        if wg is not None and wg.enabled:
            wg.enabled = False
            try:
                with GemmRoofline(ops) as gr2:
                    for i in range(3):
                        trainer.training_step(bs_[i % len(bs_)])
                    torch.cuda.synchronize()
                gemm_lines_serial = gr2.summary(3, MFMA_PEAK[args.dtype])
                other_clock_serial = gr2.other
            finally:
                wg.enabled = True
    barrier()

    k = args.negs + 1
    T_padded = args.bs * k * args.seq
    T = sum(rows) / max(1, len(rows))  # rows per launch: the ragged layout skips padding tokens
    H, V = cfg.hidden_size, cfg.vocab_size
    head_flops = 2.0 * T * H * V  # executed FLOPs of one fused decoder launch (2THV, T = computed rows)
    achieved = head_flops / (head_ms * 1e-3)
    peak = MFMA_PEAK[args.dtype]
    traffic, traffic_src, traffic_git, traffic_kernel = None, None, None, None
    if args.bs == 32 and args.negs == 15 and args.seq == 128 and args.dtype == "bf16":
        traffic, traffic_src, traffic_git, traffic_kernel = latest_traffic(HEAD_KERNELS, args.layout)
    sps = lambda el: world * args.bs * args.steps / el
    result = {
        "metric": "training samples/sec (q+1pos+15neg, seq128)",
        "value": sps(elapsed),
        "unit": "samples/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic MS-MARCO-shaped triples, random-init weights",
        "value_layout": args.layout,
        # liveness of the timed region itself: the RANKING loss of the first / last timed step on rank 0
        "loss_first": live[args.layout]["loss_first"], "loss_last": live[args.layout]["loss_last"],
        "finite": live[args.layout]["finite"] and live[args.layout]["distinct_losses"] > min(2, args.steps - 1),
        "liveness": live,
        "value_dense_layout": sps(elapsed if args.layout == "dense" else elapsed_other),
        "value_ragged_layout": sps(elapsed if args.layout == "ragged" else elapsed_other),
        "config": {"workload": "configs[1]: config_infonce.yaml recipe, v2-mini-shaped encoder (6L/384H/12A/1536I/V30522), "
                               f"bs={args.bs} x (1 pos + {args.negs} negs), seq {args.seq}, inference-free queries, "
                               "InfoNCE in-batch negatives + FLOPS, dropout " + ("off" if args.no_dropout else "0.1") + ", fused AdamW" + (", fp32 residual stream (bf16 GEMM operands)" if not args.bf16_storage else ", all-bf16 activation storage") + "; "
                               f"documents (lengths ~N(80,30) in [16,{args.seq}]) padded to {args.seq} by the collator; `value` is the "
                               + (f"ragged layout: padding tokens skipped on the device ({T:.0f} of {T_padded} token rows computed per step, "
                                  "identical outputs); value_dense_layout computes all of them" if args.layout == "ragged" else
                                  f"dense layout: all {T_padded} token rows computed (attention skips key tiles that hold only masked keys: exact); value_ragged_layout skips padding tokens"),
                   "global_batch": world * args.bs, "docs_per_query": k, "seq_len": args.seq,
                   "parallelism": f"dp{world}" + ("" if world == 1 else " (in-batch negatives across ranks over RCCL; `value`: exchange mode "
                                                      "gather = the reference's all-gather of the document representations, every rank "
                                                      "evaluates the loss head on the gathered batch; value_scores_exchange: all-gather of the "
                                                      "queries (under the document encoder) and of the score blocks + all-reduce of the FLOPS "
                                                      "column means; both: flat-gradient all-reduce in slices overlapped with backward)")},
    }
    if single:
        result["single_rank_rccl"] = ("the step through the N > 1 code path with a process group of ONE rank (SM_DIST_SINGLE_RANK=1): every collective "
                                      "of the distributed step (query all-gather on the communication stream, loss-head exchange, slice-wise gradient "
                                      "all-reduce from inside the backward) executed by the backend named in `dist`; no bytes cross a link -- this is "
                                      "the cost of the distributed plumbing on one GPU, a lower bound of the N-GPU step, NOT a scaling number")
    if dist_on:  # self-describing multi-GPU record: what carried the collectives
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001  (a build without the binding: say so instead of failing the run)
            ver = f"unavailable ({type(e).__name__})"
        result["value_scores_exchange"] = sps(elapsed_scores)
        result["ms_per_step_scores_exchange"] = elapsed_scores / args.steps * 1e3
        result["value_gather_second_sample"] = sps(elapsed_again)
        result["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
                          "exchange": "gather (value), scores (value_scores_exchange)", "exchange_env_at_start": env_exchange, "gpus_visible": ndev,
                          "env": {k: os.environ[k] for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_ALGO", "NCCL_PROTO") if k in os.environ}}
    head_line = {"kernel": "fused MLM decoder + seq-max + log1p(relu) (sm_sparse_head_fwd)", "bound": "mfma",
                 "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                 "traffic": traffic, "traffic_unit": "bytes/launch (HBM-side reads x2-corrected + writes)", "traffic_source": traffic_src,
                 "traffic_measured_at_git": traffic_git, "traffic_kernel": traffic_kernel,
                 "kernel_ms": head_ms, "rows_per_launch": T}
    if gemm_lines is not None:
        # the step's dominant kernel class: every encoder GEMM launch of a step (forward, input gradients, weight gradients; the
        # fused feed-forward forward counted by its two GEMMs), timed with HIP events on the stream each launch runs on, in the
        # step as it runs (weight gradients on the side queue share the chip with the backward chain)
        fl = sum(g["gflop_per_step"] for g in gemm_lines)
        ms = sum(g["ms_per_step"] for g in gemm_lines)
        result["roofline"] = {"kernel": "encoder GEMMs, in-step (all launches of sm_gemm_nt / sm_gemm_nt_ln_bwd / sm_gemm_tn_acc / sm_gemm_tn_group / "
                                        "sm_ffn_pc_fwd / sm_ffn_pc_bwd of one training step)",
                              "bound": "mfma", "achieved": fl / ms, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": fl / ms * 1e12 / peak,
                              "traffic": None, "gflop_per_step": fl, "ms_per_step": ms, "per_op": gemm_lines,
                              "other_kernels_clock": other_clock_lines,
                              "clock_how": "sm_clock_stamp before and after every launch on its stream: d(s_memtime) / d(s_memrealtime) x 100 MHz, "
                                           "median over the compute units both stamps reached; in-step = both queues running; one_queue.per_op_clock_ghz = each kernel alone"}
        if args.bs == 32 and args.negs == 15 and args.seq == 128 and args.dtype == "bf16":
            tr, src, tgit, tk = latest_gemm_traffic(args.layout)
            result["roofline"].update({"traffic": tr, "traffic_unit": "bytes/step over all encoder GEMM launches (HBM-side reads x2-corrected "
                                                                     "+ writes)", "traffic_source": src, "traffic_measured_at_git": tgit,
                                       "traffic_kernels": tk})
        if args.bs == 32 and args.negs == 15 and args.seq == 128 and args.dtype == "bf16":
            # the OTHER roofline of the step: HBM-side bytes of all its kernels (PMC summary of the same kernel sources) over the
            # measured step time -- the step moves ~33 GB, which is a larger share of the HBM rate than its FLOPs are of the MFMA rate
            st_bytes, st_top, st_src = latest_step_traffic(args.layout)
            if st_bytes:
                sec = elapsed / args.steps
                result["roofline"]["hbm_view_of_the_step"] = {
                    "bound": "hbm", "traffic": st_bytes, "unit": "GB/s", "achieved": st_bytes / sec / 1e9, "peak": 8000.0,
                    "frac": st_bytes / sec / 8e12, "frac_of_measured_copy_rate": None, "largest": st_top, "traffic_source": st_src,
                    "what": "HBM-side bytes per step over every kernel of the step (FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits counted) / ms_per_step"}
        if gemm_lines_serial:
            fl2 = sum(g["gflop_per_step"] for g in gemm_lines_serial)
            ms2 = sum(g["ms_per_step"] for g in gemm_lines_serial)
            result["roofline"]["one_queue"] = {
                "how": "same ops, weight-gradient launches on the main queue (no kernel shares the chip with another)",
                "per_op_tflops": {g["op"]: g["achieved_tflops"] for g in gemm_lines_serial},
                "per_op_clock_ghz": {g["op"]: g["clock_ghz"] for g in gemm_lines_serial + other_clock_serial},
                "gflop_per_step": fl2, "ms_per_step": ms2, "achieved": fl2 / ms2, "frac": fl2 / ms2 * 1e12 / peak}
        result["roofline_head_fwd"] = head_line
    else:  # N > 1 or --no-gemm-roofline: the largest single kernel of the step
        result["roofline"] = head_line
    if world == 1 and not single and not args.no_extras and args.bs == 32 and args.negs == 15 and args.seq == 128 and args.dtype == "bf16":
        # two more records beside `value` (outside every timed region above; each guarded: a failure here never loses the line)
        result["sparse_regime"] = sparse_regime_leg(trainer, batches[args.layout], args.layout)
        result["c5_per_gpu"] = c5_leg()
        result["seq_sweep"] = seq_sweep_leg(args, device, elapsed / args.steps * 1e3 if args.layout == "dense" else elapsed_other / args.steps * 1e3)
    if rank == 0:
        if world == 1:
            pm = measured_peaks(device)
            result["roofline"]["peak_measured"] = pm
            result["roofline"]["frac_of_measured_peak"] = (result["roofline"]["achieved"] / pm["mfma_bf16_tflops"]
                                                           if args.dtype == "bf16" else None)
            if "hbm_view_of_the_step" in result["roofline"]:
                hv = result["roofline"]["hbm_view_of_the_step"]
                hv["frac_of_measured_copy_rate"] = hv["achieved"] / pm["hbm_copy_gbs"]
        if world == 1 and not single and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if not result["finite"]:
        print("bench.py: the timed region did not train (non-finite parameters / losses, or a loss that does not move): " + json.dumps(live),
              file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
