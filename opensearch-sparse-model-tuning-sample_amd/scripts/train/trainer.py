"""ModelWrapper and SparseModelTrainer with the reference's surface
(scripts/train/trainer.py:18-218) driving the MI355X kernels.

The reference subclasses ``transformers.Trainer``; the step it executes
(hf trainer.py:1740-1800: H2D copy -> compute_loss -> backward (+ DDP all-reduce) ->
optimizer.step -> scheduler.step -> zero_grad -> log/save) is restated here as a small
loop so that (i) the data-parallel gradient all-reduce runs on a side HIP stream over
contiguous slices of the flat gradient buffer, overlapped with the rest of backward,
(ii) AdamW is one fused kernel over the flat parameter buffer, and (iii) no step performs
a device->host sync (the reference's per-step ``.item()`` at trainer.py:121 is replaced by
a device-side moving average that is only read when a log line is due).
"""
from __future__ import annotations

import json
import logging
import os
from typing import Optional

import torch
import torch.distributed as dist

from sparse_hip import functional as F
from sparse_hip import ops
from ..utils import gather_rep
from .loss import InfoNCELoss, KLDivLoss, MarginMSELoss

logger = logging.getLogger(__name__)


class ModelWrapper(torch.nn.Module):
    """One dict in, (d_rep, q_rep) out -- reference trainer.py:18-49."""

    def __init__(self, sparse_model, inf_free=True):
        super().__init__()
        self.sparse_model = sparse_model
        self.inf_free = inf_free
        self.on_q_rep = None  # N > 1: called with q_rep BEFORE the document encoder runs (the trainer starts its all-gather there)

    def forward(self, inputs):
        q_rep = None
        if self.inf_free and self.on_q_rep is not None:  # inference-free queries do not touch the encoder: the order is free
            q_rep = self.sparse_model(inf_free=True, input_ids=inputs["q_input_ids"], attention_mask=inputs["q_attention_mask"])
            self.on_q_rep(q_rep)
        d_rep = self.sparse_model(inf_free=False, input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                                  packed=inputs.get("packed"), grad_cache_chunks=inputs.get("grad_cache_chunks"))
        if q_rep is None:
            q_rep = self.sparse_model(inf_free=self.inf_free, input_ids=inputs["q_input_ids"],
                                      attention_mask=inputs["q_attention_mask"])
        return d_rep, q_rep

    def save(self, output_dir, **kwargs):
        kwargs.pop("state_dict", None)
        self.sparse_model.backbone.save_pretrained(output_dir, **kwargs)
        tok = self.sparse_model.tokenizer
        if tok is not None:
            tok.save_pretrained(output_dir)
        if self.sparse_model.idf_requires_grad:
            idf_vector = self.sparse_model.idf_vector.detach().cpu()
            idf_json = {}
            for idx in idf_vector.nonzero().flatten().tolist():
                key = tok._convert_id_to_token(idx) if tok is not None else str(idx)
                idf_json[key] = float(idf_vector[idx])
            with open(os.path.join(output_dir, "idf.json"), "w") as f:
                json.dump(idf_json, f)


class TrainerState:
    def __init__(self):
        self.global_step = 0


class ProcessInfo:
    """The slice of accelerate.Accelerator the reference reads (num_processes,
    local_process_index, is_main_process, gather, unwrap_model)."""

    def __init__(self, device):
        on = dist.is_available() and dist.is_initialized()
        self.num_processes = dist.get_world_size() if on else 1
        self.process_index = dist.get_rank() if on else 0
        self.local_process_index = int(os.environ.get("LOCAL_RANK", self.process_index)) if on else 0
        # single-node launch: the reference indexes the gathered tensor by the LOCAL rank
        if on and self.local_process_index != self.process_index:
            raise RuntimeError("multi-node runs are not supported (gather_rep indexes by local rank, utils.py:21)")
        self.is_main_process = self.process_index == 0
        self.device = device
        # The N > 1 code path -- collectives on the communication stream, gradient slices all-reduced from inside the backward, the
        # loss-head exchange -- with a communicator of ONE rank: SM_DIST_SINGLE_RANK=1 and an initialised process group.  Every collective
        # then runs for real through the backend (RCCL with backend "nccl": the only way that path can execute on a one-GPU box) and the
        # step must equal the plain single-process step.  Never set in production runs.
        self.distributed = self.num_processes > 1 or (on and os.environ.get("SM_DIST_SINGLE_RANK", "0") == "1")

    def gather(self, t):
        return gather_rep(t.detach(), self)

    def unwrap_model(self, model):
        return model


def linear_schedule_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """transformers.get_linear_schedule_with_warmup (train_ir.py:103-107) at optimiser step `step`."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


class SparseModelTrainer:
    def __init__(self, model_args, data_args, loss_functions, **kwargs):
        self.model_args = model_args
        self.data_args = data_args
        self.loss_functions = loss_functions
        self.args = kwargs["args"]
        sparse_model = kwargs["model"]
        self.model = ModelWrapper(sparse_model, model_args.inf_free)
        self.train_dataset = kwargs.get("train_dataset")
        self.data_collator = kwargs.get("data_collator")
        self.optimizer, self.lr_scheduler = kwargs.get("optimizers", (None, None)) or (None, None)
        self.state = TrainerState()
        self.accelerator = ProcessInfo(sparse_model.backbone.device)
        self._ma = torch.zeros(1, device=sparse_model.backbone.device)  # device-side moving average
        self._last = {}
        self._adam = None
        self._comm_stream = None
        self._pending = []
        self._q_prefetch = None
        self._in_compute_loss = False
        self._step_done = []  # events at the end of the last steps (training_step: bounded host lead)
        if self.accelerator.distributed:
            if (sparse_model.backbone.device.type == "cuda" and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8
                    and dist.get_backend() == "nccl"):  # (gloo = several ranks on one GPU in the tests: keep the default there)
                # (train_ir.py and bench.py set it before HIP starts; a caller that builds the trainer itself has to)
                logger.warning("N > 1 with GPU_MAX_HW_QUEUES < 8: the gradient all-reduce's wait for the weight gradients can share a hardware "
                               "queue with the backward chain and stall it (~0.3 ms per step); export GPU_MAX_HW_QUEUES=8 before the process starts")
            self._setup_grad_overlap()
            if sparse_model.backbone.device.type == "cuda":
                self.model.on_q_rep = self._prefetch_q_gather

    # ------------------------------------------------------------------ reference surface
    @property
    def ranking_loss_moving_avg(self) -> float:
        return float(self._ma.item())

    @ranking_loss_moving_avg.setter
    def ranking_loss_moving_avg(self, value: float) -> None:
        self._ma.fill_(float(value))

    def flops_value(self, representation, group_num=1):
        return F.flops_value(representation, group_num, self.data_args.flops_threshold)

    def get_lambda(self, lambda_value, lambda_T):
        if self.state.global_step >= lambda_T:
            return lambda_value
        step = self.state.global_step + 1
        return lambda_value * (step / lambda_T) ** 2

    def _prefetch_q_gather(self, q_rep):
        """N > 1: the all-gather of q_rep (reference trainer.py:101-104 gathers it after both encoders) is issued on the
        communication stream as soon as q_rep exists, i.e. under the document encoder; the loss waits for it where it reads
        the gathered queries.  Same collective, same place in every rank's collective order (the first of the step)."""
        if not self._in_compute_loss:  # a forward outside compute_loss (evaluation, a user's own call) must not launch a
            return                     # collective that nobody consumes
        q = q_rep.detach().contiguous()
        n = self.accelerator.num_processes
        q_all = torch.empty((n * q.shape[0],) + tuple(q.shape[1:]), dtype=q.dtype, device=q.device)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ev)
            work = dist.all_gather_into_tensor(q_all, q, async_op=True)
        self._q_prefetch = (q_all, work, q)  # q stays referenced until the wait: the side stream reads it

    def _take_q_prefetch(self):
        pre, self._q_prefetch = self._q_prefetch, None
        return pre[:2] if pre is not None else None

    def compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None):
        self._q_prefetch = None
        self._in_compute_loss = True
        try:
            return self._compute_loss(model, inputs, return_outputs)
        finally:
            self._in_compute_loss = False

    def _compute_loss(self, model, inputs, return_outputs=False):
        if hasattr(self, "bi_encoder_teacher"):
            inputs["scores"] = self.bi_encoder_teacher.get_scores_batch(
                q_features_list=inputs["query"][1:], d_features_list=inputs["docs"][1:])
        model_wrapper_input = {
            "q_input_ids": inputs["query"][0]["input_ids"],
            "q_attention_mask": inputs["query"][0]["attention_mask"],
            "input_ids": inputs["docs"][0]["input_ids"],
            "attention_mask": inputs["docs"][0]["attention_mask"],
            "packed": inputs["docs"][0].get("packed"),  # host-packed ragged layout (see _prepare_inputs)
            "grad_cache_chunks": self._grad_cache_chunks(inputs["docs"][0]),
        }
        d_rep, q_rep = model(model_wrapper_input)
        # inference-free queries have at most one non-zero per query token: let the losses use the
        # sparse-query score kernels (identical values, no V-length dense dot products)
        cap = model_wrapper_input["q_input_ids"].shape[1] if self.model_args.inf_free else None
        for loss_function in self.loss_functions:
            loss_function.sparse_query_cap = cap
        if self._use_fused_loss():
            return self._compute_loss_fused(d_rep, q_rep, inputs, cap, return_outputs)
        d_rep = gather_rep(d_rep, self.accelerator)
        q_rep = gather_rep(q_rep, self.accelerator, prefetched=self._take_q_prefetch())
        if "scores" in inputs:
            inputs["scores"] = gather_rep(inputs["scores"].to(d_rep.device, torch.float32), self.accelerator)
        d_flops = self.flops_value(d_rep, d_rep.shape[0] // q_rep.shape[0])
        flops_loss = d_flops * self.get_lambda(self.data_args.flops_d_lambda, self.data_args.flops_d_T)
        if not self.model_args.inf_free:
            flops_loss = flops_loss + self.flops_value(q_rep) * self.get_lambda(
                self.data_args.flops_q_lambda, self.data_args.flops_q_T)
        ranking_loss = 0
        for loss_function in self.loss_functions:
            ranking_loss = ranking_loss + loss_function.get_loss(q_rep=q_rep, d_rep=d_rep, inputs=inputs)
        # moving average kept on the device: ma = 0.01 * ranking + 0.99 * ma   (trainer.py:120-122)
        rl = ranking_loss.detach().reshape(1).float()
        ops.axpby(0.01, rl, 0.99, self._ma, self._ma)
        loss = ranking_loss + flops_loss
        outputs = {"q_rep": q_rep, "d_rep": d_rep}
        self._last = {"d_flops": d_flops.detach(), "flops_loss": flops_loss.detach(), "d_rep": d_rep.detach()}
        if self.state.global_step % self.args.logging_steps == 0:
            self._log_step()
        loss = loss * self.accelerator.num_processes  # DDP averages, trainer.py:139-141
        return (loss, outputs) if return_outputs else loss

    def _grad_cache_chunks(self, enc):
        """data_args.grad_cache_chunk > 0: the student's documents as chunks for HipBertMLM.encode_cached -- the host-packed
        chunks of _prepare_inputs when they exist, otherwise slices of the padded device tensors"""
        n = int(getattr(self.data_args, "grad_cache_chunk", 0) or 0)
        if n <= 0:
            return None
        if enc.get("packed_chunks") is not None:
            return enc["packed_chunks"]
        ids, mask = enc["input_ids"], enc["attention_mask"]
        return [(ids[a:a + n], mask[a:a + n], None) for a in range(0, ids.shape[0], n)]

    def _exchange_mode(self) -> str:
        """N > 1: data_args.dist_exchange / SM_EXCHANGE.
        "gather" (default) is the reference's form and the one north_star names -- RCCL all-gather of the document representations
        (utils.py:16-23) -- run as ONE autograd node (sparse_hip.functional.distributed_loss, exchange = "gather"): the all-gather goes
        out in row chunks and each chunk's FLOPS column sums and score block are computed while the next chunk is on the wire.
        "gather_ref" is the same exchange written exactly as the reference writes it (gather_rep of both representations, then the
        loss objects on the gathered batch): the parity form, also what a user-supplied loss class gets.
        "scores" (opt-in) exchanges queries, score blocks and FLOPS column means instead: same loss, same gradients, ~100x less
        traffic, every V-length loss kernel stays on the local documents."""
        mode = os.environ.get("SM_EXCHANGE", getattr(self.data_args, "dist_exchange", "gather"))
        if mode not in ("scores", "gather", "gather_ref"):
            raise KeyError(mode)
        return mode

    def _use_score_exchange(self) -> bool:
        return self.accelerator.distributed and self._exchange_mode() == "scores"

    _BUILTIN_LOSSES = {InfoNCELoss: "infonce", KLDivLoss: "kldiv", MarginMSELoss: "marginmse"}

    def _use_fused_loss(self) -> bool:
        """The whole loss head as ONE autograd node (sparse_hip.functional.distributed_loss: same values and gradients as the
        reference-form code below it, one launch for the scalar tail, no autograd add of the two d_rep gradients): a single
        process, or N > 1 with the score exchange -- when every loss object is one of the built-in classes (a user subclass
        with its own get_loss takes the reference-form path)."""
        if any(type(lf) not in self._BUILTIN_LOSSES for lf in self.loss_functions) or len(self.loss_functions) > 4:
            return False
        return not self.accelerator.distributed or self._exchange_mode() in ("scores", "gather")

    def _compute_loss_fused(self, d_rep, q_rep, inputs, cap, return_outputs):
        losses = [(self._BUILTIN_LOSSES[type(lf)], lf.weight, bool(lf.use_in_batch_negatives), float(getattr(lf, "temperature", 1.0)))
                  for lf in self.loss_functions]
        n = self.accelerator.num_processes
        teacher = None
        if "scores" in inputs:
            teacher = gather_rep(inputs["scores"].to(d_rep.device, torch.float32), self.accelerator)
        # (the moving average ma = 0.01 * ranking + 0.99 * ma of trainer.py:120-122 is updated by the same launch)
        cfg = {"losses": losses, "q_cap": cap, "flops_threshold": self.data_args.flops_threshold, "moving_avg": self._ma,
               "q_all": self._take_q_prefetch(), "exchange": self._exchange_mode() if self.accelerator.distributed else None,
               "distributed": self.accelerator.distributed,
               "lambda_d": self.get_lambda(self.data_args.flops_d_lambda, self.data_args.flops_d_T),
               "lambda_q": None if self.model_args.inf_free else self.get_lambda(self.data_args.flops_q_lambda,
                                                                                 self.data_args.flops_q_T)}
        loss = F.distributed_loss(d_rep, q_rep, teacher, cfg)
        out = cfg["out"]
        self._last = {"d_flops": out["d_flops"], "total": loss.detach(), "ranking": out["ranking"], "d_rep": d_rep.detach()}
        if self.state.global_step % self.args.logging_steps == 0:
            self._log_step()
        if n > 1:
            loss = loss * n  # DDP averages, trainer.py:139-141
        if not return_outputs:
            return loss
        with torch.no_grad():  # the reference returns the gathered representations
            outputs = {"q_rep": gather_rep(q_rep.detach(), self.accelerator), "d_rep": gather_rep(d_rep.detach(), self.accelerator)}
        return loss, outputs

    def _log_step(self):
        d_rep = self._last["d_rep"]
        if "flops_loss" not in self._last:
            self._last["flops_loss"] = self._last["total"] - self._last["ranking"]
        nz = d_rep[d_rep > 0]
        logger.info(
            "Step %d. ranking loss moving avg:%s, d_flops: %s, flops_loss: %s avg doc length: %s",
            self.state.global_step, self.ranking_loss_moving_avg, float(self._last["d_flops"]),
            float(self._last["flops_loss"]), float((d_rep > 0).sum() / d_rep.shape[0]))
        if nz.numel():
            logger.info("nonzero entries: %s %s %s", float(nz.mean()), float(nz.mean()), float(nz.max()))

    def _save(self, output_dir=None, state_dict=None):
        output_dir = output_dir if output_dir is not None else self.args.output_dir
        os.makedirs(output_dir, exist_ok=True)
        logger.info("Saving model checkpoint to %s", output_dir)
        if self.accelerator.is_main_process:
            self.accelerator.unwrap_model(self.model).save(
                output_dir, state_dict=state_dict, safe_serialization=getattr(self.args, "save_safetensors", True))
            # what a resume needs besides the weights: the step counter and the optimiser state (the fused AdamW's moments, or
            # the caller-supplied optimiser's / scheduler's own state_dict)
            st = {"global_step": self.state.global_step}
            if self._adam is not None:
                st.update({k: v.detach().cpu() for k, v in self._adam.items()})
            if self.optimizer is not None:
                st["optimizer"] = self.optimizer.state_dict()
                if self.lr_scheduler is not None:
                    st["lr_scheduler"] = self.lr_scheduler.state_dict()
            torch.save(st, os.path.join(output_dir, "trainer_state.pt"))

    def set_bi_encoder_teacher(self, embedding_service=None):
        from .bi_encoder_wrapper import BiEncoderWrapper

        kw = self.data_args.kd_ensemble_teacher_kwargs
        bb = self.model.sparse_model.backbone
        self.bi_encoder_teacher = BiEncoderWrapper(
            types=kw["types"], model_ids=kw["model_ids"], use_in_batch_negatives=self.data_args.use_in_batch_negatives,
            score_scale=kw.get("score_scale", 30), embedding_service=embedding_service,
            compute_dtype=bb.compute_dtype, device=bb.device, cache_scores=bool(kw.get("cache_scores", False)))
        self.bi_encoder_teacher.accelerator = self.accelerator

    def get_train_dataloader(self):
        if self.train_dataset is None:
            raise ValueError("Trainer: training requires a train_dataset.")
        from torch.utils.data import DataLoader, RandomSampler
        from torch.utils.data.distributed import DistributedSampler

        common = dict(collate_fn=self.data_collator, num_workers=self.args.dataloader_num_workers, pin_memory=True)
        from ..dataset.dataset import CombinedDataset, CombinedRandomSampler
        if isinstance(self.train_dataset, CombinedDataset):
            # every batch from one member dataset, members already sharded by rank (reference trainer.py:204-217)
            logger.info("Combined dataset. Set combined sampler.")
            return DataLoader(self.train_dataset, batch_sampler=CombinedRandomSampler(
                self.train_dataset.datasets, batch_size=self.args.per_device_train_batch_size), **common)
        if self.accelerator.num_processes > 1 and not getattr(self.train_dataset, "no_prepare", False):
            sampler = DistributedSampler(self.train_dataset, num_replicas=self.accelerator.num_processes,
                                         rank=self.accelerator.process_index, shuffle=True, seed=self.args.seed,
                                         drop_last=self.args.dataloader_drop_last)
        else:
            g = torch.Generator()
            g.manual_seed(self.args.seed)
            sampler = RandomSampler(self.train_dataset, generator=g)
        return DataLoader(self.train_dataset, batch_size=self.args.per_device_train_batch_size, sampler=sampler,
                          drop_last=self.args.dataloader_drop_last, **common)

    # ------------------------------------------------------------------ step driver
    def _prepare_inputs(self, obj):
        """H2D copy of the collator output.  The student's document encoding is additionally packed on
        the host into the ragged layout (padding tokens are then never computed on the device)."""
        out = self._to_device(obj)
        # only the collators' own layout is packed: {"docs": [{"input_ids", "attention_mask"}, ...]} with HOST tensors (anything
        # else -- a caller's own dict, device tensors -- goes through unchanged and takes the dense layout); a failure INSIDE
        # the packing is a bug and propagates
        from collections.abc import Mapping  # (the text collators return transformers.BatchEncoding, a UserDict)
        docs = obj.get("docs") if isinstance(obj, Mapping) else None
        enc = docs[0] if isinstance(docs, (list, tuple)) and docs and isinstance(docs[0], Mapping) else None
        if enc is None or not all(isinstance(enc.get(k), torch.Tensor) for k in ("input_ids", "attention_mask")) or enc["input_ids"].is_cuda:
            return out
        bb = self.model.sparse_model.backbone
        n = int(getattr(self.data_args, "grad_cache_chunk", 0) or 0)
        if not getattr(bb, "varlen", True):
            if not n:
                from sparse_hip.encoder import dense_embed_hints  # dense layout: only the embedding backward's sorted row lists
                hints = dense_embed_hints(enc["input_ids"], enc["attention_mask"], self.accelerator.device,
                                          bb.padded_len(enc["input_ids"].shape[1]))
                if hints is not None:
                    out["docs"][0]["packed"] = hints
            return out
        from sparse_hip.encoder import pack_documents
        if n > 0:  # rep-level gradient caching: every chunk gets its own ragged layout
            ids, mask = enc["input_ids"], enc["attention_mask"]
            chunks = []
            for a in range(0, ids.shape[0], n):
                pk = pack_documents(ids[a:a + n], mask[a:a + n], self.accelerator.device, bb.config.pad_token_id)
                chunks.append((out["docs"][0]["input_ids"][a:a + n], out["docs"][0]["attention_mask"][a:a + n], pk))
            out["docs"][0]["packed_chunks"] = chunks
        else:
            packed = pack_documents(enc["input_ids"], enc["attention_mask"], self.accelerator.device, bb.config.pad_token_id)
            if packed is not None:
                out["docs"][0]["packed"] = packed
        return out

    def _to_device(self, obj):
        dev = self.accelerator.device
        if isinstance(obj, torch.Tensor):
            return obj.to(dev, non_blocking=True)
        if isinstance(obj, dict) or hasattr(obj, "items"):
            return {k: self._to_device(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [self._to_device(v) for v in obj]
        return obj

    def _setup_grad_overlap(self):
        """Side-stream all-reduce of each layer's gradient slice as soon as backward has produced it."""
        bb = self.model.sparse_model.backbone
        self._comm_stream = torch.cuda.Stream(device=bb.device)
        if not self.model_args.inf_free:
            return  # the encoder's backward runs twice per step (queries and documents): reduce once at the end
        # (gradient caching runs one backward per chunk: the encoder calls the hook in the LAST chunk's backward only)
        layout = bb._layout
        names = [n for n, _ in layout]
        self._slices = {}
        for l in range(bb.config.num_hidden_layers):
            first = next(n for n in names if n.startswith(f"bert.encoder.layer.{l}."))
            last = [n for n in names if n.startswith(f"bert.encoder.layer.{l}.")][-1]
            o0, _ = bb._offsets[first]
            o1, s1 = bb._offsets[last]
            n1 = 1
            for d in s1:
                n1 *= d
            self._slices[l] = (o0, o1 + (n1 + 3) // 4 * 4)
        o_cls, _ = bb._offsets["cls.predictions.transform.dense.weight"]
        self._slices["head"] = (o_cls, bb.n_flat)
        self._slices["emb"] = (0, self._slices[0][0])
        bb._layer_hook = self._reduce_slice_async

    def _reduce_slice_async(self, key, wgrad_event=None):
        """All-reduce one slice of the flat gradient on the communication stream, ordered after the backward
        kernels enqueued so far on the main stream and (``wgrad_event``) on the weight-gradient stream."""
        bb = self.model.sparse_model.backbone
        if isinstance(key, tuple):  # two adjacent encoder layers whose weight gradients went out in one grouped launch
            a, b = self._slices[min(key)][0], self._slices[max(key)][1]
        else:
            a, b = self._slices[key]
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ev)
            if wgrad_event is not None:
                self._comm_stream.wait_event(wgrad_event)
            self._pending.append(dist.all_reduce(bb.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def _finish_grad_reduce(self):
        sm = self.model.sparse_model
        bb = sm.backbone
        if not self.accelerator.distributed:
            return
        if bb._layer_hook is not None:
            self._reduce_slice_async("emb")
        else:
            self._pending.append(dist.all_reduce(bb.flat_grad, op=dist.ReduceOp.SUM, async_op=True))
        if sm.idf_vector.requires_grad and sm.idf_vector.grad is not None:
            self._pending.append(dist.all_reduce(sm.idf_vector.grad, op=dist.ReduceOp.SUM, async_op=True))
        for w in self._pending:
            w.wait()
        self._pending = []

    def _optimizer_step(self):
        sm = self.model.sparse_model
        bb = sm.backbone
        a = self.args
        n = self.accelerator.num_processes
        if a.max_grad_norm:  # hf trainer.py:1780-1782 clip_grad_norm_ on the averaged gradients, for the built-in AND a caller-supplied
            # optimiser; no host sync: the factor stays on the device (the gradients hold the SUM over ranks here: norm / n)
            grads = [bb.flat_grad] + ([sm.idf_vector.grad] if sm.idf_vector.requires_grad and sm.idf_vector.grad is not None else [])
            norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads])) / n
            coef = torch.clamp(float(a.max_grad_norm) / (norm + 1e-6), max=1.0).reshape(1)
            for g in grads:
                ops.scale_by(g.view(-1), coef)
        if self.optimizer is not None:  # caller-supplied torch optimiser (train_ir.py:85-107 style)
            if n > 1:  # SUM all-reduce -> the reference's DDP mean of (loss x N) gradients
                bb.flat_grad.div_(n)
                if sm.idf_vector.requires_grad and sm.idf_vector.grad is not None:
                    sm.idf_vector.grad.div_(n)
            self.optimizer.step()
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            bb.mark_weights_dirty()
            return
        step = self.state.global_step  # scheduler has been stepped `step` times so far
        lr = linear_schedule_lr(step, a.learning_rate, a.warmup_steps, a.max_steps)
        if self._adam is None:
            self._adam = {"m": torch.zeros_like(bb.flat_param), "v": torch.zeros_like(bb.flat_param)}
            if sm.idf_vector.requires_grad:
                self._adam["im"] = torch.zeros_like(sm.idf_vector)
                self._adam["iv"] = torch.zeros_like(sm.idf_vector)
        ops.adamw(bb.flat_param, bb.flat_grad, self._adam["m"], self._adam["v"], lr, a.adam_beta1, a.adam_beta2,
                  a.adam_epsilon, a.weight_decay, step + 1, 1.0 / n)
        if sm.idf_vector.requires_grad and sm.idf_vector.grad is not None:
            idf_lr = self.data_args.idf_lr if self.data_args.idf_lr is not None else a.learning_rate
            lr_i = linear_schedule_lr(step, idf_lr, a.warmup_steps, a.max_steps)
            ops.adamw(sm.idf_vector.data, sm.idf_vector.grad, self._adam["im"], self._adam["iv"], lr_i, a.adam_beta1,
                      a.adam_beta2, a.adam_epsilon, a.weight_decay, step + 1, 1.0 / n)
        bb.mark_weights_dirty()

    def resume_from_checkpoint(self, checkpoint_dir: str) -> None:
        """Everything a checkpoint-N directory of this trainer holds: HF-layout weights, the trained IDF vector (idf.json, written
        by ModelWrapper.save when idf_requires_grad), the optimiser state and the step counter.  train() then continues the
        DATA stream where the interrupted run stopped (epoch and position inside it follow from global_step)."""
        if not os.path.exists(os.path.join(checkpoint_dir, "trainer_state.pt")):
            raise FileNotFoundError(f"resume_from_checkpoint={checkpoint_dir!r}: no trainer_state.pt there (not a checkpoint of this trainer)")
        sm = self.model.sparse_model
        if os.path.exists(os.path.join(checkpoint_dir, "model.safetensors")):
            from safetensors.torch import load_file
            sd = load_file(os.path.join(checkpoint_dir, "model.safetensors"))
        else:
            sd = torch.load(os.path.join(checkpoint_dir, "pytorch_model.bin"), map_location="cpu")
        sm.backbone.load_hf_state_dict(sd)
        if sm.idf_requires_grad:  # the moments restored below belong to the TRAINED vector, not to model_args.idf_path's
            path = os.path.join(checkpoint_dir, "idf.json")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path}: an idf_requires_grad run cannot resume without its trained IDF vector")
            with open(path) as f:
                idf_json = json.load(f)
            tok = sm.tokenizer
            vec = torch.zeros(sm.idf_vector.shape[0])  # save() writes the non-zero entries only
            for key, value in idf_json.items():
                idx = tok._convert_token_to_id_with_added_voc(key) if tok is not None else int(key)
                vec[idx] = float(value)
            with torch.no_grad():
                sm.idf_vector.data.copy_(vec.to(sm.idf_vector.device))
        self.load_trainer_state(checkpoint_dir)

    def load_trainer_state(self, checkpoint_dir: str) -> None:
        """resume: AdamW moments + global_step written by _save next to the HF-format weights"""
        st = torch.load(os.path.join(checkpoint_dir, "trainer_state.pt"), map_location="cpu")
        self.state.global_step = int(st.pop("global_step"))
        dev = self.model.sparse_model.backbone.device
        opt, sch = st.pop("optimizer", None), st.pop("lr_scheduler", None)
        if opt is not None:
            if self.optimizer is None:
                raise ValueError("checkpoint holds a caller-supplied optimizer's state but this trainer has none")
            self.optimizer.load_state_dict(opt)
            if sch is not None and self.lr_scheduler is not None:
                self.lr_scheduler.load_state_dict(sch)
        if st:
            self._adam = {k: v.to(dev) for k, v in st.items()}

    def zero_grad(self):
        sm = self.model.sparse_model
        sm.backbone.zero_grad()
        if sm.idf_vector.grad is not None:
            sm.idf_vector.grad.zero_()

    def training_step(self, inputs) -> torch.Tensor:
        """One optimisation step (hf trainer.py:1892-1963 + 1780-1797). Returns the detached loss.
        SM_TRACE_RANGES=1 brackets the phases with roctx ranges (rocprofv3 --marker-trace)."""
        self.model.train()
        bb = self.model.sparse_model.backbone
        # The host never runs more than two steps ahead of the device.  Nothing in a step synchronises, so an unthrottled host
        # queues dozens of steps; tensors handed to the weight-gradient stream (record_stream) cannot be recycled until the device
        # reaches them, the caching allocator then grows by fresh hipMalloc calls (slow, synchronising) for as many steps as the
        # host is ahead -- measured as 2x slower steps at the start of a run.  Two steps of lead hide every host gap.
        if bb.device.type == "cuda":
            if len(self._step_done) >= 2:
                self._step_done.pop(0).synchronize()
        bb.set_dropout_seed(self.args.seed * 1000003 + self.state.global_step * 64 + self.accelerator.process_index)
        bb.check_finite = bool(getattr(self.args, "check_finite", False) or _CHECK_FINITE)
        with _trace_range("forward+loss"):
            loss = self.compute_loss(self.model, inputs)
        with _trace_range("backward"):
            loss.backward()
        with _trace_range("grad_reduce"):
            self._finish_grad_reduce()
        check = getattr(self.args, "check_finite", False) or _CHECK_FINITE
        if check:
            self._raise_if_nonfinite(grad=True, loss=loss)
        with _trace_range("optimizer"):
            self._optimizer_step()
            if check:
                self._raise_if_nonfinite(grad=False)
            self.zero_grad()
        self.state.global_step += 1
        if bb.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._step_done.append(ev)
        return loss.detach()

    def _raise_if_nonfinite(self, grad: bool, loss=None) -> None:
        """check_finite mode: name every parameter (or gradient) tensor that holds a NaN / Inf at this point of the step"""
        sm = self.model.sparse_model
        what = "gradient" if grad else "parameter"
        bad = [f"{n} ({c} of {tot})" for n, c, tot in sm.backbone.nonfinite_report(grad=grad)]
        idf = sm.idf_vector.grad if grad else sm.idf_vector.data
        if sm.idf_vector.requires_grad and idf is not None and not bool(torch.isfinite(idf).all()):
            bad.append("idf_vector")
        if loss is not None and not bool(torch.isfinite(loss.detach()).all()):
            bad.insert(0, "loss")
        if bad:
            raise FloatingPointError(f"step {self.state.global_step}: non-finite {what} values in " + ", ".join(bad))

    def train(self):
        a = self.args
        _cap_host_threads(a.dataloader_num_workers)
        dl = self.get_train_dataloader()
        self.zero_grad()
        loss = None
        # a resumed run (global_step > 0) continues the data stream: epoch = steps done // batches per epoch (the samplers are
        # seeded by the epoch), and the batches of that epoch already consumed are skipped -- as hf trainer.py does on resume
        epoch, skip = divmod(self.state.global_step, max(1, len(dl))) if self.state.global_step > 0 else (0, 0)
        feed = (_InputPrefetcher(self, dl, epoch=epoch, skip=skip)
                if os.environ.get("SM_PREFETCH", "1") != "0" and torch.cuda.is_available() else None)
        it = None
        if feed is None:
            it = _epoch_iter(dl, epoch, skip)
        while self.state.global_step < a.max_steps:
            if feed is not None:
                inputs = feed.get()
            else:
                try:
                    batch = next(it)
                except StopIteration:
                    epoch += 1
                    it = _epoch_iter(dl, epoch, 0)
                    batch = next(it)
                inputs = self._prepare_inputs(batch)
            loss = self.training_step(inputs)
            step = self.state.global_step
            if step % a.logging_steps == 0:
                logger.info("{'loss': %.4f, 'learning_rate': %.3e, 'step': %d}", float(loss),
                            linear_schedule_lr(step, a.learning_rate, a.warmup_steps, a.max_steps), step)
            if a.save_strategy == "steps" and a.save_steps and step % a.save_steps == 0:
                self._save(os.path.join(a.output_dir, f"checkpoint-{step}"))
        if feed is not None:
            feed.close()
        return loss


def _epoch_iter(dl, epoch: int, skip: int):
    """iterator over epoch `epoch` of the loader with its first `skip` batches dropped.  Samplers that are seeded per epoch
    (DistributedSampler, CombinedRandomSampler) get set_epoch; a RandomSampler draws from its own generator, which a resumed run
    advances by replaying the permutations of the epochs already done."""
    import itertools
    sampler = getattr(dl, "batch_sampler", None)
    sampler = sampler if hasattr(sampler, "set_epoch") else getattr(dl, "sampler", None)
    if hasattr(sampler, "set_epoch"):
        sampler.set_epoch(epoch)
    elif epoch > 0 and getattr(dl, "_sm_epochs_drawn", 0) < epoch and getattr(sampler, "generator", None) is not None:
        for _ in range(epoch - getattr(dl, "_sm_epochs_drawn", 0)):  # (only on resume: a running loop draws one permutation per epoch itself)
            for _ in sampler:
                pass
    dl._sm_epochs_drawn = epoch + 1
    it = iter(dl)
    return itertools.islice(it, skip, None) if skip else it


_TRACE_RANGES = os.environ.get("SM_TRACE_RANGES", "0") == "1"
_CHECK_FINITE = os.environ.get("SM_CHECK_FINITE", "0") == "1"


class _trace_range:
    """roctx range around a phase of the step (torch.cuda.nvtx is roctx on ROCm); a no-op unless SM_TRACE_RANGES=1"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _TRACE_RANGES:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if _TRACE_RANGES:
            torch.cuda.nvtx.range_pop()


# ---------------------------------------------------------------------------------------
# Input pipeline (collator.py:135-177 + trainer.py:180-218 feed the path).  The device step
# takes ~12 ms at config 2; collation, host packing and the H2D copies must stay off its critical path.
def _usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cap_host_threads(num_workers: int) -> None:
    """torch sizes its intra-op pool from the VISIBLE cores; under a cgroup CPU quota (16 of 256 on the test box)
    every small CPU op then costs milliseconds (measured: 59.7 -> 17.8 ms/step).  The loop's host work is tiny:
    a handful of threads is plenty."""
    want = max(1, min(8, _usable_cores() // max(1, num_workers + 1)))
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)


def _record_streams(obj, stream) -> None:
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_streams(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_streams(v, stream)
    elif hasattr(obj, "__dict__"):  # PackedDocs / ops.Ragged
        for v in vars(obj).values():
            _record_streams(v, stream)


class _InputPrefetcher:
    """Background thread: next batch from the DataLoader -> host packing -> H2D on its own HIP stream, two batches
    ahead.  The consumer waits for the copy's event on the compute stream (no host sync) and tells the caching
    allocator that the tensors are now used there."""

    def __init__(self, trainer, dataloader, depth: int = 2, epoch: int = 0, skip: int = 0):
        import queue
        import threading
        self.trainer, self.dl = trainer, dataloader
        self.epoch0, self.skip0 = epoch, skip
        self.q = queue.Queue(maxsize=depth)
        self.stop = threading.Event()
        self.device = trainer.accelerator.device
        self.thread = threading.Thread(target=self._run, name="sm-input-prefetch", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            stream = torch.cuda.Stream(device=self.device)
            epoch = self.epoch0
            it = _epoch_iter(self.dl, epoch, self.skip0)
            while not self.stop.is_set():
                try:
                    batch = next(it)
                except StopIteration:
                    epoch += 1
                    it = _epoch_iter(self.dl, epoch, 0)
                    batch = next(it)
                with torch.cuda.stream(stream):
                    inputs = self.trainer._prepare_inputs(batch)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                self._put((inputs, ev, None))
        except BaseException as e:  # surfaced by get()
            self._put((None, None, e))

    def _put(self, item):
        import queue
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.2)
                return
            except queue.Full:
                continue

    def get(self):
        inputs, ev, err = self.q.get()
        if err is not None:
            raise err
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        _record_streams(inputs, cur)
        return inputs

    def close(self):
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except Exception:
            pass
        self.thread.join(timeout=5)
