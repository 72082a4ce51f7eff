#!/usr/bin/env python3
"""Training entry point: ``torchrun --nproc_per_node=N train_ir.py cfg.yaml`` or
``python train_ir.py cfg.yaml`` -- the reference's train_ir.py:30-150 launch contract on the
MI355X kernels.  ``train_file: synthetic`` selects the built-in MS-MARCO-shaped generator."""
import logging
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it (multi-process GPU work)
# N > 1: the step keeps FIVE HIP streams busy (backward chain, weight gradients, gradient all-reduce, RCCL's own, input prefetch); with the
# runtime's default of 4 hardware queues two of them share one, and when the communication stream's wait for a layer's weight gradients
# lands in the main stream's queue the backward chain stalls for the length of that weight-gradient kernel -- 275 us behind every layer,
# 0.36 ms of a 12.4 ms step (measured with a one-rank RCCL communicator, profiles/r6_single_rank_rccl.txt).  Must be set before HIP starts.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from scripts.args import parse_args  # noqa: E402
from scripts.dataset.collator import COLLATOR_CLS_MAP  # noqa: E402
from scripts.train.loss import LOSS_CLS_MAP  # noqa: E402
from scripts.train.trainer import SparseModelTrainer  # noqa: E402
from scripts.utils import get_model, set_logging  # noqa: E402

logger = logging.getLogger(__name__)


def init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
    return world, local


def load_training_dataset(data_args, training_args, model):
    """train_file: synthetic -> the built-in MS-MARCO-shaped generator; otherwise the reference's two forms
    (train_ir.py:109-127): one dataset (`train_file`) or a directory of datasets sharded by rank (`train_file_dir`)"""
    if data_args.train_file == "synthetic":
        from scripts.dataset.synthetic import SyntheticTriplesDataset

        k = data_args.sample_num_one_query + 1 if data_args.data_type == "posnegs" else data_args.sample_num_one_query
        return SyntheticTriplesDataset(data_args.synthetic_samples, k, data_args.max_seq_length,
                                       data_args.synthetic_query_len, model.vocab_size,
                                       with_scores=data_args.data_type == "kd" and not data_args.kd_ensemble_teacher_kwargs)
    from scripts.dataset.dataset import load_dataset, load_datasets

    kw = dict(cls=data_args.data_type, swap_times=data_args.swap_times, sample_num_one_query=data_args.sample_num_one_query,
              first_rank_thresh=data_args.first_rank_thresh)
    if data_args.train_file is not None:
        return load_dataset(path=data_args.train_file, **kw)
    if data_args.train_file_dir is not None:
        return load_datasets(path=data_args.train_file_dir, training_args=training_args, **kw)
    raise ValueError("train_file or train_file_dir must be specified")


def main():
    model_args, data_args, training_args = parse_args()
    world, local = init_distributed()
    os.makedirs(training_args.output_dir, exist_ok=True)
    if len(sys.argv) == 2 and sys.argv[1].endswith((".yaml", ".yml")) and local == 0:
        shutil.copyfile(sys.argv[1], os.path.join(training_args.output_dir, "train_config.yaml"))
    set_logging(training_args, "train.log")
    torch.manual_seed(training_args.seed)

    model = get_model(model_args, compute_dtype=training_args.compute_dtype)
    collator_key = "synthetic" if data_args.train_file == "synthetic" else data_args.data_type
    data_collator = COLLATOR_CLS_MAP[collator_key](
        model.tokenizer, data_args.max_seq_length,
        data_args.kd_ensemble_teacher_kwargs.get("teacher_tokenizer_ids", []))
    loss_functions = [LOSS_CLS_MAP[t](use_in_batch_negatives=data_args.use_in_batch_negatives,
                                      weight=data_args.ranking_loss_weight, temperature=data_args.temperature)
                      for t in data_args.loss_types]
    training_args.world_size, training_args.local_process_index = world, local
    dataset = load_training_dataset(data_args, training_args, model)
    trainer = SparseModelTrainer(model_args=model_args, data_args=data_args, model=model, args=training_args,
                                 train_dataset=dataset, data_collator=data_collator, loss_functions=loss_functions)
    if len(data_args.kd_ensemble_teacher_kwargs) != 0:
        logger.info("Set bi-encoder teacher. %s", data_args.kd_ensemble_teacher_kwargs)
        trainer.set_bi_encoder_teacher()
    if training_args.resume_from_checkpoint:
        trainer.resume_from_checkpoint(training_args.resume_from_checkpoint)
        logger.info("Resumed from %s at step %d", training_args.resume_from_checkpoint, trainer.state.global_step)
    trainer.train()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
