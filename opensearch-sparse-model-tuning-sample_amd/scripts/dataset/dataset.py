"""Training datasets of the fine-tuning entry point: what feeds the collators of scripts/dataset/collator.py.

Host-side plumbing, no kernels.  Same record formats, item tuples and sampling rules as the reference
(scripts/dataset/dataset.py:124-148 rank shards, :151-217 kd, :220-284 kd-ids, :329-358 posnegs, :389-444 combined
dataset + sampler, :454-523 loaders), so a data directory prepared for the reference trains here unchanged:

  kd       records {"query", "docs", "scores"?, "first_rank"?}   -> (query, [docs], [scores | None])
  kd-ids   records {"query", "q_id", "docs", "d_ids", "scores"?} -> [query, q_id, [docs], [d_ids], [scores | None]]
  posnegs  records {"query", "pos", "negs"}                      -> [query, pos, [negs]]

`load_dataset(path, ...)` reads a HuggingFace `datasets` directory (`Dataset.load_from_disk`) or, for tests and small
corpora, a `.jsonl` file of such records.
"""
from __future__ import annotations

import json
import logging
import os
import random
from typing import List, Sequence

import numpy as np
from torch.utils.data import BatchSampler, Dataset, RandomSampler, Sampler

from ..utils import is_ddp_enabled

logger = logging.getLogger(__name__)


def partial_shuffle(items: Sequence, swap_times: float) -> list:
    """`swap_times` random transpositions (a full shuffle once that is at least half the list): perturbs the teacher's
    ranking a little before documents are dealt into groups (reference dataset.py:22-40; numpy global RNG as there)"""
    if swap_times <= 0:
        return list(items)
    arr = np.array(items)
    n = len(arr)
    if swap_times >= n // 2:
        np.random.shuffle(arr)
        return arr.tolist()
    for i, j in np.random.randint(0, n, size=(int(swap_times), 2)):
        arr[i], arr[j] = arr[j], arr[i]
    return arr.tolist()


class RecordList:
    """The slice of the `datasets.Dataset` interface the classes below use (len, [i], iteration, column_names, filter) over
    a list of dicts: lets a .jsonl file stand in for a `datasets` directory."""

    def __init__(self, records: List[dict]):
        self.records = records
        self.column_names = sorted({k for r in records for k in r})

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        return self.records[i]

    def __iter__(self):
        return iter(self.records)

    def filter(self, fn):
        return RecordList([r for r in self.records if fn(r)])

    @staticmethod
    def from_jsonl(path: str) -> "RecordList":
        with open(path) as f:
            return RecordList([json.loads(line) for line in f if line.strip()])


def _keep_first_rank(all_data, thresh):
    """records whose positive the first-stage retriever ranked inside [0, thresh] (kd / kd-ids only, when the column exists)"""
    if "first_rank" not in all_data.column_names:
        return all_data
    kept = all_data.filter(lambda ex: 0 <= ex.get("first_rank", 1) <= thresh)
    logger.info("first_rank <= %s keeps %d of %d examples", thresh, len(kept), len(all_data))
    return kept


def _deal_groups(n_docs: int, sample_num: int, swap_times: float) -> List[List[int]]:
    """The documents of one record (ranked by the teacher) are dealt into n // sample_num groups of sample_num: group i
    takes positions i, step + i, 2 step + i, ... -- every group spans the whole ranking."""
    order = partial_shuffle(range(n_docs), swap_times)
    step = n_docs // sample_num
    return [[order[k * step + i] for k in range(sample_num)] for i in range(step)]


class KnowledgeDistillDataset(Dataset):
    def __init__(self, all_data, sample_num=2, swap_times=0, first_rank_thresh=1000, score_scale=1.0, **kwargs):
        assert sample_num >= 2
        all_data = _keep_first_rank(all_data, first_rank_thresh)
        self.all_data, self.score_scale = all_data, score_scale
        self.has_scores = "scores" in all_data.column_names
        self.idxs = [(ex_idx, grp) for ex_idx, ex in enumerate(all_data) for grp in _deal_groups(len(ex["docs"]), sample_num, swap_times)]
        logger.info("KnowledgeDistillDataset: %d records -> %d samples (score_scale %s)", len(all_data), len(self.idxs), score_scale)

    def __len__(self):
        return len(self.idxs)

    def __getitem__(self, idx):
        ex_idx, grp = self.idxs[idx]
        ex = self.all_data[ex_idx]
        scores = [ex["scores"][i] * self.score_scale for i in grp] if self.has_scores else [None] * len(grp)
        return ex["query"], [ex["docs"][i] for i in grp], scores


class KnowledgeDistillIdsDataset(Dataset):
    def __init__(self, all_data, sample_num=2, swap_times=0, first_rank_thresh=1000, **kwargs):
        assert sample_num >= 2
        all_data = _keep_first_rank(all_data, first_rank_thresh)
        self.all_data = all_data
        self.has_scores = "scores" in all_data.column_names
        self.idxs = [(ex_idx, grp) for ex_idx, ex in enumerate(all_data) for grp in _deal_groups(len(ex["docs"]), sample_num, swap_times)]
        logger.info("KnowledgeDistillIdsDataset: %d records -> %d samples", len(all_data), len(self.idxs))

    def __len__(self):
        return len(self.idxs)

    def __getitem__(self, idx):
        ex_idx, grp = self.idxs[idx]
        ex = self.all_data[ex_idx]
        scores = [ex["scores"][i] for i in grp] if self.has_scores else [None] * len(grp)
        return [ex["query"], ex["q_id"], [ex["docs"][i] for i in grp], [ex["d_ids"][i] for i in grp], scores]


class PosNegsDataset(Dataset):
    """one sample per full chunk of `sample_num` negatives of a record (a trailing partial chunk is dropped)"""

    def __init__(self, data, sample_num=3, **kwargs):
        assert sample_num >= 1
        self.data = []
        for rec in data:
            negs = rec.get("negs", [])
            for i in range(0, len(negs) - sample_num + 1, sample_num):
                self.data.append([rec["query"], rec["pos"], negs[i:i + sample_num]])
        logger.info("PosNegsDataset: %d records -> %d samples", len(data), len(self.data))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx]


class DDPDatasetWithRank(Dataset):
    """rank r of `world_size` sees samples r, r + world_size, ... (`drop`: the tail that does not divide is dropped;
    `shuffle`: a per-rank permutation seeded by the rank, leaving the global `random` state untouched)"""

    def __init__(self, inner_dataset, local_rank, world_size, drop=False, shuffle=False):
        self.inner_dataset = inner_dataset
        n = len(inner_dataset)
        if drop:
            n -= n % world_size
        self.idxs = list(range(local_rank, n, world_size))
        if shuffle:
            random.Random(local_rank).shuffle(self.idxs)
        logger.info("rank %d of %d: %d local samples", local_rank, world_size, len(self.idxs))

    def __len__(self):
        return len(self.idxs)

    def __getitem__(self, idx):
        return self.inner_dataset[self.idxs[idx]]


class CombinedRandomSampler(Sampler):
    """Batch sampler over several datasets: every batch comes from ONE dataset (its own random batch sampler); the order in
    which datasets take turns is a shuffled list with as many entries per dataset as it has batches -- under DDP shuffled
    with a fixed seed so that all ranks draw from the same dataset at the same step.  Yields [[dataset index, sample
    index], ...], the index form CombinedDataset takes."""

    def __init__(self, datasets, batch_size, drop_last=True):
        self.datasets, self.batch_size = datasets, batch_size
        self.samplers = [BatchSampler(RandomSampler(d), batch_size=batch_size, drop_last=drop_last) for d in datasets]
        self.dataset_sequences = None

    def set_dataset_sequences(self, dataset_sequences=None):
        if dataset_sequences is None:
            dataset_sequences = [i for i, s in enumerate(self.samplers) for _ in range(len(s))]
            (random.Random(0) if is_ddp_enabled() else random).shuffle(dataset_sequences)
        self.dataset_sequences = dataset_sequences

    def __iter__(self):
        if self.dataset_sequences is None:
            self.set_dataset_sequences()
        its = [iter(s) for s in self.samplers]
        for d in self.dataset_sequences:
            yield [[d, i] for i in next(its[d])]

    def __len__(self):
        return sum(len(s) for s in self.samplers)


class CombinedDataset(Dataset):
    """indexed by [dataset index, sample index]; `no_prepare` tells the trainer that the members are already rank-sharded
    (no DistributedSampler on top, reference trainer.py:214-217)"""

    def __init__(self, datasets):
        self.datasets = datasets
        self.length = sum(len(d) for d in datasets)
        self.no_prepare = True

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        dataset_idx, data_idx = idx
        return self.datasets[dataset_idx][data_idx]


DATASET_CLS_MAP = {"kd": KnowledgeDistillDataset, "posnegs": PosNegsDataset, "kd-ids": KnowledgeDistillIdsDataset}


def _read_records(path: str):
    if path.endswith(".jsonl"):
        return RecordList.from_jsonl(path)
    from datasets import Dataset as DatasetsDataset
    return DatasetsDataset.load_from_disk(path)


def load_dataset(path, cls, swap_times=0, sample_num_one_query=2, first_rank_thresh=1000, score_scale=1.0):
    logger.info("load dataset from %s as %s", path, DATASET_CLS_MAP[cls].__name__)
    return DATASET_CLS_MAP[cls](_read_records(path), sample_num=sample_num_one_query, swap_times=swap_times,
                                first_rank_thresh=first_rank_thresh, score_scale=score_scale)


def load_datasets(path, cls, training_args, swap_times=0, sample_num_one_query=2, first_rank_thresh=1000, score_scale=1.0):
    """every entry of the directory (or directories) `path` is one dataset; each is sharded by rank, then all are combined"""
    roots = [path] if isinstance(path, str) else list(path)
    world, rank = int(training_args.world_size), int(training_args.local_process_index)
    members = []
    for root in roots:
        for name in sorted(os.listdir(root)):
            ds = load_dataset(os.path.join(root, name), cls, swap_times, sample_num_one_query, first_rank_thresh, score_scale)
            members.append(DDPDatasetWithRank(ds, rank, world, drop=world != 1, shuffle=world != 1))
    combined = CombinedDataset(members)
    logger.info("total data: %d", len(combined))
    return combined
