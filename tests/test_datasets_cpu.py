"""Host-side datasets that feed the collators (reference scripts/dataset/dataset.py:124-523): item shapes, the dealing of a
record's ranked documents into groups, rank sharding, the combined sampler, and the loaders' .jsonl path."""
import json
import os
import random

import numpy as np
import pytest

from scripts.args import TrainingArguments, parse_yaml_file
from scripts.dataset.dataset import (CombinedDataset, CombinedRandomSampler, DDPDatasetWithRank, KnowledgeDistillDataset,
                                     KnowledgeDistillIdsDataset, PosNegsDataset, RecordList, load_dataset, load_datasets,
                                     partial_shuffle)


def kd_records(n=5, docs=6, with_rank=False):
    recs = []
    for i in range(n):
        r = {"query": f"q{i}", "docs": [f"d{i}_{j}" for j in range(docs)], "scores": [float(docs - j) for j in range(docs)]}
        if with_rank:
            r["first_rank"] = i * 10 - 5  # -5, 5, 15, ...
        recs.append(r)
    return recs


def test_kd_groups_span_the_ranking_and_scale_scores():
    ds = KnowledgeDistillDataset(RecordList(kd_records()), sample_num=3, score_scale=2.0)
    assert len(ds) == 5 * (6 // 3)
    q, docs, scores = ds[0]
    assert q == "q0" and docs == ["d0_0", "d0_2", "d0_4"] and scores == [12.0, 8.0, 4.0]  # positions i, step + i, 2 step + i
    q, docs, scores = ds[1]
    assert docs == ["d0_1", "d0_3", "d0_5"]
    no_scores = [{k: v for k, v in r.items() if k != "scores"} for r in kd_records()]
    assert KnowledgeDistillDataset(RecordList(no_scores), sample_num=2)[0][2] == [None, None]


def test_first_rank_filter_and_ids_items():
    ds = KnowledgeDistillDataset(RecordList(kd_records(with_rank=True)), sample_num=2, first_rank_thresh=20)
    assert {ds[i][0] for i in range(len(ds))} == {"q1", "q2"}  # ranks 5 and 15; -5 (not retrieved) and > 20 dropped
    recs = [dict(r, q_id=i, d_ids=list(range(100 * i, 100 * i + 6))) for i, r in enumerate(kd_records(2))]
    item = KnowledgeDistillIdsDataset(RecordList(recs), sample_num=2)[4]
    assert item == ["q1", 1, ["d1_1", "d1_4"], [101, 104], [5.0, 2.0]]


def test_posnegs_chunks_drop_the_partial_tail():
    recs = [{"query": "q", "pos": "p", "negs": [f"n{i}" for i in range(7)]}, {"query": "r", "pos": "s", "negs": ["x"]}]
    ds = PosNegsDataset(recs, sample_num=3)
    assert [ds[i] for i in range(len(ds))] == [["q", "p", ["n0", "n1", "n2"]], ["q", "p", ["n3", "n4", "n5"]]]


def test_partial_shuffle_is_a_permutation_with_few_moves():
    np.random.seed(0)
    out = partial_shuffle(list(range(100)), 3)
    assert sorted(out) == list(range(100)) and sum(a != b for a, b in zip(out, range(100))) <= 6
    assert partial_shuffle([3, 1, 2], 0) == [3, 1, 2]


def test_rank_shards_partition_the_dataset():
    base = list(range(23))
    shards = [DDPDatasetWithRank(base, r, 4, drop=True, shuffle=True) for r in range(4)]
    seen = sorted(x for s in shards for x in (s[i] for i in range(len(s))))
    assert seen == list(range(20)) and all(len(s) == 5 for s in shards)
    state = random.getstate()
    DDPDatasetWithRank(base, 1, 4, shuffle=True)
    assert random.getstate() == state, "the per-rank shuffle must not disturb the global RNG"


def test_combined_sampler_draws_every_batch_from_one_member():
    members = [list(range(10)), list(range(100, 107))]
    ds = CombinedDataset(members)
    sampler = CombinedRandomSampler(members, batch_size=3)
    batches = list(sampler)
    assert len(batches) == len(sampler) == 10 // 3 + 7 // 3 and ds.no_prepare and len(ds) == 17
    for b in batches:
        assert len({d for d, _ in b}) == 1 and len(b) == 3
        assert all(ds[idx] == members[idx[0]][idx[1]] for idx in b)
    assert sorted(d for b in batches for d, _ in b[:1]) == [0, 0, 0, 1, 1]


def test_loaders_read_jsonl_and_directories(tmp_path):
    for name in ("a.jsonl", "b.jsonl"):
        with open(tmp_path / name, "w") as f:
            for r in kd_records(4):
                f.write(json.dumps(r) + "\n")
    one = load_dataset(str(tmp_path / "a.jsonl"), "kd", sample_num_one_query=2)
    assert len(one) == 4 * 3
    targs = TrainingArguments()
    targs.world_size, targs.local_process_index = 2, 1
    both = load_datasets(str(tmp_path), "kd", targs, sample_num_one_query=2)
    assert isinstance(both, CombinedDataset) and len(both.datasets) == 2 and len(both) == 2 * 6


def test_unsupported_training_arguments_fail_loudly(tmp_path):
    cfg = tmp_path / "c.yaml"
    cfg.write_text("gradient_accumulation_steps: 4\nmax_steps: 10\n")
    with pytest.raises(ValueError, match="gradient_accumulation_steps"):
        parse_yaml_file(str(cfg))
    cfg.write_text("lr_scheduler_type: cosine\n")
    with pytest.raises(ValueError, match="lr_scheduler_type"):
        parse_yaml_file(str(cfg))
    cfg.write_text("gradient_accumulation_steps: 1\nbeir_datasets: scifact\nmax_grad_norm: null\n")
    assert parse_yaml_file(str(cfg))[2].extra["beir_datasets"] == "scifact"
