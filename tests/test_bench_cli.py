"""bench.py's contract: ONE JSON line with the fields the driver and the judge read (task statement + SURVEY 8d), at
N = 1 and -- two ranks sharing the test GPU over gloo -- at N = 2 through torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "value_layout", "value_dense_layout", "value_ragged_layout",
            "loss_first", "loss_last", "finite"}  # (the last three: liveness of the timed region, GPUTEST_r03)


def _run(cmd, env=None, timeout=240):
    """own process group, killed as a whole on expiry: rank processes left behind would hold the GPU for every later test"""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        pytest.fail(f"timed out after {timeout}s: {' '.join(cmd[-8:])}\n{(out + err)[-6000:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_cli_defaults_follow_the_measurement_protocol():
    """>= 50 timed steps after >= 10 warm-up by default, layout switch present"""
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert a.gpus == 1 and a.steps >= 50 and a.warmup >= 10
    assert a.layout == "dense"            # the headline computes every padded token row; the ragged rate is reported beside it
    assert not a.cpu_baseline_bounded     # the CPU leg runs configs[1] at full shape by default


@pytest.mark.gpu
def test_bench_one_gpu_line():
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = _line(r.stdout)
    # the two records that ride beside `value` (round 5): the step at ~1 % live activations, configs[4]'s per-GPU shape on one GPU
    sr, c5 = j["sparse_regime"], j["c5_per_gpu"]
    assert "error" not in sr and 0.003 < sr["alive_fraction"] < 0.03 and sr["alive_fraction_random_init"] > 0.5 and sr["ms_per_step"] > 0, sr
    assert "error" not in c5 and c5["ms_per_step"] > 0 and c5["peak_memory_gib"] < 200, c5
    # round 6: the same model and token count per step at the reference's shipped sequence lengths (config_infonce.yaml:9)
    sw = j["seq_sweep"]
    for key in ("seq256", "seq512"):
        assert "error" not in sw[key] and sw[key]["finite"] and sw[key]["tokens_per_sec"] > 0.3 * sw["seq128"]["tokens_per_sec"], sw
    assert REQUIRED <= set(j), REQUIRED - set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["unit"] == "samples/sec" and j["dtype"] == "bf16" and j["vs_baseline"] is None
    assert abs(j["value"] - 32 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    pm = rf["peak_measured"]
    assert 500 < pm["mfma_bf16_tflops"] < 2600 and 2000 < pm["hbm_copy_gbs"] < 8200, pm
    assert j["value_ragged_layout"] > j["value_dense_layout"] > 0
    # the roofline object is the step's dominant kernel class, the encoder GEMMs in-step; the head forward rides beside it
    assert rf["kernel"].startswith("encoder GEMMs, in-step") and 0 < rf["one_queue"]["frac"] < 1 and len(rf["per_op"]) >= 3
    # round 6: the shader clock under every GEMM op (sm_clock_stamp around each launch) and the cross-check on the bare MFMA loop,
    # whose rate fixes its clock (rate / 2.5 PFLOP/s x 2.4 GHz): the stamps must read that clock
    for g in rf["per_op"]:
        assert 0.8 < g["clock_ghz"] < 2.6 and 0 < g["frac_at_clock"] < 1, g
    assert abs(pm["mfma_loop_clock_ghz"] - pm["mfma_loop_clock_implied_by_its_rate_ghz"]) < 0.15 * pm["mfma_loop_clock_ghz"], pm
    assert j["value_layout"] == "dense" and abs(j["value"] - j["value_dense_layout"]) < 1e-9
    assert j["finite"] is True and j["loss_first"] == j["loss_first"] and j["loss_first"] != j["loss_last"]  # the timed region trained
    hd = j["roofline_head_fwd"]
    assert hd["bound"] == "mfma" and 0 < hd["frac"] < 1


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo():
    """the N > 1 launch contract (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run, barrier + max over ranks, one line
    from rank 0); RCCL needs one GPU per rank, so on the single test GPU the transport is gloo (SM_BENCH_BACKEND)"""
    env = dict(os.environ, SM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", SM_FAULTHANDLER_S="240")
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
              "127.0.0.1", "--master-port", "29591", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
              "--warmup", "1"], env=env, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and j["scaling"] == "weak"
    assert "cpu_baseline" not in j and j["value"] > 0 and j["finite"] is True
    assert j["dist"]["world_size"] == 2 and j["dist"]["backend"] == "gloo" and j["dist"]["exchange"].startswith("gather (value)")  # self-describing N > 1 record
    # both exchange modes in one command (round 6): `value` = north_star's all-gather of the representations, the score-block
    # exchange beside it, each with its own liveness record
    assert j["value_scores_exchange"] > 0 and j["ms_per_step_scores_exchange"] > 0 and j["value_gather_second_sample"] > 0
    assert j["liveness"]["scores_exchange"]["finite"] is True


@pytest.mark.gpu
def test_bench_single_rank_rccl_line():
    """bench.py --single-rank-rccl: the N > 1 code path through RCCL with a communicator of one rank (the only RCCL execution a
    one-GPU box allows): both exchange modes timed, the dist record names the backend and the RCCL version"""
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-rank-rccl", "--steps", "4", "--warmup", "2"], timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["finite"] is True and j["value"] > 0
    d = j["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["rccl_version"][0].isdigit(), d
    assert j["value_scores_exchange"] > 0 and j["liveness"]["scores_exchange"]["finite"] is True
    assert "single_rank_rccl" in j
