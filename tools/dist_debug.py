"""Reproduce / root-cause stalls of the 2-rank step (tests/test_distributed.py) on one GPU.

Runs the test's worker under torch.distributed.run with (i) faulthandler dumping every thread's stack
if a rank is still alive after SM_DBG_DUMP seconds, (ii) a progress line (stderr, flushed) before and after
every torch.distributed collective.  Usage on the GPU box:

    python tools/dist_debug.py [case] [mode] [repeats] [backend]

Logs go to gpurun_out/dist_debug/."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))

PRELUDE = r"""
import faulthandler, os, sys, time
faulthandler.enable()
faulthandler.dump_traceback_later(float(os.environ.get("SM_DBG_DUMP", "60")), exit=True)
import torch, torch.distributed as dist
_t0 = time.time()
def _wrap(name):
    fn = getattr(dist, name)
    def inner(*a, **k):
        r = os.environ.get("RANK", "0")
        shapes = [tuple(x.shape) for x in a if hasattr(x, "shape")]
        print(f"[{time.time()-_t0:7.3f}] rank {r} -> {name} {shapes} async={k.get('async_op', False)}", file=sys.stderr, flush=True)
        out = fn(*a, **k)
        print(f"[{time.time()-_t0:7.3f}] rank {r} <- {name}", file=sys.stderr, flush=True)
        return out
    setattr(dist, name, inner)
for _n in ("all_gather_into_tensor", "all_reduce", "barrier", "all_gather"):
    _wrap(_n)
"""


def main():
    from test_distributed import WORKER
    case = sys.argv[1] if len(sys.argv) > 1 else "infonce_ibn"
    mode = sys.argv[2] if len(sys.argv) > 2 else "scores"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    out = os.path.join(ROOT, "gpurun_out", "dist_debug")
    os.makedirs(out, exist_ok=True)
    script = os.path.join(out, "worker_dbg.py")
    with open(script, "w") as f:
        f.write(PRELUDE + WORKER.replace('dist.init_process_group("gloo")', f'dist.init_process_group("{backend}")'))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", SM_TEST_CASE=case, SM_EXCHANGE=mode)
    for i in range(reps):
        t0 = time.time()
        log = os.path.join(out, f"{case}_{mode}_{backend}_{i}.log")
        with open(log, "w") as lf:
            try:
                r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                    "--master-addr", "127.0.0.1", "--master-port", str(29600 + i), script, ROOT, PKG,
                                    os.path.join(out, f"two_{i}.npz")], stdout=lf, stderr=subprocess.STDOUT, timeout=150, env=env)
                rc = r.returncode
            except subprocess.TimeoutExpired:
                rc = "timeout"
        print(f"{case} {mode} {backend} run {i}: rc={rc} {time.time()-t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
