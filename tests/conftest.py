import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# kernel / end-to-end parity first; the multi-process cases (every leg bounded to 120 s in its own process group) run BEFORE the
# slow full-size files: in round 3 one failure in test_fullsize_gpu hid all twelve distributed legs behind `-x`
_ORDER = ("test_oracle_golden", "test_host_cpu", "test_datasets_cpu", "test_asm_hazards_cpu", "test_asan_cpu", "test_kernels_gpu", "test_ffn_pc_gpu", "test_e2e_gpu",
          "test_distributed", "test_bench_cli", "test_baseline_configs_gpu", "test_fullsize_gpu")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 1
    items.sort(key=rank)  # stable: the order inside a file is kept
