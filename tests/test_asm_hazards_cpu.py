"""Static gate for the bug class behind GPUTEST_r03's training collapse: the kernels issue LDS reads from
inline assembly and wait for them with hand-counted `s_waitcnt` statements; the compiler, which does not know that those
destination registers are still being written, may copy them BEFORE the hand-placed wait (round 3's `ffn_pc_fwd_kernel` did, with
its inline-assembly bias prefetch; `sparse_head_fwd_vs_kernel<512 / 768>` did at its loop exit).  The check replays the device
assembly of the CURRENT build (csrc/Makefile keeps it under build/obj) with a model of the two in-order counters over every path
of each kernel's control-flow graph (tools/asm_hazard_check.py) -- no GPU needed, runs wherever the library was built."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_hazard_check as A  # noqa: E402

OBJ = os.path.join(ROOT, "build", "obj")
CSRC = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd", "csrc")


def _units():
    return sorted(os.path.splitext(f)[0] for f in os.listdir(CSRC) if f.endswith(".hip"))


@pytest.fixture(scope="module")
def asm_dir():
    want = [os.path.join(OBJ, u + "-hip-amdgcn-amd-amdhsa-gfx950.s") for u in _units()]
    if not all(os.path.exists(p) for p in want):
        subprocess.check_call(["make", "-j6", "-C", CSRC])
    missing = [p for p in want if not os.path.exists(p)]
    assert not missing, f"the build did not leave the device assembly of {missing}"
    return OBJ


@pytest.mark.parametrize("unit", _units())
def test_no_instruction_touches_a_register_an_inline_asm_load_is_still_writing(asm_dir, unit):
    path = os.path.join(asm_dir, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    src = os.path.join(CSRC, unit + ".hip")
    assert os.path.getmtime(path) >= os.path.getmtime(src), f"{path} is older than its source: rebuild (make -C csrc)"
    found, n_kernels = A.check_source(path)
    assert n_kernels > 0, f"no kernels parsed from {path}"
    assert not found, "\n".join(found[:20])


def test_the_checker_sees_the_round_3_bug():
    """the pattern of round 3's ffn_pc_fwd_kernel, reduced: a copy of an inline-asm load's destination ahead of the counted wait"""
    asm = """
_Z6kernelv:
	;;#ASMSTART
	global_load_dwordx4 v[134:137], v[146:147], off
	;;#ASMEND
	global_store_dwordx4 v[152:153], v[22:25], off
	global_store_dwordx4 v[152:153], v[18:21], off
	s_barrier
	v_mov_b64_e32 v[18:19], v[134:135]
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_mfma_f32_32x32x16_f16 v[18:33], v[150:153], v[6:9], v[18:33]
	s_endpgm
	.section	.rodata
"""
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(asm)
    try:
        found, n = A.check_source(f.name)
    finally:
        os.unlink(f.name)
    assert n == 1 and len(found) == 1 and "v_mov_b64_e32 v[18:19], v[134:135]" in found[0]
    good = asm.replace("\tv_mov_b64_e32 v[18:19], v[134:135]\n", "").replace("\tv_mfma", "\tv_mov_b64_e32 v[18:19], v[134:135]\n\tv_mfma")
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(good)
    try:
        found, n = A.check_source(f.name)
    finally:
        os.unlink(f.name)
    assert n == 1 and not found
