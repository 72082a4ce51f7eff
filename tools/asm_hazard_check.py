"""Static check of the gfx950 assembly of csrc/*.hip for the bug class behind GPUTEST_r03's training collapse.

The kernels here issue LDS reads / LDS-DMA / (until round 4) global loads from INLINE ASSEMBLY and wait for them with hand-counted
`s_waitcnt` statements, because the compiler's own wait-count pass would drain the whole LDS-DMA ring at every read.  The price:
the compiler does not know that the destination registers of such an instruction are still being written.  It treats them as
finished values from the asm statement on -- and may copy them (register coalescing, live-range splitting) BEFORE the hand-placed
wait.  `ffn_pc_fwd_kernel` did exactly that with its bias prefetch: `v_mov_b64 v[18:19], v[134:135]` ... `s_waitcnt vmcnt(2)`.

This tool compiles every translation unit to assembly (`hipcc -S --cuda-device-only`), replays each kernel's instruction stream
with a model of the two in-order counters (vmcnt: vector memory; lgkmcnt: LDS / scalar memory) and reports every instruction
that READS or WRITES a register while an inline-assembly load into it is still outstanding by the counters' arithmetic.  Loops are
replayed a second time with the state at their back edge.  (Compiler-inserted spill traffic inside a hand-counted region is NOT a
hazard: the counters are in order, an extra operation can only make a counted wait stricter.  It is listed as a note because the
compiler's own wait for a reload drains the LDS-DMA ring.)

  python tools/asm_hazard_check.py [file.hip | file.s ...]        (default: every csrc/*.hip)
Exit code 1 when a hazard is found.  `check_source(path)` is what tests/test_asm_hazards_cpu.py calls.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd", "csrc")

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_WAIT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def _regs(text: str):
    out = set()
    for m in _REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            out.update((kind, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def _is_vm(mn: str) -> bool:
    return mn.startswith(("global_", "buffer_", "flat_", "scratch_", "image_", "tbuffer_"))


def _is_lgkm(mn: str) -> bool:
    return mn.startswith(("ds_", "s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_sendmsg", "s_dcache", "s_scratch_load", "flat_"))


def _has_dest(mn: str) -> bool:
    if mn.startswith("ds_"):
        return mn.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in mn
    if "_lds_" in mn or mn.endswith("_lds"):
        return False
    return "_load" in mn or ("_atomic" in mn and False)


class Kernel:
    def __init__(self, name):
        self.name = name
        self.ins = []  # (line_no, text, in_asm)
        self.labels = {}


def parse(path: str):
    kernels, cur, in_asm = [], None, False
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            line = raw.split(";", 1)[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
            s = line.strip()
            if raw.lstrip().startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if raw.lstrip().startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s:
                continue
            m = re.match(r"^([A-Za-z_][\w$.]*):", s)
            if m and not s.startswith(".L"):
                if cur is not None and cur.ins:
                    kernels.append(cur)
                cur = Kernel(m.group(1))
                continue
            if cur is None:
                continue
            m = re.match(r"^(\.L[\w$.]*):", s)
            if m:
                cur.labels[m.group(1)] = len(cur.ins)
                continue
            if s.startswith("."):
                if s.startswith(".end_amdhsa_kernel") or s.startswith(".section"):
                    if cur.ins:
                        kernels.append(cur)
                    cur = None
                continue
            cur.ins.append((no, s, in_asm))
    if cur is not None and cur.ins:
        kernels.append(cur)
    return [k for k in kernels if any(t.startswith("s_endpgm") for _, t, _ in k.ins)]


def _step(k: Kernel, idx: int, vm, lgkm, found, seen):
    """one instruction against the pending queues vm / lgkm (tuples of (asm, dest regs, line)); returns the new queues"""
    no, text, in_asm = k.ins[idx]
    mn = text.split()[0]
    if mn == "s_waitcnt":
        waits = dict((c, int(n)) for c, n in _WAIT.findall(text))
        if not waits and re.search(r"s_waitcnt\s+0\b", text):
            waits = {"vmcnt": 0, "lgkmcnt": 0}
        if "vmcnt" in waits:
            vm = vm[len(vm) - waits["vmcnt"]:] if waits["vmcnt"] else ()
        if "lgkmcnt" in waits:
            lgkm = lgkm[len(lgkm) - waits["lgkmcnt"]:] if waits["lgkmcnt"] else ()
        return vm, lgkm
    regs = k.regs[idx]
    for q in (vm, lgkm):
        for asm, dest, l0 in q:
            if asm and dest and regs & dest:
                key = (no, l0)
                if key not in seen:
                    seen.add(key)
                    found.append(f"{k.name}: line {no} `{text}` touches {sorted(regs & dest)[:4]} while the inline-asm load of "
                                 f"line {l0} `{k.text_of[l0]}` is outstanding")
    vmop, lgop = _is_vm(mn), _is_lgkm(mn)
    if vmop or lgop:
        dest = frozenset()
        if in_asm and _has_dest(mn):
            dest = frozenset(_regs(text[len(mn):].split(",")[0]))
        ent = (in_asm and bool(dest), dest, no if dest else 0)  # compiler-visible operations only count
        if vmop:
            vm = (vm + (ent,))[-64:]
        if lgop and not (vmop and not mn.startswith("flat_")):
            lgkm = (lgkm + (ent,))[-64:]
    return vm, lgkm


def check_kernel(k: Kernel, max_states_per_block: int = 48):
    """path-sensitive replay over the kernel's control-flow graph: every basic block is entered with every distinct state of the
    two queues that reaches it (bounded per block)"""
    found, seen = [], set()
    k.regs = [_regs(t) for _, t, _ in k.ins]
    k.text_of = {no: t for no, t, _ in k.ins}
    n = len(k.ins)
    starts = {0} | set(k.labels.values())
    br = re.compile(r"(s_cbranch_\w+|s_branch)\s+(\.L[\w$.]*)")
    for i, (_, t, _) in enumerate(k.ins):
        if br.match(t) or t.startswith("s_endpgm"):
            starts.add(i + 1)
    starts = sorted(x for x in starts if x < n)
    end_of = {s0: (starts[j + 1] if j + 1 < len(starts) else n) for j, s0 in enumerate(starts)}
    visited = {s0: set() for s0 in starts}
    work = [(0, (), ())]
    while work:
        b, vm, lgkm = work.pop()
        if b >= n:
            continue
        sig = (vm, lgkm)
        if sig in visited[b] or len(visited[b]) >= max_states_per_block:
            continue
        visited[b].add(sig)
        e = end_of[b]
        for idx in range(b, e):
            vm, lgkm = _step(k, idx, vm, lgkm, found, seen)
        last = k.ins[e - 1][1]
        m = br.match(last)
        if last.startswith("s_endpgm"):
            continue
        if m:
            tgt = k.labels.get(m.group(2))
            if tgt is not None:
                work.append((tgt, vm, lgkm))
            if m.group(1) != "s_branch":
                work.append((e, vm, lgkm))
        else:
            work.append((e, vm, lgkm))
    return found


def compile_to_asm(src: str, out_dir: str) -> str:
    out = os.path.join(out_dir, os.path.splitext(os.path.basename(src))[0] + ".s")
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, cwd=os.path.dirname(src), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def spill_notes(k: Kernel):
    hand_vm = [no for no, t, a in k.ins if a and "vmcnt" in t]
    if not hand_vm:
        return []
    return [f"{k.name}: line {no} `{t}` (spill traffic ahead of a hand-counted vmcnt wait: performance only)"
            for no, t, a in k.ins if not a and t.startswith("scratch_") and no < hand_vm[-1]]


def check_source(path: str, out_dir=None):
    """hazards (list of strings) of one .hip (compiled here) or .s file"""
    if path.endswith(".s"):
        asm = path
    else:
        out_dir = out_dir or tempfile.mkdtemp(prefix="sm_asm_")
        asm = compile_to_asm(path, out_dir)
    found = []
    kernels = parse(asm)
    for k in kernels:
        found += check_kernel(k)
    return found, len(kernels)


def main(argv):
    files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        found, n = check_source(f)
        print(f"{os.path.basename(f)}: {n} kernels, {len(found)} hazards")
        for h in found[:40]:
            print("   ", h)
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
