"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes of the same bench command:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
   rocprofv3 --kernel-trace --pmc WRITE_SIZE ... -- (same)
(FETCH_SIZE needs 3 of the 4 TCC slots and WRITE_SIZE 2: one counter per pass.)  Both counters are in units
of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, 'HBM').  usage: pmc_summary.py fetch.csv write.csv out.json [git revision the passes were measured at] [steps run] [layout]"""
import csv, collections, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def load(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name']
        n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        n = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_Z\d+', '', n)
        n = re.sub(r'[<(].*', '', n)
        n = re.sub(r'(I(DF16b|f)|PK).*', '', n)
        d[n][0] += 1
        d[n][1] += float(r['Counter_Value'])
    return d

f, w = load(sys.argv[1]), load(sys.argv[2])
out = {}
for n in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
    kf, vf = f.get(n, [0, 0.0]); kw, vw = w.get(n, [0, 0.0])
    if max(kf, kw) == 0: continue
    rd = 2.0 * vf * 1024 / max(kf, 1); wr = vw * 1024 / max(kw, 1)
    out[n] = {"launches": max(kf, kw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr}
git = sys.argv[4] if len(sys.argv) > 4 else "unknown"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else None    # training steps the profiled command ran (warm-up + timed), layout it ran on
layout = sys.argv[6] if len(sys.argv) > 6 else None
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --only-value-layout "
                     "--no-cpu-baseline --no-gemm-roofline; FETCH_SIZE x2 (gfx950)",
           "git": git, "csrc_sha": __import__("bench").csrc_sha(), "steps": steps, "layout": layout, "kernels": out}, open(sys.argv[3], "w"), indent=1)
for n, v in list(out.items())[:12]:
    print(f"{n:36s} x{v['launches']:4d}  read {v['read_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB per launch")
