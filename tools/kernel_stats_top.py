"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kernel_stats_top.py <dir with s_kernel_stats.csv> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    t = float(r["TotalDurationNs"])
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), f"{t / 1e6:9.1f} ms {100 * t / tot:5.1f} %  avg {float(r['AverageNs']) / 1e3:8.1f} us")
print(f"total {tot / 1e6:.1f} ms")
