"""Summarise a rocprofv3 rocpd database: per-kernel time per step, stream occupancy, idle time."""
import sqlite3, re, collections, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
c = db.cursor()
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:60]
d = collections.defaultdict(lambda: [0, 0])
for n, s, e, st in rows:
    k = (short(n), st)
    d[k][0] += 1; d[k][1] += e - s
tot = sum(v[1] for v in d.values())
print(f"kernel time/step {tot/steps/1e6:.3f} ms")
for (n, st), (k, t) in sorted(d.items(), key=lambda x: -x[1][1])[:22]:
    print(f"{n:60s} stream {st} calls/step {k/steps:6.1f} ms/step {t/steps/1e6:7.3f} avg_us {t/k/1e3:8.1f} {100*t/tot:5.1f}%")
