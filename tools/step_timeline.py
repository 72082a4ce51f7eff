"""One training step of a rocprofv3 kernel trace (bench.py run) as a timeline: per queue busy time, gaps on the main queue,
and the kernels in launch order with start offsets.  usage: step_timeline.py s_kernel_trace.csv [step index from the end]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_Z\d+', '', n)
    return re.sub(r'[<(].*', '', n)[:28]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(r['Kernel_Name'])) for r in rows))
# a step starts at each embed_fwd of the document encoder
starts = [i for i, e in enumerate(ev) if e[3].startswith('embed_fwd')]
i0, i1 = starts[-back - 1], starts[-back]
step = ev[i0:i1]
t0 = step[0][0]; t1 = max(e[1] for e in step)
print(f"step wall {(ev[i1][0] - t0) / 1e3:.1f} us, kernels {len(step)}")
byq = collections.defaultdict(list)
for s, e, q, n in step: byq[q].append((s, e, n))
for q, l in byq.items():
    busy = sum(e - s for s, e, _ in l)
    print(f"queue {q}: {len(l)} kernels, busy {busy / 1e3:.1f} us")
mainq = max(byq, key=lambda q: len(byq[q]))
l = byq[mainq]
gaps = [(l[i + 1][0] - l[i][1], l[i][2], l[i + 1][2]) for i in range(len(l) - 1)]
print(f"main queue gaps: total {sum(g for g, _, _ in gaps) / 1e3:.1f} us; largest:")
for g, a, b in sorted(gaps, reverse=True)[:12]: print(f"   {g / 1e3:7.1f} us between {a} and {b}")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, q, n in step: agg[(q, n)][0] += 1; agg[(q, n)][1] += e - s
for (q, n), (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]: print(f"   q{q} {n:28s} x{k:3d} {t / 1e3:8.1f} us")
if len(sys.argv) > 3 and sys.argv[3] == "all":  # every kernel of the step in start order: offset, duration, queue, gap to its queue predecessor
    last_end = {}
    print("   offset_us  dur_us  gap_us  q  kernel")
    for s, e, q, n in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        print(f"   {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  {'M' if q == mainq else 's'}  {n}")
