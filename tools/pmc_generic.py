"""Per-kernel means of every counter in a rocprofv3 counter_collection.csv (any --pmc set).  usage: pmc_generic.py file.csv [name filter]"""
import csv, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))[:70]
    if len(sys.argv) > 2 and sys.argv[2] not in n: continue
    acc[n][r['Counter_Name']] += float(r['Counter_Value']); cnt[n][r['Counter_Name']] += 1
for n in acc:
    print(n)
    for c in sorted(acc[n]): print(f"   {c:28s} {acc[n][c] / cnt[n][c]:16.0f}  (x{cnt[n][c]})")
