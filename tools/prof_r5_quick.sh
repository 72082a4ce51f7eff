out=$PWD/gpurun_out/r5d; mkdir -p $out; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --only-value-layout --no-cpu-baseline --no-extras > $out/stats.log 2>&1
cp $(find $out/stats -name 's_kernel_stats.csv' | head -1) $out/kernel_stats.csv
python3 tools/step_timeline.py $(find $out/stats -name 's_kernel_trace.csv' | head -1) 5 all > $out/step_timeline_all.txt 2>&1
rm -rf $out/stats
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_f -o f --output-format csv -- python3 tools/tn2_bench.py > $out/pmc_f.log 2>&1
python3 - <<'PY' > $out/tn2_fetch.txt 2>&1
import csv, glob, collections
f = glob.glob('gpurun_out/r5d/pmc_f/**/f_counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE':
        agg[(r['Kernel_Name'][:60], r['Grid_Size'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(k, 'launches', len(v), 'FETCH_SIZE(KB->x2 for 16B loads) mean', sum(v) / len(v), 'x2 in MB:', 2 * sum(v) / len(v) / 1024)
PY
rm -rf $out/pmc_f
head -30 $out/kernel_stats.csv | cut -c1-160; cat $out/tn2_fetch.txt; head -40 $out/step_timeline_all.txt
