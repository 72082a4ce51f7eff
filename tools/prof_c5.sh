# rocprofv3 kernel statistics of configs[4]'s per-GPU shape (tools/c5_shape_smoke.py): bash tools/prof_c5.sh [mode]
mode=${1:-fp8}; out=$PWD/gpurun_out/c5prof; mkdir -p $out; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 tools/c5_shape_smoke.py 64 248 2 $mode > $out/run_$mode.log 2>&1
cp $(find $out/stats -name 's_kernel_stats.csv' | head -1) $out/kernel_stats_$mode.csv
rm -rf $out/stats
tail -2 $out/run_$mode.log; head -32 $out/kernel_stats_$mode.csv | cut -c1-150
