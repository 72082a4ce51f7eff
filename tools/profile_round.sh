#!/bin/bash
# Box-side profile set of a round (run through gpurun from the repo root):  bash tools/profile_round.sh <tag> <git revision>
#   1. default bench line                                  -> gpurun_out/<tag>/bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of the bench        -> gpurun_out/<tag>/stats/
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE, SQ set), each in its own run with --kernel-trace only
#      -> gpurun_out/<tag>/pmc_traffic.json, pmc_mfma_util.txt
# The caller copies what it wants judged into profiles/.
tag=${1:-r4}; git=${2:-${SM_GIT_REV:-unknown}}   # (the box has no .git: pass the revision, e.g. $(git rev-parse --short HEAD) expanded on the submitting side)
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 2 --warmup 1 --only-value-layout --no-cpu-baseline --no-gemm-roofline --no-extras"
timeout 900 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --only-value-layout --no-cpu-baseline --no-extras --no-gemm-roofline > $out/stats.log 2>&1   # (--no-gemm-roofline: its HIP events around every GEMM launch are marker packets, ~0.5 ms of main-queue gaps per step that the timed steps of the bench do not have)
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- python3 $B > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- python3 $B > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $out/pmc_sq -o s --output-format csv -- python3 $B > $out/pmc_sq.log 2>&1
f=$(find $out/pmc_fetch -name 'f_counter_collection.csv' | head -1); w=$(find $out/pmc_write -name 'w_counter_collection.csv' | head -1)
s=$(find $out/pmc_sq -name 's_counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $f $w $out/pmc_traffic.json $git 3 dense > $out/pmc_traffic.txt 2>&1
python3 tools/sq_summary.py $s $git > $out/pmc_mfma_util.txt 2>&1
cp $(find $out/stats -name 's_kernel_stats.csv' | head -1) $out/kernel_stats.csv
python3 tools/step_timeline.py $(find $out/stats -name 's_kernel_trace.csv' | head -1) > $out/step_timeline.txt 2>&1
# the raw per-dispatch CSVs are large: keep the summaries only
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/stats
head -12 $out/kernel_stats.csv; cat $out/pmc_traffic.txt | head -8; head -8 $out/pmc_mfma_util.txt; cat $out/bench_n1.json
