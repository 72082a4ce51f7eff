"""Per-kernel MFMA-pipe utilisation, wave-state shares and LDS bank conflicts from one rocprofv3 PMC pass:
   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
             SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --only-value-layout --no-cpu-baseline
MfmaUtil = MFMA_BUSY / (4 SIMDs x BUSY_CU_CYCLES); wave-state shares are of WAVE_CYCLES over ALL waves of a kernel (loader / finisher
waves of the head forward park by design).  usage: sq_summary.py counter_collection.csv [git revision]"""
import csv, collections, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_Z\d+', '', n)
    n = re.sub(r'[<(].*', '', n)
    n = re.sub(r'(I(DF16b|f)|PK).*', '', n)
    acc[n][r['Counter_Name']] += float(r['Counter_Value'])
print(f"# git {sys.argv[2] if len(sys.argv) > 2 else 'unknown'}; see tools/sq_summary.py for the command and the formulas")
for n, c in sorted(acc.items(), key=lambda kv: -kv[1]['SQ_BUSY_CU_CYCLES'])[:16]:
    wc = max(c['SQ_WAVE_CYCLES'], 1.0)
    parked = c['SQ_WAIT_ANY'] / wc; issuing = c['SQ_ACTIVE_INST_ANY'] / wc
    print(f"{n[:34]:34s} MfmaUtil {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(4 * c['SQ_BUSY_CU_CYCLES'], 1):5.1f} %   waves: issuing {100 * issuing:5.1f} %  "
          f"parked (waitcnt/barrier) {100 * parked:5.1f} %  issue-stalled {100 * max(0.0, 1 - parked - issuing):5.1f} %   "
          f"LDS bank conflicts {100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):4.1f} % of LDS cycles")
