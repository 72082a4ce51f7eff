#!/bin/bash
# round 6, GPU call 16: the whole GPU suite and smoke() at the round's final revision, then the default bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 4200 python -m pytest tests -q -m gpu -s > $O/gputest_final.txt 2>&1; echo "rc $?" >> $O/gputest_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.txt 2>&1; echo "rc $?" >> $O/smoke_final.txt
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "rc $?" >> $O/bench_final.err
grep -E "passed|failed|^FAILED|^ERROR" $O/gputest_final.txt | tail -8; tail -2 $O/smoke_final.txt | cut -c1-200; tail -c 400 $O/bench_final.json
