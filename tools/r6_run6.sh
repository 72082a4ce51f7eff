#!/bin/bash
# round 6, GPU call 6: the whole GPU suite at the round's kernel sources, then smoke()
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 4200 python -m pytest tests -q -m gpu -s > $O/gputest_full.txt 2>&1; echo "rc $?" >> $O/gputest_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "rc $?" >> $O/smoke.txt
grep -E "passed|failed|^FAILED|^ERROR" $O/gputest_full.txt | tail -15; tail -3 $O/smoke.txt
