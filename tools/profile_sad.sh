#!/usr/bin/env bash
# rocprofv3 kernel statistics of the TransE-L1 exact-fast path (tools/bench_transe_eval.py, NQ test triples).
# usage (on the GPU box): bash tools/profile_sad.sh [NQ]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sadprof
mkdir -p $OUT
NQ=${1:-2048} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sad -- python3 tools/bench_transe_eval.py > $OUT/log.txt 2>&1 < /dev/null
grep "L1\|L2\|image" $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [[ -n "$f" ]]; then head -14 "$f" | cut -c1-220 | tee $OUT/kernel_stats_head.txt; fi
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
