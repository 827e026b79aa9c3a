#!/usr/bin/env bash
# kernel table of the Zipf-skewed C3 step (hub entities: long segments in the apply).  usage: bash tools/profile_zipf.sh
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/zipfprof
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o z -- python3 bench.py --workload C3z --quick --no-eval --no-cpu --steps 300 --warmup 30 > $OUT/log.txt 2>&1 < /dev/null
tail -1 $OUT/log.txt | cut -c1-300
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [[ -n "$f" ]]; then head -14 "$f" | cut -c1-200 | tee $OUT/kernel_stats_head.txt; fi
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
