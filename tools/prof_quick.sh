#!/usr/bin/env bash
# rocprofv3 kernel trace of one bench configuration -> gpurun_out/TAG_kernel_stats.md.  usage: bash tools/prof_quick.sh TAG [bench args...]
TAG="$1"; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o p -- python3 bench.py --no-cpu --no-eval --no-others --sustained-seconds 0 --no-ceilings "$@" > gpurun_out/prof_$TAG.log 2>&1
db=$(ls gpurun_out/prof_$TAG/*/*results.db gpurun_out/prof_$TAG/*results.db 2>/dev/null | head -1)
python3 profiles/summarize_rocpd.py "$db" gpurun_out/${TAG}_kernel_stats.md > /dev/null
# the first line names the binary the table measured (bench.py::profiled_avg_us quotes a table only for the library it runs)
H=$(python3 -c "from emgraph_amd import _lib; print(_lib.load().emg_source_hash().decode())")
sed -i "1i <!-- source_hash: $H | rocprofv3 --kernel-trace --stats of: bench.py --no-cpu --no-eval --no-others --sustained-seconds 0 --no-ceilings $* -->" gpurun_out/${TAG}_kernel_stats.md
grep '^{"metric"' gpurun_out/prof_$TAG.log | tail -1 > gpurun_out/${TAG}_profiled_bench.json
rm -rf gpurun_out/prof_$TAG
