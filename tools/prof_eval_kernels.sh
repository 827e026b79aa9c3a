#!/usr/bin/env bash
# rocprofv3 kernel trace of tools/bench_exact_fast.py (the three ranking precisions at C4 size) -> gpurun_out/TAG_eval_kernel_stats.md
# usage: bash tools/prof_eval_kernels.sh TAG   (environment switches such as EMG_BF16_V4 pass through)
TAG="$1"; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o p -- python3 tools/bench_exact_fast.py "$@" > gpurun_out/prof_$TAG.log 2>&1
db=$(ls gpurun_out/prof_$TAG/*/*results.db gpurun_out/prof_$TAG/*results.db 2>/dev/null | head -1)
python3 profiles/summarize_rocpd.py "$db" gpurun_out/${TAG}_eval_kernel_stats.md > /dev/null
H=$(python3 -c "from emgraph_amd import _lib; print(_lib.load().emg_source_hash().decode())")
sed -i "1i <!-- source_hash: $H | rocprofv3 --kernel-trace --stats of: tools/bench_exact_fast.py $* -->" gpurun_out/${TAG}_eval_kernel_stats.md
rm -rf gpurun_out/prof_$TAG
