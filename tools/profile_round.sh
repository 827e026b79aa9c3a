#!/usr/bin/env bash
# The round's judged evidence in one call on the GPU box:  bash tools/profile_round.sh TAG
#   gpurun_out/bench_TAG.json           python3 bench.py --gpus 1 --steps 20 --warmup 5           (the driver's command)
#   gpurun_out/TAG_kernel_stats.md      rocprofv3 --kernel-trace --stats of the same command (--no-cpu)
#   gpurun_out/TAG_pmc_traffic.json     tools/pmc_traffic.sh (FETCH_SIZE / WRITE_SIZE, separate passes)
TAG="${1:-r2_x}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/prof_$TAG.log 2>&1
db=$(ls gpurun_out/prof_$TAG/*/*results.db gpurun_out/prof_$TAG/*results.db 2>/dev/null | head -1)
python3 profiles/summarize_rocpd.py "$db" gpurun_out/${TAG}_kernel_stats.md
tail -1 gpurun_out/prof_$TAG.log > gpurun_out/${TAG}_profiled_bench.json
rm -rf gpurun_out/prof_$TAG
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc_traffic
