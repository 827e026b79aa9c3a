#!/usr/bin/env bash
# Round 6's judged evidence in one call on the GPU box:  bash tools/profile_round6.sh TAG
#   gpurun_out/TAG_bench.json                 python3 bench.py --gpus 1 --steps 20 --warmup 5          (the driver's command)
#   gpurun_out/TAG_<wl>_kernel_stats.md       rocprofv3 --kernel-trace --stats of each workload ALONE (C3 C3a C3g C1 C2 C5)
#   gpurun_out/TAG_pmc_traffic.json           tools/pmc_traffic.sh (C3: the file bench.py's roofline.traffic reads)
#   gpurun_out/TAG_pmc_traffic_all.json       tools/pmc_traffic_all.sh (FETCH_SIZE / WRITE_SIZE of every kernel of C3 C3a C3g C3r C1 C2 C5)
TAG="${1:-r6_x}"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# (the headline workload's kernel table FIRST, on the fresh box, like the driver's bench: a box that has run for ~20 s drops to a
#  slower state — the same binary's scoring kernel measures 0.221 ms first and 0.254 ms half a minute later, r5_z)
bash tools/prof_quick.sh ${TAG}_c3 --workload C3 --steps 100 --warmup 20
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
for wl in C3a C3g C3r C1 C2 C5; do
  lw=$(echo $wl | tr 'A-Z' 'a-z')
  steps=100; [[ $wl == C1 || $wl == C2 || $wl == C5 ]] && steps=300
  bash tools/prof_quick.sh ${TAG}_$lw --workload $wl --steps $steps --warmup 20
done
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc_traffic
bash tools/pmc_traffic_all.sh gpurun_out/${TAG}_pmc_traffic_all.json C3 C3a C3g C3r C1 C2 C5 > gpurun_out/${TAG}_pmc_traffic_all.log 2>&1
tail -c 600 gpurun_out/${TAG}_bench.json
# the ranking path's kernels (three precisions at C4 size; tools/bench_exact_fast.py)
bash tools/prof_eval_kernels.sh ${TAG}
