#!/usr/bin/env bash
# timing ablations of the v4 count kernel on the GPU box (wrong results by design): bash tools/ablate_v4.sh 0 1 2 3 ...
# (V4_ABLATE bits: 1 no refills, 2 no compare epilogue, 4 no MFMAs, 8 no LDS reads, 16 no wait / barrier; EMG_BF16_V4=2: both modes through v4)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for ab in "$@"; do
  touch emgraph_amd/csrc/emg_rank_bf16.hip
  EMG_EXTRA_FLAGS="-DV4_ABLATE=$ab" bash emgraph_amd/csrc/build.sh >/dev/null 2>gpurun_out/ablate_build.err || { echo "build failed $ab"; tail -3 gpurun_out/ablate_build.err; continue; }
  echo "ablate $ab: $(EMG_BF16_V4=2 python3 tools/bench_bf16_count.py --reps 3 2>&1 | tr "\n" "|")"
done
