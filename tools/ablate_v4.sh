#!/usr/bin/env bash
# timing ablations of the v4 count kernel on the GPU box (wrong results by design): bash tools/ablate_v4.sh 0 1 2 3 ...
# (V4_ABLATE bits: 1 no refills, 2 no compare epilogue, 4 no MFMAs, 8 no LDS reads, 16 no wait / barrier; EMG_BF16_V4=2: both modes through v4)
# The ablated libraries are built into gpurun_out/ablate/ with their own object directory and loaded through EMGRAPH_HIP_LIB:
# emgraph_amd/lib/libemgraph_hip.so and csrc/_obj are never touched (a round-5 form of this script rebuilt the product library in place
# and left the last ablation there).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ablate
for ab in "$@"; do
  EMG_OUT_DIR="$PWD/gpurun_out/ablate/lib_$ab" EMG_OBJ_DIR="$PWD/gpurun_out/ablate/obj_$ab" EMG_EXTRA_FLAGS="-DV4_ABLATE=$ab" \
    bash emgraph_amd/csrc/build.sh >/dev/null 2>gpurun_out/ablate_build.err || { echo "build failed $ab"; tail -3 gpurun_out/ablate_build.err; continue; }
  echo "ablate $ab: $(EMGRAPH_HIP_LIB="$PWD/gpurun_out/ablate/lib_$ab/libemgraph_hip.so" EMG_BF16_V4=2 python3 tools/bench_bf16_count.py --reps 3 2>&1 | tr "\n" "|")"
  rm -rf "gpurun_out/ablate/obj_$ab" "gpurun_out/ablate/lib_$ab"
done
