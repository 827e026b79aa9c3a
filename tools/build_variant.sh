#!/usr/bin/env bash
# Build an experimental variant of libemgraph_hip.so: tools/build_variant.sh NAME FILE "-DFLAG=1 ..." [FILE2 "FLAGS2"]
# -> emgraph_amd/lib/variants/libemgraph_hip_NAME.so (select with EMGRAPH_HIP_LIB=...).  A/B timing aid only.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; shift
CS="${ROOT}/emgraph_amd/csrc"; OBJ="${CS}/_obj"; OUT="${ROOT}/emgraph_amd/lib/variants"; TMP="${OBJ}/var_${NAME}"
mkdir -p "${OUT}" "${TMP}"
bash "${CS}/build.sh" >/dev/null
declare -A REPL
while (( "$#" )); do
  f="$1"; flags="$2"; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"${ROOT}/include" -Wall -Wno-unused-function ${flags} -c "${CS}/${f}.hip" -o "${TMP}/${f}.o"
  REPL[$f]=1
done
objs=()
for src in "${CS}"/emg_*.hip; do   # every source build.sh links (derived, so a new file cannot be forgotten here)
  f="$(basename "${src}" .hip)"
  if [[ -n "${REPL[$f]:-}" ]]; then objs+=("${TMP}/${f}.o"); else objs+=("${OBJ}/${f}.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libemgraph_hip_${NAME}.so" "${objs[@]}"
echo "built ${OUT}/libemgraph_hip_${NAME}.so"
