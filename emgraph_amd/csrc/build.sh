#!/usr/bin/env bash
# Build libemgraph_hip.so for gfx950 (MI355X), in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${EMG_OUT_DIR:-${HERE}/../lib}"      # (EMG_OUT_DIR / EMG_OBJ_DIR: timing ablations build beside the product, never over it)
OBJ="${EMG_OBJ_DIR:-${HERE}/_obj}"
mkdir -p "${OUT}" "${OBJ}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"${HERE}/../../include" -Wall -Wno-unused-function ${EMG_EXTRA_FLAGS:-})   # EMG_EXTRA_FLAGS: timing ablations (tools/ablate_v4.sh)
# the hash of every kernel source goes into the library (emg_source_hash): profiles/ name the binary they measured
# (with the extra flags, if any: an ablated or variant build never carries the product's hash)
SRC_HASH="$(cd "${HERE}" && { cat emg_*.hip emg_*.hpp emg_*.inc ../../include/emgraph_hip.h; printf '%s' "${EMG_EXTRA_FLAGS:-}"; } | sha256sum | cut -c1-16)"
# a changed hash (sources OR flags) rebuilds everything when the flags changed, emg_abi.o (which carries the hash) otherwise
FLAG_SIG="$(printf '%s' "${EMG_EXTRA_FLAGS:-}" | sha256sum | cut -c1-16)"
if [[ ! -f "${OBJ}/flag_sig.txt" || "$(cat "${OBJ}/flag_sig.txt")" != "${FLAG_SIG}" ]]; then rm -f "${OBJ}"/emg_*.o; echo "${FLAG_SIG}" > "${OBJ}/flag_sig.txt"; fi
if [[ ! -f "${OBJ}/src_hash.txt" || "$(cat "${OBJ}/src_hash.txt")" != "${SRC_HASH}" ]]; then rm -f "${OBJ}/emg_abi.o"; echo "${SRC_HASH}" > "${OBJ}/src_hash.txt"; fi
pids=()
for f in emg_abi emg_score emg_fused_m0 emg_fused_m1 emg_fused_m2 emg_fused_m3 emg_fused_m4 emg_train emg_group emg_group_bucket emg_apply emg_rank emg_rank_bf16 emg_rank_sad emg_api emg_plan; do
  src="${HERE}/${f}.hip"; obj="${OBJ}/${f}.o"
  if [[ ! -f "${obj}" || "${src}" -nt "${obj}" || "${HERE}/emg_common.hpp" -nt "${obj}" || "${HERE}/emg_group.hpp" -nt "${obj}" || "${HERE}/emg_group_kernels.hpp" -nt "${obj}" || "${HERE}/emg_score_kernels.hpp" -nt "${obj}" || "${HERE}/emg_fused_inst.inc" -nt "${obj}" || "${HERE}/../../include/emgraph_hip.h" -nt "${obj}" ]]; then
    "${HIPCC}" "${FLAGS[@]}" -DEMG_SRC_HASH="\"${SRC_HASH}\"" -c "${src}" -o "${obj}" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "${p}" ]] && wait "${p}"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libemgraph_hip.so" "${OBJ}"/emg_*.o
echo "built ${OUT}/libemgraph_hip.so"
