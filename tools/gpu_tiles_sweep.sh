#!/usr/bin/env bash
# GPU box: tools/gpu_tiles_perf.py over a list of settings; every line of stdin = "LIBNAME VAR=value ..." (LIBNAME: base or a variant of
# tools/build_variant.sh).  usage: tools/gpu_tiles_sweep.sh [workload] [steps] < settings
WL="${1:-synth-400}"; STEPS="${2:-20}"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
while read -r lib rest; do
  [ -z "$lib" ] && continue
  L="$ROOT/tuna_amd/libtunafock.so"; [ "$lib" != base ] && L="$ROOT/tuna_amd/libtunafock_$lib.so"
  out=$(env TUNAFOCK_LIB="$L" $rest timeout -k 10 120 python "$ROOT/tools/gpu_tiles_perf.py" "$WL" "$STEPS" 2>/dev/null | cut -c30-170)
  echo "$lib $rest :: $out"
done
