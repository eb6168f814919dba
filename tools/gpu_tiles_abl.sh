#!/usr/bin/env bash
# GPU box: the wide tile kernel under the timing-only ablations of tf_jktile.hip.h (variants built by tools/build_variant.sh NAME "-DTJ_ABL_...";
# wrong numbers by construction).  usage: bash tools/gpu_tiles_abl.sh [names...]  ->  kernel ms at 4 and 8 densities per variant
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
for v in "" "$@"; do
  lib="$ROOT/tuna_amd/libtunafock${v:+_$v}.so"
  [ -f "$lib" ] || continue
  echo "## ${v:-as shipped}"
  TUNAFOCK_LIB="$lib" timeout -k 10 200 python3 "$ROOT/tools/gpu_tiles_nd.py" synth-400 6 2>&1 | grep -E "nd=4|nd=8" | sed -e 's/rel err[^|]*| //'
done
