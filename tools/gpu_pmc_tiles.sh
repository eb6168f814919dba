#!/usr/bin/env bash
# GPU box: PMC passes over a short run of tools/gpu_tiles_perf.py (one pass per counter group; kernel-trace only, as the pool requires),
# summarised by tools/pmc_summary.py into gpurun_out/pmc_<tag>.json.   usage: tools/gpu_pmc_tiles.sh TAG [workload] [script = gpu_tiles_perf.py]
# (script gpu_tiles_nd.py: the wide passes over 4 and 8 densities)
set -uo pipefail
TAG="$1"; WL="${2:-synth-400}"; SCRIPT="${3:-gpu_tiles_perf.py}"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/$SCRIPT" "$WL" 3 > "$OUT/p$i.log" 2>&1 || echo "pass $i ($group) failed" >> "$OUT/fail.log"
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/pmc_$TAG.json" "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "jk_tile|jk_edge" "$OUT/summary.txt" | cut -c1-2500
