#!/usr/bin/env bash
# GPU box: PMC passes over a short bench run (one pass per counter group; kernel-trace only, as the pool requires), summarised by
# tools/pmc_summary.py into gpurun_out/pmc_<tag>.json.   usage: tools/gpu_pmc_jk.sh TAG [bench args...]
set -uo pipefail
TAG="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-scf "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($group) failed" >> "$OUT/fail.log"
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE GRBM_COUNT
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_GATE_EN1_sum
GROUPS
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/pmc_$TAG.json" "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "jk_packed_kernel|jk_reduce" "$OUT/summary.txt" | cut -c1-1500
