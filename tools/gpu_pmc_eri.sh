#!/usr/bin/env bash
# GPU box: SQ counter passes over tensor builds (tools/gpu_eri_build.py), one counter group per pass, kernel trace only;
# summary by tools/pmc_eri_summary.py.   usage: tools/gpu_pmc_eri.sh TAG [workload] [builds]
set -uo pipefail
TAG="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/pmc_eri_$TAG"
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/gpu_eri_build.py" "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($group) failed" >> "$OUT/fail.log"
done <<'GROUPS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GROUPS
python3 "$ROOT/tools/pmc_eri_summary.py" "$ROOT/gpurun_out/pmc_eri_$TAG.json" "$OUT" | tee "$OUT/summary.txt"
