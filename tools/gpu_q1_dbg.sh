#!/usr/bin/env bash
# GPU box: the first-quarter kernel under its timing switches (TF_Q1_DBG), kernel time from rocprofv3
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  OUT="$ROOT/gpurun_out/q1dbg_$d"
  TF_Q1_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/gpu_mp2_repeat.py" > "$OUT.log" 2>&1
  python3 - "$OUT" "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "mo_q1" in r["Name"]: print("TF_Q1_DBG=" + sys.argv[2], f'{float(r["AverageNs"])/1e6:8.3f} ms')
PY
done
