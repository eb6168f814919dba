#!/usr/bin/env bash
# GPU box: jk_reduce_kernel time of library variants (tools/build_variant.sh NAME ...): bench run under rocprofv3 per variant
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT="$ROOT/gpurun_out/jkrab_$v"
  LIB="$ROOT/tuna_amd/libtunafock_$v.so"; [ "$v" = base ] && LIB="$ROOT/tuna_amd/libtunafock.so"
  TUNAFOCK_LIB=$LIB timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 20 --no-cpu-baseline --no-scf > "$OUT.log" 2>&1
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "jk_reduce" in r["Name"]: print(sys.argv[2], r["Name"][:40], f'{float(r["AverageNs"])/1e6:8.4f} ms')
try:
    d = json.loads(open(sys.argv[1] + ".log").read().strip().splitlines()[-1]); print(sys.argv[2], "builds/s", d["value"], "ms/step", d["ms_per_step"])
except Exception as e: print("no json", e)
PY
done
