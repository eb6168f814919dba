#!/usr/bin/env bash
# GPU box: jk_reduce_kernel under its timing switch (TF_JKR_DBG: 1 = exchange blocks only, 2 = Jt blocks only), kernel time from rocprofv3
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  OUT="$ROOT/gpurun_out/jkrdbg_$d"
  TF_JKR_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 20 --no-cpu-baseline --no-scf > "$OUT.log" 2>&1
  python3 - "$OUT" "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "jk_reduce" in r["Name"] or "jk_packed_final" in r["Name"] or "pack_density" in r["Name"]: print("TF_JKR_DBG=" + sys.argv[2], r["Name"][:40], f'{float(r["AverageNs"])/1e6:8.4f} ms')
PY
done
