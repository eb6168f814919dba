#!/usr/bin/env bash
# GPU box: the rocprofv3 evidence of a round -- kernel stats of a default-workload bench run, the J/K span per build, the PMC passes
# of the J/K kernel and of the tensor build -- under gpurun_out/profiles_<TAG>/ (copy what is to be judged into profiles/).
# usage: tools/gpu_profiles_round.sh TAG
set -uo pipefail
TAG="$1"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 20 --no-cpu-baseline --no-scf > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
cp "$OUT"/stats/*/*kernel_stats.csv "$OUT/bench_kernel_stats.csv" 2>/dev/null
python3 "$ROOT/tools/jk_span_from_trace.py" "$OUT/stats" > "$OUT/jk_span.txt" 2>&1
echo "kernel stats done"
bash "$ROOT/tools/gpu_pmc_jk.sh" "$TAG" > "$OUT/pmc_jk.log" 2>&1
cp "$ROOT/gpurun_out/pmc_$TAG.json" "$OUT/pmc_jk.json" 2>/dev/null
echo "jk pmc done"
bash "$ROOT/tools/gpu_pmc_eri.sh" "$TAG" synth-400 2 > "$OUT/pmc_eri.log" 2>&1
cp "$ROOT/gpurun_out/pmc_eri_$TAG.json" "$OUT/pmc_eri.json" 2>/dev/null
echo "eri pmc done"
cd "$ROOT" && timeout -k 10 400 python3 bench.py > "$OUT/bench_default_run.json" 2> "$OUT/bench_default_run.err"
echo "default bench done"
head -c 400 "$OUT/jk_span.txt"
