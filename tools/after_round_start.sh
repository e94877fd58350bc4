#!/bin/bash
# Build-container side, AFTER `gpurun -- bash tools/round_start.sh <tag>` has merged its results into gpurun_out/: copy what is judged into
# profiles/ (tracked) under per-round names and print the numbers profiles/<tag>_summary.md is written from.
#     bash tools/after_round_start.sh r06 [reference kernel stats, default profiles/r04_bench_kernel_stats.csv]
TAG=${1:-r06}
REF=${2:-profiles/r04_bench_kernel_stats.csv}
SRC=gpurun_out/$TAG
RAW=gpurun_out/profiles_raw/$TAG
cd "$(dirname "$0")/.." || exit 1
[ -d "$SRC" ] || { echo "no $SRC: the call has not run (gpurun_out/.last_call.json: $(cut -c1-200 gpurun_out/.last_call.json 2>/dev/null))"; exit 1; }
cp -f "$SRC/box.txt" "profiles/${TAG}_box.txt" 2>/dev/null
for f in gpu_tests gpu_tests_next gpu_tests_next_fps_busy gpu_tests_next_pk_busy; do
  [ -f "$SRC/$f.log" ] && { grep -vE "^\s*$" "$SRC/$f.log" | tail -60 > "profiles/${TAG}_$f.log"; echo "$f: $(grep -E "passed|failed|error" "$SRC/$f.log" | tail -1)"; }
done
for f in bench bench_next bench_mfma_short bench_mfma_long bench_mfma_short_bwd bench_lib_losses bench_tokenizer_bf16 bench_c3 bench_c3_rows; do
  if [ -s "$SRC/$f.json" ]; then
    grep '"metric"' "$SRC/$f.json" | tail -1 > "profiles/${TAG}_$f.json"
    python - "$SRC/$f.json" "$f" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{") and '"metric"' in l]
d = json.loads(line[-1])
c = d.get("config", {})
print("%-24s %8.1f %s  %7.3f ms/step  mode %s%s  chain: %s" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"], c.get("step_mode"),
      " (fell back: %s)" % c["mode_fallback"] if c.get("mode_fallback") else "", (c.get("projection_chain") or {}).get("selected")))
PY
  else
    echo "$f: no line ($(tail -c 200 "$SRC/$f.err" 2>/dev/null | tr '\n' ' '))"
  fi
done
[ -f "$SRC/bench_tables.json" ] && cp -f "$SRC/bench_tables.json" "profiles/${TAG}_bench_tables.json"
[ -f "$SRC/mb_proj_ln.log" ] && cp -f "$SRC/mb_proj_ln.log" "profiles/${TAG}_mb_proj_ln.log" && cat "$SRC/mb_proj_ln.log"
[ -f "$SRC/pk_hazard.log" ] && cp -f "$SRC/pk_hazard.log" "profiles/${TAG}_pk_hazard.log"
[ -f "$SRC/smoke.log" ] && tail -1 "$SRC/smoke.log"
A=$(ls $RAW/bench/*kernel_stats.csv 2>/dev/null | head -1); B=$(ls $RAW/bench_next/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$A" ]; then
  cp -f "$A" "profiles/${TAG}_bench_kernel_stats.csv"
  [ -n "$B" ] && cp -f "$B" "profiles/${TAG}_bench_next_kernel_stats.csv" && python tools/ab_kernel_table.py "$REF" "$A" "$B" > "profiles/${TAG}_ab_kernels.md" && tail -14 "profiles/${TAG}_ab_kernels.md"
fi
echo "copied into profiles/${TAG}_*: now write profiles/${TAG}_summary.md section A from them and commit"
