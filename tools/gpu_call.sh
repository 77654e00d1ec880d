#!/bin/bash
# ONE parameterised recipe for everything this repo runs on the GPU box (replaces the per-call scripts of rounds 1-3).
# Run it through gpurun from the repo root; steps are joined with `--` and stop at the first failure:
#
#   gpurun --timeout 1200 -- 'bash tools/gpu_call.sh <tag> <step> [-- <step> ...]'
#
# steps (every step writes under gpurun_out/<tag>/):
#   tests [pytest args...]              python -m pytest -x -q -m gpu <args>            -> pytest_<n>.log
#   bench <name> [bench.py args...]     one bench.py line                                -> bench_<name>.json(.err)
#   ab <ENVVAR> <v1,v2,...> <rounds> [bench.py args...]
#                                       same-box A/B of an environment switch: `rounds` interleaved passes over the values
#                                       ("-" = unset)                                    -> ab_<ENVVAR>_<value>_<round>.json
#   stats <workload> [bench.py args...] rocprofv3 --kernel-trace --stats of bench.py     -> bench_<workload>_kernel_stats.csv
#   pmc <workload> [bench.py args...]   HBM traffic of the workload's kernels: SEPARATE --pmc FETCH_SIZE / WRITE_SIZE passes
#                                       (they do not share a pass), reduced by tools/pmc_summary.py -> pmc_<workload>_<counter>.csv
#   py <script> [args...]               python <script> <args>                           -> <script basename>.log
# The program after rocprofv3's `--` is python3 itself (no env / bash hop: the profiler's preload initialises the GPU).
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
n=0
run_step() {
  local kind=$1; shift
  n=$((n + 1))
  case $kind in
    tests)
      python -m pytest -x -q -m gpu "$@" > "$OUT/pytest_$n.log" 2>&1 || { tail -30 "$OUT/pytest_$n.log"; return 1; }
      tail -2 "$OUT/pytest_$n.log" ;;
    bench)
      local name=$1; shift
      python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { tail -20 "$OUT/bench_$name.err"; return 1; }
      python - "$OUT/bench_$name.json" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = l.get("roofline") or {}
print(sys.argv[1], l["value"], l["unit"], "ms/step", l["ms_per_step"], "roofline", r.get("frac"), (r.get("isolated") or {}).get("frac"),
      {k: v.get("value", v.get("error")) for k, v in (l.get("secondary") or {}).items()})
PY
      ;;
    ab)
      local var=$1 vals=$2 rounds=$3; shift 3
      for r in $(seq 1 "$rounds"); do
        for v in ${vals//,/ }; do
          if [ "$v" = "-" ]; then
            env -u "$var" python bench.py "$@" > "$OUT/ab_${var}_unset_$r.json" 2> "$OUT/ab_${var}.err" || { tail -20 "$OUT/ab_${var}.err"; return 1; }
            echo "$var unset round $r: $(grep -o '"value": [0-9.]*' "$OUT/ab_${var}_unset_$r.json" | head -1)"
          else
            env "$var=$v" python bench.py "$@" > "$OUT/ab_${var}_${v}_$r.json" 2> "$OUT/ab_${var}.err" || { tail -20 "$OUT/ab_${var}.err"; return 1; }
            echo "$var=$v round $r: $(grep -o '"value": [0-9.]*' "$OUT/ab_${var}_${v}_$r.json" | head -1)"
          fi
        done
      done ;;
    stats)
      local w=$1; shift
      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$w" -o b -- python3 bench.py --workload "$w" --no-cpu-baseline --secondary none "$@" \
        > "$OUT/bench_${w}_profiled.json" 2> "$OUT/bench_${w}_profiled.err" || { tail -20 "$OUT/bench_${w}_profiled.err"; return 1; }
      find "$OUT/stats_$w" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_${w}_kernel_stats.csv" \;
      rm -rf "$OUT/stats_$w"
      head -6 "$OUT/bench_${w}_kernel_stats.csv" | cut -c1-200 ;;
    pmc)
      local w=$1; shift
      for C in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_${w}_$C" -o p -- python3 bench.py --workload "$w" --steps 1 --warmup 0 \
          --no-cpu-baseline --no-kernel-events --secondary none "$@" > /dev/null 2> "$OUT/pmc_${w}_$C.err" || { tail -20 "$OUT/pmc_${w}_$C.err"; return 1; }
        CSV=$(find "$OUT/pmc_${w}_$C" -name '*counter_collection.csv' | head -1)
        python3 tools/pmc_summary.py "$CSV" "$OUT/pmc_${w}_$C.csv" conv_kernel conv1x1r attn gn_ || return 1
        rm -rf "$OUT/pmc_${w}_$C"
      done ;;
    py)
      local script=$1; shift
      python "$script" "$@" > "$OUT/$(basename "$script" .py).log" 2>&1 || { tail -30 "$OUT/$(basename "$script" .py).log"; return 1; }
      tail -15 "$OUT/$(basename "$script" .py).log" ;;
    *) echo "unknown step $kind" >&2; return 2 ;;
  esac
}
args=()
for a in "$@" --; do
  if [ "$a" = "--" ]; then
    if [ ${#args[@]} -gt 0 ]; then
      echo "== step: ${args[*]}" >&2
      run_step "${args[@]}" || exit 1
    fi
    args=()
  else
    args+=("$a")
  fi
done
echo "all steps done" >&2
