#!/bin/bash
# Collect this round's rocprofv3 summaries on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag> [workloads...]        e.g.  bash tools/profile_round.sh v2 guided adm256 sd
# Per workload W: (1) kernel stats of `bench.py --workload W` (kernel trace only), (2)+(3) HBM traffic counters in two
# SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# reduced by tools/pmc_summary.py.  Writes gpurun_out/r03_<tag>/...; tools/pmc_traffic_json.py turns (2)+(3) into
# profiles/r03/pmc_dominant_kernel_traffic.json.  The program after `--` is python3 itself (no env / bash hop).
set -o pipefail
TAG=${1:-v}
shift
WL=${@:-guided}
OUT=gpurun_out/r03_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for W in $WL; do
  STEPS=3; [ "$W" = "adm256" ] && STEPS=2
  echo "== $W: kernel stats" >&2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o b -- python3 bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/bench_${W}_profiled.json.log 2> $OUT/bench_${W}_profiled.err || exit 1
  find $OUT/stats_$W -name '*kernel_stats.csv' -exec cp {} $OUT/bench_${W}_kernel_stats.csv \;
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== $W: pmc $C" >&2
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -o p -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/pmc_${W}_$C.err || exit 1
    CSV=$(find $OUT/pmc_${W}_$C -name '*counter_collection.csv' | head -1)
    python3 tools/pmc_summary.py $CSV $OUT/pmc_${W}_$C.csv conv_kernel conv1x1r attn gn_ || exit 1
    rm -rf $OUT/pmc_${W}_$C
  done
  rm -rf $OUT/stats_$W
done
echo done >&2
