# final build: rocprofv3 kernel stats + PMC traffic passes per workload, and the sequential guided attribution
set -o pipefail
bash tools/profile_round.sh fold guided adm128 adm256 sd 2> gpurun_out/profile_round_fold.err; echo "profile rc $?"
O=gpurun_out/r03_fold
export TMPDIR=/tmp
ADM_OVERLAP_GUIDANCE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o b -- python3 bench.py --workload guided --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/bench_guided_seq_profiled.json.log 2> $O/bench_guided_seq_profiled.err
find $O/stats_seq -name '*kernel_stats.csv' -exec cp {} $O/seq_guided_kernel_stats.csv \; ; rm -rf $O/stats_seq
ls -la $O | head -40
head -6 $O/seq_guided_kernel_stats.csv | cut -c1-150
