# tile width for the 512- / 1024-channel levels: 192-wide (padded) vs 128-wide
set -o pipefail
O=gpurun_out/r03w
mkdir -p $O
for V in 6 5 6 5; do
  echo "== VARIANT=$V" >> $O/conv_bench_wide.log
  SHAPESET=wide VARIANT=$V REPS=10 timeout -k 10 300 python tools/conv_bench.py >> $O/conv_bench_wide.log 2>&1 || exit 1
done
cat $O/conv_bench_wide.log
