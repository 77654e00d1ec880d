set -o pipefail
O=gpurun_out/r03v
mkdir -p $O
CHECK_VARIANT=9 timeout -k 10 300 python tools/conv_v8_check.py > $O/v9_check.log 2>&1; grep -c "out equal True" $O/v9_check.log
ONLY="3x3" VARIANT=0 REPS=10 timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep -v Traceback | head -5 > $O/conv_bench_v0.log; ONLY="3x3" VARIANT=9 REPS=10 timeout -k 10 300 python tools/conv_bench.py 2>&1 | head -5 > $O/conv_bench_v9.log
cat $O/conv_bench_v0.log $O/conv_bench_v9.log
