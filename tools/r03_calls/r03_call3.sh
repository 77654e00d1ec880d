set -o pipefail
O=gpurun_out/r03c
mkdir -p $O
timeout -k 10 300 python tools/conv_v8_check.py > $O/v8_check.log 2>&1; echo "check rc $?" >> $O/v8_check.log; cat $O/v8_check.log
ONLY="3x3" VARIANT=0 REPS=10 timeout -k 10 300 python tools/conv_bench.py > $O/conv_bench_v0.log 2>&1; ONLY="3x3" VARIANT=8 REPS=10 timeout -k 10 300 python tools/conv_bench.py > $O/conv_bench_v8.log 2>&1
cat $O/conv_bench_v0.log $O/conv_bench_v8.log | grep -v Traceback
