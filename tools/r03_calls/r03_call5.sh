set -o pipefail
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 300 python tools/conv_v8_check.py > $O/v8_check.log 2>&1; echo "check rc $?" >> $O/v8_check.log; grep -c "out equal True" $O/v8_check.log; tail -2 $O/v8_check.log
export ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so
PRE=20 VARIANT=0 timeout -k 10 300 python tools/conv_timing.py > $O/timing_v0.log 2>&1
PRE=20 VARIANT=8 timeout -k 10 300 python tools/conv_timing.py > $O/timing_v8.log 2>&1
unset ADM_HIP_LIB
cut -c1-330 $O/timing_v0.log $O/timing_v8.log | grep -v "skipped\|no stamps"
ONLY="3x3" VARIANT=0 REPS=10 timeout -k 10 300 python tools/conv_bench.py 2>&1 | grep -v Traceback | head -6 > $O/conv_bench_v0.log; ONLY="3x3" VARIANT=8 REPS=10 timeout -k 10 300 python tools/conv_bench.py 2>&1 | head -6 > $O/conv_bench_v8.log
cat $O/conv_bench_v0.log $O/conv_bench_v8.log
