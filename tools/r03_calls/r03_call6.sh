set -o pipefail
O=gpurun_out/r03f
mkdir -p $O
python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_sd.py tests/test_hip_classifier.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
PREV=autodiffusion_amd/libadm_hip_prev.so
REPS=10 ADM_HIP_LIB=$PREV python tools/conv_bench.py > $O/conv_bench_prev.log 2>&1
REPS=10 python tools/conv_bench.py > $O/conv_bench_new.log 2>&1
paste <(cut -c1-75 $O/conv_bench_prev.log) <(cut -c37-75 $O/conv_bench_new.log)
for i in 1 2; do
ADM_HIP_LIB=$PREV python bench.py --steps 5 --no-cpu-baseline > $O/guided_prev$i.json.log 2>/dev/null
python bench.py --steps 5 --no-cpu-baseline > $O/guided_new$i.json.log 2>/dev/null
done
for f in prev1 new1 prev2 new2; do python -c "import json,sys; d=json.loads([l for l in open('$O/guided_$f.json.log') if l.startswith('{')][0]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['frac'], r['isolated']['frac'], r['avg_launch_us'])"; done
