# gn_finalize with 32 lanes per group (grid n x 4): tests, then same-box A/B against the previous kernel (ADM_HIP_LIB)
set -o pipefail
O=gpurun_out/r03_gnfin
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_kernels.py tests/test_hip_classifier.py tests/test_hip_fullsize.py tests/test_hip_sd.py tests/test_variants.py tests/test_hip_bigbatch.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_prevnorm.so python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_prev_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_new_$i.json.log 2>> $O/bench.err || exit 1
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_prevnorm.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_prev_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_new_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_gnfin/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
