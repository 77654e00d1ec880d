# 1x1 loop of the 128-pixel tiles without early exits: tests, then same-box A/B against the build with the exits (ADM_HIP_LIB)
set -o pipefail
O=gpurun_out/r03_noexit
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_kernels.py tests/test_hip_sd.py tests/test_hip_unet.py tests/test_hip_switches.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_exit.so python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_exit_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_noexit_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_noexit/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
