set -o pipefail
O=gpurun_out/r03_centre2
mkdir -p $O
for i in 1 2 3; do
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_prevfold.so python bench.py --workload adm128 --steps 3 --warmup 1 --no-cpu-baseline > $O/adm128_prev_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload adm128 --steps 3 --warmup 1 --no-cpu-baseline > $O/adm128_centre_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_centre2/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
