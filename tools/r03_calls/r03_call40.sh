# centre-only staging of the skip steps: same-box A/B against the previous build (ADM_HIP_LIB), guided / adm256 / adm128
set -o pipefail
O=gpurun_out/r03_centre
mkdir -p $O
for i in 1 2; do
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_prevfold.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_prev_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/guided_centre_$i.json.log 2>> $O/bench.err || exit 1
done
for W in adm256 adm128; do
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_prevfold.so python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/${W}_prev.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/${W}_centre.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_centre/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
