# 1x1 loop of the 128-pixel tiles: exits + ring of 2 (libadm_hip_exit.so) vs no exits + 4 in flight (libadm_hip_deep4.so) vs 8 in flight (default build), one box
set -o pipefail
O=gpurun_out/r03_deep
mkdir -p $O
for i in 1 2 3; do
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_exit.so python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_exit_$i.json.log 2>> $O/bench.err || exit 1
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_deep4.so python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_deep4_$i.json.log 2>> $O/bench.err || exit 1
  python bench.py --workload sd --steps 3 --warmup 1 --no-cpu-baseline > $O/sd_deep8_$i.json.log 2>> $O/bench.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_deep/*.json.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'])
PY
