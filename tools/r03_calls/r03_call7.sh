set -o pipefail
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 600 python tools/mixed_torso_probe.py > $O/mixed_torso_probe.log 2>&1; echo "rc $?" >> $O/mixed_torso_probe.log; grep -v "WARNING\|UserWarning\|warn_compute\|load_filled\|amdgpu.ids" $O/mixed_torso_probe.log | cut -c1-260
