# tile timing of the folded out_layers convs vs the same layers as two launches (timing build)
set -o pipefail
O=gpurun_out/r03z4
mkdir -p $O
export ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so
for F in 0 1; do
  echo "== FOLD=$F" >> $O/conv_tile_timing_fold.log
  FOLD=$F PRE=20 timeout -k 10 300 python tools/conv_timing.py >> $O/conv_tile_timing_fold.log 2>&1 || exit 1
done
cat $O/conv_tile_timing_fold.log
