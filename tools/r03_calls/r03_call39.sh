# what bounds a skip step?  ablation builds of the folded conv (diagnostic: wrong results by design)
set -o pipefail
O=gpurun_out/r03_abl
mkdir -p $O
for A in timing abl1 abl2 abl3 abl4; do
  echo "== libadm_hip_$A.so (abl1 no activation loads, abl2 no weight loads, abl3 no LDS park, abl4 no MFMAs in the skip steps)" >> $O/fold_ablation.log
  ADM_HIP_LIB=autodiffusion_amd/libadm_hip_$A.so FOLD=1 PRE=20 timeout -k 10 300 python tools/conv_timing.py 2>&1 | grep -v amdgpu.ids | cut -c1-175 >> $O/fold_ablation.log || exit 1
done
cat $O/fold_ablation.log
