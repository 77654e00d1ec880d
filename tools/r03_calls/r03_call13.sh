set -o pipefail
O=gpurun_out/r03l
mkdir -p $O
export ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so
XCD_REPORT=1 PRE=20 VARIANT=0 timeout -k 10 300 python tools/conv_timing.py > $O/timing_xcd.log 2>&1
cut -c1-260 $O/timing_xcd.log
