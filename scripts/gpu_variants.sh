#!/bin/bash
# parity of every build of the G2P adjoint + A/B.  usage: scripts/gpu_variants.sh <tag>
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
for V in 1 2 3; do
  echo "== FE_G2P_GRAD_V=$V"; FE_G2P_GRAD_V=$V timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_random_scenes.py tests/test_kernel_golden.py -m gpu -q 2>&1 | tail -2 | tee -a $OUT/variants.txt
done
timeout 900 python scripts/ab_phases.py --reps 2 "g2p_grad_v=1" "g2p_grad_v=2" "g2p_grad_v=3" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt > /dev/null
