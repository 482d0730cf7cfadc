#!/bin/bash
# profiles/<tag>_kernel_resources.txt: registers, spills, scratch, LDS and occupancy of every kernel, as the compiler reports them
# (hipcc -Rpass-analysis=kernel-resource-usage with the flags of all_is_cubes_amd/csrc/Makefile). Runs anywhere hipcc does.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
C=all_is_cubes_amd/csrc
OUT=profiles/${TAG}_kernel_resources.txt
{
  echo "# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the flags of all_is_cubes_amd/csrc/Makefile (aic_trace.hip, then aic_light.hip)"
  echo "# trace_image_kernel<VOL, LMODE, DIAG, BIG, XC>: the production variants are DIAG=false (XC: with the lane exchange) (launch bound 4 waves/SIMD = 128 VGPRs); DIAG=true"
  echo "# (aux records, counters) is compiled for 2 waves/SIMD. No scratch and no VGPR spill in any production variant."
  for f in aic_trace.hip aic_light.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $C/$f -o /dev/null 2>&1 |
      grep "remark:" | sed -e 's/^.*remark: //' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | c++filt
  done
} > $OUT
grep -c "Function Name" $OUT
