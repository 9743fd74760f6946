#!/bin/bash
# Builds tools/probe/pk_hazard.hip twice (the SLP-vectorised form that failed in round 3 and the
# shipped form), counts the packed f32 ops of fps_regs_kernel<4,8> in each and runs both under
# the three co-runner settings.  Output: gpurun_out/pk_hazard.txt (copy into profiles/).
set -u
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/pk_hazard.txt}
mkdir -p "$(dirname "$OUT")"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -DBTR_FMAD=1 -Ibacktoreality_amd/csrc -Iinclude -Wno-unused-function"
{
  echo "# pk_hazard: $(/opt/rocm/bin/hipcc --version | head -1)"
  for form in slp ok; do
    if [ $form = slp ]; then FL="-fslp-vectorize -DBTR_PK_REPRO_NO_FENCE"; else FL="-fno-slp-vectorize -fno-vectorize"; fi
    /opt/rocm/bin/hipcc $COMMON $FL tools/probe/pk_hazard.hip -o /tmp/pk_hazard_$form --save-temps=obj 2>/dev/null \
      || /opt/rocm/bin/hipcc $COMMON $FL tools/probe/pk_hazard.hip -o /tmp/pk_hazard_$form || exit 1
    /opt/rocm/bin/hipcc $COMMON $FL --offload-device-only -S tools/probe/pk_hazard.hip -o /tmp/pk_hazard_$form.s 2>/dev/null
    n=$(awk '/^_ZN3btr15fps_regs_kernelILi4ELi8EEEviiiiPKfPiPKii:/{p=1} p&&/s_endpgm/{exit} p' /tmp/pk_hazard_$form.s | grep -c 'v_pk_[a-z]*_f32')
    echo "form=$form flags='$FL' packed_f32_ops_in_fps_regs_kernel<4,8>=$n"
    for load in 0 1 2; do
      echo -n "form=$form  "
      timeout 300 /tmp/pk_hazard_$form ${LAUNCHES:-4000} $load
    done
  done
} | tee "$OUT"
