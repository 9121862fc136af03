#!/bin/bash
# build an ablation / instrumentation variant of the library next to the product .so:  tools/build_variant.sh <suffix> <extra hipcc flags...>
# (FDC_PK=+ tools/build_variant.sh ...: WITH packed fp32 instructions, the pre-r3 code generation; see __graft_entry__.build)
set -e
cd "$(dirname "$0")/.."
sfx=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -Xclang -target-feature -Xclang ${FDC_PK:--}packed-fp32-ops $([ "${FDC_PK:--}" = "+" ] && echo -DFDC_BUILD_ALLOW_PK_F32 || echo -DFDC_BUILD_NO_PK_F32) "$@" \
  -o 4dcapture-fpv_amd/libfdcap_hip_$sfx.so 4dcapture-fpv_amd/csrc/fdcap.hip
