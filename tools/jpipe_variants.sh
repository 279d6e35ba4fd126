#!/bin/bash
# builds libbicgstab_hip.so variants of k_spmm_jpipe (tail entries per trip, LDS reads in flight, wavefronts per SIMD) into
# mpi-bicgstab_amd/variants/ (git-ignored, travels with gpurun); select one with BICG_HIP_LIB=...
set -e
cd "$(dirname "$0")/../mpi-bicgstab_amd"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -Wno-unused-function"
OBJ=$(ls build/*.o | grep -v bicg_spmm_jag.o)
for v in ${VARIANTS:-"u8:-DJPIPE_U=8" "xr8:-DJPIPE_XR=8" "u8xr8:-DJPIPE_U=8_-DJPIPE_XR=8" "w2:-DJPIPE_WAVES=2" "w2u8xr8:-DJPIPE_WAVES=2_-DJPIPE_U=8_-DJPIPE_XR=8"}; do
  name=${v%%:*}; defs=$(echo "${v#*:}" | tr '_' ' ' | sed 's/JPIPE /JPIPE_/g')
  ( /opt/rocm/bin/hipcc $FLAGS $defs -Rpass-analysis=kernel-resource-usage -c csrc/bicg_spmm_jag.hip -o variants/spmm_jag_$name.o 2> variants/res_$name.txt
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/lib_$name.so $OBJ variants/spmm_jag_$name.o -ldl -lpthread
    echo "$name ($defs): $(grep -E 'VGPRs:|ScratchSize' variants/res_$name.txt | awk '{print $(NF-1)}' | paste -sd' ')" ) &
done
wait
