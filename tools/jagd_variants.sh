#!/bin/bash
# builds libbicgstab_hip.so variants of k_spmv_jagd (entries per batch, wavefronts per SIMD, third-batch prefetch) into
# mpi-bicgstab_amd/variants/ (git-ignored, travels with gpurun); select one with BICG_HIP_LIB=...
set -e
cd "$(dirname "$0")/../mpi-bicgstab_amd"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -Wno-unused-function"
OBJ=$(ls build/*.o | grep -v bicg_jagw.o)
for v in "u6w5p1:-DJAGD_U=6 -DJAGD_WAVES=5 -DJAGD_PREFETCH=1" "u6w5p2:-DJAGD_U=6 -DJAGD_WAVES=5 -DJAGD_PREFETCH=2" "u7w5p1:-DJAGD_U=7 -DJAGD_WAVES=5 -DJAGD_PREFETCH=1" "u8w4p1:-DJAGD_U=8 -DJAGD_WAVES=4 -DJAGD_PREFETCH=1" "u5w5p2:-DJAGD_U=5 -DJAGD_WAVES=5 -DJAGD_PREFETCH=2" "u6w4p2:-DJAGD_U=6 -DJAGD_WAVES=4 -DJAGD_PREFETCH=2"; do
  name=${v%%:*}; defs=${v#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -Rpass-analysis=kernel-resource-usage -c csrc/bicg_jagw.hip -o variants/jagw_$name.o 2> variants/res_$name.txt
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/lib_$name.so $OBJ variants/jagw_$name.o -ldl -lpthread
    echo "$name: $(grep -A8 'k_spmv_jagdILi1ELb0ELi0ELb1' variants/res_$name.txt | grep -E 'VGPRs:|ScratchSize' | awk '{print $(NF-1)}' | paste -sd' ')" ) &
done
wait
