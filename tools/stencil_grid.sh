export BICG_HIP_LIB=$GRAFT_REPO_ROOT/mpi-bicgstab_amd/variants/lib_x.so PRODUCT_ONLY=1
ARGS=()
for w in 2 4; do for x in 0 1 2; do for l in 0 36864 49152 65536; do for p in 16 32 64; do
  ARGS+=("wide=$w planes=$p BICG_STENCIL_XCD=$x BICG_STENCIL_LDS=$l")
done; done; done; done
timeout 1000 python tools/stencil_sweep.py 512 "${ARGS[@]}" 2>&1 | cut -c1-120 | tee gpurun_out/stencil_grid.txt | sort -k5 -n -t']' | awk '{print}' | sort -t't' -k3 | head -0
sort -k2 -t']' gpurun_out/stencil_grid.txt | head -20
