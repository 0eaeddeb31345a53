# development: a variant build of the library with extra -D flags -> gpurun_<name>_libgvamp.so at the repo root (git-ignored, travels
# to the GPU box; scripts pick it up through GV_DBG_LIB).   bash scripts/build_variant.sh <name> -DGV_TILE_SCHED=2 ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
H=$(python3 -c "from gvamp_amd import build; print(build.kernel_src_hash()[:16])")
O=/tmp/gv_var_$name; mkdir -p $O
for s in gv_kernels gv_mfma gv_capi gv_solvers; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Iinclude -Igvamp_amd/csrc -I/opt/rocm/include \
        "$@" -DGV_KERNEL_SRC_HASH="\"$H\"" -c gvamp_amd/csrc/$s.hip -o $O/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_${name}_libgvamp.so $O/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la gpurun_${name}_libgvamp.so
