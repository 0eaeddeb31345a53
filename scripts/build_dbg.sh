# development: the -DGV_WGTIME build of the library (per-workgroup clocks, scripts/wgtime.py) -> gpurun_dbg_libgvamp.so at the repo root
# (git-ignored, travels to the GPU box).  Run here in the container: bash scripts/build_dbg.sh
set -e
cd "$(dirname "$0")/.."
H=$(python3 -c "from gvamp_amd import build; print(build.kernel_src_hash()[:16])")
O=/tmp/gv_dbg_objs; mkdir -p $O
for s in gv_kernels gv_mfma gv_capi gv_solvers; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Iinclude -Igvamp_amd/csrc -I/opt/rocm/include \
        -DGV_WGTIME -DGV_KERNEL_SRC_HASH="\"$H\"" -c gvamp_amd/csrc/$s.hip -o $O/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_dbg_libgvamp.so $O/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la gpurun_dbg_libgvamp.so
