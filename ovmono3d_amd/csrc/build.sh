#!/bin/bash
# Builds libovm3d.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libovm3d.so
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
# OVM_DIAG=1: diagnostic build (in-kernel s_memtime stamps and the timing-only attention ablations of scratch/; never shipped) - delete build/ when switching
if [ -n "$OVM_DIAG" ]; then FLAGS="$FLAGS -DOVM_DIAG"; fi
pids=()
for f in gemm gemm256 gemm_small elementwise attn attn64 roi_cube det2d ops gops gdino_kernels dec_chain gdino box3d resize jpeg api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer build/$f.o)" ] || [ ../../include/ovm3d.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/gemm.o build/gemm256.o build/gemm_small.o build/elementwise.o build/attn.o build/attn64.o build/roi_cube.o build/det2d.o build/ops.o build/gops.o build/gdino_kernels.o build/dec_chain.o build/gdino.o build/box3d.o build/resize.o build/jpeg.o build/api.o -L/opt/rocm/lib -lrccl
echo "built $OUT"
