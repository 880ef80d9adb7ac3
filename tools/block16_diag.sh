#!/bin/bash
# A/B of diagnostic builds of kernels_block16.hip (VISP_BLOCK16_DBG bits, see the kernel): what each part of the stream costs
set -e
cd vision.cpp_amd/csrc
for dbg in ${DBGS:-0 1 3 8 11 32 64}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 ${EXTRA} -DVISP_BLOCK16_DBG=$dbg -c kernels_block16.hip -o build/kernels_block16.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libvisioncpp_dbg$dbg.so build/*.o -Wl,--no-undefined
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 -c kernels_block16.hip -o build/kernels_block16.o
cd ../..
for dbg in ${DBGS:-0 1 3 8 11 32 64}; do
  echo "== DBG=$dbg"
  VISP_LIBRARY=vision.cpp_amd/lib/libvisioncpp_dbg$dbg.so python tools/bench_block.py --only16 2>&1 | grep block16 | head -3
done
rm -f vision.cpp_amd/lib/libvisioncpp_dbg*.so
