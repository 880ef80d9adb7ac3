#!/bin/bash
# A/B of builds of kernels_block16.hip on ONE box: VARIANTS is a list of name=flags (flags: -DVISP_BLOCK16_DBG=n diagnostic bits,
# -DVISP_BLOCK16_PF=n fragment window, -DVISP_BLOCK16_DEFER=n / _DEFER_MLP=n groups run after the next pair's boundary, ...); launch times at batch 23 / 32 / 11
set -e
cd vision.cpp_amd/csrc
IFS=';' read -ra VS <<< "${VARIANTS:-base=;nofeed=-DVISP_BLOCK16_DBG=1;nofrag=-DVISP_BLOCK16_DBG=8;nogelu=-DVISP_BLOCK16_DBG=32;nobarrier=-DVISP_BLOCK16_DBG=64}"
for v in "${VS[@]}"; do
  name=${v%%=*}; flags=${v#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c kernels_block16.hip -o build/kernels_block16.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libvisioncpp_v_$name.so $(make -s print-obj) -Wl,--no-undefined
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 -c kernels_block16.hip -o build/kernels_block16.o
cd ../..
for rnd in ${ROUNDS:-1 2}; do
for v in "${VS[@]}"; do
  name=${v%%=*}
  if [ -n "$STAMPS" ]; then echo "== $name"; VISP_LIBRARY=vision.cpp_amd/lib/libvisioncpp_v_$name.so python tools/bench_block.py --stamps16 2>&1 | grep -E "out-proj|MLP|QKV|lifetime|clock"; else
  echo "== $name: $(VISP_LIBRARY=vision.cpp_amd/lib/libvisioncpp_v_$name.so python tools/bench_block.py --only16 2>&1 | grep block16 | sed 's/.*qkv=1://' | tr '\n' '|')"; fi
done
done
rm -f vision.cpp_amd/lib/libvisioncpp_v_*.so
