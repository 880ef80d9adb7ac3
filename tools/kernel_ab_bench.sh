#!/bin/bash
# end-to-end A/B on ONE box of two sources of one kernel file (OBJ=kernels_attn ...): $1, $2 = source files in csrc/; bench.py step time with each
set -e
OBJ=${OBJ:-kernels_block16}
cd vision.cpp_amd/csrc
i=0
for src in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 -x hip -c "$src" -o build/$OBJ.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libvisioncpp_ab_$i.so $(make -s print-obj) -Wl,--no-undefined
  i=$((i+1))
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form=1 -c $OBJ.hip -o build/$OBJ.o
cd ../..
for rnd in 1 2; do
  for j in $(seq 0 $((i-1))); do
    echo "== variant $j: $(VISP_LIBRARY=vision.cpp_amd/lib/libvisioncpp_ab_$j.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --min-seconds 0 --no-pipeline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
  done
done
rm -f vision.cpp_amd/lib/libvisioncpp_ab_*.so
