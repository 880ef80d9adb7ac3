#!/bin/bash
# The host side of the library (GGUF reader, image code, graph lowering / planner, model loaders, C ABI) under AddressSanitizer + UBSan on the CPU
# (GPU sanitizers are not available on this pool): the host .cpp files are rebuilt with -fsanitize=address,undefined, linked with the already built
# kernel objects into /tmp/asan/libvisioncpp_asan.so, and the CPU suite runs against that library (VISP_LIBRARY). Reports go to /tmp/asan/log*.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/vision.cpp_amd/csrc
make -s -C "$src"
mkdir -p /tmp/asan && rm -f /tmp/asan/*.o /tmp/asan/log* /tmp/asan/ublog*
host="gguf image image_resize depthany esrgan tinyvit swin birefnet graph c_api"
for f in $host; do g++ -O1 -g -fPIC -std=c++17 -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer -c "$src/$f.cpp" -o /tmp/asan/$f.o; done
objs=$(make -s -C "$src" print-obj | tr ' ' '\n' | grep -v -E "build/($(echo $host | tr ' ' '|'))\.o" | sed "s|^|$src/|" | tr '\n' ' ')
g++ -shared -fPIC -o /tmp/asan/libvisioncpp_asan.so $objs /tmp/asan/*.o -fsanitize=address,undefined -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=/tmp/asan/log UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/asan/ublog
cd "$root"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" VISP_LIBRARY=/tmp/asan/libvisioncpp_asan.so python -m pytest tests -q -m "not gpu" "$@"
if ls /tmp/asan/log* /tmp/asan/ublog* >/dev/null 2>&1; then echo "SANITIZER REPORTS:"; cat /tmp/asan/log* /tmp/asan/ublog* | head -60; exit 1; fi
echo "no sanitizer report"
