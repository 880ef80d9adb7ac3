#!/bin/bash
# Builds libvisioncpp.so of another commit into gpurun_out/variants/<name>.so for same-box A/B runs:
#   tools/build_variant.sh <commit> <name>   then   VISP_LIBRARY=gpurun_out/variants/<name>.so python tools/bench_esrgan.py
set -e
commit=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$commit" vision.cpp_amd/csrc include | tar -x -C "$tmp"
make -s -C "$tmp/vision.cpp_amd/csrc" -j4 >/dev/null 2>&1 || make -C "$tmp/vision.cpp_amd/csrc"
mkdir -p "$root/variants"
cp "$tmp/vision.cpp_amd/lib/libvisioncpp.so" "$root/variants/$name.so"
rm -rf "$tmp"
echo "built variants/$name.so from $commit"
