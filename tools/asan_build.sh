#!/bin/bash
# Host-side AddressSanitizer build of libgsx + the facade driver (device code is not instrumented: GPU ASan is not
# available on this pool).  Builds into tools/asan/ (git-ignored); run on the GPU box as
#   cd tools/asan && ASAN_OPTIONS=verify_asan_link_order=0:protect_shadow_gap=0 ./facade_test scene.splat out 224 160
# Round 1 result: no finding in libgsx / the facade; the only reports are leaks inside libhsa-runtime64.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/asan
CL=/opt/rocm/lib/llvm/bin/clang++
cd gaussian-splatting_amd/csrc
for f in gsx_api gsx_sort gsx_lod gsx_raster; do
  extra=""; [ $f = gsx_sort -o $f = gsx_lod ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer $extra -c -o ../../tools/asan/$f.o $f.hip
done
$CL -std=c++17 -O1 -g -ffp-contract=off -fPIC -fsanitize=address -fno-omit-frame-pointer -c -o ../../tools/asan/gsx_host.o gsx_host.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -shared-libsan -o ../../tools/asan/libgsx.so ../../tools/asan/gsx_*.o
cd ../../tools/asan
$CL -std=c++17 -O1 -g -ffp-contract=off -fsanitize=address -shared-libsan -fno-omit-frame-pointer -o facade_test ../../tests/cpp/facade_test.cpp \
    -L. -lgsx -Wl,-rpath,'$ORIGIN' -Wl,-rpath,$(dirname $($CL -print-file-name=libclang_rt.asan-x86_64.so))
