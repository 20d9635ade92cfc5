#!/bin/bash
# Host-side AddressSanitizer build of the library and of the plain-C host (device code is built as usual), into
# build/asan/.  Build here, then on the GPU box:
#   ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:protect_shadow_gap=0 build/asan/cabi_demo_asan capture.txt      (and build/asan/cabi_tour_asan, build/asan/cabi_batch_asan, no arguments)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rt=$(dirname "$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
mkdir -p "$root/build/asan" && cd "$root/build/asan"
for f in abi k_frame k_correct k_render k_token k_synth; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -fPIC -std=c++17 -ffp-contract=off -Xarch_host -fsanitize=address \
        -c "$root/slam-process_amd/csrc/$f.hip" -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address -shared-libasan -o libslamhip_asan.so *.o
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -shared-libasan -I"$root/include" "$root/examples/cabi_demo.c" \
    -o cabi_demo_asan -L. -lslamhip_asan -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$rt"
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -shared-libasan -I"$root/include" "$root/examples/cabi_tour.c" \
    -o cabi_tour_asan -L. -lslamhip_asan -lm -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$rt"
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -shared-libasan -D__HIP_PLATFORM_AMD__ -I"$root/include" -I/opt/rocm/include \
    "$root/examples/cabi_batch.c" -o cabi_batch_asan -L. -lslamhip_asan -L/opt/rocm/lib -lamdhip64 -lm \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$rt" -Wl,-rpath,/opt/rocm/lib
echo "built $root/build/asan/cabi_demo_asan, cabi_tour_asan and cabi_batch_asan"
