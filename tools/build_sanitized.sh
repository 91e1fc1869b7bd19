#!/bin/bash
# Host-side sanitizer build (CPU box only): libatx.so and the test-only RCCL stand-in with AddressSanitizer + UBSan applied to the
# HOST compilation only (-Xarch_host: the gfx950 device code is compiled as usual — GPU ASan / xnack+ code objects are not available
# on this pool).  What it covers is everything the library does before a launch: argument validation, table builders
# (atx_vector_program), error strings, the dlopen'ed RCCL binding and the communicator bookkeeping.
#   bash tools/build_sanitized.sh            -> gpurun_out/hostsan/libatx_hostsan.so, gpurun_out/hostsan/librccl_stub_hostsan.so
# (gpurun_out/ is scratch: git-ignored and outside the snapshot gpurun sends to the GPU box, which refuses sanitizer builds)
# Run by tests/test_host_sanitizers.py, which then drives tests/c_abi/sanitize_check.c (built with the same flags) through both.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/hostsan
OBJ=$ROOT/gpurun_out/hostsan/obj
mkdir -p $OUT $OBJ
SAN="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize-recover=all -Xarch_host -fno-omit-frame-pointer"
cd $ROOT/anemoi-transform_amd/csrc
# one object per source, in parallel (the gather kernels' file alone is half of the build)
ls *.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-function $SAN -c {} -o $OBJ/{}.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libasan -o $OUT/libatx_hostsan.so $OBJ/*.o
cd $ROOT/tests/rccl_stub
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared $SAN -fsanitize=address,undefined -shared-libasan -o $OUT/librccl_stub_hostsan.so rccl_stub.cpp -lpthread
echo built $OUT/libatx_hostsan.so $OUT/librccl_stub_hostsan.so
