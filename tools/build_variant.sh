#!/bin/bash
# tools/build_variant.sh NAME FILE -DMACRO=VALUE...  ->  mrfp_amd/csrc/libmrfp_hip_NAME.so with csrc/FILE.hip compiled
# under the given macros (the other objects are reused).  Run with MRFP_HIP_LIB=mrfp_amd/csrc/libmrfp_hip_NAME.so for
# A/B runs of build-time variants on one GPU box (boxes differ by several per cent).
#   tools/build_variant.sh m32 conv -DMRFP_M16=0        32x32x16 MFMA in the forward / dgrad kernel
#   tools/build_variant.sh nt affine -DMRFP_NT=1        non-temporal last reads in the normalisation apply kernels
set -e
cd "$(dirname "$0")/.."
python -m mrfp_amd.build >/dev/null
cd mrfp_amd/csrc
name=$1; file=$2; shift; shift
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off "$@" -c $file.hip -o ${file}_$name.variant.o
objs=$(ls *.o | grep -v "variant.o" | grep -v "^$file.o" | tr '\n' ' ')
hipcc --offload-arch=gfx950 -shared -fPIC -o libmrfp_hip_$name.so ${file}_$name.variant.o $objs
echo "$(pwd)/libmrfp_hip_$name.so"
