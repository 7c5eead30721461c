#!/bin/bash
# tools/build_variant.sh NAME -DMACRO=VALUE...  ->  mrfp_amd/csrc/libmrfp_hip_NAME.so with conv.hip compiled under the
# given macros (the other objects are reused).  Run with MRFP_HIP_LIB=mrfp_amd/csrc/libmrfp_hip_NAME.so for A/B.
set -e
cd "$(dirname "$0")/../mrfp_amd/csrc"
name=$1; shift
python -m mrfp_amd.build >/dev/null 2>&1 || (cd ../.. && python -m mrfp_amd.build >/dev/null)
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off "$@" -c conv.hip -o conv_$name.o
objs=$(ls *.o | grep -v '^conv' | tr '\n' ' ')
hipcc --offload-arch=gfx950 -shared -fPIC -o libmrfp_hip_$name.so conv_$name.o $objs
echo "$(pwd)/libmrfp_hip_$name.so"
