#!/bin/bash
# Where does the time of the pointwise kernel go?  Device time of conv_micro shapes with parts of the kernel compiled out
# (libmrfp_hip_pwdN.so: -DMRFP_PW_DBG=N, built by mrfp_amd.build.build_variant).  Results are garbage, timing only.
cd "$(dirname "$0")/../.."
L=$PWD/mrfp_amd/csrc
for shape in "$@"; do
  bash tools/prof_micro.sh $shape "X=0" "MRFP_HIP_LIB=$L/libmrfp_hip_pwd1.so" "MRFP_HIP_LIB=$L/libmrfp_hip_pwd2.so" "MRFP_HIP_LIB=$L/libmrfp_hip_pwd4.so" "MRFP_HIP_LIB=$L/libmrfp_hip_pwd6.so" "MRFP_HIP_LIB=$L/libmrfp_hip_pwd7.so" "MRFP_DEBUG_DROP=5"
done
