cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-l3_3x3}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_a_$S -o p -- python3 $R/tools/conv_micro.py $S 5 fwd > $R/gpurun_out/pmc_a_$S.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_b_$S -o p -- python3 $R/tools/conv_micro.py $S 5 fwd > $R/gpurun_out/pmc_b_$S.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_c_$S -o p -- python3 $R/tools/conv_micro.py $S 5 fwd > $R/gpurun_out/pmc_c_$S.log 2>&1 || exit 1
