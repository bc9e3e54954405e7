cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc5
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc5/fetch -o p -- python tools/pmc_conv.py > gpurun_out/pmc5/fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc5/write -o p -- python tools/pmc_conv.py > gpurun_out/pmc5/write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc5/hit -o p -- python tools/pmc_conv.py > gpurun_out/pmc5/hit.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc5/sq -o p -- python tools/pmc_conv.py > gpurun_out/pmc5/sq.log 2>&1
true
