cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
timeout 60 rocprofv3 -L > gpurun_out/pmc2/counters.txt 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc2/a -o p -- python tools/pmc_conv.py > gpurun_out/pmc2/a.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc2/b -o p -- python tools/pmc_conv.py > gpurun_out/pmc2/b.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc2/c -o p -- python tools/pmc_conv.py > gpurun_out/pmc2/c.log 2>&1
ls gpurun_out/pmc2/*
