timeout 100 python tools/bench_abl.py
WGS_LIB=$PWD/tools/_bin/libwgs_abl5.so timeout 100 python tools/bench_abl.py 2>&1 | grep TF
export TMPDIR=/tmp
WGS_LIB=$PWD/tools/_bin/libwgs_abl5.so timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc3/hit5 -o p -- python tools/pmc_conv.py > gpurun_out/pmc3/hit5.log 2>&1
