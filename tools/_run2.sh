timeout 300 python -m pytest tests -m gpu -q -x -k "bf16 or conv" 2>&1 | tail -3
timeout 100 python tools/bench_small.py
WGS_LIB=$PWD/tools/_bin/libwgs_abl13.so timeout 100 python tools/bench_small.py
