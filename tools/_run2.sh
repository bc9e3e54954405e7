timeout 100 python tools/bench_abl.py
for n in 11; do WGS_LIB=$PWD/tools/_bin/libwgs_abl$n.so timeout 100 python tools/bench_abl.py 2>&1 | grep TF; done
timeout 300 python -m pytest tests -m gpu -q -x -k "bf16 or conv" 2>&1 | tail -2
