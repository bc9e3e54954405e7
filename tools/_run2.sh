timeout 600 python -m pytest tests -m gpu -q -x -k "patch_form or lds_dma" 2>&1 | tail -2
timeout 100 python tools/bench_dma.py 2>&1 | grep TF
