timeout 900 python -m pytest tests -m gpu -q -x -k "conv or recon or step" 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2
