timeout 600 python -m pytest tests -m gpu -q -x -k "conv or bf16" 2>&1 | tail -2
timeout 100 python tools/bench_phase4.py 2>&1 | grep TF
WGS_PHASE_PATCH=1 timeout 100 python tools/bench_phase4.py 2>&1 | grep TF | sed 's/^/phase-patch /'
WGS_PHASE_PATCH=1 timeout 600 python -m pytest tests -m gpu -q -x -k "conv or bf16" 2>&1 | tail -2
