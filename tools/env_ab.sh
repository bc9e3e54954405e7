cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, env...
  name=$1; shift
  v=$(env "$@" timeout 200 python bench.py --precision auto --steps 80 --warmup 20 --no-product-run --no-extra --no-cpu-baseline --no-roofline --extra-out gpurun_out/envab_extra.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$name: $v"
}
for rep in 1 2; do
  run base A=1
  run hwq8 GPU_MAX_HW_QUEUES=8
  run hwq2 GPU_MAX_HW_QUEUES=2
  run devkernarg0 HIP_FORCE_DEV_KERNARG=0
  run devkernarg1 HIP_FORCE_DEV_KERNARG=1
  run nodirect AMD_DIRECT_DISPATCH=0
done
