#!/usr/bin/env python
"""Copy the summaries of a tools/run_round.sh pass from gpurun_out/ (scratch) into profiles/ (tracked).
usage: python tools/publish_profiles.py <tag> [round=r3]"""
import json, os, sys
tag = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else 'r3'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
d = json.load(open(os.path.join(go, tag + '_bench.json')))
json.dump(d, open(os.path.join(pr, rnd + '_bench_default.json'), 'w'), indent=1)
names = {'fp32w': ('fp32w', 'fp32 everywhere, Winograd F(2x2,3x3) form of the 3x3 stride-1 convs (the headline arithmetic)', '--precision fp32w'),
         'fp32': ('fp32', 'direct-form exact fp32 everywhere (extra[1] of the bench line)', '--precision fp32'),
         'auto': ('mixed', "the product's default arithmetic (auto = mixed fp16 / split-bf16 per layer; extra[0] of the bench line)", '--precision auto')}
for mode, (out, what, flag) in names.items():
    src = os.path.join(go, '%s_step_%s_kernel_stats.md' % (tag, mode))
    if not os.path.exists(src):
        continue
    body = open(src).read()
    head = ("# Round 3: per-kernel time of the training step, %s\n\nStyleGAN2-256 K=128 N=32 B=32, ResNet-18 R, 1x MI355X.  Command (tools/run_round.sh):\n"
            "`rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_%s_%s -o bench -- python bench.py --steps 5 --warmup 2 "
            "--no-cpu-baseline --no-extra --no-product-run --single-stream %s`\nsummarised by tools/prof_summary.py.  12 steps in the trace "
            "(2 warm-up + 5 timed + 2 with per-launch HIP events + 3 host-enqueue timing steps); `--single-stream` so that kernel times add up to "
            "the step (the multi-stream step of the same build is `profiles/%s_bench_default.json`: %s).\n\n" % (
                what, tag, mode, flag, rnd,
                ('%.2f img/s, %.2f ms/step' % (d['value'], d['ms_per_step'])) if mode == 'fp32w' else
                ('%.2f img/s, %.2f ms/step' % (d['extra'][0 if mode == 'auto' else 1]['value'], d['extra'][0 if mode == 'auto' else 1]['ms_per_step']))))
    open(os.path.join(pr, '%s_step_%s_kernel_stats.md' % (rnd, out)), 'w').write(head + body)
pmc = os.path.join(go, tag + '_conv_pmc.json')
if os.path.exists(pmc):
    import shutil
    shutil.copy(pmc, os.path.join(pr, rnd + '_conv_pmc.json'))
    table = open(os.path.join(go, tag + '_conv_pmc_table.md')).read()
    open(os.path.join(pr, rnd + '_conv_pmc.md'), 'w').write(
        "# Round 3: PMC counters of the dominant conv kernels (rocprofv3 --pmc, five separate passes, no tracing besides --kernel-trace)\n\n"
        "Commands (tools/run_round.sh): one `rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python tools/pmc_r3.py` per counter group\n"
        "(`SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE` | "
        "`SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA` | `FETCH_SIZE` | "
        "`WRITE_SIZE` | `TCC_HIT_sum TCC_MISS_sum`), summarised by `tools/pmc_r3.py --summarise`.  Every shape is launched twice; the second launch is "
        "read.  B = 32.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM section), WRITE_SIZE is in KB; "
        "algorithmic bytes = input tensor + weights + output tensor, each once.  MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs).\n"
        "Rows 1-3: the Winograd fp32 kernel (conv_wino_f32.hip, DESIGN.md section 3.9; its weight operand is U = 16 * Cin * Cout floats); rows 4-7: the direct exact-fp32 template (conv_igemm_f32.hip, section 3.7); rows 8-9: the fp16 patch kernel; rows 10-11: the fused up-sampling kernel (fp16 x2).\n\n" + table)
r = d['roofline']
print('headline', d['value'], d['ms_per_step'], d['dtype'], '| roofline', r['kernel'], r['achieved'], r['frac'])
for i in (0, 1):
    e = d['extra'][i]; r = e['roofline']
    print('extra[%d]' % i, e['value'], e['ms_per_step'], e['precision'], '| roofline', r['kernel'], r['achieved'], r['frac'])
