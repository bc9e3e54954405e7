#!/usr/bin/env python
"""Copy the summaries of a tools/run_round.sh pass from gpurun_out/ (scratch) into profiles/ (tracked).
usage: python tools/publish_profiles.py <tag> [round=r5]"""
import json, os, shutil, sys
tag = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else 'r5'
RN = 'Round %s' % rnd.lstrip('r')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
line = json.load(open(os.path.join(go, tag + '_bench.json')))
json.dump(line, open(os.path.join(pr, rnd + '_bench_default.json'), 'w'), indent=1)
side = os.path.join(go, tag + '_bench_extra.json')
if os.path.exists(side):
    shutil.copy(side, os.path.join(pr, rnd + '_bench_extra.json'))
names = {'fp32w': ('fp32w', 'fp32 everywhere, Winograd F(2x2,3x3) form of the 3x3 stride-1 convs (the headline arithmetic)', '--precision fp32w', line),
         'fp32': ('fp32', 'direct-form exact fp32 everywhere (`direct_fp32` of the bench line)', '--precision fp32', line.get('direct_fp32') or {}),
         'auto': ('auto', "the product's default arithmetic (auto = the per-layer table the engine calibrated on its own generator; `product` of the bench line; steps only: the calibration's launches are not counted)", '--precision auto', line.get('product') or {})}
for mode, (out, what, flag, rec) in names.items():
    for kind, title in (('step', 'per-kernel time of the training step'), ('phases', 'phases of one single-stream training step')):
        src = os.path.join(go, '%s_%s_%s%s.md' % (tag, kind, mode, '_kernel_stats' if kind == 'step' else ''))
        if not os.path.exists(src):
            continue
        head = ("# " + RN + ": %s, %s\n\nStyleGAN2-256 K=128 N=32 B=32, ResNet-18 R, 1x MI355X.  Command (tools/run_round.sh):\n"
                "`rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_%s_%s -o bench -- python bench.py --steps 5 --warmup 2 "
                "--no-cpu-baseline --no-extra --no-product-run --single-stream %s`\nsummarised by tools/%s.  12 steps in the trace "
                "(2 warm-up + 5 timed + 2 with per-launch HIP events + 3 host-enqueue timing steps); `--single-stream` so that kernel times add up to "
                "the step (the multi-stream step of the same build, `profiles/%s_bench_default.json`: %s img/s, %s ms/step).\n\n" % (
                    title, what, tag, mode, flag, 'prof_summary.py' if kind == 'step' else 'phase_breakdown.py', rnd, rec.get('value'), rec.get('ms_per_step')))
        open(os.path.join(pr, '%s_%s_%s%s.md' % (rnd, kind, out, '_kernel_stats' if kind == 'step' else '')), 'w').write(head + open(src).read())
pmc = os.path.join(go, tag + '_conv_pmc.json')
if os.path.exists(pmc):
    shutil.copy(pmc, os.path.join(pr, rnd + '_conv_pmc.json'))
    table = open(os.path.join(go, tag + '_conv_pmc_table.md')).read()
    open(os.path.join(pr, rnd + '_conv_pmc.md'), 'w').write(
        "# " + RN + ": PMC counters of the dominant conv kernels (rocprofv3 --pmc, five separate passes, no tracing besides --kernel-trace)\n\n"
        "Commands (tools/run_round.sh): one `rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python tools/pmc_r3.py` per counter group\n"
        "(`SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE` | "
        "`SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA` | `FETCH_SIZE` | "
        "`WRITE_SIZE` | `TCC_HIT_sum TCC_MISS_sum`), summarised by `tools/pmc_r3.py --summarise`.  Every shape is launched twice; the second launch is "
        "read.  B = 32 (B = 8 for the StyleGAN2-1024 shapes).  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM section), WRITE_SIZE is in KB; "
        "algorithmic bytes = input tensor + weights + output tensor, each once.  MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs).\n"
        "Rows 1-3: the Winograd fp32 kernel; 4-7: the direct exact-fp32 template; 8-9: the fp16 patch kernel fed with fp32 activations; 10-11: the fused "
        "up-sampling kernel (fp16 x2); 12: the patch kernel fed with a producer-written fp16 plane (XF16 form); 13-14: the few-channel halo kernel; "
        "15: the fused up-sampling kernel in split-bf16 (round 5).\n\n" + table)
he = os.path.join(go, tag + '_host_enqueue_8.json')
if os.path.exists(he) and os.path.getsize(he) > 10:
    shutil.copy(he, os.path.join(pr, rnd + '_host_enqueue_8_processes.json'))
for src, dst in ((tag + '_wino16_bench.txt', rnd + '_wino16_bench.txt'), (tag + '_ab_schedule.txt', rnd + '_ab_schedule.txt'),
                 (tag + '_upfused_bench.txt', rnd + '_upfused_bench.txt'), (tag + '_wgrad_bench.txt', rnd + '_wgrad_bench.txt'), (tag + '_wgrad_pmc.txt', rnd + '_wgrad_pmc.txt'),
                 (tag + '_precision_schemes.json', rnd + '_precision_schemes.json')):
    if os.path.exists(os.path.join(go, src)) and os.path.getsize(os.path.join(go, src)) > 10:
        shutil.copy(os.path.join(go, src), os.path.join(pr, dst))
r = line['roofline']
print('headline', line['value'], line['ms_per_step'], line['dtype'], '| roofline', r['kernel'], r['achieved'], r['frac'], '| bytes', len(json.dumps(line)))
for k in ('product', 'direct_fp32'):
    e = line.get(k) or {}
    print(k, e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('kernel'), (e.get('roofline') or {}).get('frac'))
print(line.get('others_images_per_sec'))
