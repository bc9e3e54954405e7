"""CPU: the static tables behind bench.py's JSON line stay consistent with the library's precision codes and with SURVEY.md 8(d)'s
algorithmic FLOP counts (no GPU work: bench.py itself refuses to run without an MI355X)."""
import importlib.util
import os

import pytest

from warpedganspace_amd import conv as C
from warpedganspace_amd import reconstructor as RR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('wgs_bench', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_dtype_text_covers_every_concrete_mode(bench):
    for name, code in C.PRECISION_NAMES.items():
        if code >= 0:
            assert name in bench.DTYPE_TEXT and bench.DTYPE_TEXT[name]
            assert name in bench.DTYPE and bench.DTYPE[name]
    assert bench.DTYPE['fp32'] == 'fp32'                      # the headline's dtype field: the reference's arithmetic, verbatim
    for code in (None, 0, 1, 2, 3, 4):
        assert tuple(RR.r_arith('auto', code)) in bench.R_TEXT


def test_flop_table_matches_the_survey(bench):
    # cfg3 285.8 GFLOP/img = 2 x 90.24 (two forwards) + 90.24 (dgrad) + 5.05 + 10.09 (R fwd + bwd)
    assert abs(bench.GFLOP_PER_IMG['stylegan2-256'] - (3 * 90.24 + 5.05 + 10.09)) < 0.2
    assert bench.GFLOP_PER_IMG['stylegan2-1024'] == 687.8 and bench.GFLOP_PER_IMG['proggan-256'] == 184.3


def test_extra_runs_are_well_formed(bench):
    names = set()
    keys = set()
    for name, gan, size, K, N, B, prec, r_prec, w_space, steps, gkey, short in bench.EXTRA:
        assert name not in names and short not in keys and len(short) <= 24
        names.add(name); keys.add(short)
        assert gan in ('stylegan2', 'proggan', 'biggan') and size in (128, 256, 1024) and K > 0 and N > 0 and B > 0 and steps >= 3
        assert prec in C.PRECISION_NAMES and r_prec in ('auto', 'fp32', 'bf16x3')
        assert gkey is None or gkey in bench.GFLOP_PER_IMG
    assert sum('[R fp32]' in n for n in names) == 1


def test_roofline_groups_by_kernel_symbol(bench):
    """The dominant kernel is chosen by SYMBOL (summed over its launch shapes), and its rate is summed FLOPs / summed time."""
    recs = [('conv f16x2 256->128 @128x128 up-conv + blur fused B32', 'upconv_blur_kernel<3, 16>', 618.5e9, 2.2, 2.0),
            ('conv f16x2 512->256 @64x64 up-conv + blur fused B32', 'upconv_blur_kernel<3, 16>', 618.5e9, 2.0, 2.0),
            ('conv f16 128->128 @256x256 9 taps B32', 'igemm_patch_kernel<1, 128, 128, 2, 2, 1, 0>', 1855.4e9, 2.7, 3.0),
            ('wgrad fp32 64->64 @64x64 9 taps B32', 'igemm_wgrad_kernel<64, 64, 2, 2, false>', 19.3e9, 0.3, 4.0)]
    r = bench.roofline_of(recs, 1000.0, 285.8, pmc_files=[])
    assert r['kernel'] == 'upconv_blur_kernel<3, 16>'                       # 4.2 ms summed beats the heaviest single shape (2.7 ms)
    assert abs(r['achieved'] - 2 * 618.5 / 4.2) < 0.5 and r['peak'] == bench.F16_MFMA_PEAK_TF
    assert abs(r['frac'] - r['achieved'] / 2500.0) < 1e-3
    assert abs(r['mfma_pipe_frac'] - 2 * r['frac']) < 1e-3                  # f16x2: two MFMAs per product occupy the pipe, one is credited
    assert r['traffic'] is None and r['traffic_algorithmic'] is None
    assert [s['symbol'] for s in r['by_symbol']][:2] == ['upconv_blur_kernel<3, 16>', 'igemm_patch_kernel<1, 128, 128, 2, 2, 1, 0>']
    fp = bench.roofline_of([('conv fp32 128->128 @256x256 9 taps B32', 'igemm_nt_kernel<128, 128, 32, 2, 2, true>', 1855.4e9, 17.0, 3.0)], 340.0, 285.8, pmc_files=[])
    assert fp['peak'] == bench.FP32_MFMA_PEAK_TF and abs(fp['achieved'] - 1855.4 / 17.0) < 0.1 and fp['achieved'] == fp['direct_equiv_TFLOPs']


def test_no_roofline_fraction_exceeds_one(bench):
    """a Winograd launch is rated on the MFMA FLOPs it EXECUTES (16/36 of the direct form's); the direct-equivalent rate has
    its own key.  With r3's measured figures (15.59 ms for 4 097.4 GFLOP of direct-form work) frac is 0.74, not 1.67."""
    w = bench.roofline_of([('conv fp32w 512->512 @64x64 9 taps B32', 'wino_f32_kernel<1, 4, true, 8>', 4097.4e9, 15.59, 10.0),
                           ('conv fp32 512->256 @64x64 up-conv x4 phases B32', 'igemm_nt16_kernel<4, 256, 256, 2, 4, 2, false>', 631.4e9, 4.6, 2.0)],
                          566.0, 285.8, pmc_files=[])
    assert w['kernel'].startswith('wino_f32_kernel') and w['peak'] == bench.FP32_MFMA_PEAK_TF
    assert abs(w['direct_equiv_TFLOPs'] - 262.8) < 0.5 and abs(w['achieved'] - 262.8 * 16 / 36) < 0.3 and abs(w['frac'] - 0.743) < 0.005
    assert w['frac'] <= 1.0 and w['step_frac'] <= 1.0 and w['step_direct_equiv_TFLOPs'] > w['step_achieved_TFLOPs']
    assert all(r['frac'] <= 1.0 for r in w['by_symbol'])
    assert 'cudnn' not in (bench.__doc__ + ' '.join(bench.DTYPE_TEXT.values())).lower()
    assert bench.DTYPE['fp32w'] == 'fp32'


def test_traffic_is_the_launch_weighted_mean_over_the_symbols_shapes(bench, tmp_path):
    import json
    sym = 'wino_f32_kernel<1, 4, true, 8>'
    rows = [dict(symbol=sym, shape='conv fp32w 512->512 @64x64 9 taps B32', hbm_bytes_per_launch=2.727e9, algorithmic_bytes_per_launch=0.554e9),
            dict(symbol=sym, shape='conv fp32w 128->128 @256x256 9 taps B32', hbm_bytes_per_launch=2.633e9, algorithmic_bytes_per_launch=2.149e9),
            dict(symbol='other', shape='x', hbm_bytes_per_launch=9e9, algorithmic_bytes_per_launch=9e9)]
    f = tmp_path / 'r9_conv_pmc.json'
    f.write_text(json.dumps({'kernels': rows}))
    recs = [('conv fp32w 512->512 @64x64 9 taps B32', sym, 1237.0e9, 4.4, 1.0), ('conv fp32w 128->128 @256x256 9 taps B32', sym, 1855.4e9, 7.2, 3.0),
            ('conv fp32w 256->256 @128x128 9 taps B32', sym, 1855.4e9, 6.2, 2.0)]       # the third shape has no PMC row: left out of the mean
    r = bench.roofline_of(recs, 566.0, 285.8, pmc_files=[str(f)])
    assert abs(r['traffic'] - (1 * 2.727 + 3 * 2.633) / 4) < 2e-3 and abs(r['traffic_algorithmic'] - (1 * 0.554 + 3 * 2.149) / 4) < 2e-3
    assert '2 of the symbol\'s 3 shapes' in r['traffic_note']


def _canned_full(bench, world=1):
    """A full record shaped like a real run with every section populated and long strings everywhere."""
    import types
    shapes = [{"shape": 'conv fp32w %d->%d @%dx%d 9 taps B32' % (c, c, h, h), "gflop_per_step": 1234.5, "ms_per_step": 4.321, "TFLOP/s": 123.4, "launches": 3.0}
              for c, h in ((512, 64), (256, 128), (128, 256), (512, 32), (512, 16))]
    roof = {"bound": "mfma", "kernel": "wino_f32_kernel<1, 4, true, 8>", "achieved": 116.8, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.7425,
            "direct_equiv_TFLOPs": 262.8, "traffic": 2.52, "traffic_algorithmic": 1.26, "traffic_note": "x" * 300, "mfma_pipe_frac": 0.7425,
            "kernel_launches_per_step": 10.0, "kernel_ms_per_step": 15.59, "kernel_avg_launch_ms": 1.559, "kernel_gflop_per_step": 4097.4,
            "kernel_shapes": shapes, "all_conv_launches": {"TFLOP/s": 120.0, "direct_equiv_TFLOPs": 180.0, "ms_per_step": 50.0, "gflop_per_step": 9000.0},
            "by_symbol": [{"symbol": "igemm_nt16_kernel<4, 256, 256, 2, 4, 2, false>" + str(i), "ms_per_step": 1.0, "frac": 0.5} for i in range(10)],
            "method": "y" * 400, "step_achieved_TFLOPs": 116.0, "step_frac": 0.74, "step_direct_equiv_TFLOPs": 163.0}
    comm = {"backend": "RCCL (torch.distributed 'nccl')", "world_size_observed": world, "allreduce_bytes_per_step": 63800000, "collectives_per_step": 2,
            "exposed_wait_ms_per_step": 0.123, "per_gpu_images_per_sec": 71.59, "n1_same_job": {"value": 590.12, "ms_per_step": 54.23, "steps": 30},
            "note": "z" * 300} if world > 1 else None
    host = {"library_launches_per_step": 426.0, "host_enqueue_ms_per_step": 6.6, "host_enqueue_ms_per_step_mean": 7.0, "note": "n" * 200}
    same = [dict({"config": "headline workload in the product's default arithmetic (--precision auto), same steps / warmup"}, precision='mixed-strict', value=1145.0,
                 ms_per_step=27.9, dtype=bench.DTYPE['mixed-strict'], dtype_detail='d' * 500, roofline=dict(roof, kernel='igemm_patch_kernel<0, 256, 256, 2, 4, 1, 0>'),
                 host=host, comm=comm, last_stats={}, n_gpus=world, steps=100,
                 strict_calibration={"table": "128:w,3;256:w,3", "tried": [("default 128:2,3;256:2,3", 1.1e-3, 256), ("128:w,3;256:w,3", 8.1e-4, 2304)], "images": 2304,
                                     "margin": 0.95, "gate": 1e-3, "fp16_layers": 2},
                 precision_check={"batch_max": 5.9e-4, "image_median": 4.1e-4, "image_p99": 8.4e-4, "image_max": 9.06e-4, "over_gate_frac": 0.0, "gate": 1e-3, "n": 2304}),
            dict({"config": "headline workload in direct-form exact fp32 (--precision fp32: no Winograd), same steps / warmup"}, precision='fp32', value=410.0,
                 ms_per_step=78.0, dtype='fp32', dtype_detail='d' * 500, roofline=roof, host=host, last_stats={}, n_gpus=world, steps=100)]
    others = [dict({"config": e[0], "key": e[-1]}, precision=e[6], value=123.45, ms_per_step=1.0, r_arith=[1, 1, 1]) for e in bench.EXTRA]
    for o in others:
        if o["key"] == "cfg3_mixed_uncalibrated":
            o["precision_check"] = {"batch_max": 8.5e-4, "image_median": 4.5e-4, "image_p99": 1.0e-3, "image_max": 1.3e-3, "over_gate_frac": 0.012, "gate": 1e-3, "n": 2304}
    assert bench.EXTRA[5][-1] == 'cfg3_auto_Rfp32'
    others[5] = {"config": bench.EXTRA[4][0], "key": bench.EXTRA[4][-1], "precision": bench.EXTRA[4][6], "error": "RuntimeError('" + "e" * 300 + "')"}
    args = types.SimpleNamespace(size=256, gan='stylegan2', K=128, N=32, batch=32, w_space=False, steps=100, warmup=20, precision='fp32w')
    head = {"value": 572.7, "ms_per_step": 55.9, "dtype": "fp32", "dtype_detail": bench.DTYPE_TEXT['fp32w'], "precision": "fp32w", "r_arith": [5, 5, 0],
            "roofline": roof, "host": host}
    cpu = {"value": 0.393, "unit": "images/sec", "cores": 128, "kind": "port", "sample": "s" * 600}
    hbm = {k: {"bytes": 1, "us": 1.0, "GB/s": 1.0, "note": "h" * 80} for k in ('rbf_fwd', 'rbf_bwd', 'adam', 'blur', 'torgb', 'bn')}
    return bench.full_record(args, world, head, {"accuracy": 0.1, "classification_loss": 4.8, "regression_loss": 0.3, "total_loss": 4.9}, comm, hbm,
                             same + (others if world == 1 else []), cpu if world == 1 else None, 'stylegan2-256')


@pytest.mark.parametrize('world', [1, 8])
def test_final_line_is_under_4kb_and_round_trips(bench, world):
    """the driver keeps ~8 KB of stdout and r3's 21 KB line left BENCH_r03.parsed = null.  The last stdout line is built by
    final_line() alone and stays < 4 KB with every section populated; everything else lives in the side file."""
    import json
    full = _canned_full(bench, world)
    assert len(json.dumps(full)) > 8000                         # the complete record would not have survived
    line = bench.final_line(full, 'gpurun_out/bench_extra.json')
    assert len(line) < 4096 and '\n' not in line
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline', 'comm'):
        assert k in d
    assert d['value'] == 572.7 and d['n_gpus'] == world and d['config']['workload'].startswith('StyleGAN2-FFHQ-256') and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['kernel'] and r['frac'] == round(r['achieved'] / r['peak'], 4) <= 1.0 and r['unit'] == 'TFLOP/s'
    assert r['traffic'] == 2.52 and r['traffic_algorithmic'] == 1.26 and 'kernel_shapes' not in r and 'by_symbol' not in r
    assert d['product']['value'] == 1145.0 and d['direct_fp32']['value'] == 410.0 and d['extra_file'] == 'gpurun_out/bench_extra.json'
    if world == 1:
        c = d['cpu_baseline']
        assert c['value'] == 0.393 and c['cores'] == 128 and c['kind'] == 'port' and len(c['sample']) <= 300
        assert len(d['others_images_per_sec']) == len(bench.EXTRA) - 1 and 'error' in d['others_images_per_sec'].values()
        assert d['comm'] is None
        # the measured image error of the timed arithmetic — the table the engine calibrated on its own generator —
        # and the un-calibrated default table's rate + error beside it
        pc = d['product']['precision_check']
        assert pc['n'] == 2304 and pc['image_max'] == 9.06e-4 and pc['over_gate_frac'] == 0.0 and pc['batch_max'] == 5.9e-4 and pc['image_p99'] == 8.4e-4
        assert d['product']['precision'] == 'mixed-strict' and d['product']['table'] == '128:w,3;256:w,3' and d['product']['fp16_layers'] == 2
        un = d['product']['uncalibrated']
        assert un['precision'] == 'mixed' and un['value'] == 123.45 and un['precision_check']['over_gate_frac'] == 0.012
    else:
        assert d['comm']['world_size_observed'] == 8 and d['comm']['exposed_wait_ms_per_step'] == 0.123 and d['product']['exposed_wait_ms_per_step'] == 0.123
        # VERDICT r4 #6c: the N = 1 rate of the same job beside the per-GPU rate and the exposed wait
        assert d['comm']['per_gpu_images_per_sec'] == 71.59 and d['comm']['n1_same_job']['value'] == 590.12 and d['product']['n1_same_job']['steps'] == 30


def test_no_gpu_means_a_loud_failure(bench, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    monkeypatch.setattr('sys.argv', ['bench.py', '--steps', '1', '--warmup', '0'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'MI355X' in str(e.value)
