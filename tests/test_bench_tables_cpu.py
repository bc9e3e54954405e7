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
    for name, gan, size, K, N, B, prec, r_prec, w_space, steps, gkey in bench.EXTRA:
        assert name not in names
        names.add(name)
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
    r = bench.roofline_of(recs, 1000.0, 285.8)
    assert r['kernel'] == 'upconv_blur_kernel<3, 16>'                       # 4.2 ms summed beats the heaviest single shape (2.7 ms)
    assert abs(r['achieved'] - 2 * 618.5 / 4.2) < 0.5 and r['peak'] == bench.F16_MFMA_PEAK_TF
    assert abs(r['frac'] - r['achieved'] / 2500.0) < 1e-3
    assert abs(r['executed_mfma_frac'] - 2 * r['frac']) < 1e-3             # f16x2: two MFMAs per product
    assert [s['symbol'] for s in r['by_symbol']][:2] == ['upconv_blur_kernel<3, 16>', 'igemm_patch_kernel<1, 128, 128, 2, 2, 1, 0>']
    fp = bench.roofline_of([('conv fp32 128->128 @256x256 9 taps B32', 'igemm_nt_kernel<128, 128, 32, 2, 2, true>', 1855.4e9, 17.0, 3.0)], 340.0, 285.8)
    assert fp['peak'] == bench.FP32_MFMA_PEAK_TF and abs(fp['achieved'] - 1855.4 / 17.0) < 0.1
    # Winograd launches: rated on the direct form's multiplies against the fp32 MFMA peak (frac may pass 1), the executed share beside it
    w = bench.roofline_of([('conv fp32w 512->512 @64x64 9 taps B32', 'wino_f32_kernel<1, 4, true>', 1237.0e9, 4.4, 2.0),
                           ('conv fp32 512->256 @64x64 up-conv x4 phases B32', 'igemm_nt16_kernel<4, 256, 256, 2, 4, 2, false>', 631.4e9, 4.6, 2.0)], 560.0, 285.8)
    assert w['kernel'] == 'igemm_nt16_kernel<4, 256, 256, 2, 4, 2, false>' and w['peak'] == bench.FP32_MFMA_PEAK_TF and 'frac_note' not in w
    wk = [r for r in w['by_symbol'] if r['symbol'].startswith('wino')][0]
    assert wk['peak'] == bench.FP32_MFMA_PEAK_TF and wk['frac'] > 1.0
    w2 = bench.roofline_of([('conv fp32w 512->512 @64x64 9 taps B32', 'wino_f32_kernel<1, 4, true>', 1237.0e9, 4.4, 2.0)], 560.0, 285.8)
    assert abs(w2['executed_mfma_frac'] - w2['frac'] * 16 / 36) < 1e-3 and 'frac_note' in w2
    assert bench.DTYPE['fp32w'] == 'fp32'


def test_no_gpu_means_a_loud_failure(bench, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    monkeypatch.setattr('sys.argv', ['bench.py', '--steps', '1', '--warmup', '0'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'MI355X' in str(e.value)
