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
    assert set(bench.R_TEXT) == {0, 1}
    for code in (None, 0, 1, 2, 3, 4):
        assert RR.forward_precision(code) in bench.R_TEXT


def test_flop_table_matches_the_survey(bench):
    # cfg3 285.8 GFLOP/img = 2 x 90.24 (two forwards) + 90.24 (dgrad) + 5.05 + 10.09 (R fwd + bwd)
    assert abs(bench.GFLOP_PER_IMG['stylegan2-256'] - (3 * 90.24 + 5.05 + 10.09)) < 0.2
    assert bench.GFLOP_PER_IMG['stylegan2-1024'] == 687.8 and bench.GFLOP_PER_IMG['proggan-256'] == 184.3


def test_extra_runs_are_well_formed(bench):
    names = set()
    for name, gan, size, K, N, B, prec, w_space, steps, gkey in bench.EXTRA:
        assert name not in names
        names.add(name)
        assert gan in ('stylegan2', 'proggan', 'biggan') and size in (128, 256, 1024) and K > 0 and N > 0 and B > 0 and steps >= 3
        assert prec is None or prec in C.PRECISION_NAMES
        assert gkey is None or gkey in bench.GFLOP_PER_IMG
    assert sum('[R fp32]' in n for n in names) == 1


def test_no_gpu_means_a_loud_failure(bench, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    monkeypatch.setattr('sys.argv', ['bench.py', '--steps', '1', '--warmup', '0'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'MI355X' in str(e.value)
