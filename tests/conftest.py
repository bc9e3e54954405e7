import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __init__(self):
            self._c = {}

        def __call__(self, name):
            if name not in self._c:
                self._c[name] = dict(np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz')))
            return self._c[name]
    return G()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # step engines built with precision 'auto' / 'mixed-strict' on a StyleGAN2-256 / -1024 calibrate their per-layer table on their generator
    # (trainer.TrainStep.calibrate_strict: 2 304 / 576 latent codes by default): the suite's engines use a small sample (the calibration
    # itself is tested with the full one in tests/test_precision_schemes_gpu.py)
    from warpedganspace_amd.trainer import TrainStep
    TrainStep.calibrate_images_default = 96
    return torch.device('cuda:0')


@pytest.fixture()
def dev_flags(monkeypatch):
    """Set development switches of libwgs_hip.so (read once from the environment) for one test: dev_flags(WGS_DMA_ALWAYS='1')."""
    from warpedganspace_amd import _lib as L

    def _set(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        L.lib().wgs_dev_reload_flags()
    yield _set
    monkeypatch.undo()
    L.lib().wgs_dev_reload_flags()
