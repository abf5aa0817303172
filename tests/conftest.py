import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


_sd_cache = {}


@pytest.fixture(scope="session")
def calibrated_sd():
    """model_type -> BN-calibrated seeded state_dict (built once per session, CPU fp32)."""
    import frmap_amd
    from frmap_amd import synth
    from oracle import weights

    def get(mt):
        if mt not in _sd_cache:
            model = frmap_amd.get_model(mt, 36)
            _sd_cache[mt] = weights.calibrated_state_dict(mt, synth.shapes_of(model), weights.SEEDS[mt][0])
        return _sd_cache[mt]

    return get
