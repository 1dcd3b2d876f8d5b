"""Short runs of the randomised differential soaks under tools/ (every one against the oracle, through the C-ABI):
the long runs are recorded in DESIGN.md section 2; here a few hundred random cases each (about half a minute in all), a seed of their own."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("tool,cases", [("soak_score", 500), ("soak_resample", 300), ("soak_cycle", 100),
                                        ("soak_cloud", 30), ("soak_motion", 600), ("soak_stats", 100),
                                        ("soak_lut", 300), ("soak_rays", 80), ("soak_recovery", 60)])
def test_randomised_soak_against_the_oracle(engine, tool, cases):
    mod = importlib.import_module(tool)
    assert mod.run(cases, seed=977, e=engine) == 0
