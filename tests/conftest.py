import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    # tests/test_switches_gpu.py re-runs the bf16-vs-f32 step comparison under a retained switch and states ITS loss gate here
    parser.addoption('--stj-loss-gate', action='store', default=None, help='relative loss gate of test_bench_step_bf16_vs_f32_mode_cfg256_b8')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def lib_built():
    """Make sure the in-tree HIP library exists (cross-compiles without a GPU)."""
    from strajnet_amd import build
    return build.build(verbose=False)
