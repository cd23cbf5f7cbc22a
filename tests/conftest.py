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


@pytest.fixture(autouse=True)
def _release_device_objects():
    """After every test: collect garbage and drain the GPU, so that captured hipGraphs (each holds internal streams of its own) and their
    memory pools are destroyed where the test ends, not whenever Python's cycle collector gets to them -- with dozens of dead graphs still
    alive in one process a LATER graph replay died inside hip::Graph::UpdateStreams (ROCm 7.2)."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:
        pass

