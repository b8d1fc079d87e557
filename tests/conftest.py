import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, os.cpu_count() or 1))))     # (inherited by the tests' subprocesses)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the fp64 oracle is thousands of tiny per-graph tensor ops: on the GPU box's 256 logical CPUs torch's default
    # intra-op pool (one thread per core) spends its time in fork/join -- 8 threads are an order of magnitude faster
    try:
        import torch
        torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    except Exception:                                         # noqa: BLE001
        pass


EMU = os.environ.get("DGCNN_EMU") == "1"      # run the GPU tests on the CPU SIMT emulation (tests/emu_util.py; small sizes only)
# GPU tests that cannot mean anything under the emulation: second streams / processes, timing, the test builds of the library
EMU_SKIP = ("two_trainers_on_two_streams", "busy_second_stream", "two_processes", "one_shot", "three_waves_stalled", "isa_audit",
            "in_a_subprocess", "two_rank")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and they were not
    explicitly selected, so a plain ``pytest tests`` works in the CPU container."""
    import torch
    if EMU:
        import emu_util
        emu_util.install_global()
        skip = pytest.mark.skip(reason="not meaningful on the CPU emulation")
        for it in items:
            if any(k in it.name for k in EMU_SKIP):
                it.add_marker(skip)
        return
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
