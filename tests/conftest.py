import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    import pyoracle
    pyoracle.build()
    return pyoracle.lib()


def pytest_sessionfinish(session, exitstatus):
    """The random sweeps' bookkeeping (tests/util.py::sweep_record) as a JSON summary: how many draws needed a bar wider than
    the flat one and why, the worst error ratios -- the evidence behind "N draws green" (copied into profiles/ per round)."""
    try:
        import json
        import util
        if not util.SWEEP_LOG:
            return
        out = os.environ.get("SYLDET_SWEEP_SUMMARY", os.path.join(ROOT, "gpurun_out", "sweep_summary.json"))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump({"exitstatus": int(exitstatus), "draws_env": os.environ.get("SYLDET_FUZZ_DRAWS", "default"),
                   "summary": util.sweep_summary()}, open(out, "w"), indent=1)
    except Exception as e:                                        # bookkeeping must never fail a run
        print("sweep summary not written:", e)
