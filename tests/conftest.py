import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """The parity allowances this run consumed (tests/helpers.py, record_allowance) -> gpurun_out/parity_allowances.json."""
    try:
        import json
        import helpers
        if helpers.ALLOWANCES:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            name = os.environ.get("CARMA_ALLOWANCE_FILE", "parity_allowances.json")
            with open(os.path.join(out, name), "w") as f:
                json.dump(dict(exitstatus=int(exitstatus), records=helpers.ALLOWANCES), f, indent=1)
    except Exception as ex:                                   # bookkeeping must never fail a run
        print("parity allowance table not written: %r" % (ex,))
