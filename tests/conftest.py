import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


# Measured errors of the GPU parity tests: every comparison calls errlog(test, case, key, err, tol); the table is
# written to gpurun_out/parity_errors.json at the end of the session (scripts/parity_table.py check turns it into
# profiles/rNN_parity_errors.md and verifies it against the frozen table tests/golden/tolerances.json).
_ERRLOG = []


@pytest.fixture(scope='session')
def errlog():
    def log(test, case, key, err, tol):
        _ERRLOG.append({'test': test, 'case': case, 'key': key, 'err': float(err), 'tol': float(tol)})
        return float(err)
    return log


def pytest_sessionfinish(session, exitstatus):
    if not _ERRLOG:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_errors.json'), 'w') as f:
            json.dump(_ERRLOG, f, indent=0)
    except OSError:
        pass
