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


def dev_switches(fn):
    """A/B tests that flip the developers' CERB_* environment switches inside the schedules: those switches are compiled OUT of the product
    library (cerberus_amd/csrc/cerb_dev.h) and exist only in libcerberus_hip_dev.so, built from the same sources with -DCERB_DEV_SWITCHES.
    The decorated test re-runs itself in a child process whose cerberus_amd loads that library (CERB_DEV_LIB=1); the parent -- which holds the
    product library -- only checks the child's verdict."""
    import functools
    import subprocess

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("CERB_DEV_LIB") == "1":
            return fn(*args, **kwargs)
        node = os.environ["PYTEST_CURRENT_TEST"].rsplit(" ", 1)[0]
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", node], cwd=ROOT, capture_output=True, text=True,
                           timeout=1800, env=dict(os.environ, CERB_DEV_LIB="1"))
        assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:] + r.stderr[-2000:])

    return wrapper
