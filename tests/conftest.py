import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fatal signal inside a native call must leave a Python traceback behind (the plugin is on by default; make sure of it even under -p no:...)
    import faulthandler
    if not faulthandler.is_enabled():
        faulthandler.enable(file=sys.stderr, all_threads=True)


def pytest_runtest_setup(item):
    # GPU tests call hand-written kernels through a C ABI: if the process dies with a signal, pytest's own report dies with it.  The test id is
    # therefore written (unbuffered) to stderr -- and to $BBGPU_TEST_TRACE when set, a file under gpurun_out/ that is pulled back -- BEFORE the
    # test body runs, so the last line names the test that was running.  tools/gpu_suite.sh runs the suite that way.
    if item.get_closest_marker("gpu") is None:
        return
    line = "[gpu-test] %s\n" % item.nodeid
    os.write(2, line.encode())
    trace = os.environ.get("BBGPU_TEST_TRACE")
    if trace:
        with open(trace, "a") as fh:
            fh.write(line)
            fh.flush()
            os.fsync(fh.fileno())


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json

    def load(name):
        with open(os.path.join(ROOT, "tests", "golden", name)) as fh:
            return json.load(fh)
    return load
