import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu on the GPU box")
    config.addinivalue_line("markers", "wall_clock(seconds): this test's own wall-clock bound (default: CELO_TEST_TIMEOUT, 900 s)")


# Every test runs under a wall-clock bound.  A kernel that never returns blocks the interpreter inside hipStreamSynchronize, where no Python-level
# timeout can reach it (round 5 lost a 50-minute GPU lease to one hung k_combine_big<G_761>): faulthandler's watchdog THREAD dumps the traceback
# of the stuck test - naming it - and ends the process with os._exit, so a hang costs minutes, not the lease.
DEFAULT_WALL_CLOCK_S = float(os.environ.get("CELO_TEST_TIMEOUT", "900"))


def _real_stderr_fd(config):
    """The terminal's stderr: pytest's fd-level capture has redirected fd 2 by the time a test runs, and a process ended by os._exit never
    reports what it captured.  Falls back to a log file under gpurun_out/ (merged back from the GPU box)."""
    try:
        capman = config.pluginmanager.getplugin("capturemanager")
        return os.dup(capman._global_capturing.err.targetfd_save)
    except Exception:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        return os.open(os.path.join(ROOT, "gpurun_out", "pytest_wall_clock.txt"), os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)


_WATCHDOG_FILE = None


@pytest.fixture(autouse=True)
def _wall_clock_bound(request):
    import faulthandler
    global _WATCHDOG_FILE
    m = request.node.get_closest_marker("wall_clock")
    limit = float(m.args[0]) if m and m.args else DEFAULT_WALL_CLOCK_S
    if limit <= 0:
        yield
        return
    if _WATCHDOG_FILE is None:
        _WATCHDOG_FILE = os.fdopen(_real_stderr_fd(request.config), "w")
    out, nodeid = _WATCHDOG_FILE, request.node.nodeid

    def expire():      # a Python thread runs while the main thread waits inside a ctypes call (the GIL is released there)
        try:
            faulthandler.dump_traceback(file=out, all_threads=True)
        except Exception:
            pass
        out.write("\nWALL-CLOCK BOUND: %s did not finish within %.0f s; ending the test process (tests/conftest.py)\n" % (nodeid, limit))
        out.flush()
        os._exit(1)

    import threading
    timer = threading.Timer(limit, expire)
    timer.daemon = True
    timer.start()
    # backstop without the GIL (a C-level watchdog thread), a few seconds later: its traceback names the test function that was running
    faulthandler.dump_traceback_later(limit + 5, exit=True, file=out)
    try:
        yield
    finally:
        timer.cancel()
        faulthandler.cancel_dump_traceback_later()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu():
    """Initialises the HIP library on device 0; fails loudly (no fallback) if it cannot."""
    from celo_bls_snark_rs_amd import ffi
    ffi.init(0)
    return ffi
