import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def g2g_session_counters():
    """the scheduler's counters over this pytest process (None when libg2g.so was never loaded)"""
    from prrn_aln_amd import _lib
    if _lib._lib is None:
        return None
    from prrn_aln_amd import engine
    return engine.process_counters()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line at the tail of every run: the persistent kernels' waits that ran into their limit, the DPs re-run because of it
    and the pauses waiting waves saw -- a recovery the library hid inside g2g_batch_run is visible here whatever -q captured
    (tests/test_zz_recovery_visible.py turns an unexpected one into a failure)."""
    try:
        c = g2g_session_counters()
    except Exception as e:                      # the summary must never break a run
        terminalreporter.write_line("g2g: counters unavailable (%s)" % e)
        return
    if c is None:
        return
    terminalreporter.write_line(
        "g2g: runs=%d wait_timeouts=%d recovered_dps=%d (on v1: %d) gaps=%d | injected by tests: wait_timeouts=%d recovered_dps=%d"
        % (c["runs"], c["wait_timeouts"], c["recovered_dps"], c["recovered_on_v1"], c["wait_gaps"], c["injected_timeouts"],
           c["injected_recovered_dps"]))
    if c["recovered_dps"] > c["injected_recovered_dps"]:
        from prrn_aln_amd import engine
        terminalreporter.write_line("g2g: UNEXPECTED RECOVERY -- last report: " + engine.process_last_timeout())
