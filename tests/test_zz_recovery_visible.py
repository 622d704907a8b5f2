"""Runs LAST in the GPU suite (file order): every time-out of the persistent kernels' waits that happened in this pytest process
and was recovered inside g2g_batch_run must have been asked for by a test (the INJECT_STALL hook).  A recovery nobody injected is
the stall DESIGN.md 4.2 describes: results stay correct (the DPs were re-run), but the event must not hide in captured stdout --
this test fails and prints the kept report."""
import pytest

pytestmark = pytest.mark.gpu


def test_no_unreported_recovery():
    from prrn_aln_amd import engine
    c = engine.process_counters()
    if c["runs"] == 0:
        pytest.skip("no batch ran in this process (this file judges a whole -m gpu session, not itself)")
    unexpected_dps = c["recovered_dps"] - c["injected_recovered_dps"]
    unexpected_waits = c["wait_timeouts"] - c["injected_timeouts"]
    assert unexpected_dps == 0 and unexpected_waits == 0, (
        "%d wait(s) timed out and %d DP(s) were re-run without a test asking for it; last report: %s"
        % (unexpected_waits, unexpected_dps, engine.process_last_timeout()))
