"""Helpers for the tests that run TWO rank processes on the ONE GPU of the test box.

Round 3 found that such runs were not perfectly repeatable, and why (tools/flake_trace.py, tools/two_stream_trace.py; DESIGN.md
section 7): on the MI355X boxes of this pool a packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) now and then
returns a wrong low element in lanes 48-63 while ANOTHER kernel keeps the matrix pipe of the same CU busy -- a second stream
or a second process running this library's MFMA kernels next to its one-wave pose kernels.  One process with one stream (the
product's arrangement) never has two kernels resident at once and never showed it.  The library is now built without packed
fp32 instructions (__graft_entry__.build), after which 0 of 280 traced two-rank / two-stream fits differ (before: 23-30 of 30).

The comparison helper below stays as a tripwire: a two-rank comparison that fails is run once more, and a pass on the second
attempt is REPORTED (a warning), not hidden."""
import warnings


def retry_on_shared_gpu_glitch(check, attempts=2):
    """Run `check()` (which raises AssertionError on a mismatch) up to `attempts` times; the last failure propagates."""
    for k in range(attempts):
        try:
            out = check()
            if k:
                warnings.warn("a two-rank comparison on the shared GPU passed only on attempt %d (tests/shared_gpu.py)" % (k + 1))
            return out
        except AssertionError:
            if k == attempts - 1:
                raise
