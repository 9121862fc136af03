"""CPU oracle for the global-optimisation hot path of aptx4869lm/4DCapture-FPV.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the `4dcapture-fpv_amd` package, the
C-ABI library, bench.py's timed GPU leg) imports or calls this package.  Allowed users:
`tests/`, `__graft_entry__.smoke()` (as the checker) and `bench.py`'s `cpu_baseline` leg.

What it is: a PyTorch-CPU (fp32 / fp64) restatement of
  * torchgeometry 0.1.2 aa<->rotmat          (oracle/tgm.py,     used at cvae.py:83,:92)
  * cvae.ContinousRotReprDecoder             (oracle/rotrepr.py, cvae.py:46-93)
  * VPoser v1.0 decoder                      (oracle/vposer.py,  global_optimization.py:270)
  * smplx SMPLX.forward / lbs                (oracle/smplx.py,   global_optimization.py:280)
  * ChamferDistancePytorch chamferDist       (oracle/chamfer.py, global_optimization.py:292)
  * FittingOP.cal_loss / init / fitting      (oracle/fitting.py, global_optimization.py:191-312,
                                              :450-489, :558-593, :632-635; modes 'local' :315-447, :499-556
                                              and 'dct' :232-246, :595-630)
  * optimization.py's per-frame smoother     (oracle/smoother.py, optimization.py:155-238, :334-348)
  * per-frame inner fit with a 2D reprojection term -- NOT in the reference (external SMPLify-X step): the autograd
    twin of csrc/fdc_fit2d.h, parity unpinned              (oracle/innerfit.py)

Parity pinning: the reference ships no tests, golden vectors or fixtures, and
the third-party packages above are absent from /root/reference and from this image, so their
arithmetic is restated from the published algorithms.  What IS pinned:
the reference's OWN code (global_optimization.py + cvae.py, imported unmodified through stub
modules in tests/golden/make_golden.py) drives these restatements and its outputs are committed
as tests/golden/*.npz; tests/test_oracle_golden.py checks oracle/fitting.py against them.
So: reference-file arithmetic = pinned by reference-generated goldens; third-party arithmetic
(tgm / smplx / VPoser / Chamfer ext) = "parity unpinned" beyond self-consistency checks.
"""
