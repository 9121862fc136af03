"""GPU tests of the batched L-BFGS / strong-Wolfe state machine (csrc/fdc_lbfgs.h, include/fdcap.h fdcap_lbfgs_*) and of
the inner fit that uses it (fdcap_opt_fit2d_lbfgs).

The checker is torch.optim.LBFGS(line_search_fn="strong_wolfe") ITSELF, on the CPU, one instance per problem, fed the same
objective: the kernel restates that published algorithm, so the two must take the same decisions.  Objective values and
gradients are computed by the SAME CPU code for both sides (the state machine only consumes f and g), so what differs is
the optimiser's own arithmetic: the summation order of its dot products and torch's mixed float / double scalars."""
import ctypes

import numpy as np
import pytest
import torch

import fdcap_amd  # noqa: F401
from fdcap_amd import capi

pytestmark = pytest.mark.gpu


class Quadratic:
    """f = 1/2 x'Ax - b'x, A symmetric positive definite with condition number ~ 300"""
    dim = 78

    def __init__(self, rng):
        q, _ = np.linalg.qr(rng.standard_normal((self.dim, self.dim)))
        self.A = torch.tensor((q * np.geomspace(1.0, 300.0, self.dim)) @ q.T, dtype=torch.float32)
        self.b = torch.tensor(rng.standard_normal(self.dim), dtype=torch.float32)
        self.x0 = torch.tensor(rng.standard_normal(self.dim), dtype=torch.float32)

    def __call__(self, x):
        return 0.5 * x @ (self.A @ x) - self.b @ x


class Rosenbrock:
    dim = 10

    def __init__(self, rng):
        self.x0 = torch.tensor(rng.uniform(-1.5, 1.5, self.dim), dtype=torch.float32)

    def __call__(self, x):
        return torch.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2)


class RobustFit:
    """Geman-McClure residuals of a linear model + an L2 prior: the shape of the inner fit's objective"""
    dim = 40

    def __init__(self, rng):
        self.M = torch.tensor(rng.standard_normal((120, self.dim)), dtype=torch.float32)
        xt = rng.standard_normal(self.dim)
        y = self.M.numpy() @ xt + 0.05 * rng.standard_normal(120)
        y[::9] += 8.0                                               # outliers
        self.y = torch.tensor(y, dtype=torch.float32)
        self.x0 = torch.tensor(xt + 0.7 * rng.standard_normal(self.dim), dtype=torch.float32)

    def __call__(self, x):
        r2 = (self.M @ x - self.y) ** 2
        return torch.sum(4.0 * r2 / (r2 + 4.0)) + 0.01 * torch.sum(x ** 2)


def _value_and_grad(fun, x):
    x = x.detach().clone().requires_grad_(True)
    f = fun(x)
    f.backward()
    return float(f.detach()), x.grad.detach()


def _as_double(fun):
    """The same objective with every tensor it holds in float64 (torch.optim.LBFGS then runs in float64 throughout)."""
    import copy
    f64 = copy.copy(fun)
    for k, v in vars(fun).items():
        if torch.is_tensor(v):
            setattr(f64, k, v.double())
    return f64


def _torch_lbfgs(fun, cfg):
    """SMPLify-X's loop around torch.optim.LBFGS.step (oracle/innerfit.py fitting_lbfgs has the same loop)."""
    x = fun.x0.clone().requires_grad_(True)
    opt = torch.optim.LBFGS([x], lr=cfg["lr"], max_iter=cfg["max_iter"], max_eval=None, history_size=cfg["history"],
                            tolerance_grad=cfg["tolerance_grad"], tolerance_change=cfg["tolerance_change"], line_search_fn="strong_wolfe")
    calls = [0]

    def closure():
        opt.zero_grad()
        f = fun(x)
        f.backward()
        calls[0] += 1
        return f
    prev, steps = None, 0
    for n in range(cfg["max_steps"]):
        loss = float(opt.step(closure).detach())
        steps += 1
        if not np.isfinite(loss):
            break
        if n > 0 and cfg["ftol"] > 0 and abs(prev - loss) / max(abs(prev), abs(loss), 1.0) <= cfg["ftol"]:
            break
        if float(_value_and_grad(fun, x)[1].abs().max()) < cfg["gtol"]:
            break
        prev = loss
    f_end = float(fun(x.detach()))
    return x.detach().clone(), f_end, opt.state[x]["n_iter"], calls[0] - (steps - 1)      # (the kernel does not repeat the closure call that opens a later step)


def _run_kernel(funs, cfg, max_rounds=5000, finalize=False):
    lib = capi.load_library()
    n, dim = len(funs), funs[0].dim
    stride = dim + 3                                                # rows wider than the problem: the strides are honoured
    cf = capi.LbfgsConfig(dim, cfg["history"], cfg["max_iter"], 0, cfg["max_steps"], 25, cfg["lr"], cfg["tolerance_grad"],
                          cfg["tolerance_change"], cfg["ftol"], cfg["gtol"])
    h = ctypes.c_void_p()
    capi.check(lib.fdcap_lbfgs_create(n, ctypes.byref(cf), ctypes.byref(h)), "fdcap_lbfgs_create")
    X = torch.full((n, stride), 7.0, device="cuda")
    X[:, :dim] = torch.stack([f.x0 for f in funs]).cuda()
    active = torch.zeros(1, dtype=torch.int32, device="cuda")
    rounds = 0
    try:
        while rounds < max_rounds:
            xc = X[:, :dim].cpu()
            fg = [_value_and_grad(f, xc[i]) for i, f in enumerate(funs)]
            F = torch.tensor([v for v, _ in fg], dtype=torch.float32, device="cuda")
            G = torch.zeros(n, stride, device="cuda")
            G[:, :dim] = torch.stack([g for _, g in fg]).cuda()
            capi.check(lib.fdcap_lbfgs_advance(h, capi.dptr(X), stride, capi.dptr(F), capi.dptr(G), stride, capi.dptr(active),
                                               capi.current_stream()), "fdcap_lbfgs_advance")
            rounds += 1
            if int(active.item()) == 0:
                break
        if finalize:
            x_trial = X[:, :dim].cpu()
            unf = torch.zeros(1, dtype=torch.int32, device="cuda")
            capi.check(lib.fdcap_lbfgs_finalize(h, capi.dptr(X), stride, capi.dptr(unf), capi.current_stream()), "fdcap_lbfgs_finalize")
        it = torch.zeros(n, dtype=torch.int32, device="cuda")
        ev = torch.zeros(n, dtype=torch.int32, device="cuda")
        loss = torch.zeros(n, device="cuda")
        capi.check(lib.fdcap_lbfgs_get_stats(h, capi.dptr(it), capi.dptr(ev), capi.dptr(loss), capi.current_stream()), "fdcap_lbfgs_get_stats")
        torch.cuda.synchronize()
        assert torch.all(X[:, dim:] == 7.0), "the kernel wrote outside its problem's columns"
        if finalize:
            return X[:, :dim].cpu(), it.cpu().numpy(), ev.cpu().numpy(), loss.cpu().numpy(), rounds, int(unf.item()), x_trial
        return X[:, :dim].cpu(), it.cpu().numpy(), ev.cpu().numpy(), loss.cpu().numpy(), rounds
    finally:
        lib.fdcap_lbfgs_destroy(h)


CFG = dict(history=100, max_iter=30, max_steps=1, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, ftol=0.0, gtol=0.0)


@pytest.mark.parametrize("family,cfg", [
    (Quadratic, dict(CFG, max_iter=40)),
    (Quadratic, dict(CFG, max_iter=40, history=5)),                 # the ring buffer turns over
    (Rosenbrock, dict(CFG, max_iter=15)),                           # the first directions of a curved valley: the same path
    (Rosenbrock, dict(CFG, max_iter=400)),                          # ... and run to convergence: the same minimum
    (RobustFit, dict(CFG, max_iter=50)),
], ids=["quadratic", "quadratic-history5", "rosenbrock-first15", "rosenbrock-converged", "robust"])
def test_one_optimizer_step_takes_torchs_decisions(family, cfg):
    """One optimizer.step(closure) of up to max_iter directions: same number of directions and objective calls, same point."""
    rng = np.random.Generator(np.random.PCG64(11))
    funs = [family(rng) for _ in range(12)]
    x, it, ev, loss, rounds = _run_kernel(funs, cfg)
    same_counts, any_converged, off_path, ties = 0, False, 0, 0
    for i, f in enumerate(funs):
        xr, fr, itr, evr = _torch_lbfgs(f, cfg)
        err = float((x[i] - xr).abs().max()) / max(1.0, float(xr.abs().max()))
        same = it[i] == itr and ev[i] == evr
        same_counts += int(same)
        converged = itr < cfg["max_iter"]                           # torch stopped on a tolerance, not on the cap
        any_converged |= converged
        note = ""
        if not same and not converged:
            # Host-independent classification: the same problem through torch.optim.LBFGS in float64.  Where torch's
            # own two precisions take different decisions, a comparison on the path sits within fp32 rounding of a tie and no fp32
            # implementation is "the" path; where they agree and the kernel does not, the kernel left the path.
            _, _, it64, ev64 = _torch_lbfgs(_as_double(f), cfg)
            tie = (it64, ev64) != (itr, evr)
            ties += int(tie)
            off_path += int(not tie)
            note = f"  [torch float64: {it64} / {ev64} -> {'tie-sensitive problem' if tie else 'KERNEL OFF THE PATH'}]"
        print(f"{family.__name__} {i}: directions {it[i]} / {itr}, objective calls {ev[i]} / {evr}, loss {loss[i]:.6g} / {fr:.6g}, |dx| {err:.2e}{note}")
        # Stopped on a tolerance: the same minimum (the last, rounding-sized steps need not be the same in number).
        # Stopped by the cap: the same decisions all the way, hence the same point -- unless a comparison sat within rounding
        # of a tie and sent the two down different, equally valid, paths (counted below).
        if converged or same:
            assert abs(float(f(x[i])) - fr) <= 1e-4 * max(1.0, abs(fr)) + 1e-5
            assert err < (1e-3 if converged else 2e-3)
    if not any_converged:
        # (measured on this pool's hosts: 11, 12 and 11 of 12 equal in the three cap-limited families.)  The bar does not depend
        # on the host's summation order: a count mismatch is excused only where torch in float64 disagrees with torch in float32
        # on the same problem; elsewhere at most two of twelve may differ (the bar before r4's widening).
        print(f"{family.__name__}: {same_counts} of {len(funs)} on torch's path, {ties} tie-sensitive, {off_path} off the path")
        assert off_path <= 2, f"{off_path} of {len(funs)} problems left torch.optim.LBFGS's path where torch's own precisions agree"
        assert same_counts + ties >= len(funs) - 2
    assert rounds == ev.max()                                        # one objective call per round for the slowest problem, no more


def test_the_loop_around_optimizer_step_stops_where_smplifyx_would():
    """max_steps > 1 with SMPLify-X's ftol rule: a problem stops when two successive step() calls start at (nearly) the same loss."""
    rng = np.random.Generator(np.random.PCG64(5))
    cfg = dict(CFG, max_iter=8, max_steps=30, ftol=2e-9, gtol=1e-9)
    funs = [RobustFit(rng) for _ in range(6)]
    x, it, ev, loss, rounds = _run_kernel(funs, cfg)
    for i, f in enumerate(funs):
        xr, fr, itr, evr = _torch_lbfgs(f, cfg)
        print(f"robust {i}: directions {it[i]} / {itr}, objective calls {ev[i]} / {evr}, loss {loss[i]:.6g} / {fr:.6g}")
        assert abs(float(f(x[i])) - fr) <= 1e-4 * max(1.0, abs(fr))
        assert abs(int(it[i]) - itr) <= max(3, itr // 5)
    assert it.max() < 8 * 30                                          # nobody ran to the cap: the stopping rules work


def test_finished_problems_stay_put_and_bad_arguments_are_refused():
    lib = capi.load_library()
    h = ctypes.c_void_p()
    E_ARG = -1
    bad = capi.LbfgsConfig(200, 100, 30, 0, 1, 25, 1.0, 1e-7, 1e-9, 0.0, 0.0)              # dim > 128
    assert lib.fdcap_lbfgs_create(4, ctypes.byref(bad), ctypes.byref(h)) == E_ARG
    bad = capi.LbfgsConfig(10, 500, 30, 0, 1, 25, 1.0, 1e-7, 1e-9, 0.0, 0.0)               # history > 128
    assert lib.fdcap_lbfgs_create(4, ctypes.byref(bad), ctypes.byref(h)) == E_ARG
    rng = np.random.Generator(np.random.PCG64(3))
    funs = [Quadratic(rng) for _ in range(3)]
    cfg = dict(CFG, max_iter=3)
    x, it, ev, loss, rounds = _run_kernel(funs, cfg)
    for i, f in enumerate(funs):                                     # (max_eval = 3 * 5 // 4 ends the step after one direction, as in torch)
        xr, fr, itr, evr = _torch_lbfgs(f, cfg)
        assert it[i] == itr and ev[i] == evr and float((x[i] - xr).abs().max()) < 1e-5
    # a problem whose gradient is already zero finishes in its first round without moving
    class Flat:
        dim = 10
        x0 = torch.arange(10, dtype=torch.float32)
        def __call__(self, x):
            return torch.sum(x * 0.0) + 1.5
    x, it, ev, loss, rounds = _run_kernel([Flat()], CFG)
    assert rounds == 1 and it[0] == 0 and torch.equal(x[0], Flat.x0) and loss[0] == 1.5


def test_a_round_budget_returns_accepted_points_not_trial_points():
    """a caller that stops asking mid line search used to keep the TRIAL point in its rows.  fdcap_lbfgs_finalize puts
    the last accepted point back: the objective there equals the `loss` the stats report, and it is not above the start's."""
    rng = np.random.Generator(np.random.PCG64(17))
    funs = [Rosenbrock(rng) for _ in range(8)]
    for budget in (2, 3, 5, 9):
        x, it, ev, loss, rounds, unfinished, x_trial = _run_kernel(funs, dict(CFG, max_iter=400), max_rounds=budget, finalize=True)
        assert rounds == budget and unfinished > 0
        moved = 0
        for i, f in enumerate(funs):
            fx = float(f(x[i]))
            assert abs(fx - float(loss[i])) <= 1e-5 * max(1.0, abs(fx)), (budget, i, fx, float(loss[i]))
            assert fx <= float(f(f.x0)) * (1 + 1e-6)
            moved += int(not torch.equal(x[i], x_trial[i]))
        assert moved > 0, "no row held a trial point: the case under test did not occur"


def _px_err(orc, rows, kp):
    from oracle import rotrepr
    with torch.no_grad():
        uv = orc.project(orc.joints_cam(rotrepr.convert_to_6D_rot(torch.tensor(rows)))).numpy()
    w = kp[..., 2] > 0
    return float(np.sqrt(((uv - kp[..., :2]) ** 2).sum(-1))[w].mean())


def _stage1_lbfgs_case(with64=True):
    """stage-1 inner fit of 8 frames with the library's L-BFGS and with torch.optim.LBFGS on the oracle's objective (float32 and
    float64) -> (library rows, oracle rows fp32, oracle rows fp64, library's per-frame objective, InnerFitOP, fp64 oracle, kp, stage)"""
    from fdcap_amd.innerfit import DEFAULT_STAGES, InnerFitOP
    from oracle.innerfit import InnerFitOracle
    from oracle.smplx import SMPLXOracle
    from oracle.vposer import VPoserDecoder
    from tests.test_gpu_innerfit import _case
    n = 8
    bm, vp, gt, init, kp = _case(n, 21)
    stages = DEFAULT_STAGES[0:1]
    op = InnerFitOP(bm, vp, n, stages=stages, optimizer="lbfgs")
    out = op.fitting(init, kp, log_every=1).cpu().numpy()
    orc = InnerFitOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp))
    ref = orc.fitting_lbfgs(init, kp, stages).numpy()
    o64 = InnerFitOracle(SMPLXOracle(bm, dtype=torch.float64), VPoserDecoder.from_data(vp, dtype=torch.float64), dtype=torch.float64)
    ref64 = o64.fitting_lbfgs(init, kp, stages).numpy() if with64 else None
    return out, ref, ref64, op, orc, o64, kp, stages[0]


def _objective64(o64, rows75, kp, stage):
    """per-frame objective and largest gradient entry of the float64 oracle at the given [N,75] rows (format-independent yardstick)"""
    from oracle import rotrepr
    x = rotrepr.convert_to_6D_rot(torch.tensor(rows75, dtype=torch.float64)).detach().requires_grad_(True)
    k = torch.tensor(kp, dtype=torch.float64)
    f, g = [], []
    for i in range(x.shape[0]):
        xi = x[i:i + 1].detach().clone().requires_grad_(True)
        li = sum(o64.loss(xi, k[i:i + 1], stage))
        gi, = torch.autograd.grad(li, xi)
        f.append(float(li.detach())); g.append(float(gi.abs().max()))
    return np.array(f), np.array(g)


def test_inner_fit_with_lbfgs_ends_where_torchs_lbfgs_ends_on_the_oracles_objective():
    """fdcap_opt_fit2d_lbfgs (every frame its own L-BFGS problem) against oracle/innerfit.py fitting_lbfgs = torch.optim.LBFGS
    per frame on the oracle's autograd objective, SMPLify-X's settings (30 x 30, ftol 2e-9), in the first stage: its strong priors
    make the minimum unique enough that two arithmetics reach the same one.

    ADVICE r5: the bars of the default product format (two fp16 planes) must not be derived from that format's own deviation.
    They are now FORMAT-INDEPENDENT: the float64 objective at the library's end point, the stationarity of that point, and its
    distance to torch's float64 run measured with torch's float32 run as the yardstick (two arithmetics of the SAME optimiser).
    The old parameter bars (3e-2 / 2e-3, loss 1e-3) stay on the exact-fp32 form: the next test, in a child process."""
    out, ref, ref64, op, orc, o64, kp, stage = _stage1_lbfgs_case()
    fl, ofl = op.frame_loss[0], np.array(orc.final_loss[0])
    f_out, g_out = _objective64(o64, out, kp, stage)
    f_ref, g_ref = _objective64(o64, ref, kp, stage)
    f_r64, g_r64 = _objective64(o64, ref64, kp, stage)
    err64, yard = np.abs(out - ref64), np.abs(ref - ref64)
    print("L-BFGS inner fit, stage 1: float64 objective at the end points, worst frame relative to torch-fp64's: library",
          (f_out / f_r64 - 1).max(), "torch fp32", (f_ref / f_r64 - 1).max(), "| largest gradient entry: library", g_out.max(), "torch fp32",
          g_ref.max(), "torch fp64", g_r64.max(), "| parameters vs torch fp64: library max", err64.max(), "q90", np.quantile(err64, 0.9),
          "; torch fp32 max", yard.max(), "q90", np.quantile(yard, 0.9), "| rounds", op.rounds, "directions", op.frame_iterations[0])
    # (1) the end point is as good a minimiser as torch's, frame by frame, in float64 -- fixed bar, whatever the product format
    assert np.all(f_out <= f_r64 * (1 + 2e-3)), (f_out / f_r64 - 1)   # (measured: library +7.1e-4, torch's own float32 run +3.4e-4)
    # (2) the objective the library reports is the objective there (its own arithmetic against float64)
    np.testing.assert_allclose(fl, f_out, rtol=5e-4)
    np.testing.assert_allclose(op.log[0][0] + op.log[0][1], f_out.sum(), rtol=5e-4)    # re-evaluated at the returned rows
    # (3) ... and a stationary point to the degree torch's own float32 run is one (objective ~5e3 per frame, parameters O(1))
    assert g_out.max() <= max(10.0 * g_ref.max(), 1.0), (g_out.max(), g_ref.max())
    # (4) the bulk of the parameters sits where torch's float64 run ends (fixed); the flattest direction may differ by what
    #     separates torch's own float32 and float64 runs, times a margin that is NOT read off the library
    assert np.quantile(err64, 0.9) < 2e-3
    assert err64.max() <= max(25.0 * yard.max(), 0.1), (err64.max(), yard.max())
    assert op.rounds[0] < 30 * 38                                                       # stopped by the rules, not by the cap
    op.close()


_EXACT_CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
from tests.test_gpu_lbfgs import _stage1_lbfgs_case
out, ref, ref64, op, orc, o64, kp, stage = _stage1_lbfgs_case(with64=False)
err = np.abs(out - ref)
fl, ofl = op.frame_loss[0], np.array(orc.final_loss[0])
print("RESULT " + json.dumps({"max": float(err.max()), "q90": float(np.quantile(err, 0.9)), "loss_rel": float(np.abs(fl / ofl - 1).max())}))
"""


def test_exact_fp32_inner_fit_keeps_the_round_4_parameter_bars():
    """The bars this comparison carried before the two-plane fp16 products became the default (r4: err.max 3e-2, q90 2e-3,
    objective rtol 1e-3) are kept on the exact-fp32 product form (FDCAP_GEMM_SPLIT3=0; the switch is read once per process:
    child process), so a regression of the optimiser itself cannot hide behind a product format's rounding."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _EXACT_CHILD % root], env=dict(os.environ, FDCAP_GEMM_SPLIT3="0"), capture_output=True,
                       text=True, timeout=900, cwd=root)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert p.returncode == 0 and line, p.stderr[-2000:]
    r = json.loads(line[-1][7:])
    print("exact fp32 products vs torch.optim.LBFGS (fp32):", r)
    assert r["max"] < 3e-2 and r["q90"] < 2e-3 and r["loss_rel"] < 1e-3, r


def test_five_stage_lbfgs_fit_reduces_the_reprojection_error_like_the_oracles():
    """BASELINE config 4's schedule with L-BFGS.  The late stages' weak priors leave a flat, non-convex objective: the minimum a
    run ends in depends on rounding (the oracle in fp32 and fp64: parameters up to 0.7 apart, objective 2.5 %), so the comparison
    is of the objective reached and of the reprojection error, with that yardstick."""
    from fdcap_amd.innerfit import DEFAULT_STAGES, InnerFitOP
    from oracle.innerfit import InnerFitOracle
    from oracle.smplx import SMPLXOracle
    from oracle.vposer import VPoserDecoder
    from tests.test_gpu_innerfit import _case
    n = 4
    bm, vp, gt, init, kp = _case(n, 21)
    op = InnerFitOP(bm, vp, n, optimizer="lbfgs")
    out = op.fitting(init, kp).cpu().numpy()
    orc = InnerFitOracle(SMPLXOracle(bm), VPoserDecoder.from_data(vp))
    ref = orc.fitting_lbfgs(init, kp, DEFAULT_STAGES).numpy()
    fl, ofl = op.frame_loss[-1], np.array(orc.final_loss[-1])
    e0, e1, e1o = _px_err(orc, init, kp), _px_err(orc, out, kp), _px_err(orc, ref, kp)
    print("five stages: objective per frame", fl, "oracle", ofl, "| rounds per stage", op.rounds, "oracle closure calls per stage",
          [max(e) for e in orc.evals], "| mean reprojection error (px): start", e0, "end", e1, "oracle", e1o)
    # (measured: one of four frames has two minima, 274 and 312: this library and the fp64 oracle end in the lower one, the fp32
    # oracle in either depending on the host CPU -- so no frame may end WORSE than the oracle's by more than the yardstick, and the
    # clip's total agrees to it)
    # (per frame the yardstick is 15 %: the fp32 oracle on two hosts and the fp64 oracle end frame 2 at 312.7 / 279.9 / 273.1)
    assert np.all(fl <= 1.15 * ofl) and abs(fl.sum() / ofl.sum() - 1) < 0.08
    assert e1 < 0.5 * e0 and abs(e1 - e1o) < 0.25 * e1o + 0.5
    op.close()
