"""Per-field parity envelopes of the HIP stepper against the fp32 CPU oracle (tests/golden/parity_envelopes.json).

The envelopes derive from what was measured on an MI355X over 2048 envs x 3 rollout depths per workload
(tools/gpu_parity_stats.py -> tools/make_parity_envelopes.py; the JSON keeps the measured values and the hash of the kernel
sources they were measured on).  err = |hip - oracle| / max(1, |oracle|_inf of that env's field), per env.  For every
(workload, phase, field), with max = max(1e-5, 5 x the measured maximum):
  samples of >= 1000 envs:  99.9 % of the envs within max, none beyond min(30 x max, max(3 x max, 10 x the fp32-oracle-to-fp64-oracle
                            distance measured on the field)) (hard_cap), quantile(err, 0.99) <= p99;
  smaller samples:          at most ONE env above max and none beyond 5 x max (the 4-env goldens, the 64-env wrapper tests:
                            a small sample has no quantiles to speak of, so it is held to the bound itself);
  and at most 1 % of the envs above 1e-5 where the measurement found none (obs, reward, xpos ... on the Airbot envs).
The JSON's _provenance.csrc_sha16 is the hash of the kernel sources the envelopes were measured on;
test_parity_gpu.py::test_envelopes_were_measured_on_these_kernel_sources fails when it is not the hash of the sources in the tree.
Phases: "reset" = the env-step straight after reset, "rollout" = any later step.

What the numbers say (north_star: 1e-5 relative fp32):
  * everything the learner consumes on the Airbot envs in a rollout -- obs, reward, metrics, xpos, site_xpos, info -- is
    within 1e-5 on EVERY measured env (cube obs max 4e-6, T-shape 9e-7);
  * qvel / qacc_warmstart are documented deviations: acceleration-level quantities of an ill-conditioned solve (joint
    inertia 5e-5, |qacc| ~ 1e3..1e4): p99 1e-4, max 5e-3 against the fp32 oracle, which is itself 10-100x farther from
    its own fp64 build (column f32_vs_f64_max);
  * T-shape straight after reset: the reference's reset pose has the T block 6-12 mm inside the table, the solver's cost
    is ~4e4 with row terms up to 1e8 and its minimum is flat at fp32 resolution, so any two fp32 solvers (and fp32 vs
    fp64) land up to 1e-4 relative apart in qacc (tools/gpu_tshape_reset_diag.py); test_tshape_reset_solver_quality pins the
    kernel's answer by its optimality gap instead;
  * Go2: one Newton iteration / five line-search iterations is an unconverged solve by design; velocities carry that noise
    (obs p99 5e-5); rough terrain adds true discontinuities where a foot is equidistant from two facets."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_envelopes.json")
ENV = json.load(open(_PATH))


def scaled_err(a, b):
    n = a.shape[0]
    a, b = np.asarray(a).reshape(n, -1).astype(np.float64), np.asarray(b).reshape(n, -1).astype(np.float64)
    return (np.abs(a - b) / np.maximum(1.0, np.abs(b).max(axis=1, keepdims=True))).max(axis=1)


def bound(kind, phase, field, what="max"):
    return ENV[kind][phase][field][what]


def hard_cap(e):
    """What no env of a large sample may pass: 30 x the bound (150 x the measured maximum) at most, and where the physics says less,
    less -- ten times the distance between the fp32 oracle and its own fp64 build on that field (a HIP result farther from the fp32
    oracle than fp32 arithmetic is from fp64 arithmetic is not rounding), but never under 3 x the bound."""
    spread = e["measured"].get("f32_vs_f64_max")
    if spread is None:
        return 30.0 * e["max"]
    return min(30.0 * e["max"], max(3.0 * e["max"], 10.0 * spread))


def check(kind, phase, field, got, want, tag="", quantiles=True, outliers=0):
    """Asserts the envelope of (kind, phase, field) on the per-env scaled error of `got` against `want`; returns the errors.
    Fields without an envelope are values the step only passes through (targets, constants): exact to 1e-6.  The 99 % quantile
    is asked for on samples of >= 1000 envs (it is too noisy below); the share of envs above 1e-5 only where the measurement
    found essentially none (<= 0.2 %; the bound is the envelope's `frac`, 3 x that + 1 %): there it is the north_star's own bar, elsewhere 1e-5 sits inside the bulk of the
    distribution and the share says nothing the quantile does not.  `outliers`: envs exempt from the hard cap of a large sample (the Go2
    models with many contact pairs: a touch-down that falls on the other side of a step boundary is an O(1e-3) position change in a
    one-iteration solve; the quantile clauses still hold them to one env in a thousand)."""
    err = scaled_err(got, want)
    assert np.isfinite(err).all(), (tag, kind, phase, field, "non-finite")
    if field not in ENV[kind][phase]:
        assert err.max() <= 1e-6, (tag, kind, phase, field, "pass-through field", float(err.max()))
        return err
    e = ENV[kind][phase][field]
    # the tail is heavy (ill-conditioned solves, contact-mode switches: one env in ~3000 env-steps lands 20-30 x beyond the
    # measured maximum of a 6000-sample run): in a large sample one env in a thousand may pass the bound, none may pass 30 x it;
    # a small sample is held to the bound itself -- one env may pass it, by less than 5 x
    if len(err) >= 1000:
        cap = hard_cap(e)
        over = int((err > cap).sum())
        assert over <= outliers, (tag, kind, phase, field, "max", float(err.max()), cap, int(np.argmax(err)), over)
        assert np.quantile(err, 0.999) <= e["max"], (tag, kind, phase, field, "p99.9", float(np.quantile(err, 0.999)), e["max"])
    else:
        assert int((err > e["max"]).sum()) <= 1, (tag, kind, phase, field, "envs above max in a small sample", np.nonzero(err > e["max"])[0].tolist(), e["max"])
        assert err.max() <= 5.0 * e["max"], (tag, kind, phase, field, "max (small sample)", float(err.max()), 5.0 * e["max"], int(np.argmax(err)))
    if quantiles and len(err) >= 1000:
        assert np.quantile(err, 0.99) <= e["p99"], (tag, kind, phase, field, "p99", float(np.quantile(err, 0.99)), e["p99"])
    if quantiles and len(err) >= 200 and e["measured"]["frac"] <= 0.002:
        assert np.mean(err > 1e-5) <= e["frac"], (tag, kind, phase, field, "share above 1e-5", float(np.mean(err > 1e-5)), e["frac"])
    return err
