"""Shared helpers for the parity tests."""
import numpy as np

from solver2d_amd import wire

# (velIters, posIters) the reference GUI uses by default (samples/settings.h:17-19) and the
# BASELINE TGS_Soft setting
DEFAULT_ITERS = {name: (4, 2) for name in wire.SOLVER_NAMES}
DEFAULT_ITERS["TGS_Soft"] = (8, 4)
DEFAULT_ITERS["SoftStep"] = (8, 4)

BODY_OUT = ["position", "rot", "linearVelocity", "angularVelocity", "deltaPosition"]
JOINT_OUT = ["impulse", "motorImpulse", "lowerImpulse", "upperImpulse"]
POINT_OUT = ["normalImpulse", "tangentImpulse", "frictionAnchorA", "frictionAnchorB", "frictionNormalA", "frictionNormalB"]


def bits(a):
    """Raw 32-bit words; every NaN is mapped to one pattern (x86 and gfx950 generate different NaN
    sign/payload bits for the same invalid operation -- a NaN must still meet a NaN)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        w = a.view(np.uint32).copy()
        w[np.isnan(a)] = 0x7FC00000
        return w
    return a


def diff_report(name, a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    bad = bits(a) != bits(b)
    if not bad.any():
        return None
    idx = np.argwhere(bad)
    first = tuple(idx[0])
    with np.errstate(invalid="ignore"):
        err = np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64))) if a.dtype.kind == "f" else 0
    return "%s: %d/%d words differ, first at %s: %r vs %r, max abs err %.3g" % (
        name, int(bad.sum()), bad.size, first, a[first], b[first], err)


def compare_exact(got, want, what=""):
    """Bitwise comparison of every solver output field of (bodies, contacts, joints)."""
    gb, gc, gj = got
    wb, wc, wj = want
    problems = []
    live = wb["type"] >= 0
    for f in BODY_OUT:
        r = diff_report("body." + f, gb[f][live], wb[f][live])
        if r:
            problems.append(r)
    for f in POINT_OUT:
        r = diff_report("contact.points." + f, gc["points"][f], wc["points"][f])
        if r:
            problems.append(r)
    r = diff_report("contact.frictionPersisted", gc["frictionPersisted"], wc["frictionPersisted"])
    if r:
        problems.append(r)
    # the reference leaves a stale constraintIndex in skipped contacts; the wire format reports -1 there
    act = wc["pointCount"] > 0
    r = diff_report("contact.constraintIndex", gc["constraintIndex"][act], wc["constraintIndex"][act])
    if r:
        problems.append(r)
    livej = wj["type"] >= 0
    for f in JOINT_OUT:
        r = diff_report("joint." + f, gj[f][livej], wj[f][livej])
        if r:
            problems.append(r)
    assert not problems, what + "\n  " + "\n  ".join(problems)


def max_abs_diff(got, want):
    """Largest absolute difference over all float solver outputs (for tolerance-based links)."""
    gb, gc, gj = got
    wb, wc, wj = want
    live = wb["type"] >= 0
    out = {}
    for f in BODY_OUT:
        out["body." + f] = float(np.max(np.abs(gb[f][live].astype(np.float64) - wb[f][live].astype(np.float64)), initial=0.0))
    for f in ["normalImpulse", "tangentImpulse"]:
        out["point." + f] = float(np.max(np.abs(gc["points"][f].astype(np.float64) - wc["points"][f].astype(np.float64)), initial=0.0))
    livej = wj["type"] >= 0
    for f in JOINT_OUT:
        out["joint." + f] = float(np.max(np.abs(gj[f][livej].astype(np.float64) - wj[f][livej].astype(np.float64)), initial=0.0))
    return out


def copy3(t):
    return tuple(x.copy() for x in t)


# ---- tolerance mode (libs2amd_fast.so: FMA contraction on) ----
# SURVEY.md 8c, link L2: "CPU restatement <=> HIP kernels, same colour order, tolerance <= 1e-5 relative ... per sweep".
# One s2Solve_* is `sweeps` passes over the constraints (warm starts, solve and relax / position sweeps: everything that
# touches a body), and a contracted a*b+c differs from the separately rounded one by at most half an ulp of the result, so
# the stated bound on every output field is
#     |fast - oracle| <= FAST_RTOL_PER_SWEEP * sweeps * scale(field)
# with scale(field) = max(|oracle field| over the scene, the field's floor below): a NORM-wise relative bound (an element
# that cancels to ~0 is measured against the field's magnitude in the scene, not against itself).
FAST_RTOL_PER_SWEEP = 1e-5
# floors: one contact slop / one slop per substep of velocity / the impulse of a 1 kg body at that velocity
FAST_SCALE_FLOOR = {"position": 1.0, "rot": 1.0, "deltaPosition": 1.0, "linearVelocity": 1.0, "angularVelocity": 1.0,
                    "normalImpulse": 1.0, "tangentImpulse": 1.0, "impulse": 1.0, "motorImpulse": 1.0, "lowerImpulse": 1.0, "upperImpulse": 1.0}


def sweeps_touching_bodies(params):
    """Passes over the constraints in one s2Solve_* call: the solve sweeps of wire.solve_sweeps_per_step plus one warm
    start per sub-step (or per step) -- the count the per-sweep tolerance is multiplied by."""
    name = wire.SOLVER_NAMES[params.solverType]
    solve = wire.solve_sweeps_per_step(name, params.velIters, params.posIters)
    substepping = name in ("TGS_Soft", "SoftStep", "TGS_Sticky", "TGS_NGS", "XPBD")
    return solve + (params.velIters if substepping else 1)


def relative_errors(got, want, params=None):
    """{field: (max abs error, scale, error / scale)} over every float solver output (norm-wise, see above).

    params of an s2Solve_XPBD call: its two impulse fields get the scale that fits what they are.  solve_xpbd.c:152 / :524 report
    lambda * inv_h of the LAST position iteration -- lambda = -C / (kA + kB + compliance) with C the residual separation of an
    almost converged contact: a difference of nearly equal numbers whose error is the POSITION error, whatever its own size.  A
    position error e_p reaches the reported impulse as e_p * inv_h * (effective mass <= the heaviest body's), so that is the scale:
    scale(position) * inv_h * max mass (a 400 kg box on a 1 kg box, high_mass_ratio3: 7 units of error on an impulse of 400)."""
    gb, gc, gj = got
    wb, wc, wj = want
    live = wb["type"] >= 0
    out = {}
    xpbd_scale = None
    if params is not None and wire.SOLVER_NAMES[params.solverType] == "XPBD":
        inv_h = params.velIters / float(params.dt)
        pos_scale = max(float(np.max(np.abs(wb["position"][live]), initial=0.0)), FAST_SCALE_FLOOR["position"])
        xpbd_scale = pos_scale * inv_h * max(float(np.max(wb["mass"][live], initial=0.0)), 1.0)

    def note(name, g, w, floor):
        g = np.asarray(g, dtype=np.float64)
        w = np.asarray(w, dtype=np.float64)
        if g.size == 0:
            return
        finite = np.isfinite(w)
        assert np.array_equal(finite, np.isfinite(g)), name + ": NaN / inf where the oracle has none (or the reverse)"
        err = float(np.max(np.abs(g[finite] - w[finite]), initial=0.0))
        scale = max(float(np.max(np.abs(w[finite]), initial=0.0)), floor)
        out[name] = (err, scale, err / scale)
    for f in BODY_OUT:
        note("body." + f, gb[f][live], wb[f][live], FAST_SCALE_FLOOR[f])
    for f in ["normalImpulse", "tangentImpulse"]:
        note("point." + f, gc["points"][f], wc["points"][f], xpbd_scale if xpbd_scale is not None else FAST_SCALE_FLOOR[f])
    livej = wj["type"] >= 0
    for f in JOINT_OUT:
        note("joint." + f, gj[f][livej], wj[f][livej], FAST_SCALE_FLOOR[f])
    return out


def compare_close(got, want, sweeps, what="", rtol_per_sweep=FAST_RTOL_PER_SWEEP, params=None):
    """The tolerance-mode comparison: every output field within rtol_per_sweep * sweeps of the oracle, norm-wise; integer
    outputs (constraintIndex, frictionPersisted) exactly."""
    errs = relative_errors(got, want, params)
    bound = rtol_per_sweep * sweeps
    problems = ["%s: |err| %.3g / scale %.3g = %.3g > %.3g" % (k, e, sc, r, bound) for k, (e, sc, r) in errs.items() if r > bound]
    act = want[1]["pointCount"] > 0
    if not np.array_equal(got[1]["constraintIndex"][act], want[1]["constraintIndex"][act]):
        problems.append("contact.constraintIndex differs")
    assert not problems, what + "\n  " + "\n  ".join(problems)
    return errs
