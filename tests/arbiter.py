"""Step-by-step parity of a fitting run against the oracle with DEMONSTRATED bounds (test infrastructure).

What is compared and why.  A run of K Adam iterations is not comparable entry by entry at the end: Adam's step lr * m / (sqrt(v) + 1e-8) turns
a 1e-9 difference of a near-zero gradient entry into a 1e-2 difference of the parameter, so two correct fp32 evaluations of the same loop (the
reference's CPU and CUDA runs, this oracle, the HIP kernels) drift apart on exactly those entries.  Earlier rounds answered that with
allowances ("up to three loose bodies", "90 % of the entries within 1e-3").  Here every iteration is checked ON ITS OWN, from the state the
implementation under test actually had (teacher forcing), against two evaluations of the reference's arithmetic at that state:

  * ``oracle/psi_oracle.py`` in fp32 (the restatement that is pinned to the reference's recorded numbers), and
  * the same code in fp64 (the ARBITER, ``SMPLXOracle(dtype=torch.float64)``).

Per iteration:

  * the four loss values at the iteration's parameters (fitting_proxe.py:101-162): |implementation - arbiter| <= K_NOISE x |oracle_fp32 -
    arbiter| + 3e-6 relative;
  * the gradient (recovered from Adam's first moment: g = (m_after - beta1 m_before) / (1 - beta1)), per body (largest entry), passes if
      (a) it is within 1e-4 of the batch's largest gradient entry of the fp32 ORACLE's gradient — the north star's tolerance against the
          pinned restatement — or
      (b) it is no further from the ARBITER than K_NOISE x the fp32 oracle's own distance from the arbiter in that body (the reference's
          rotation-matrix -> quaternion -> angle-axis chain, cvae.py:128-137, is ill-conditioned for some orientations: there the fp32
          oracle itself is 1e-3..1e-2 from the exact gradient, and an implementation that rounds differently must be allowed the same), or
      (c) the body contains a vertex whose fp64 SDF value is within tau of zero, tau = K_NOISE x the largest |sdf_fp32 - sdf_fp64| the fp32
          ORACLE itself shows over all vertices (>= AMBIGUOUS) — the ``sdf < 0`` mask (fitting_proxe.py:155) is discontinuous and a
          correct fp32 evaluation may count such a vertex either way — and (a) or (b) holds against the oracle /
          arbiter evaluated with those vertices counted in or counted out (both are tried; bodies without such a vertex see the mask only
          through the global count N, a relative change of n_ambiguous / N that is added to their bounds);
    the test reports how many bodies needed (b) or (c);
  * Adam's update (torch.optim.Adam, fitting_proxe.py:73-74,188-189) recomputed in fp64 from the implementation's OWN state and gradient:
    m, v and the parameters to fp32 rounding of the formula (the gradient having been checked above, this isolates the optimiser).
"""
import numpy as np
import torch

import psi_oracle as O

K_NOISE = 4.0          # implementation-vs-arbiter may be this many times the fp32 oracle's own distance from the arbiter
AMBIGUOUS = 1e-6       # floor of the ambiguity threshold tau (below): |sdf_fp64| < tau: the sdf < 0 decision of the vertex is not determined in fp32
BETA1, BETA2, EPS = 0.9, 0.999, 1e-8    # torch.optim.Adam defaults (fitting_proxe.py:73-74)


def engine_state(op):
    """(x [B,75] in the 6D representation, Adam m, Adam v) of a FittingOP as float64 arrays (zeros before the first step)."""
    B = op.batch_size
    if op.engine == 'fused':
        eng = op._fused
        x, _, _ = eng.read(0)
        m, v = eng.buffer('adam_m', (B, 75)), eng.buffer('adam_v', (B, 75))
    else:
        x = op.xhr_rec.detach()
        st = op.optimizer.state.get(op.xhr_rec, {})
        m = st.get('exp_avg', torch.zeros_like(x))
        v = st.get('exp_avg_sq', torch.zeros_like(x))
    f = lambda t: t.detach().cpu().numpy().astype(np.float64)
    return f(x), f(m), f(v)


def gpu_trace(op, bodies, iters):
    """Run ``iters`` single iterations of the product and record the state around each of them."""
    runner = op.make_step_runner(bodies)
    trace = []
    for _ in range(iters):
        x0, m0, v0 = engine_state(op)
        runner.step()
        losses = np.asarray(runner.last_losses(), np.float64)
        x1, m1, v1 = engine_state(op)
        trace.append(dict(x0=x0, m0=m0, v0=v0, x1=x1, m1=m1, v1=v1, losses=losses))
    runner.finish()
    return trace


def save_trace(path, trace):
    np.savez(path, **{'%s_%d' % (k, i): v for i, t in enumerate(trace) for k, v in t.items()})


def load_traces(paths):
    """Traces of the ranks of a data-parallel run (rows sharded, loss values global) -> one trace over the global batch."""
    parts = [np.load(p) for p in paths]
    n = len([k for k in parts[0].files if k.startswith('losses_')])
    out = []
    for i in range(n):
        t = {k: np.concatenate([p['%s_%d' % (k, i)] for p in parts]) for k in ('x0', 'm0', 'v0', 'x1', 'm1', 'v1')}
        for p in parts[1:]:
            assert np.array_equal(p['losses_%d' % i], parts[0]['losses_%d' % i])      # every rank reports the GLOBAL loss values
        t['losses'] = parts[0]['losses_%d' % i]
        out.append(t)
    return out


def _evaluate(fo, x, xhr, cam, force=None):
    """Loss values, gradient and SDF values of FittingOracle ``fo`` at parameters x ([B,75], 6D form).  ``force`` = (mask [B,V] of
    vertices, bool value): those vertices are counted as penetrating (True) or not (False) whatever the sign of their SDF value."""
    dt = fo.dtype
    fo.xhr_rec.data = torch.as_tensor(x, dtype=dt).clone()
    fo.xhr_rec.grad = None
    fo.pen_override = None if force is None else (torch.as_tensor(force[0]), bool(force[1]))
    losses = fo.cal_loss(torch.as_tensor(xhr, dtype=dt), torch.as_tensor(cam, dtype=dt))
    sum(losses).backward()
    fo.pen_override = None
    return (np.array([float(l.detach()) for l in losses]), fo.xhr_rec.grad.detach().numpy().astype(np.float64),
            fo.last.sdf.detach().numpy().reshape(x.shape[0], -1).astype(np.float64))


def _adam(x0, m0, v0, g, t, lr):
    m = BETA1 * m0 + (1 - BETA1) * g
    v = BETA2 * v0 + (1 - BETA2) * g * g
    denom = np.sqrt(v) / np.sqrt(1 - BETA2 ** t) + EPS
    return x0 - lr / (1 - BETA1 ** t) * m / denom, m, v


def check_gradient(g_gpu, f32, f64, x, xhr, cam):
    """The gradient rules (a) / (b) / (c) of the module docstring for ONE evaluation point x ([B,75], 6D form): g_gpu [B,75] is the
    implementation's gradient there, f32 / f64 the fp32 oracle and the fp64 arbiter (FittingOracle).  Asserts every body passes; returns
    the evaluations and a summary."""
    l32, g32, sdf32 = _evaluate(f32, x, xhr, cam)
    l64, g64, sdf64 = _evaluate(f64, x, xhr, cam)
    # how far an fp32 evaluation of a vertex's SDF value is from the exact one (vertex coordinates of a few metres carry ~5e-7 m of
    # rounding, times |grad sdf| ~ 1): measured on the fp32 oracle, not assumed
    tau = max(AMBIGUOUS, K_NOISE * float(np.abs(sdf32 - sdf64).max()))
    amb = np.abs(sdf64) < tau
    n_pen = max(int((sdf64 < 0).sum()), 1)
    scale = np.abs(g64).max()
    slack = (amb.sum() / n_pen) * scale                        # the global count N seen by bodies without an ambiguous vertex
    bmax = lambda a: np.abs(a).max(axis=1)

    def rules(r32, r64):
        a = bmax(g_gpu - r32) <= 1e-4 * scale + slack
        b = bmax(g_gpu - r64) <= K_NOISE * bmax(r32 - r64) + 2e-6 * scale + slack
        return a, b
    ok_a, ok_b = rules(g32, g64)
    ok_a_strict = bmax(g_gpu - g32) <= 1e-4 * scale             # rule (a) WITHOUT the ambiguity slack: the north star's literal 1e-4 (reported and pinned, record())
    ok_c = np.zeros_like(ok_a)
    bodies_amb = np.nonzero(amb.any(axis=1))[0]
    if len(bodies_amb) and not np.all(ok_a | ok_b):
        alts32 = [_evaluate(f32, x, xhr, cam, force=(amb, val))[1] for val in (True, False)]
        alts64 = [_evaluate(f64, x, xhr, cam, force=(amb, val))[1] for val in (True, False)]
        for r32 in alts32:
            for r64 in alts64:
                a, b = rules(r32, r64)
                ok_c[bodies_amb] |= (a | b)[bodies_amb]
    ok = ok_a | ok_b | ok_c
    assert np.all(ok), (np.nonzero(~ok)[0].tolist(), (bmax(g_gpu - g32) / scale)[~ok], (bmax(g_gpu - g64) / scale)[~ok],
                        (bmax(g32 - g64) / scale)[~ok], bodies_amb.tolist(), tau)
    info = dict(grad_vs_oracle32_rel=float(np.median(bmax(g_gpu - g32)) / scale), grad_vs_oracle32_worst_rel=float(bmax(g_gpu - g32).max() / scale),
                oracle32_vs_arbiter_worst_rel=float(bmax(g32 - g64).max() / scale),
                bodies_by_rule=dict(a=int(ok_a.sum()), b_only=int((ok_b & ~ok_a).sum()), c_only=int((ok_c & ~ok_a & ~ok_b).sum())),
                bodies_within_1e4_of_oracle32_no_slack=int(ok_a_strict.sum()), bodies=int(len(ok_a)),
                grad_vs_arbiter_worst_rel=float(bmax(g_gpu - g64).max() / scale), slack_rel=float(slack / scale),
                ambiguous_vertices=int(amb.sum()), ambiguity_threshold=tau)
    return (l32, l64, amb, n_pen, tau), info


# Fraction of the bodies of EVERY checked evaluation that must pass by rule (a) alone — within 1e-4 of the fp32 oracle, the north star's
# tolerance — before rules (b) / (c) may carry the rest.  Measured on the GPU runs of round 5 (profiles/r05_arbiter.json: the counts of
# every arbiter-checked test) and pinned below that: a change that pushes more bodies onto the looser rules fails here.
MIN_RULE_A = {'default': 1.0,       # measured: every body of every test passes by rule (a) ...
              # ... but one of 64 in iteration 2 of the habitat sweep since the blend products run as fp16 split products (round 6): a body whose fp32
              # ORACLE is 5.1e-4 from the fp64 arbiter there (an sdf ~ 0 vertex on the other side of the mask) while the product is 2.0e-5 from it
              'configs4_habitat_64_fused': 0.98}
# Rule (a) carries a slack term (n_ambiguous / N_pen x scale: bodies without an ambiguous vertex see the sdf < 0 mask through the global count).
# The share of bodies within the LITERAL 1e-4 of the fp32 oracle — no slack — is reported per evaluation (`bodies_within_1e4_of_oracle32_no_slack`)
# and pinned here over a whole test (all its evaluations added up): measured on the GPU runs of round 6 (profiles/r06_arbiter.json), floor set
# just below the smallest share any test showed.
MIN_STRICT_A = {'default': 0.98}           # measured shares: 0.9896 .. 1.0 (one or two bodies of 136 / 137 / 192, each with an ambiguous vertex: sdf ~ 0 on the mask's edge)


def record(name, report, min_rule_a=None):
    """Keep the per-iteration summaries of an arbiter-checked test (which rule every body passed by, how much of each bound was used) as
    JSON under gpurun_out/arbiter/ (merged back from the GPU box; profiles/r05_arbiter.json is the committed copy) and hold the share of
    bodies that needed no more than rule (a) to its pinned floor."""
    import json
    import os
    rows = report if isinstance(report, list) else [report]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'arbiter')
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, name + '.json'), 'w') as f:
            json.dump(rows, f, indent=1, default=float)
    except OSError:
        pass
    floor = MIN_RULE_A.get(name, MIN_RULE_A['default']) if min_rule_a is None else min_rule_a
    for r in rows:
        by = r['bodies_by_rule']
        n = by['a'] + by['b_only'] + by['c_only']
        assert by['a'] >= floor * n, ('too many bodies needed the fp64-arbiter rules (b) / (c)', name, r.get('step'), by, floor)
    strict = sum(r.get('bodies_within_1e4_of_oracle32_no_slack', 0) for r in rows)
    total = sum(r.get('bodies', 0) for r in rows)
    if total:
        assert strict >= MIN_STRICT_A.get(name, MIN_STRICT_A['default']) * total, ('too few bodies within the literal 1e-4 of the fp32 oracle', name, strict, total)
    return rows


def check_trace(trace, make_oracle, cam, lr=0.1, first_step=1):
    """``make_oracle(dtype)`` -> FittingOracle on the GLOBAL batch; cam [B,4,4].  The fixed 6D target ``xhr`` of the reconstruction loss is the
    implementation's own starting point (the loop starts AT the target, fitting_proxe.py:171-175, where |xhr - x| has its kink: the target
    must be the very same fp32 numbers, or iteration 1 sees sign(+-1 ulp) instead of sign(0)).  Returns a per-iteration summary (which rule
    every body passed by, how much of the bounds was used) for the test's log."""
    assert first_step == 1, 'the trace must start at the target'
    xhr = trace[0]['x0']
    f32, f64 = make_oracle(torch.float32), make_oracle(torch.float64)
    report = []
    for i, t in enumerate(trace):
        step = first_step + i
        g_gpu = (t['m1'] - BETA1 * t['m0']) / (1 - BETA1)
        (l32, l64, amb, n_pen, tau), info = check_gradient(g_gpu, f32, f64, t['x0'], xhr, cam)
        # --- loss values (continuous in the parameters: no event rule needed; a vertex counted the other way moves the penetration MEAN by
        # about mean / N: the count changes by one, the sum by < tau)
        l_bound = K_NOISE * np.abs(l32 - l64) + 3e-6 * np.maximum(np.abs(l64), 1e-2)
        l_bound[3] += amb.sum() * 1.5 * (np.abs(l64[3]) + tau) / n_pen
        assert np.all(np.abs(t['losses'] - l64) <= l_bound), (step, t['losses'], l64, l32, l_bound)
        # --- Adam's update from the implementation's own state and gradient
        x_ref, m_ref, v_ref = _adam(t['x0'], t['m0'], t['v0'], g_gpu, step, lr)
        assert np.abs(t['m1'] - m_ref).max() <= 1e-6 * np.abs(m_ref).max(), (step, 'm')
        assert np.all(np.abs(t['v1'] - v_ref) <= 2e-6 * np.abs(v_ref) + 1e-37), (step, 'v')
        xerr = np.abs(t['x1'] - x_ref)
        x_tol = 2e-5 * np.abs(x_ref - t['x0']) + 5e-7 * np.maximum(np.abs(x_ref), 1.0)
        assert np.all(xerr <= x_tol), (step, float((xerr / x_tol).max()), np.unravel_index(np.argmax(xerr / x_tol), xerr.shape))
        report.append(dict(step=step, loss_err=float(np.abs(t['losses'] - l64).max()), oracle32_loss_err=float(np.abs(l32 - l64).max()),
                           x_update_err=float(xerr.max()), **info))
    return report
