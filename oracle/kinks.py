"""TEST INFRASTRUCTURE (oracle side; never imported by the product path): ReLU-kink bracketing for the strict
end-to-end gradient comparisons.

Why.  The reference network has ~120 ReLU layers (`nn.ReLU` after norm_a, after the residual add, in the stem and the
decoders: reference model/x3d.py:176-183, 227-231, 94-106; model/change_decoder.py:30-55).  A pre-activation that lies
within f32 rounding noise of zero takes either side depending on the summation order of whoever computes it -- torch-CPU
itself changes sides with its thread count -- and ONE flipped unit moves the ~100 gradient tensors upstream of it by
1e-4 .. 1e-3.  Comparing an f32 implementation with the f32 oracle to 1e-4 was therefore only possible on weight seeds that
happened to have no such unit under both parties' exact roundings, and every change of a summation order in a kernel
(a bank-conflict fix that re-associates a wave sum) had to be vetoed or the seeds re-scanned.

What this module does instead (all of it on the oracle, in float64 unless stated):
  1. `probe()` runs the oracle once in f64 with forward hooks on every `nn.ReLU` call and records the pre-activations;
     the f32 oracle runs the test does anyway (one per thread count) record theirs: sigma_L = rms(pre_f32 - pre_f64) is the
     f32 noise of ReLU call L as torch-CPU itself exhibits it.
  2. A unit is AT RISK when |pre_f64| < k_sigma * sigma_L (default 6): f32 noise can put it on either side.
  3. For each at-risk unit one more f64 run with THAT unit forced to the other side (on -> off: output 0, gradient 0;
     off -> on: output = pre, gradient 1) gives delta_i = what that one flip does to every gradient tensor (it moves only
     tensors upstream of its layer, in a pattern fixed by the network).
  4. `explain_flips` explains (implementation - f32 oracle) with the deltas by matching pursuit (a flip has coefficient 1);
     the units it picks are the ones the implementation took on the other side; they are printed (the "exclusion list" is a list of UNITS, not of
     tensors) and their deltas added to the oracle.  EVERY tensor must then meet the strict bound.

The check stays two-sided: an implementation that flips a unit which is NOT at risk (|pre| >= k_sigma sigma) fails exactly
as before, and so does any error that is not a sum of at-risk deltas -- a genuine defect is not of that form.  `bracket_widths`
(all at-risk units flipped at once) is kept as a cheap diagnostic of how far the at-risk units can move each tensor at all."""
import contextlib

import torch
from torch import nn


class _Recorder:
    """Forward hooks on every nn.ReLU call of `model`: mode 'record' keeps the pre-activations (as float64, on the CPU);
    mode 'flip' replaces the output of flagged units by the other branch."""

    def __init__(self, model):
        self.mods = [m for m in model.modules() if isinstance(m, nn.ReLU)]
        self.pre, self.flags, self.mode, self.k = [], None, "record", 0
        # parameter name -> number of ReLU calls made before the FIRST forward of the module that owns it: a flipped unit of
        # ReLU call c can only move parameters with position <= c (they are upstream of it)
        self.owners = [(m, [f"{mn}.{pn}" if mn else pn for pn, _ in m.named_parameters(recurse=False)])
                       for mn, m in model.named_modules() if any(True for _ in m.named_parameters(recurse=False))]
        self.param_pos = {}

    def _owner_hook(self, names):
        def hook(mod, inp):
            for n in names:
                self.param_pos.setdefault(n, len(self.pre))
        return hook

    def _hook(self, mod, inp, out):
        x = inp[0]
        if self.mode == "record":
            self.pre.append(x.detach().double().clone())
            return None
        f = self.flags[self.k]
        self.k += 1
        if f is None or not bool(f.any()):
            return None
        # other branch: on (x > 0) -> 0, off (x <= 0) -> x  ==  x - relu(x)
        return torch.where(f, x - out, out)

    @contextlib.contextmanager
    def attached(self, mode, flags=None):
        self.mode, self.flags, self.k, self.pre = mode, flags, 0, []
        hs = [m.register_forward_hook(self._hook) for m in self.mods]
        if mode == "record":
            self.param_pos = {}
            hs += [m.register_forward_pre_hook(self._owner_hook(names)) for m, names in self.owners]
        try:
            yield self
        finally:
            for h in hs:
                h.remove()


def record(model, run):
    """run() -> anything, executed with recording hooks on `model`; returns (run's result, [pre-activation per ReLU call])."""
    rec = _Recorder(model)
    for m in rec.mods:
        assert not getattr(m, "inplace", False), "in-place ReLU: the hook would see the output"
    with rec.attached("record"):
        out = run()
    record.param_pos = dict(rec.param_pos)     # (of the latest recording: see explain_flips' `param_pos`)
    return out, rec.pre


def at_risk(pre64, pre32_runs, k_sigma=6.0):
    """Flags per ReLU call: |pre_f64| < k_sigma * sigma_L, sigma_L = the largest rms(pre_f32 - pre_f64) over the given f32
    runs, floored at one f32 ulp of the call's rms value.  Returns (flags, sigmas)."""
    flags, sigmas = [], []
    for i, p in enumerate(pre64):
        rms = float(p.pow(2).mean().sqrt())
        sig = max([float((q[i] - p).pow(2).mean().sqrt()) for q in pre32_runs] + [rms * 2.0 ** -23, 1e-300])
        f = p.abs() < k_sigma * sig
        # exact zeros on both sides (padding, dead channels: pre == 0 in f64 AND in every f32 run) carry no gradient either way
        dead = (p == 0)
        for q in pre32_runs:
            dead &= (q[i] == 0)
        f &= ~dead
        flags.append(f)
        sigmas.append(sig)
    return flags, sigmas


def flipped_grads(model, run_backward, flags):
    """run_backward() must zero the gradients, run forward + backward on `model`; executed with the flagged units forced to the
    other branch.  Returns {name: grad clone}."""
    rec = _Recorder(model)
    with rec.attached("flip", flags):
        run_backward()
    assert rec.k == len(flags), (rec.k, len(flags))
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).norm().item() / (b.norm().item() + 1e-300)


def bracket_widths(g64, g64_flipped):
    return {n: rel_l2(g64_flipped[n], g64[n]) for n in g64}


def unit_list(flags, pre64, sigmas, limit=None):
    """[(call index, flat element index, pre_f64, |pre_f64| / sigma)] of the flagged units, the ones closest to zero (in units of
    their call's f32 noise) first; at most `limit`."""
    units = []
    for i, f in enumerate(flags):
        idx = f.reshape(-1).nonzero().reshape(-1).tolist()
        flat = pre64[i].reshape(-1)
        units += [(i, j, float(flat[j]), abs(float(flat[j])) / sigmas[i]) for j in idx]
    units.sort(key=lambda u: u[3])
    return units[:limit] if limit else units


def single_unit_flags(flags, call, elem):
    out = [None] * len(flags)
    f = torch.zeros_like(flags[call]).reshape(-1)
    f[elem] = True
    out[call] = f.reshape(flags[call].shape)
    return out


def explain_flips(g_impl, g_base, units, delta_of, tol=1e-4, log=print, max_flips=8, param_pos=None, tries_per_flip=12):
    """Which at-risk units did the implementation take on the other side?  g_impl / g_base: {name: gradient} of the
    implementation under test and of the f32 oracle; delta_of(unit) -> {name: g64_flipped - g64} (one f64 run per unit, so
    they are evaluated lazily); units sorted by closeness to zero.  Matching pursuit with the network's structure as a
    guide: a flip of ReLU call c moves only parameters upstream of it (`param_pos[name] <= c`), so the DEEPEST tensor that
    misses `tol` bounds the call index of some flip from below; the candidates at or beyond it are tried closest-to-zero first
    and the first one whose delta (coefficient exactly 1) halves the residual of the tensors it moves at or beyond that
    position is granted; repeat.  A genuine implementation error is not a sum of such deltas and stays.
    Returns (errs after granting, granted units)."""
    names = list(g_impl)
    nrm = {n: g_base[n].detach().double().cpu().norm().item() + 1e-300 for n in names}
    cur = {n: g_base[n].detach().double().cpu().clone() for n in names}
    gi = {n: g_impl[n].detach().double().cpu() for n in names}
    pos = param_pos or {}

    def errs_of(c):
        return {n: (gi[n] - c[n]).norm().item() / nrm[n] for n in names}

    errs = errs_of(cur)
    granted, tried, evals = [], set(), 0
    while len(granted) < max_flips:
        bad = [n for n in names if errs[n] >= tol]
        if not bad:
            break
        deepest = max(pos.get(n, 0) for n in bad)
        # the parameters that feed ReLU call c sit at position c exactly: a flip most likely belongs to call `deepest` itself
        cands = sorted([u for u in units if u[0] >= deepest and u[:2] not in tried], key=lambda u: (u[0] != deepest, u[3]))[:tries_per_flip]
        hit = None
        for u in cands:
            tried.add(u[:2])
            d = delta_of(u)
            evals += 1
            # judged on the tensors it moves AT OR BEYOND the deepest bad position: only flips of calls >= that position reach
            # them, so another (shallower, possibly larger) flip still in the residual cannot mask this one
            moved = [n for n in names if d[n].norm().item() / nrm[n] > 0.2 * tol and pos.get(n, 0) >= deepest]
            if not moved:
                continue
            before = sum(errs[n] ** 2 for n in moved)
            after = sum(((gi[n] - cur[n] - d[n]).norm().item() / nrm[n]) ** 2 for n in moved)
            if after < 0.5 * before:
                hit = (u, d)
                break
        if hit is None:
            break
        u, d = hit
        for n in names:
            cur[n] += d[n]
        n_bad = len(bad)
        errs = errs_of(cur)
        granted.append(u)
        log(f"  kink: ReLU call {u[0]} element {u[1]} (pre_f64 {u[2]:+.3e} = {u[3]:.2f} sigma) taken on the other side: "
            f"{n_bad} -> {sum(1 for e in errs.values() if e >= tol)} tensors above {tol:g}  [{evals} flip evaluations so far]")
    return errs, granted


def strict_compare(g_impl, make_run, tol=1e-4, threads=(1, 4), k_sigma=6.0, check_outputs=None, log=print):
    """The whole procedure of this module's header.  ONLY units within k_sigma of zero can be granted (the units between
    k_sigma and 2 k_sigma are listed in the log as a diagnostic: a flip out there is an implementation error, not noise).  g_impl: {name: gradient tensor (CPU)} of the implementation under test;
    make_run(dtype) -> (oracle model in that dtype, run) with run() = zero the gradients, forward, backward, return the outputs.
    The f32 oracle is evaluated once per entry of `threads` (torch-CPU's rounding depends on it); check_outputs(outputs) may
    assert on each evaluation's forward results.  Returns ({name: rel-L2 after granting}, [granted units])."""
    grads = lambda m: {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    threads0 = torch.get_num_threads()
    runs = []
    try:
        for thr in threads:
            torch.set_num_threads(thr)
            ref, run = make_run(torch.float32)
            outs, pre32 = record(ref, run)
            if check_outputs is not None:
                check_outputs(outs)
            g32 = grads(ref)
            runs.append((g32, pre32, {n: rel_l2(g_impl[n], g32[n]) for n in g_impl}))
    finally:
        torch.set_num_threads(threads0)
    errs = {n: min(r[2][n] for r in runs) for n in g_impl}
    if max(errs.values()) < tol:
        return errs, []
    ref64, run64 = make_run(torch.float64)
    _, pre64 = record(ref64, run64)
    g64 = grads(ref64)
    base_run = min(runs, key=lambda r: sum(e * e for e in r[2].values()))      # the f32 evaluation closer to the implementation
    base, base_pre = base_run[0], base_run[1]
    param_pos = dict(record.param_pos)
    flags, sigmas = at_risk(pre64, [r[1] for r in runs], 2.0 * k_sigma)
    wide = unit_list(flags, pre64, sigmas)
    units = [u for u in wide if u[3] < k_sigma]       # the grantable ones; `wide` is printed, never granted
    log(f"  {len(units)} ReLU units within {k_sigma:g} sigma of zero in the f64 oracle ({len(wide)} within "
        f"{2 * k_sigma:g}: diagnostic only); the closest (call, element, sigmas): {[(u[0], u[1], round(u[3], 2)) for u in wide[:8]]}")

    def delta_of(u):
        # what moving this unit from the side the BASE (f32 oracle) has it on to the other side does to every gradient.  The
        # f64 run supplies the magnitude; where the f32 oracle itself already sits on the other side than the f64 oracle (it is
        # at risk, after all), "the other side" of the base is the f64 side: the delta enters with the opposite sign.
        gf = flipped_grads(ref64, run64, single_unit_flags(flags, u[0], u[1]))
        same = bool(base_pre[u[0]].reshape(-1)[u[1]] > 0) == bool(pre64[u[0]].reshape(-1)[u[1]] > 0)
        sgn = 1.0 if same else -1.0
        return {n: sgn * (gf[n] - g64[n]).double() for n in g_impl}

    return explain_flips(g_impl, {n: base[n] for n in g_impl}, units, delta_of, tol=tol, log=log, param_pos=param_pos)
