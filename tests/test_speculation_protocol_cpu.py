"""Executable specification of the reject-chain speculation of k_track_step (lsd_slam_amd/csrc/tracker.hip), on CPU.

The reference's LM loop (C/Tracking/SE3Tracker.cpp:316-447) is transcribed twice in Python over the ORACLE's residual
evaluation: once literally, and once as the launch protocol the kernel implements — a launch evaluates the next C trial poses of
the "increase lambda and retry" chain side by side (they depend only on A, b and lambda), the next launch finds, trial by trial
in parallel, the first one at which the loop stops (diverged / accepted / rejected with a step below stepSizeMin), advances
lambda and incTry past the plain rejections before it in closed form, and runs ONE LM step on that trial.  For every setting
(lambdaInitial, lambdaFailFac, lambdaSuccessFac, stepSizeMin, convergenceEps, iteration caps) and every C the two must make
the same decisions: same accepted poses, same lambda / error history, same number of evaluations the reference would have
executed, same "last executed evaluation" (whose statistics and refPixelWasGood the frame keeps).  The HIP code itself is
checked against the one-evaluation-per-launch run on the GPU (tests/test_gpu_parity.py); this file pins the protocol."""
import itertools

import numpy as np
import pytest

from common import sequence

F = np.float32
MIN_ABSMIN = 0.01   # MIN_GOODPERALL_PIXEL_ABSMIN


class Evaluator:
    """one residual evaluation (K1 + K2 + K3) at a pose through the oracle; counts calls"""

    def __init__(self, oracle, w, h):
        frames, depth0, K, gt = sequence(w, h, 4)
        self.o, self.w, self.h = oracle, w, h
        kf = oracle.Frame(0, frames[0], K)
        kf.set_depth_gt(depth0)
        self.ref = oracle.TrackingReference()
        self.ref.import_frame(kf)
        self.kf = kf
        self.tr = oracle.SE3Tracker(w, h, K, mode=oracle.SSE)
        self.frame = oracle.Frame(2, frames[2], K)
        self.calls = 0

    def __call__(self, T, lvl, a, b):
        self.calls += 1
        r = self.tr.evaluate(self.ref, self.frame, np.asarray(T, np.float32), lvl, float(a), float(b))
        return dict(err=F(r.weightedError), M=r.warped_size, A=np.array(r.A, np.float32).reshape(6, 6), b=np.array(r.b, np.float32),
                    a_lastIt=F(r.affine_a_lastIt), b_lastIt=F(r.affine_b_lastIt), usage=F(r.pointUsage))


def solve(A, b, lam):
    Ad = A.astype(np.float64).copy()
    for i in range(6):
        Ad[i, i] = F(Ad[i, i]) * F(1 + lam)
    return np.linalg.solve(Ad, -b.astype(np.float64)).astype(np.float32)


def lam_fail(lam, incTry, fail):
    if lam == 0:
        return F(0.2)
    return F(np.float64(lam) * np.float64(fail) ** incTry)


def step_pose(o, inc, T):
    return o.se3_mul(o.se3_exp(inc.astype(np.float64)), T)


def reference_level(ev, o, lvl, T, a, b, S):
    """SE3Tracker.cpp:316-447 for one level; returns the state and a log of what happened"""
    log = []
    e = ev(T, lvl, a, b)
    if e["M"] < MIN_ABSMIN * (ev.w >> lvl) * (ev.h >> lvl):
        return None, log + ["diverged"]
    a, b = e["a_lastIt"], e["b_lastIt"]
    lastErr, lam, last_exec, last_residual = e["err"], F(S["lambdaInitial"]), e, None
    iteration = 0
    while iteration < S["maxIts"]:
        A, bb = last_acc_A(e), e["b"]
        incTry = 0
        while True:
            inc = solve(A, bb, lam)
            incTry += 1
            Tn = step_pose(o, inc, T)
            en = ev(Tn, lvl, a, b)
            last_exec = en
            if en["M"] < MIN_ABSMIN * (ev.w >> lvl) * (ev.h >> lvl):
                return None, log + ["diverged"]
            if en["err"] < lastErr:
                T, e = Tn, en
                a, b = en["a_lastIt"], en["b_lastIt"]
                log.append(("A", float(lam), incTry))
                if en["err"] / lastErr > S["convergenceEps"]:
                    iteration = S["maxIts"]
                last_residual = lastErr = en["err"]
                lam = F(0) if lam <= F(0.2) else F(lam * F(S["lambdaSuccessFac"]))
                break
            log.append(("R", float(lam), incTry))
            if not (float(np.dot(inc, inc)) > S["stepSizeMin"]):
                iteration = S["maxIts"]
                break
            lam = lam_fail(lam, incTry, S["lambdaFailFac"])
        iteration += 1
    return dict(T=T, a=a, b=b, lastErr=lastErr, last_residual=last_residual, last_exec=last_exec), log


def last_acc_A(e):
    return e["A"]


def speculative_level(ev, o, lvl, T, a, b, S, C):
    """the launch protocol: every `launch` evaluates a list of trial poses; the next one consumes them"""
    log, launches = [], 0
    e = ev(T, lvl, a, b)          # launch: the level's first evaluation (one trial)
    launches += 1
    if e["M"] < MIN_ABSMIN * (ev.w >> lvl) * (ev.h >> lvl):
        return None, log + ["diverged"], launches
    a, b = e["a_lastIt"], e["b_lastIt"]
    lastErr, lam, last_exec, last_residual = e["err"], F(S["lambdaInitial"]), e, None
    iteration, incTry = 0, 0
    propose = iteration < S["maxIts"]
    A, bb = e["A"], e["b"]
    while propose:
        # ---- a launch: trial c uses lambda advanced c times past the proposal (closed form), its own solve, its own pose
        trials, l, it = [], lam, incTry
        for c in range(C):
            inc = solve(A, bb, l)
            it += 1
            Tn = step_pose(o, inc, T)
            trials.append(dict(lam=l, incTry=it, inc=inc, T=Tn, e=ev(Tn, lvl, a, b)))
            l = lam_fail(l, it, S["lambdaFailFac"])
        launches += 1
        # ---- the next launch's finishing half: which trial stops the loop?
        minW = MIN_ABSMIN * (ev.w >> lvl) * (ev.h >> lvl)
        stop = [t["e"]["M"] < minW or t["e"]["err"] < lastErr or not (float(np.dot(t["inc"], t["inc"])) > S["stepSizeMin"]) for t in trials]
        pc = stop.index(True) if any(stop) else C - 1
        for t in trials[:pc]:
            log.append(("R", float(t["lam"]), t["incTry"]))       # plain rejections: nothing kept but lambda / incTry / counters
        t = trials[pc]
        lam, incTry, last_exec = t["lam"], t["incTry"], t["e"]
        # ---- ONE LM step on trial pc (lm_wave)
        if t["e"]["M"] < minW:
            return None, log + ["diverged"], launches
        if t["e"]["err"] < lastErr:
            T, e = t["T"], t["e"]
            a, b = e["a_lastIt"], e["b_lastIt"]
            A, bb = e["A"], e["b"]
            log.append(("A", float(lam), incTry))
            if e["err"] / lastErr > S["convergenceEps"]:
                iteration = S["maxIts"]
            last_residual = lastErr = e["err"]
            lam = F(0) if lam <= F(0.2) else F(lam * F(S["lambdaSuccessFac"]))
            iteration += 1
            incTry = 0
            propose = iteration < S["maxIts"]
        else:
            log.append(("R", float(lam), incTry))
            if not (float(np.dot(t["inc"], t["inc"])) > S["stepSizeMin"]):
                iteration = S["maxIts"] + 1
                propose = False
            else:
                lam = lam_fail(lam, incTry, S["lambdaFailFac"])   # the chain goes on: the next launch starts at this lambda
                propose = True
    return dict(T=T, a=a, b=b, lastErr=lastErr, last_residual=last_residual, last_exec=last_exec), log, launches


SETTINGS = [dict(lambdaInitial=li, lambdaFailFac=ff, lambdaSuccessFac=sf, stepSizeMin=sm, convergenceEps=ce, maxIts=mi)
            for li, ff, sf, sm, ce, mi in [(0, 2, 0.5, 1e-8, 0.999, 20), (0, 2, 0.5, 1e-8, 0.999, 1), (0, 2, 0.5, 1e-8, 0.999, 3),
                                           (0.5, 2, 0.5, 1e-8, 0.999, 20), (5, 3, 0.25, 1e-8, 0.99, 20), (0, 3, 0.5, 1e-5, 0.999, 20),
                                           (0.3, 2, 0.5, 1e-6, 0.9999, 6)]]


@pytest.mark.parametrize("si", range(len(SETTINGS)))
def test_launch_protocol_makes_the_reference_decisions(oracle, si):
    S = SETTINGS[si]
    w, h = 176, 144
    ev = Evaluator(oracle, w, h)
    o = oracle
    starts = [np.array([1.0, 0, 0, 0, 0, 0, 0]), o.se3_exp(np.array([0.01, -0.004, 0.003, 0.002, -0.003, 0.004]))]
    saved = 0
    for T0, lvl in itertools.product(starts, (3, 2, 1)):
        st_ref, log_ref = reference_level(ev, o, lvl, T0.copy(), F(1), F(0), S)
        n_ref = len([x for x in log_ref if x != "diverged"]) + 1
        for C in (1, 2, 3, 5, 6):
            st, log, launches = speculative_level(ev, o, lvl, T0.copy(), F(1), F(0), S, C)
            assert log == log_ref, (S, lvl, C, log, log_ref)              # same accept / reject sequence at the same lambdas
            if st_ref is None:
                assert st is None
                continue
            assert np.array_equal(st["T"], st_ref["T"]) and st["a"] == st_ref["a"] and st["b"] == st_ref["b"]
            assert st["lastErr"] == st_ref["lastErr"] and st["last_residual"] == st_ref["last_residual"]
            # the evaluation whose statistics / refPixelWasGood survive is the one the reference executed last
            assert st["last_exec"]["err"] == st_ref["last_exec"]["err"] and st["last_exec"]["usage"] == st_ref["last_exec"]["usage"]
            assert launches <= n_ref
            if C == 1:
                assert launches == n_ref
            saved += n_ref - launches
    assert saved > 0 or S["maxIts"] == 1
