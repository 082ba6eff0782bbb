"""Host-side arithmetic of the product's Sim3 LM step (liblsdhip.so, pure CPU entry points) against the oracle and numpy:
this part of the product runs without a GPU, so the CPU suite covers it directly."""
import ctypes as C

import numpy as np
import pytest

TANGENTS = [[0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0, 0.1], [0, 0, 1, 0, 1, 0, 0.1],
            [-1, 1, 0, 0, 0, 1, -0.1], [20, -1, 0, -1, 1, 0, -0.1], [30, 5, -1, 20, -1, 0, 1.5]]   # Sophus test_sim3.cpp:72-85


@pytest.fixture(scope="module")
def L():
    from lsd_slam_amd import capi
    return capi.lib()


def step(L, a, T):
    a = np.ascontiguousarray(a, np.float64)
    T = np.ascontiguousarray(T, np.float64)
    out = np.zeros(8)
    assert L.lsdhip_host_sim3_step(a.ctypes.data, T.ctypes.data, out.ctypes.data) == 0
    return out


def test_sim3_exp_matches_the_oracle(L, oracle):
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0, 1.0])
    rng = np.random.default_rng(3)
    for a in TANGENTS + [rng.normal(0, 0.3, 7) for _ in range(20)] + [[1e-12, 0, 0, 1e-11, 0, 0, 1e-12], [0.1, 0.2, 0.3, 0, 0, 0, 0.05]]:
        a = np.array(a, np.float64)
        got, want = step(L, a, ident), oracle.sim3_exp(a)
        assert np.allclose(got, want, rtol=1e-13, atol=1e-13), a


def test_sim3_left_multiplication(L, oracle):
    rng = np.random.default_rng(4)
    for _ in range(10):
        a, b = rng.normal(0, 0.2, 7), rng.normal(0, 0.4, 7)
        T = oracle.sim3_exp(b)
        got = step(L, a, T)
        E = oracle.sim3_exp(a)
        Re, Rt = oracle.quat_to_rot(E[:4]), oracle.quat_to_rot(T[:4])
        assert np.allclose(oracle.quat_to_rot(got[:4]), Re @ Rt, atol=1e-12)
        assert np.allclose(got[4:7], E[7] * (Re @ T[4:7]) + E[4:7], atol=1e-12)
        assert got[7] == pytest.approx(E[7] * T[7], rel=1e-14)
        assert np.linalg.norm(got[:4]) == pytest.approx(1.0, abs=1e-14)          # re-normalised after the product


def test_ldlt7_solves_damped_normal_equations(L):
    rng = np.random.default_rng(5)
    for k in range(20):
        J = rng.normal(size=(40, 7)) * rng.uniform(0.01, 100.0, 7)               # badly scaled columns: pivoting matters
        A = (J.T @ J).astype(np.float32)
        A[np.diag_indices(7)] *= 1.2
        b = rng.normal(size=7).astype(np.float32)
        x = np.zeros(7, np.float32)
        A_ = np.ascontiguousarray(A)
        assert L.lsdhip_host_ldlt7(A_.ctypes.data, b.ctypes.data, x.ctypes.data) == 0
        want = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        res = A.astype(np.float64) @ x - b
        assert np.linalg.norm(res) <= 1e-3 * max(1.0, np.linalg.norm(b)), k
        assert np.allclose(x, want, rtol=5e-2, atol=1e-3 * np.abs(want).max()), k
    # singular direction: zero row / column is ignored instead of producing NaN (the d != 0 guards)
    A = np.diag([1, 2, 0, 4, 5, 6, 7]).astype(np.float32)
    b = np.arange(1, 8, dtype=np.float32)
    x = np.zeros(7, np.float32)
    assert L.lsdhip_host_ldlt7(A.ctypes.data, b.ctypes.data, x.ctypes.data) == 0
    assert np.all(np.isfinite(x)) and x[2] == 0 and x[0] == pytest.approx(1.0) and x[6] == pytest.approx(1.0)


def test_se3f_exp_and_ldlt6_match_the_oracle_bit_for_bit(L, oracle):
    """pose_math.hpp (what the device LM runs) compiled for the host, against the oracle's Sophus SE3f restatement: same
    operation order, contraction off on both sides"""
    OL = oracle.lib()
    rng = np.random.default_rng(6)
    ident = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)
    for k in range(40):
        a = (rng.normal(0, 0.05 if k % 2 else 1.0, 6)).astype(np.float32)
        if k == 0:
            a[3:] = 0                                    # small-angle branch
        got, want = np.zeros(7, np.float32), np.zeros(7, np.float32)
        assert L.lsdhip_host_se3f_step(a.ctypes.data, ident.ctypes.data, got.ctypes.data) == 0
        OL.orc_se3_exp_f(a, want)
        assert np.allclose(got, want, rtol=0, atol=2e-7), (k, got, want)   # sinf/cosf of two C libraries may differ in the last bit
    for k in range(30):
        J = rng.normal(size=(50, 6)) * rng.uniform(0.01, 50.0, 6)
        A = np.ascontiguousarray((J.T @ J).astype(np.float32))
        b = rng.normal(size=6).astype(np.float32)
        x, xo = np.zeros(6, np.float32), np.zeros(6, np.float32)
        assert L.lsdhip_host_ldlt6(A.ctypes.data, b.ctypes.data, x.ctypes.data) == 0
        OL.orc_ldlt6_solve(A.ravel(), b, xo)
        assert np.array_equal(x, xo), k                                     # pure +,-,*,/ in a fixed order: identical bits
