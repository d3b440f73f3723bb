"""Pin the oracle's SE3 exp/log/Adj with the reference's own test data: the 9 group elements and
7 tangents of thirdparty/Sophus/sophus/test_se3.cpp:40-82 and the property checks of
thirdparty/Sophus/sophus/tests.hpp:43-200 (adjoint, exp(log), expmap vs hat, group action)
with its tolerance SMALL_EPS = 1e-10 (sophus.hpp:45-59)."""
import ctypes as C

import numpy as np
import scipy.linalg

from sdso_amd import abi

SMALL_EPS = 1e-10


def _so3(L, w):
    T = abi.SE3()
    xi = np.array([0, 0, 0, w[0], w[1], w[2]], np.float64)
    L.orc_se3_exp(abi.dp(xi), C.byref(T))
    return T.Rt()[0]


def _elems(L):
    pi = np.pi
    E = []
    E.append((_so3(L, (0.2, 0.5, 0.0)), np.array([0., 0, 0])))
    E.append((_so3(L, (0.2, 0.5, -1.0)), np.array([10., 0, 0])))
    E.append((_so3(L, (0., 0., 0.)), np.array([0., 100, 5])))
    E.append((_so3(L, (0., 0., 0.00001)), np.array([0., 0, 0])))
    E.append((_so3(L, (0., 0., 0.00001)), np.array([0., -0.00000001, 0.0000000001])))
    E.append((_so3(L, (0., 0., 0.00001)), np.array([0.01, 0, 0])))
    E.append((_so3(L, (pi, 0, 0)), np.array([4., -5, 0])))

    def mul(A, B):
        return A[0] @ B[0], A[0] @ B[1] + A[1]
    z = np.zeros(3)
    E.append(mul(mul((_so3(L, (0.2, 0.5, 0.0)), z), (_so3(L, (pi, 0, 0)), z)), (_so3(L, (-0.2, -0.5, -0.0)), z)))
    E.append(mul(mul((_so3(L, (0.3, 0.5, 0.1)), np.array([2., 0, -7])), (_so3(L, (pi, 0, 0)), z)),
                 (_so3(L, (-0.3, -0.5, -0.1)), np.array([0., 6, 0]))))
    return E


TANGENTS = [np.array(t, np.float64) for t in ([0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0],
                                               [-1, 1, 0, 0, 0, 1], [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0])]


def _mat(R, t):
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return M


def _hat(x):
    w = x[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = x[:3]
    return M


def _vee(M):
    return np.array([M[0, 3], M[1, 3], M[2, 3], M[2, 1], M[0, 2], M[1, 0]])


def test_adjoint(oracle):
    for R, t in _elems(oracle):
        T = abi.SE3.from_Rt(R, t)
        Ad = np.zeros(36)
        oracle.orc_se3_adj(C.byref(T), abi.dp(Ad))
        Ad = Ad.reshape(6, 6)
        M = _mat(R, t)
        for x in TANGENTS:
            ad2 = _vee(M @ _hat(x) @ np.linalg.inv(M))
            assert np.linalg.norm(Ad @ x - ad2) <= 20 * SMALL_EPS * max(1.0, np.linalg.norm(ad2))


def test_exp_log(oracle):
    for R, t in _elems(oracle):
        T = abi.SE3.from_Rt(R, t)
        xi = np.zeros(6)
        oracle.orc_se3_log(C.byref(T), abi.dp(xi))
        T2 = abi.SE3()
        oracle.orc_se3_exp(abi.dp(xi), C.byref(T2))
        R2, t2 = T2.Rt()
        assert np.linalg.norm(_mat(R, t) - _mat(R2, t2)) <= SMALL_EPS * max(1.0, np.linalg.norm(t))


def test_expmap_vs_hat(oracle):
    for x in TANGENTS:
        T = abi.SE3()
        oracle.orc_se3_exp(abi.dp(x.copy()), C.byref(T))
        ref = scipy.linalg.expm(_hat(x))
        assert np.linalg.norm(_mat(*T.Rt()) - ref) <= 10 * SMALL_EPS * max(1.0, np.linalg.norm(ref))


def test_mul_inv(oracle):
    E = _elems(oracle)
    for A in E:
        for B in E:
            TA, TB, TC = abi.SE3.from_Rt(*A), abi.SE3.from_Rt(*B), abi.SE3()
            oracle.orc_se3_mul(C.byref(TA), C.byref(TB), C.byref(TC))
            assert np.linalg.norm(_mat(*TC.Rt()) - _mat(*A) @ _mat(*B)) <= SMALL_EPS * 1e3
        TA, TI = abi.SE3.from_Rt(*A), abi.SE3()
        oracle.orc_se3_inv(C.byref(TA), C.byref(TI))
        assert np.linalg.norm(_mat(*TI.Rt()) @ _mat(*A) - np.eye(4)) <= 1e-9


def test_ldlt_and_inverse(oracle):
    rs = np.random.RandomState(5)
    for n in (6, 8, 68):
        A = rs.normal(size=(n, n))
        A = A @ A.T + np.diag(rs.uniform(0.1, 1e6, n))
        b = rs.normal(size=n)
        x = np.zeros(n)
        assert oracle.orc_ldlt_solve(n, abi.dp(np.ascontiguousarray(A)), abi.dp(b), abi.dp(x)) == 0
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    K = np.array([700.3, 0, 612.1, 0, 701.9, 180.4, 0, 0, 1], np.float32)
    Ki = np.zeros(9, np.float32)
    oracle.orc_mat3f_inv(abi.fp(K), abi.fp(Ki))
    assert np.allclose(Ki.reshape(3, 3) @ K.reshape(3, 3), np.eye(3), atol=1e-4)
