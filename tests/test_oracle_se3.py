"""Pin the oracle's (and, on the GPU box, the device's) SE3 / SO3 exp/log/Adj with the reference's own test data: the 9 group elements and
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


# ---------------------------------------------------------------------------------------------------------------------------
# SO3: the 9 group elements and 7 tangents of thirdparty/Sophus/sophus/test_so3.cpp:40-62 through the oracle's SO3 code (SE3 with t = 0)
def _quat_R(w, x, y, z):
    q = np.array([w, x, y, z], np.float64)
    w, x, y, z = q / np.linalg.norm(q)                       # SO3Group(Quaternion) normalises (so3.hpp:118-130)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _so3_elems(L):
    pi = np.pi
    return [_quat_R(0.1e-11, 0., 1., 0.), _quat_R(-1, 0.00001, 0.0, 0.0), _so3(L, (0.2, 0.5, 0.0)), _so3(L, (0.2, 0.5, -1.0)), _so3(L, (0., 0., 0.)),
            _so3(L, (0., 0., 0.00001)), _so3(L, (pi, 0, 0)),
            _so3(L, (0.2, 0.5, 0.0)) @ _so3(L, (pi, 0, 0)) @ _so3(L, (-0.2, -0.5, -0.0)),
            _so3(L, (0.3, 0.5, 0.1)) @ _so3(L, (pi, 0, 0)) @ _so3(L, (-0.3, -0.5, -0.1))]


SO3_TANGENTS = [np.array(t, np.float64) for t in ([0, 0, 0], [1, 0, 0], [0, 1, 0], [np.pi / 2, np.pi / 2, 0.0], [-1, 1, 0], [20, -1, 0], [30, 5, -1])]
POINT = np.array([1., 2, 4])                                  # test_so3.cpp:64, test_se3.cpp:84


def _hat3(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)


def test_so3_exp_log_and_expmap(oracle):
    """tests.hpp expLogTest / expMapTest on the SO3 fixtures"""
    for R in _so3_elems(oracle):
        T = abi.SE3.from_Rt(R, np.zeros(3))
        xi = np.zeros(6)
        oracle.orc_se3_log(C.byref(T), abi.dp(xi))
        assert np.all(xi[:3] == 0)
        R2 = _so3(oracle, xi[3:])
        assert np.linalg.norm(R - R2) <= SMALL_EPS
    for w in SO3_TANGENTS:
        assert np.linalg.norm(_so3(oracle, w) - scipy.linalg.expm(_hat3(w))) <= 10 * SMALL_EPS


def test_so3_adjoint_and_group_action(oracle):
    """tests.hpp adjointTest (Ad of a rotation is the rotation itself: both diagonal blocks of the SE3 adjoint, se3.hpp:131-145) and
    groupActionTest (R * p against the matrix product)"""
    for R in _so3_elems(oracle):
        T = abi.SE3.from_Rt(R, np.zeros(3))
        Ad = np.zeros(36)
        oracle.orc_se3_adj(C.byref(T), abi.dp(Ad))
        Ad = Ad.reshape(6, 6)
        assert np.linalg.norm(Ad[:3, :3] - R) <= SMALL_EPS and np.linalg.norm(Ad[3:, 3:] - R) <= SMALL_EPS and not Ad[3:, :3].any() and np.abs(Ad[:3, 3:]).max() <= SMALL_EPS
        for x in SO3_TANGENTS:
            ad2 = R @ _hat3(x) @ R.T                          # vee(T hat(x) T^-1)
            assert np.linalg.norm(Ad[3:, 3:] @ x - np.array([ad2[2, 1], ad2[0, 2], ad2[1, 0]])) <= 20 * SMALL_EPS * max(1.0, np.linalg.norm(x))
        # group action through the SE3 product: T * (I, p) has translation R p
        Tp, Tq = abi.SE3.from_Rt(np.eye(3), POINT), abi.SE3()
        oracle.orc_se3_mul(C.byref(T), C.byref(Tp), C.byref(Tq))
        assert np.linalg.norm(Tq.Rt()[1] - R @ POINT) <= SMALL_EPS


def test_se3_group_action(oracle):
    """tests.hpp groupActionTest: T * p against map(T.matrix(), p) for the SE3 fixtures and the point of test_se3.cpp:84"""
    for R, t in _elems(oracle):
        T, Tp, Tq = abi.SE3.from_Rt(R, t), abi.SE3.from_Rt(np.eye(3), POINT), abi.SE3()
        oracle.orc_se3_mul(C.byref(T), C.byref(Tp), C.byref(Tq))
        ref = (_mat(R, t) @ np.append(POINT, 1.0))[:3]
        assert np.linalg.norm(Tq.Rt()[1] - ref) <= SMALL_EPS * max(1.0, np.linalg.norm(ref))


def _ad(x):
    """ad(x) b = [x, b] = vee(hat(x) hat(b) - hat(b) hat(x)) as a 6x6 matrix (SE3Group::lieBracket, se3.hpp:560-600)"""
    A = np.zeros((6, 6))
    for j in range(6):
        e = np.zeros(6); e[j] = 1
        A[:, j] = _vee(_hat(x) @ _hat(e) - _hat(e) @ _hat(x))
    return A


def test_vee_hat_and_lie_bracket(oracle):
    """tests.hpp veeHatTest / lieBracketTest tie hat, vee and the bracket together; the oracle has none of the three as functions (DSO
    never calls them) but its Adj and exp must be consistent with them: vee(hat(x)) = x, and Ad(exp(x)) = expm(ad(x)) with ad built
    from the bracket — for every pair of test tangents whose rotation part keeps expm well conditioned."""
    for x in TANGENTS:
        assert np.array_equal(_vee(_hat(x)), x)
    for x in TANGENTS:
        if np.linalg.norm(x[3:]) > 4:                         # (rotations of 20 rad: expm of the 6x6 ad loses digits, not the oracle)
            continue
        T = abi.SE3()
        oracle.orc_se3_exp(abi.dp(x.copy()), C.byref(T))
        Ad = np.zeros(36)
        oracle.orc_se3_adj(C.byref(T), abi.dp(Ad))
        ref = scipy.linalg.expm(_ad(x))
        assert np.linalg.norm(Ad.reshape(6, 6) - ref) <= 1e3 * SMALL_EPS * max(1.0, np.linalg.norm(ref))
        for y in TANGENTS:                                    # [x, y] = d/dt Ad(exp(t x)) y at 0, by central differences of the oracle's own functions
            h = 1e-6
            d = []
            for sgn in (+1, -1):
                Th = abi.SE3()
                oracle.orc_se3_exp(abi.dp(sgn * h * x), C.byref(Th))
                A = np.zeros(36)
                oracle.orc_se3_adj(C.byref(Th), abi.dp(A))
                d.append(A.reshape(6, 6) @ y)
            assert np.linalg.norm((d[0] - d[1]) / (2 * h) - _ad(x) @ y) <= 1e-6 * max(1.0, np.linalg.norm(x) * np.linalg.norm(y))


# ---------------------------------------------------------------------------------------------------------------------------
# the same fixtures through the DEVICE's SE3 code (host_math.h compiled for gfx950: what k_track_lm / opt_step_body run)
import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_se3_matches_the_sophus_fixtures(gpu_ctx, oracle):
    """exp / inverse / product on the device for the 7 SE3 tangents of test_se3.cpp:74-82 plus the logs of its 9 group elements:
    expmap-vs-hat, G = exp(log(G)), T T^-1 = I and the product against the matrix product, with tests.hpp's tolerances; and the device
    agrees with the oracle's exp to 1e-13 (the two are separate restatements of se3.hpp:406-428 / so3.hpp:343-370)."""
    xis = [x.copy() for x in TANGENTS]
    for R, t in _elems(oracle):
        xi = np.zeros(6)
        oracle.orc_se3_log(C.byref(abi.SE3.from_Rt(R, t)), abi.dp(xi))
        xis.append(xi)
    xi = np.ascontiguousarray(np.array(xis))
    n = len(xis)
    Te, Ti, Tm = np.zeros((n, 12)), np.zeros((n, 12)), np.zeros((n, 12))
    gpu_ctx.check(gpu_ctx.L.sdso_selftest_se3(gpu_ctx.h, n, abi.dp(xi), abi.dp(Te), abi.dp(Ti), abi.dp(Tm)))
    M = [_mat(Te[i, :9].reshape(3, 3), Te[i, 9:]) for i in range(n)]
    for i in range(n):
        ref = scipy.linalg.expm(_hat(xi[i]))
        assert np.linalg.norm(M[i] - ref) <= 10 * SMALL_EPS * max(1.0, np.linalg.norm(ref))                      # expMapTest
        To = abi.SE3()
        oracle.orc_se3_exp(abi.dp(xi[i].copy()), C.byref(To))
        assert np.abs(M[i] - _mat(*To.Rt())).max() <= 1e-13 * max(1.0, np.abs(M[i]).max())                       # device == oracle
        Mi = _mat(Ti[i, :9].reshape(3, 3), Ti[i, 9:])
        assert np.linalg.norm(Mi @ M[i] - np.eye(4)) <= 1e-9 * max(1.0, np.linalg.norm(M[i]))
        Mm = _mat(Tm[i, :9].reshape(3, 3), Tm[i, 9:])
        assert np.linalg.norm(Mm - M[i] @ M[(i + 1) % n]) <= SMALL_EPS * 1e3 * max(1.0, np.linalg.norm(Mm))      # mapAndMultTest
    for k, (R, t) in enumerate(_elems(oracle)):                                                                  # expLogTest through the device exp
        assert np.linalg.norm(_mat(R, t) - M[len(TANGENTS) + k]) <= SMALL_EPS * max(1.0, np.linalg.norm(t))
