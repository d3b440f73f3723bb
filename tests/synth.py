"""Deterministic synthetic inputs of KITTI shape for the tracker / windowed-BA / static-stereo
hot path.  Pure numpy; used by tests, bench.py and __graft_entry__.smoke().

The scene is analytic (ground plane, two side walls, a far wall) with a procedural sinusoid
texture evaluated at continuous surface coordinates, so every camera of a trajectory — and the
right camera of the stereo rig — sees a photo-consistent image and the inverse depth of every
pixel is known in closed form.

Reference rules restated here (generation only, all checked against the oracle in tests/):
  pyramid + gradients   src/FullSystem/HessianBlocks.cpp:141-203   (make_pyramid)
  pyramid level count   src/util/globalCalib.cpp:52-58             (pyramid_levels)
  per-level intrinsics  src/util/globalCalib.cpp:90-107            (level_intrinsics)
  pc_* template build   src/FullSystem/CoarseTracker.cpp:360-534   (make_pc, STEP2-5)
"""
import numpy as np

PATTERN = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], dtype=np.int32)
SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_F, SCALE_C, SCALE_A, SCALE_B = 1.0, 0.5, 50.0, 50.0, 10.0, 1000.0


# ------------------------------------------------------------------ SE3 helpers (double)
def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def se3_exp(xi):
    """Sophus ordering: xi = [upsilon(3), omega(3)] -> (R, t)."""
    xi = np.asarray(xi, dtype=np.float64)
    u, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = hat(w)
    if th < 1e-10:
        R = np.eye(3) + W + 0.5 * W @ W
        V = np.eye(3) + 0.5 * W
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th**2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * W @ W
    return R, V @ u


def se3_mul(A, B):
    return A[0] @ B[0], A[0] @ B[1] + A[1]


def se3_inv(A):
    return A[0].T, -A[0].T @ A[1]


def se3_pack(T):
    return np.concatenate([np.asarray(T[0], np.float64).reshape(9), np.asarray(T[1], np.float64).reshape(3)])


# ------------------------------------------------------------------ calibration
def pyramid_levels(w, h):
    wl, hl, n = w, h, 1
    while wl % 2 == 0 and hl % 2 == 0 and wl * hl > 5000 and n < 6:
        wl //= 2
        hl //= 2
        n += 1
    return n


def kitti_calib(w, h):
    """KITTI-like pinhole for a w x h image. Values are float32 like CalibHessian::value_scaledf."""
    f = np.float32(718.856 * (w / 1241.0))
    return dict(fx=f, fy=f, cx=np.float32(w / 2 - 0.5 + 3.2), cy=np.float32(h / 2 - 0.5 - 2.9), baseline=np.float32(0.5372))


def level_intrinsics(fx, fy, cx, cy, levels):
    """CoarseTracker::makeK / setGlobalCalib: float arithmetic as in the reference."""
    fxs, fys, cxs, cys = [np.float32(fx)], [np.float32(fy)], [np.float32(cx)], [np.float32(cy)]
    for l in range(1, levels):
        fxs.append(np.float32(np.float64(fxs[l - 1]) * 0.5))
        fys.append(np.float32(np.float64(fys[l - 1]) * 0.5))
        cxs.append(np.float32((np.float64(cxs[0]) + 0.5) / (1 << l) - 0.5))
        cys.append(np.float32((np.float64(cys[0]) + 0.5) / (1 << l) - 0.5))
    return fxs, fys, cxs, cys


# ------------------------------------------------------------------ analytic scene
class Scene:
    """Ground plane y=+1.65 (y points down), walls x=-7 / x=+7, far wall z=150 (metres, frame of camera 0)."""

    def __init__(self, seed=1001, nwaves=24, lam_range=(0.15, 5.0)):
        rs = np.random.RandomState(seed)
        self.planes = []
        for _ in range(4):
            lam = np.exp(rs.uniform(np.log(lam_range[0]), np.log(lam_range[1]), nwaves))  # wavelength [m]
            ang = rs.uniform(0, 2 * np.pi, nwaves)
            ph = rs.uniform(0, 2 * np.pi, nwaves)
            amp = np.sqrt(lam)
            amp = amp / amp.sum() * 110.0
            kx = 2 * np.pi / lam * np.cos(ang)
            ky = 2 * np.pi / lam * np.sin(ang)
            self.planes.append((kx, ky, ph, amp))

    def _tex(self, pid, s, t):
        kx, ky, ph, amp = self.planes[pid]
        out = np.full(s.shape, 127.5, dtype=np.float64)
        for i in range(len(kx)):
            out += amp[i] * np.sin(kx[i] * s + ky[i] * t + ph[i])
        return out

    def render(self, w, h, K, T_cw, noise_seed=None, aff=(0.0, 0.0), exposure=1.0):
        """Render the w x h irradiance image of the camera with world-to-camera pose T_cw = (R, t).
        Returns (image float32 [h,w], idepth float32 [h,w])."""
        fx, fy, cx, cy = [np.float64(x) for x in K]
        R_cw, t_cw = T_cw
        R_wc = R_cw.T
        o = -R_wc @ t_cw
        xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        dc = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], axis=-1)
        dw = dc @ R_wc.T
        best = np.full((h, w), np.inf)
        img = np.zeros((h, w))
        cands = [
            (0, 1, 1.65, (0, 2)),   # ground: y = 1.65, tex (X, Z)
            (1, 0, -7.0, (2, 1)),   # left wall: x = -7, tex (Z, Y)
            (2, 0, 7.0, (2, 1)),    # right wall
            (3, 2, 150.0, (0, 1)),  # far wall: z = 150, tex (X, Y)
        ]
        with np.errstate(divide="ignore", invalid="ignore"):
            for pid, axis, val, tc in cands:
                s = (val - o[axis]) / dw[..., axis]
                ok = np.isfinite(s) & (s > 0.5) & (s < best)
                P = o[None, None, :] + s[..., None] * dw
                tex = self._tex(pid, P[..., tc[0]], P[..., tc[1]])
                img = np.where(ok, tex, img)
                best = np.where(ok, s, best)
        idepth = (1.0 / best).astype(np.float32)
        img = exposure * np.exp(aff[0]) * img + aff[1]
        if noise_seed is not None:
            img = img + np.random.RandomState(noise_seed).uniform(-1.5, 1.5, img.shape)
        img = np.clip(img, 0.0, 255.0)
        return img.astype(np.float32), idepth


# ------------------------------------------------------------------ pyramid (makeImages rule)
def abs_squared_grad(d, B=None):
    """absSquaredGrad of one level d = [h, w, 3] {I, dx, dy} (HessianBlocks.cpp:192), times CalibHessian::getBGradOnly(I)^2 when a
    response table B (256 floats) is given (:194-198, HessianBlocks.h:356-362).  float32 arithmetic in the reference's order."""
    a = (d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]).astype(np.float32)
    if B is not None:
        B = np.asarray(B, np.float32)
        c = np.clip((d[..., 0] + np.float32(0.5)).astype(np.int32), 5, 250)
        gw = (B[c + 1] - B[c]).astype(np.float32)
        a = (a * (gw * gw).astype(np.float32)).astype(np.float32)
    return a


def gamma_from_binv(BInv):
    """FullSystem::setGammaFunction (FullSystem.cpp:210-234): CalibHessian::B from the inverse response, float32 like the reference."""
    BInv = np.asarray(BInv, np.float32)
    B = np.zeros(256, np.float32)
    for i in range(1, 255):
        for s in range(1, 255):
            if BInv[s] <= i and BInv[s + 1] >= i:
                B[i] = np.float32(s) + (np.float32(i) - BInv[s]) / (BInv[s + 1] - BInv[s])
                break
    B[0], B[255] = 0, 255
    return B


def make_pyramid(color, levels=None):
    """color: float32 [h,w].  Returns list of AoS arrays [h_l, w_l, 3] float32 = {I, dx, dy}."""
    h, w = color.shape
    if levels is None:
        levels = pyramid_levels(w, h)
    out = []
    I = np.ascontiguousarray(color, dtype=np.float32)
    for lvl in range(levels):
        if lvl > 0:
            P = out[lvl - 1][..., 0]
            I = np.float32(0.25) * (((P[0::2, 0::2] + P[0::2, 1::2]) + P[1::2, 0::2]) + P[1::2, 1::2])
            I = I.astype(np.float32)
        hl, wl = I.shape
        d = np.zeros((hl, wl, 3), dtype=np.float32)
        d[..., 0] = I
        flat = I.reshape(-1)
        idx = np.arange(wl, wl * (hl - 1))
        dx = np.float32(0.5) * (flat[idx + 1] - flat[idx - 1])
        dy = np.float32(0.5) * (flat[idx + wl] - flat[idx - wl])
        dx = np.where(np.isfinite(dx), dx, np.float32(0)).astype(np.float32)
        dy = np.where(np.isfinite(dy), dy, np.float32(0)).astype(np.float32)
        d.reshape(-1, 3)[idx, 1] = dx
        d.reshape(-1, 3)[idx, 2] = dy
        out.append(d)
    return out


# ------------------------------------------------------------------ point selection
def select_points(dI0, n, seed, margin=6, min_grad=8.0, idepth=None, min_idepth=0.0):
    """n distinct integer pixels with gradient magnitude > min_grad inside the margin."""
    h, w, _ = dI0.shape
    rs = np.random.RandomState(seed)
    g = np.sqrt(dI0[..., 1] ** 2 + dI0[..., 2] ** 2)
    ok = g > min_grad
    ok[:margin, :] = False
    ok[-margin - 1:, :] = False
    ok[:, :margin] = False
    ok[:, -margin - 1:] = False
    if idepth is not None:
        ok &= idepth > min_idepth
    ys, xs = np.nonzero(ok)
    if len(xs) < n:
        raise ValueError("not enough textured pixels: %d < %d" % (len(xs), n))
    sel = rs.choice(len(xs), size=n, replace=False)
    sel.sort()
    return xs[sel].astype(np.int32), ys[sel].astype(np.int32)


# ------------------------------------------------------------------ tracker template (makeCoarseDepthL0 STEP2-5)
def make_pc(u, v, idepth, weight, ref_pyr):
    """Build pc_u/pc_v/pc_idepth/pc_color for every level from weighted points splatted at level 0.
    Returns a list (per level) of dicts with float32 arrays u, v, idepth, color."""
    levels = len(ref_pyr)
    h0, w0, _ = ref_pyr[0].shape
    idep = [np.zeros((ref_pyr[l].shape[0], ref_pyr[l].shape[1]), np.float32) for l in range(levels)]
    wsum = [np.zeros_like(idep[l]) for l in range(levels)]
    for i in range(len(u)):  # sequential like the reference's += (duplicates accumulate in order)
        ww = np.float32(weight[i])
        idep[0][v[i], u[i]] += np.float32(idepth[i]) * ww
        wsum[0][v[i], u[i]] += ww
    for l in range(1, levels):
        hl, wl = idep[l].shape
        for src, dst in ((idep, idep), (wsum, wsum)):
            P = src[l - 1][: 2 * hl, : 2 * wl]
            dst[l][...] = ((P[0::2, 0::2] + P[0::2, 1::2]) + P[1::2, 0::2]) + P[1::2, 1::2]

    def dilate(l, offs):
        hl, wl = idep[l].shape
        bak = wsum[l].copy()
        idl = idep[l].reshape(-1)
        wsl = wsum[l].reshape(-1)
        bakf = bak.reshape(-1)
        idx = np.arange(wl, wl * hl - wl)
        need = bakf[idx] <= 0
        s = np.zeros(len(idx), np.float32)
        num = np.zeros(len(idx), np.float32)
        numn = np.zeros(len(idx), np.float32)
        idsrc = idl.copy()  # reads only where bak>0, writes only where bak<=0: a copy is equivalent
        for o in offs:
            j = idx + o
            valid = (j >= 0) & (j < wl * hl)
            jj = np.where(valid, j, 0)
            okk = valid & (bakf[jj] > 0)
            s = np.where(okk, s + idsrc[jj], s).astype(np.float32)
            num = np.where(okk, num + bakf[jj], num).astype(np.float32)
            numn = np.where(okk, numn + 1, numn).astype(np.float32)
        upd = need & (numn > 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            idl[idx[upd]] = (s[upd] / numn[upd]).astype(np.float32)
            wsl[idx[upd]] = (num[upd] / numn[upd]).astype(np.float32)

    for l in range(min(2, levels)):
        wl = idep[l].shape[1]
        dilate(l, [1 + wl, -1 - wl, wl - 1, -wl + 1])
    for l in range(2, levels):
        wl = idep[l].shape[1]
        dilate(l, [1, -1, wl, -wl])

    out = []
    for l in range(levels):
        hl, wl = idep[l].shape
        ys, xs = np.meshgrid(np.arange(2, hl - 2), np.arange(2, wl - 2), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        ws = wsum[l][ys, xs]
        has = ws > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            idn = (idep[l][ys, xs] / ws).astype(np.float32)
        col = ref_pyr[l][ys, xs, 0]
        keep = has & np.isfinite(col) & (idn > 0)
        out.append(dict(u=xs[keep].astype(np.float32), v=ys[keep].astype(np.float32),
                        idepth=idn[keep].astype(np.float32), color=col[keep].astype(np.float32)))
    return out


# ------------------------------------------------------------------ configs
def tracker_problem(w=1232, h=368, npts=2000, seed=2002, scene_seed=1001,
                    motion=(0.02, -0.01, 0.35, 0.004, -0.006, 0.002), aff=(0.02, 1.5),
                    lam_range=(0.15, 5.0), noise=True, min_grad=8.0):
    """reference KF at identity, new frame displaced by `motion` (Sophus tangent).
    Returns dict with reference pyramid, new-frame pyramid, pc arrays per level, calibration, truth."""
    levels = pyramid_levels(w, h)
    cal = kitti_calib(w, h)
    K = (cal["fx"], cal["fy"], cal["cx"], cal["cy"])
    sc = Scene(scene_seed, lam_range=lam_range)
    T_ref = (np.eye(3), np.zeros(3))
    T_new_ref = se3_exp(motion)           # refToNew
    img_ref, id_ref = sc.render(w, h, K, T_ref, noise_seed=(seed + 11) if noise else None)
    img_new, _ = sc.render(w, h, K, T_new_ref, noise_seed=(seed + 12) if noise else None, aff=aff)
    pyr_ref = make_pyramid(img_ref, levels)
    pyr_new = make_pyramid(img_new, levels)
    u, v = select_points(pyr_ref[0], npts, seed, idepth=id_ref, min_idepth=0.0075, min_grad=min_grad)
    idp = id_ref[v, u]
    pc = make_pc(u, v, idp, np.ones(npts, np.float32), pyr_ref)
    fxs, fys, cxs, cys = level_intrinsics(*K, levels)
    return dict(w=w, h=h, levels=levels, calib=cal, K=K, fx=fxs, fy=fys, cx=cxs, cy=cys,
                pyr_ref=pyr_ref, pyr_new=pyr_new, pc=pc, refToNew_true=T_new_ref, aff_true=aff,
                points=(u, v, idp))


def ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001, scene_seed=1001, step_z=0.8,
              rot_jitter=0.01, idepth_noise=0.03, state_noise=1e-3, max_res_per_point=7, point_seed=None):
    """nf keyframes on a forward trajectory, pts_per_kf points hosted per KF, a residual
    to every other KF in which the whole 8-pixel pattern projects inside the image.
    Returns a dict of numpy arrays laid out like sdso_ba_window_t plus the rendered level-0 pyramids."""
    rs = np.random.RandomState(seed)
    levels = pyramid_levels(w, h)
    cal = kitti_calib(w, h)
    K = (cal["fx"], cal["fy"], cal["cx"], cal["cy"])
    sc = Scene(scene_seed)
    poses, affs = [], []
    for k in range(nf):
        xi = np.array([0, 0, -step_z * k, 0, 0, 0], np.float64)      # world->cam: camera moved +z
        xi[3:] = rs.normal(0, rot_jitter, 3)
        xi[0:2] = rs.normal(0, 0.02, 2)
        poses.append(se3_exp(xi))
        affs.append((rs.uniform(-0.05, 0.05), rs.uniform(-5, 5)) if k > 0 else (0.0, 0.0))
    imgs, idmaps, pyrs = [], [], []
    for k in range(nf):
        im, idm = sc.render(w, h, K, poses[k], noise_seed=seed + 100 + k, aff=affs[k])
        imgs.append(im)
        idmaps.append(idm)
        pyrs.append(make_pyramid(im, levels))

    # frame states: evalPT = true pose perturbed; state = small perturbation around zero state
    evalPT = np.zeros((nf, 12))
    state = np.zeros((nf, 10))
    state_zero = np.zeros((nf, 10))
    for k in range(nf):
        pert = rs.normal(0, state_noise, 6) if k > 0 else np.zeros(6)
        T_eval = se3_mul(se3_exp(-pert), poses[k])                   # so that exp(state_scaled)*evalPT ~ true
        evalPT[k] = se3_pack(T_eval)
        st = np.zeros(10)
        st[0:3] = pert[0:3] / SCALE_XI_TRANS
        st[3:6] = pert[3:6] / SCALE_XI_ROT
        st[6] = affs[k][0] / SCALE_A
        st[7] = affs[k][1] / SCALE_B
        state[k] = st
        sz = np.zeros(10)
        sz[6] = st[6] + (rs.normal(0, state_noise) / SCALE_A if k > 0 else 0.0)
        sz[7] = st[7] + (rs.normal(0, state_noise * 10) / SCALE_B if k > 0 else 0.0)
        state_zero[k] = sz

    fx, fy, cx, cy = [np.float64(x) for x in K]
    Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    Ki = np.linalg.inv(Kmat)
    us, vs, ids, hosts, colors, weights, ids_true = [], [], [], [], [], [], []
    res_point, res_target = [], []
    c2 = np.float32(50 * 50)
    for k in range(nf):
        # point_seed decouples the point draw from the frame draw: ranks of a sharded window share the
        # frames (same seed) and own disjoint point sets (different point_seed)
        pseed = seed if point_seed is None else point_seed
        u, v = select_points(pyrs[k][0], pts_per_kf, pseed + 500 + k, idepth=idmaps[k], min_idepth=0.0075)
        idp_true = idmaps[k][v, u].astype(np.float64)
        prs = rs if point_seed is None else np.random.RandomState(pseed + 900 + k)
        idp = (idp_true * (1 + prs.normal(0, idepth_noise, len(u)))).astype(np.float32)
        dI = pyrs[k][0]
        for i in range(len(u)):
            pi = len(us)
            col = np.array([dI[v[i] + PATTERN[j, 1], u[i] + PATTERN[j, 0], 0] for j in range(8)], np.float32)
            gx = np.array([dI[v[i] + PATTERN[j, 1], u[i] + PATTERN[j, 0], 1] for j in range(8)], np.float32)
            gy = np.array([dI[v[i] + PATTERN[j, 1], u[i] + PATTERN[j, 0], 2] for j in range(8)], np.float32)
            wgt = np.sqrt(c2 / (c2 + (gx * gx + gy * gy))).astype(np.float32)
            us.append(np.float32(u[i])); vs.append(np.float32(v[i])); ids.append(idp[i]); hosts.append(k); ids_true.append(idp_true[i])
            colors.append(col); weights.append(wgt)
            # residuals to the other keyframes where the pattern stays inside
            P_h = Ki @ np.array([u[i], v[i], 1.0]) / idp_true[i]
            nres = 0
            for t in range(nf):
                if t == k or nres >= max_res_per_point:
                    continue
                T_th = se3_mul(poses[t], se3_inv(poses[k]))
                Pt = T_th[0] @ P_h + T_th[1]
                if Pt[2] <= 0.1:
                    continue
                uv = Kmat @ (Pt / Pt[2])
                if 8 < uv[0] < w - 9 and 8 < uv[1] < h - 9:
                    res_point.append(pi)
                    res_target.append(t)
                    nres += 1
    np_, nr = len(us), len(res_point)
    n = 8 * nf + 4
    return dict(
        nf=nf, np=np_, nr=nr, w=w, h=h, levels=levels, K=K, calib=cal, pyrs=pyrs, poses=poses, affs=affs,
        calib_value_scaled=np.array([fx, fy, cx, cy], np.float64),
        calib_value_zero=np.array([fx / SCALE_F, fy / SCALE_F, cx / SCALE_C, cy / SCALE_C], np.float64),
        evalPT=evalPT, state=state, state_zero=state_zero,
        ab_exposure=np.ones(nf, np.float32), frameEnergyTH=np.full(nf, 8 * 8 * 8, np.float32),
        frameID=np.arange(nf, dtype=np.int32),
        u=np.array(us, np.float32), v=np.array(vs, np.float32), idepth=np.array(ids, np.float32),
        idepth_true=np.array(ids_true, np.float64), idepth_zero=np.array(ids, np.float32), color=np.array(colors, np.float32).reshape(np_, 8),
        weights=np.array(weights, np.float32).reshape(np_, 8), host=np.array(hosts, np.int32),
        hasDepthPrior=np.zeros(np_, np.uint8),
        res_point=np.array(res_point, np.int32), res_target=np.array(res_target, np.int32),
        res_state=np.zeros(nr, np.uint8), HM=np.zeros((n, n), np.float64), bM=np.zeros(n, np.float64),
        solverMode=128 | 2048, affineOptModeA=1e12, affineOptModeB=1e8, forceAcceptStep=1,
    )


def stereo_problem(w=1232, h=368, npts=20000, seed=4001, scene_seed=1001):
    """a static stereo pair and `npts` fresh immature points on the left image."""
    levels = pyramid_levels(w, h)
    cal = kitti_calib(w, h)
    K = (cal["fx"], cal["fy"], cal["cx"], cal["cy"])
    sc = Scene(scene_seed)
    T_l = (np.eye(3), np.zeros(3))
    T_r = (np.eye(3), np.array([-float(cal["baseline"]), 0.0, 0.0]))   # right camera 0.5372 m to the right
    img_l, id_l = sc.render(w, h, K, T_l, noise_seed=seed + 1)
    img_r, _ = sc.render(w, h, K, T_r, noise_seed=seed + 2)
    pyr_l = make_pyramid(img_l, levels)
    pyr_r = make_pyramid(img_r, levels)
    u, v = select_points(pyr_l[0], npts, seed, margin=6)
    return dict(w=w, h=h, levels=levels, calib=cal, K=K, pyr_l=pyr_l, pyr_r=pyr_r,
                u=u.astype(np.float32), v=v.astype(np.float32), idepth_true=id_l[v, u])
