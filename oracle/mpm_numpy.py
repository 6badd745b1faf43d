"""TEST INFRASTRUCTURE -- a second, independent restatement of the forward MLS-MPM substep.

Vectorised numpy (fp64) version of mpm_simulator.py's forward kernels, written separately
from oracle/fe_oracle.cpp so the two restatements can be checked against each other
(tests/test_oracle.py).  PARITY UNPINNED: the reference ships no vectors for this path.

Citations `mpm:NNN` = fluidlab/fluidengine/simulators/mpm_simulator.py.
"""
import numpy as np

MAT_LIQUID, MAT_PLASTO_ELASTIC, MAT_ELASTIC, MAT_RIGID, MAT_PLASTO_ELASTIC_DEMO = 200, 201, 202, 203, 204
EPS = 1e-12


def proper_svd(F):
    """ti.svd contract (mpm:264): U, V rotations, sign carried by the smallest singular value."""
    U, s, Vt = np.linalg.svd(F)
    V = np.swapaxes(Vt, -1, -2).copy()
    U = U.copy()
    s = s.copy()
    neg = np.linalg.det(U) < 0
    U[neg, :, 2] *= -1
    s[neg, 2] *= -1
    neg = np.linalg.det(V) < 0
    V[neg, :, 2] *= -1
    s[neg, 2] *= -1
    return U, s, V


def weights(x, inv_dx):
    base = (x * inv_dx - 0.5).astype(np.int64)          # mpm:335 (positive coordinates: trunc == floor)
    fx = x * inv_dx - base
    w = np.stack([0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1.0) ** 2, 0.5 * (fx - 0.5) ** 2], axis=0)   # [3, N, 3]
    return base, fx, w


def boundary_v(bnd, xn, v):
    """impose_x_v velocity part, boundaries.py:40-63 / 107-121."""
    v = v.copy()
    r = bnd.get('restitution', 0.0)
    if bnd['type'] == 'cube':
        lo, up = np.asarray(bnd['lower']), np.asarray(bnd['upper'])
        for i in range(3):
            hit = ((xn[:, i] >= up[i]) & (v[:, i] >= 0)) | ((xn[:, i] <= lo[i]) & (v[:, i] <= 0))
            v[hit, i] *= -r
    else:
        y0, y1 = bnd['y_range']
        hit = ((xn[:, 1] > y1) & (v[:, 1] > 0)) | ((xn[:, 1] < y0) & (v[:, 1] < 0))
        v[hit, 1] *= -r
        c = np.asarray(bnd['xz_center'])
        rv = xn[:, [0, 2]] - c
        nrm = np.sqrt((rv ** 2).sum(1) + EPS)
        out = nrm > bnd['xz_radius']
        v[out, 0] = 0.0
        v[out, 2] = 0.0
    for d in bnd.get('lock_dims', ()):
        v[:, d] = 0.0
    return v


def _sdf_sample(vox, pv):
    """static.py:34-49, vectorised over rows of pv"""
    res = vox.shape[0]
    base = np.floor(pv).astype(int)
    outside = ((base >= res - 1) | (base < 0)).any(1)
    b = np.clip(base, 0, res - 2)
    sd = np.zeros(len(pv))
    for i in range(2):
        for j in range(2):
            for k in range(2):
                vp = b + [i, j, k]
                w = np.prod(1 - np.abs(pv - vp), axis=1)
                sd += w * vox[vp[:, 0], vp[:, 1], vp[:, 2]]
    return np.where(outside, 1.0, sd)


def static_collide(static, xn, v):
    """Static.collide (static.py:82-103) for node positions xn [M,3] and velocities v [M,3]"""
    vox, T, mu_f = np.asarray(static['voxels'], np.float64), np.asarray(static['T'], np.float64), static['friction']
    pv = xn @ T[:3, :3].T + T[:3, 3]
    hit = _sdf_sample(vox, pv) <= 0
    if not hit.any():
        return v
    pv_h, vh = pv[hit], v[hit]
    g = np.zeros_like(pv_h)
    for d in range(3):
        e = np.zeros(3); e[d] = 1e-2
        g[:, d] = (_sdf_sample(vox, pv_h + e) - _sdf_sample(vox, pv_h - e)) / 2e-2
    g /= np.sqrt((g ** 2).sum(1) + EPS)[:, None]
    n = g @ np.linalg.inv(T[:3, :3]).T
    n /= np.sqrt((n ** 2).sum(1) + EPS)[:, None]
    nc = (vh * n).sum(1)
    vt = vh - np.minimum(nc, 0)[:, None] * n
    vtn = np.linalg.norm(vt, axis=1)
    flag = (nc < 0) & (vtn > EPS)
    scale = np.where(flag, np.maximum(0, vtn + nc * mu_f) / np.where(vtn > 0, vtn, 1.0), 1.0)
    out = v.copy()
    out[hit] = vt * scale[:, None]
    return out


def impose_x(bnd, x):
    """Boundary.impose_x (boundaries.py:66-78 cylinder, 123-126 cube) for one position"""
    x = np.asarray(x, np.float64)
    if bnd.get('type', 'cube') == 'cube':
        return np.maximum(np.minimum(x, np.asarray(bnd['upper'], np.float64)), np.asarray(bnd['lower'], np.float64))
    lo, hi = np.array([0.0, bnd['y_range'][0], 0.0]), np.array([1.0, bnd['y_range'][1], 1.0])
    xn = np.maximum(np.minimum(x, hi), lo)
    c = np.asarray(bnd['xz_center'], np.float64)
    r = np.array([x[0], x[2]]) - c
    rn = np.sqrt((r ** 2).sum() + EPS)
    if rn > bnd['xz_radius']:
        nxz = r / rn * bnd['xz_radius'] + c
        xn = np.array([nxz[0], xn[1], nxz[1]])
    return xn


def effector_move(bnd, pos, quat, v, w):
    """Effector.move_kernel (effector.py:157-161): pos' = impose_x(pos + v); quat' = qmul(w2quat(w), quat) (geom.py:8-28)"""
    wn = np.sqrt((np.asarray(w, np.float64) ** 2).sum() + EPS)
    a = np.concatenate([[np.cos(wn / 2)], np.asarray(w, np.float64) / wn * np.sin(wn / 2)])
    t = np.outer(np.asarray(quat, np.float64), a)                    # terms = r.outer_product(q) with q = a, r = quat
    o = np.array([t[0, 0] - t[1, 1] - t[2, 2] - t[3, 3], t[0, 1] + t[1, 0] - t[2, 3] + t[3, 2],
                  t[0, 2] + t[1, 3] + t[2, 0] - t[3, 1], t[0, 3] - t[1, 2] + t[2, 1] + t[3, 0]])
    return impose_x(bnd, np.asarray(pos, np.float64) + np.asarray(v, np.float64)), o / np.sqrt((o ** 2).sum())


def _quat_rot(v, q):
    """geom.py:86-90 transform_by_quat (rows of v by one quaternion wxyz)"""
    qv = np.asarray(q[1:], np.float64)
    uv = np.cross(qv, v)
    uuv = np.cross(qv, uv)
    return v + 2 * (q[0] * uv + uuv)


def dynamic_collide(dyn, pos, v, dt):
    """Dynamic.collide (dynamic.py:90-122) for positions pos [M,3] and material velocities v [M,3].  dyn: voxels, T (mesh ->
    voxels), friction, softness, and the carrier's pose at f and f+1 (pos0, quat0, pos1, quat1)."""
    vox, T = np.asarray(dyn['voxels'], np.float64), np.asarray(dyn['T'], np.float64)
    q0 = np.asarray(dyn['quat0'], np.float64)
    qi = np.array([q0[0], -q0[1], -q0[2], -q0[3]]) / np.linalg.norm(q0)                  # inv_quat(...).normalized(), geom.py:30-32
    pm = _quat_rot(pos - np.asarray(dyn['pos0'], np.float64), qi)                       # dynamic.py:33
    pv = pm @ T[:3, :3].T + T[:3, 3]
    sd = _sdf_sample(vox, pv)
    infl = np.minimum(np.exp(-sd * dyn['softness']), 1.0)
    hit = (sd <= 0) | ((dyn['softness'] > 0) & (infl > 0.1))
    out = v.copy()
    if not hit.any():
        return out
    pm_h, pv_h, vh, infl_h = pm[hit], pv[hit], v[hit], infl[hit]
    cv = (_quat_rot(pm_h, np.asarray(dyn['quat1'], np.float64)) + np.asarray(dyn['pos1'], np.float64) - pos[hit]) / dt     # collider_v, :84-88
    if dyn['friction'] > 10.0:
        out[hit] = cv
        return out
    rel = vh - cv
    g = np.zeros_like(pv_h)
    for d in range(3):
        e = np.zeros(3); e[d] = 1e-2
        g[:, d] = (_sdf_sample(vox, pv_h + e) - _sdf_sample(vox, pv_h - e)) / 2e-2
    g /= np.sqrt((g ** 2).sum(1) + EPS)[:, None]
    n = _quat_rot(g @ np.linalg.inv(T[:3, :3]).T, q0)
    n /= np.sqrt((n ** 2).sum(1) + EPS)[:, None]
    nc = (rel * n).sum(1)
    vt = rel - np.minimum(nc, 0)[:, None] * n
    vtn = np.linalg.norm(vt, axis=1)
    flag = (nc < 0) & (vtn > EPS)
    scale = np.where(flag, np.maximum(0, vtn + nc * dyn['friction']) / np.where(vtn > 0, vtn, 1.0), 1.0)
    out[hit] = cv + (vt * scale[:, None]) * infl_h[:, None] + rel * (1 - infl_h)[:, None]
    return out


def substep(x, v, C, F, used, mu, lam, mass, mat_cls, n_grid, dt, p_vol, gravity, bnd, body_id=None, statics=(), dynamic=None,
            collide_type=1):
    """One forward substep (mpm:515-533).  `dynamic`: a moving SDF collider (see dynamic_collide) applied at the particles
    (collide_type & 1, mpm:418-422) and/or at the grid nodes (collide_type & 2, mpm:393-395).  Returns x', v', C', F'."""
    n = n_grid
    dx, inv_dx = 1.0 / n, float(n)
    act = used.astype(bool)
    xs, vs, Cs, Fs = x[act], v[act], C[act], F[act]
    mus, lams, ms, cls = mu[act], lam[act], mass[act], mat_cls[act]
    I = np.eye(3)
    Ft = (I + dt * Cs) @ Fs                                             # mpm:258
    U, s, V = proper_svd(Ft)
    J = s.prod(1)                                                       # mpm:339
    r = U @ np.swapaxes(V, 1, 2)
    stress = 2 * mus[:, None, None] * (Ft - r) @ np.swapaxes(Ft, 1, 2) + I * (lams * J * (J - 1))[:, None, None]
    stress = (-dt * p_vol * 4 * inv_dx * inv_dx) * stress               # mpm:343
    affine = stress + ms[:, None, None] * Cs
    base, fx, w = weights(xs, inv_dx)
    g_v = np.zeros((n, n, n, 3))
    g_m = np.zeros((n, n, n))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                off = np.array([i, j, k])
                dpos = (off - fx) * dx
                wt = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                idx = base + off
                np.add.at(g_v, (idx[:, 0], idx[:, 1], idx[:, 2]), wt[:, None] * (ms[:, None] * vs + np.einsum('nab,nb->na', affine, dpos)))
                np.add.at(g_m, (idx[:, 0], idx[:, 1], idx[:, 2]), wt * ms)
    Fn = np.zeros_like(Fs)
    liq = cls == MAT_LIQUID
    Fn[liq] = I * np.cbrt(J[liq])[:, None, None]                        # J > 0 in all tests
    ela = (cls == MAT_ELASTIC) | (cls == MAT_RIGID)
    Fn[ela] = Ft[ela]
    pla = (cls == MAT_PLASTO_ELASTIC) | (cls == MAT_PLASTO_ELASTIC_DEMO)
    sn = np.clip(s, 1 - 2e-3, 1 + 3e-3)
    Fn[pla] = (U[pla] * sn[pla][:, None, :]) @ np.swapaxes(V[pla], 1, 2)
    # grid_op (mpm:380-398)
    occ = g_m > EPS
    v_out = np.zeros_like(g_v)
    ii, jj, kk = np.nonzero(occ)
    vo = g_v[occ] / g_m[occ][:, None] + dt * np.asarray(gravity)
    xn = np.stack([ii, jj, kk], 1) * dx
    for st in statics:                                                  # mpm:386-390
        vo = static_collide(st, xn, vo)
    if dynamic is not None and (collide_type & 2):                      # mpm:393-395
        vo = dynamic_collide(dynamic, xn, vo, dt)
    v_out[occ] = boundary_v(bnd, xn, vo)
    # g2p (mpm:400-426)
    nv = np.zeros_like(vs)
    nC = np.zeros_like(Cs)
    for i in range(3):
        for j in range(3):
            for k in range(3):
                off = np.array([i, j, k])
                dpos = off - fx
                wt = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                idx = base + off
                gv = v_out[idx[:, 0], idx[:, 1], idx[:, 2]]
                nv += wt[:, None] * gv
                nC += 4 * inv_dx * wt[:, None, None] * gv[:, :, None] * dpos[:, None, :]
    if dynamic is not None and (collide_type & 1):                      # mpm:418-422
        nv = dynamic_collide(dynamic, xs + dt * nv, nv, dt)
    x2, v2, C2, F2 = x.copy(), v.copy(), C.copy(), F.copy()
    xn2 = xs + dt * nv                                                  # mpm:505
    if body_id is not None and (cls == MAT_RIGID).any():
        # MAT_RIGID shape matching, mpm:449-505: means divide by the body's TOTAL particle count (mpm:201,461)
        bid_all = np.asarray(body_id)
        bid = bid_all[act]
        for b in np.unique(bid[cls == MAT_RIGID]):
            sel = (bid == b) & (cls == MAT_RIGID)
            nb = float((bid_all == b).sum())
            c0 = xs[sel].sum(0) / nb
            c1 = xn2[sel].sum(0) / nb
            H = (xs[sel] - c0).T @ (xn2[sel] - c1)
            Ub, _, Vb = proper_svd(H[None])
            R = Vb[0] @ Ub[0].T
            xn2[sel] = (xs[sel] - c0) @ R.T + c1
    x2[act] = xn2
    v2[act] = nv
    C2[act] = nC
    F2[act] = Fn
    return x2, v2, C2, F2, dict(grid_mass=g_m, grid_v_in=g_v, grid_v_out=v_out)
