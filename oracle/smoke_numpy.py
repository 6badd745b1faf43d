"""Second, independently written restatement of SmokeField.step (fluidlab/fluidengine/simulators/smoke_field.py:95-360) in
vectorised numpy, forward only -- test infrastructure like the rest of oracle/.  Positions are in cell units (cell centre =
index + 0.5).  Like oracle/fe_oracle.cpp it falls back to the CLAMPED index in compute_location when the clamped cell is not
free (the reference falls back to the unclamped one, which reads out of range for samples outside the grid)."""
import numpy as np

EPS = 1e-12


def free_space(n, lower_y, higher_y, solid_at=None):
    """compute_free_space (:187-200): the slab lower_y < j < higher_y minus the cells whose centre is inside a static"""
    free = np.zeros((n, n, n), bool)
    free[:, lower_y + 1:higher_y, :] = True
    if solid_at is not None:
        g = (np.arange(n) + 0.5) / n
        X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
        free &= ~solid_at(np.stack([X, Y, Z], -1).reshape(-1, 3)).reshape(n, n, n)
    return free


def _location(free, I, I0):
    """compute_location (:298-306): the clamped neighbour if it is free, else the (clamped) cell itself"""
    n = free.shape[0]
    Ic = np.clip(I, 0, n - 1)
    ok = free[Ic[..., 0], Ic[..., 1], Ic[..., 2]]
    return np.where(ok[..., None], Ic, np.clip(I0, 0, n - 1))


def _at(field, I):
    return field[I[..., 0], I[..., 1], I[..., 2]]


def trilerp(free, field, p):
    """trilerp (:323-343) of a [n,n,n,c] field at positions p [M,3]"""
    base = np.floor(p - 0.5).astype(int)
    pI = p - 0.5
    q = np.zeros((len(p), field.shape[-1]))
    wt = np.zeros(len(p))
    for a in range(2):
        for b in range(2):
            for c in range(2):
                gI = base + [a, b, c]
                w = np.prod(1 - np.abs(pI - gI), axis=1)
                q += w[:, None] * _at(field, _location(free, gI, gI))
                wt += w
    return q / wt[:, None]


def backtrace(free, v, p, dt):
    """RK3 (:346-357)"""
    v1 = trilerp(free, v, p)
    v2 = trilerp(free, v, p - 0.5 * dt * v1)
    v3 = trilerp(free, v, p - 0.75 * dt * v2)
    return p - dt * ((2 / 9) * v1 + (1 / 3) * v2 + (4 / 9) * v3)


def _nb(free, field, ax, sh):
    """field at compute_location(i, j, k, +-1 along ax) for every cell, and whether that neighbour is free and in range"""
    n = free.shape[0]
    I0 = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij'), -1)
    I = I0.copy(); I[..., ax] += sh
    inrange = (I[..., ax] >= 0) & (I[..., ax] <= n - 1)
    Ic = np.clip(I, 0, n - 1)
    isfree = inrange & free[Ic[..., 0], Ic[..., 1], Ic[..., 2]]
    return _at(field, _location(free, I, I0)), isfree


def step(v, q, p, free, dt, solver_iters, aircon, low_T=0.0):
    """One SmokeField.step.  v [n,n,n,3], q [n,n,n,c], p [n,n,n]; aircon: pos (world), quat (wxyz), inject_v, s, r.
    Returns v_tmp, div, and the next v, q, p."""
    n = free.shape[0]
    dx = 1.0 / n
    I0 = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij'), -1).reshape(-1, 3)
    fm = free.reshape(-1)
    cells = I0[fm]
    pb = backtrace(free, v, cells + 0.5, dt)                              # :207-209
    v_f, q_f = trilerp(free, v, pb), trilerp(free, q, pb)
    qv = np.asarray(aircon['quat'][1:], np.float64)                         # geom.py:97-102
    iv = np.asarray(aircon['inject_v'], np.float64)
    uv = np.cross(qv, iv); imp_dir = iv + 2 * (aircon['quat'][0] * uv + np.cross(qv, uv))
    dist = np.sqrt(((cells - np.asarray(aircon['pos'], np.float64) / dx) ** 2).sum(1) + EPS)
    factor = np.exp(-dist / aircon['r'])
    v_tmp = np.zeros_like(v).reshape(-1, 3)
    v_tmp[fm] = v_f + (imp_dir * aircon['s'])[None] * factor[:, None] * dt   # :212-225
    q_new = q.reshape(n ** 3, -1).copy()
    q_new[fm] = (1 - factor)[:, None] * q_f + factor[:, None] * low_T        # :228
    v_tmp = v_tmp.reshape(n, n, n, 3)
    div = np.zeros((n, n, n))                                                # :236-262
    for ax in range(3):
        lo, lo_free = _nb(free, v_tmp, ax, -1)
        hi, hi_free = _nb(free, v_tmp, ax, +1)
        lo_c = np.where(lo_free, lo[..., ax], -v_tmp[..., ax])
        hi_c = np.where(hi_free, hi[..., ax], -v_tmp[..., ax])
        div += 0.5 * (hi_c - lo_c)
    div = np.where(free, div, 0.0)
    cur = np.where(free, p, 0.0)                                             # pressure_to_swap, :265-269
    for _ in range(solver_iters):                                            # pressure_jacobi, :136-146
        tot = sum(_nb(free, cur, ax, sh)[0] for ax in range(3) for sh in (-1, 1))
        cur = np.where(free, (tot - div) / 6.0, 0.0)                         # (the swap buffers start from zero each step: non-free cells stay 0)
    p_new = np.where(free, cur, 0.0)
    grad = np.stack([_nb(free, p_new, ax, +1)[0] - _nb(free, p_new, ax, -1)[0] for ax in range(3)], -1)
    v_new = np.where(free[..., None], v_tmp - 0.5 * grad, v_tmp)              # subtract_gradient, :277-288
    return v_tmp, div, v_new, q_new.reshape(q.shape), p_new
