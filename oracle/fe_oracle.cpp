/*
 * fe_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the FluidLab MLS-MPM hot path, kernel by kernel, behind the
 * same C ABI as the HIP engine (include/fluidengine.h).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path and
 * its implementation (Python + Taichi 1.1.0 JIT) can neither be imported nor built
 * in this image.  This file follows fluidlab/fluidengine/simulators/mpm_simulator.py
 * line by line (citations as `mpm:NNN`); the adjoints, which the reference obtains
 * from Taichi's source-transform autodiff, are hand-derived here and validated
 * against central finite differences of this file's own forward (tests/).
 * Also restated: boundaries.py, effector.py / injector.py / aircon.py / rigid.py,
 * meshes/static.py + dynamic.py (SDF colliders), the agents' collector kernels,
 * smoke_field.py and shapematching_loss.py.
 * Third-party arithmetic not in /root/reference: taichi==1.1.0 `ti.svd` (McAdams
 * et al. 3x3 SVD; contract restated in svd3() below) and mesh_to_sdf 0.0.x (the
 * signed distance it approximates by virtual scans is computed exactly in
 * fe_mesh_sdf() below: point-triangle distance + generalized winding number).
 *
 * Build: see oracle/Makefile (-DFE_REAL=float -> libfe_oracle_f32.so, double -> _f64).
 */
#include <algorithm>
#include <atomic>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/fluidengine.h"

typedef fe_real R;

namespace {

const R EPS = (R)1e-12;   /* fluidlab/configs/macros.py:213 */

struct M3 { R m[3][3]; };
struct V3 { R v[3]; };

inline M3 m_zero() { M3 a; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.m[i][j] = 0; return a; }
inline M3 m_ident() { M3 a = m_zero(); a.m[0][0] = a.m[1][1] = a.m[2][2] = 1; return a; }
inline M3 m_mul(const M3& a, const M3& b) {
    M3 c;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        R s = 0; for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j]; c.m[i][j] = s;
    }
    return c;
}
inline M3 m_T(const M3& a) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.m[i][j] = a.m[j][i]; return c; }
inline M3 m_add(const M3& a, const M3& b) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] + b.m[i][j]; return c; }
inline M3 m_sub(const M3& a, const M3& b) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] - b.m[i][j]; return c; }
inline M3 m_scale(const M3& a, R s) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] * s; return c; }
inline M3 m_had(const M3& a, const M3& b) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] * b.m[i][j]; return c; }
inline R m_det(const M3& a) {
    return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1])
         - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0])
         + a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
inline R m_trace(const M3& a) { return a.m[0][0] + a.m[1][1] + a.m[2][2]; }
inline M3 m_load(const R* p) { M3 a; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.m[i][j] = p[i * 3 + j]; return a; }
inline void m_store(R* p, const M3& a) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p[i * 3 + j] = a.m[i][j]; }
inline void m_accum(R* p, const M3& a) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p[i * 3 + j] += a.m[i][j]; }

/*
 * 3x3 SVD with the contract of taichi 1.1.0 `ti.svd` (call sites mpm:264, mpm:483):
 * F = U diag(sig) V^T, U and V proper rotations (det = +1), |sig| sorted descending,
 * a negative sign (det F < 0) carried by the last (smallest) singular value.
 * Taichi's implementation is the McAdams/Sifakis fixed-sweep Jacobi; it is not in
 * /root/reference, so the published contract is restated with a one-sided (Hestenes)
 * Jacobi iterated to convergence.  Consumers (mpm:339-376) use only U V^T, det S and
 * U clamp(S) V^T, which are invariant to the residual gauge freedom.
 */
void svd3(const M3& Fin, M3& U, M3& S, M3& V) {
    R A[3][3];
    R Vm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = Fin.m[i][j];
    const R tiny = sizeof(R) == 4 ? (R)1e-30 : (R)1e-290;
    const R tol = sizeof(R) == 4 ? (R)1e-7 : (R)1e-15;
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            R alpha = 0, beta = 0, gamma = 0;
            for (int i = 0; i < 3; i++) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
            if (std::fabs(gamma) <= tol * std::sqrt(alpha * beta) || std::fabs(gamma) < tiny) continue;
            rotated = true;
            R zeta = (beta - alpha) / (2 * gamma);
            R t = (zeta >= 0 ? (R)1 : (R)-1) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
            R c = 1 / std::sqrt(1 + t * t), s = c * t;
            for (int i = 0; i < 3; i++) {
                R ap = A[i][p], aq = A[i][q];
                A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
                R vp = Vm[i][p], vq = Vm[i][q];
                Vm[i][p] = c * vp - s * vq; Vm[i][q] = s * vp + c * vq;
            }
        }
        if (!rotated) break;
    }
    R sig[3];
    for (int j = 0; j < 3; j++) sig[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    /* sort descending (swap columns of A, V together) */
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2 - a; b++) if (sig[b] < sig[b + 1]) {
        std::swap(sig[b], sig[b + 1]);
        for (int i = 0; i < 3; i++) { std::swap(A[i][b], A[i][b + 1]); std::swap(Vm[i][b], Vm[i][b + 1]); }
    }
    R Um[3][3];
    /* U columns = A columns / sigma; complete rank-deficient columns orthonormally */
    const R rel = sig[0] * (sizeof(R) == 4 ? (R)1e-6 : (R)1e-13);
    int good = 0;
    for (int j = 0; j < 3; j++) {
        if (sig[j] > rel && sig[j] > tiny) { for (int i = 0; i < 3; i++) Um[i][j] = A[i][j] / sig[j]; good = j + 1; }
        else break;
    }
    if (good == 0) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Um[i][j] = (i == j); good = 3; }
    if (good == 1) {
        /* pick any unit vector orthogonal to column 0 */
        int k = 0; for (int i = 1; i < 3; i++) if (std::fabs(Um[i][0]) < std::fabs(Um[k][0])) k = i;
        R e[3] = {0, 0, 0}; e[k] = 1;
        R d = Um[k][0];
        R w[3]; R nn = 0; for (int i = 0; i < 3; i++) { w[i] = e[i] - d * Um[i][0]; nn += w[i] * w[i]; }
        nn = std::sqrt(nn); for (int i = 0; i < 3; i++) Um[i][1] = w[i] / nn;
        good = 2;
    }
    if (good == 2) {
        Um[0][2] = Um[1][0] * Um[2][1] - Um[2][0] * Um[1][1];
        Um[1][2] = Um[2][0] * Um[0][1] - Um[0][0] * Um[2][1];
        Um[2][2] = Um[0][0] * Um[1][1] - Um[1][0] * Um[0][1];
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U.m[i][j] = Um[i][j]; V.m[i][j] = Vm[i][j]; }
    if (m_det(U) < 0) { for (int i = 0; i < 3; i++) U.m[i][2] = -U.m[i][2]; sig[2] = -sig[2]; }
    if (m_det(V) < 0) { for (int i = 0; i < 3; i++) V.m[i][2] = -V.m[i][2]; sig[2] = -sig[2]; }
    S = m_zero(); S.m[0][0] = sig[0]; S.m[1][1] = sig[1]; S.m[2][2] = sig[2];
}

/* mpm:294-302 */
inline R clamp_svd(R a) { if (a >= 0) a = std::max(a, (R)1e-8); else a = std::min(a, (R)-1e-8); return a; }

/* mpm:272-292 */
M3 backward_svd(const M3& gU, const M3& gS, const M3& gV, const M3& U, const M3& S, const M3& V) {
    M3 vt = m_T(V), ut = m_T(U);
    M3 S_term = m_mul(m_mul(U, gS), vt);
    R s[3] = {S.m[0][0] * S.m[0][0], S.m[1][1] * S.m[1][1], S.m[2][2] * S.m[2][2]};
    M3 Fm = m_zero();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Fm.m[i][j] = (i == j) ? (R)0 : (R)1 / clamp_svd(s[j] - s[i]);
    M3 u_term = m_mul(m_mul(U, m_mul(m_had(Fm, m_sub(m_mul(ut, gU), m_mul(m_T(gU), U))), S)), vt);
    M3 v_term = m_mul(U, m_mul(S, m_mul(m_had(Fm, m_sub(m_mul(vt, gV), m_mul(m_T(gV), V))), vt)));
    return m_add(m_add(u_term, v_term), S_term);
}

/* fluidlab/utils/geom.py:8-16 */
inline void qmul(const R q[4], const R r[4], R out[4]) {
    R t[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t[i][j] = r[i] * q[j];
    R w = t[0][0] - t[1][1] - t[2][2] - t[3][3];
    R x = t[0][1] + t[1][0] - t[2][3] + t[3][2];
    R y = t[0][2] + t[1][3] + t[2][0] - t[3][1];
    R z = t[0][3] - t[1][2] + t[2][1] + t[3][0];
    R nn = std::sqrt(w * w + x * x + y * y + z * z);
    out[0] = w / nn; out[1] = x / nn; out[2] = y / nn; out[3] = z / nn;
}
/* geom.py:18-28; Vector.norm(eps) = sqrt(|a|^2 + eps) */
inline void w2quat(const R aa[3], R out[4]) {
    R w = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2] + EPS);
    R sh = std::sin(w / 2);
    out[0] = std::cos(w / 2); out[1] = aa[0] / w * sh; out[2] = aa[1] / w * sh; out[3] = aa[2] / w * sh;
}
/* geom.py:97-102 */
inline void transform_by_quat(const R v[3], const R q[4], R out[3]) {
    R qv[3] = {q[1], q[2], q[3]};
    R uv[3] = {qv[1] * v[2] - qv[2] * v[1], qv[2] * v[0] - qv[0] * v[2], qv[0] * v[1] - qv[1] * v[0]};
    R uuv[3] = {qv[1] * uv[2] - qv[2] * uv[1], qv[2] * uv[0] - qv[0] * uv[2], qv[0] * uv[1] - qv[1] * uv[0]};
    for (int i = 0; i < 3; i++) out[i] = v[i] + 2 * (q[0] * uv[i] + uuv[i]);
}

/* Boundary.impose_x: boundaries.py:66-78 (cylinder), 123-126 (cube).  Also returns the
 * Jacobian d x_new / d x (3x3) under Taichi's min/max adjoint rules (Appendix B of SURVEY). */
void impose_x(const FeBoundary& b, const R x[3], R xn[3], R J[3][3]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i][j] = 0;
    for (int i = 0; i < 3; i++) {
        R m = std::min(x[i], b.upper[i]);
        xn[i] = std::max(m, b.lower[i]);
        /* min(a,b): adjoint to a iff a<b ; max(a,b): adjoint to a iff a>b */
        J[i][i] = (x[i] < b.upper[i] && m > b.lower[i]) ? (R)1 : (R)0;
    }
    if (b.type == FE_BOUNDARY_CYLINDER) {
        R rx = x[0] - b.xz_center[0], rz = x[2] - b.xz_center[1];
        R nrm = std::sqrt(rx * rx + rz * rz + EPS);
        if (nrm > b.xz_radius) {
            xn[0] = rx / nrm * b.xz_radius + b.xz_center[0];
            xn[2] = rz / nrm * b.xz_radius + b.xz_center[1];
            R k = b.xz_radius / nrm, k3 = b.xz_radius / (nrm * nrm * nrm);
            J[0][0] = k - k3 * rx * rx; J[0][2] = -k3 * rx * rz;
            J[2][0] = -k3 * rz * rx;    J[2][2] = k - k3 * rz * rz;
        }
    }
}

/* Boundary.impose_x_v velocity part: boundaries.py:40-63 (cylinder), 107-121 (cube).
 * v is modified in place; k[d] is the multiplier applied to component d (for the adjoint). */
void impose_v(const FeBoundary& b, const R x[3], R v[3], R k[3]) {
    k[0] = k[1] = k[2] = 1;
    if (b.type == FE_BOUNDARY_CUBE) {
        for (int i = 0; i < 3; i++) {
            if (x[i] >= b.upper[i] && v[i] >= 0) k[i] = -b.restitution;
            else if (x[i] <= b.lower[i] && v[i] <= 0) k[i] = -b.restitution;
        }
    } else {
        if (x[1] > b.upper[1] && v[1] > 0) k[1] = -b.restitution;
        else if (x[1] < b.lower[1] && v[1] < 0) k[1] = -b.restitution;
        R rx = x[0] - b.xz_center[0], rz = x[2] - b.xz_center[1];
        R nrm = std::sqrt(rx * rx + rz * rz + EPS);
        if (nrm > b.xz_radius) { k[0] = 0; k[2] = 0; }
    }
    for (int i = 0; i < 3; i++) if (b.lock_dims & (1 << i)) k[i] = 0;
    for (int i = 0; i < 3; i++) v[i] = (k[i] == 1) ? v[i] : ((k[i] == 0) ? (R)0 : v[i] * k[i]);
}

struct Effector {
    FeEffectorDesc d;
    std::vector<R> pos, quat, v, w, gpos, gquat, gv, gw;     /* [L+1] x {3,4,3,3} */
    std::vector<R> sa, ra, gsa, gra;                         /* AirCon strength s[f], radius r[f] + grads (aircon.py:20-21) */
    std::vector<R> abuf, gabuf, abuf_p, gabuf_p;              /* action buffers + grads */
    std::vector<int> act_id;                                   /* [L+1] */
    std::vector<R> random_vector;
    std::vector<int> act_range;
    int mesh = -1;                                             /* index into FeEngine::meshes (Rigid.setup_mesh, rigid.py:19-24) */
};

/* a Mesh with has_dynamics: SDF voxels + world->voxel map (mesh.py:57-66,120-127) */
struct Sdf { int res; std::vector<R> vox; R T[16]; R Rinv[9]; R friction, softness; };

/* SmokeField state (smoke_field.py:57-84) */
struct Smoke {
    FeSmokeConfig c;
    int n, S;                      /* res, max_steps_local */
    size_t n3;
    std::vector<R> v, vt, dv, p, q, gv, gvt, gdv, gp, gq;       /* [(S+1)][n3][{3,3,1,1,q_dim}] */
    std::vector<unsigned char> fr;                                /* is_free [(S+1)][n3] */
    std::vector<R> pc, pn, gpc, gpn;                              /* p_swap.cur / nxt + grads */
    std::vector<std::vector<R>> jac;                              /* Jacobi iterates kept for nothing: the sweep is linear */
    size_t idx(int i, int j, int k) const { return ((size_t)i * n + j) * n + k; }
    R* F(std::vector<R>& a, int s, int comps) { return &a[(size_t)s * n3 * comps]; }
    unsigned char* FR(int s) { return &fr[(size_t)s * n3]; }
};

/* body_state of mpm:181-189 plus its adjoint */
struct Body { R com0[3], com1[3]; M3 H, Rm, U, S, V; R gcom0[3], gcom1[3]; M3 gH, gR, gU, gS, gV; };

} // namespace

struct FeEngine {
    FeConfig cfg;
    Smoke* smoke = nullptr;            /* SmokeField (smoke_field.py), optional */
    int N, L, n;
    R dx, inv_dx;
    std::vector<R> x, v, C, F, gx, gv, gC, gF;   /* [(L+1), N, {3,3,9,9}] */
    std::vector<int> used;                        /* [(L+1), N] */
    std::vector<R> Ftmp, U, V, S, gFtmp, gU, gV, gS; /* per-substep temporaries [N,9] */
    std::vector<R> mu, lam, mass;
    std::vector<int> mat, mat_cls, body_id;
    /* rigid bodies, mpm:176-201: per-body particle count and class; state is transient per substep */
    int n_bodies = 0;
    bool has_rigid = false;
    std::vector<int> body_n, body_cls;
    std::vector<Body> bodies;
    std::vector<R> g_vin, g_mass, g_vout, gg_vin, gg_mass, gg_vout;
    std::vector<Effector> effs;
    std::vector<Sdf> statics;        /* statics.py */
    std::vector<Sdf> meshes;         /* Dynamic meshes of Rigid effectors (dynamic.py) */
    bool has_mesh_effector = false;
    int inject_till = -1;            /* AgentIceCreamDynamic.inject_till (agent_icecreamdynamic.py:11,23-30): no injection from this global substep on */
    int collide_type = 1;            /* Agent.collide_type (agent.py:17-26): 1 'particle', 2 'grid', 3 'both' */
    bool has_collector = false;      /* AgentPouring / AgentJetBot collector_act_kernel (agent_pouring.py:30-41, agent_jetbot.py:33-43) */
    FeBoundary collector;            /* particles outside this boundary are taken out of the simulation */
    int collector_mat = -1;          /* < 0: every material (AgentPouring); else only this one (AgentJetBot: WATER) */
    R collide_min_y = (R)-1e30;      /* AgentIceCreamDynamic.collide: only above this height (agent_icecreamdynamic.py:39-43) */
    int loss_steps = 0;
    std::vector<R> tgt;          /* [loss_steps, N, 3] */
    std::vector<R> chamfer, step_loss;
    std::string err;
    int threads = 1;
    int scatter_coloured = 0;      /* option "scatter": 1 = the particle scatters go colour by colour without atomics (scatter_loop) */
    bool initialized = false;
    std::chrono::steady_clock::time_point t0;
    bool prof_on = false;
    double prof_ms[2] = {0, 0};
    long long prof_n[2] = {0, 0};

    R* X(int f) { return &x[(size_t)f * N * 3]; }
    R* Vv(int f) { return &v[(size_t)f * N * 3]; }
    R* Cc(int f) { return &C[(size_t)f * N * 9]; }
    R* Ff(int f) { return &F[(size_t)f * N * 9]; }
    R* GX(int f) { return &gx[(size_t)f * N * 3]; }
    R* GV(int f) { return &gv[(size_t)f * N * 3]; }
    R* GC(int f) { return &gC[(size_t)f * N * 9]; }
    R* GF(int f) { return &gF[(size_t)f * N * 9]; }
    int* Us(int f) { return &used[(size_t)f * N]; }
};

static std::string g_create_err;

#define FE_FAIL(h, msg) do { (h)->err = (msg); return 1; } while (0)
#define CHECK_FRAME(h, f) do { if ((f) < 0 || (f) > (h)->L) FE_FAIL(h, "frame index out of range"); } while (0)
#define CHECK_EFF(h, e) do { if ((e) < 0 || (e) >= (int)(h)->effs.size()) FE_FAIL(h, "effector index out of range"); } while (0)

namespace {

inline size_t cell_index(int n, int i, int j, int k) { return ((size_t)i * n + j) * n + k; }

struct Stencil { int base[3]; R fx[3]; R w[3][3]; R dw[3][3]; };

/* mpm:335-337 / mpm:404-406 */
inline void make_stencil(const R* xp, R inv_dx, Stencil& s) {
    for (int d = 0; d < 3; d++) {
        s.base[d] = (int)(xp[d] * inv_dx - (R)0.5);   /* C-style truncation == Taichi cast(int) */
        s.fx[d] = xp[d] * inv_dx - (R)s.base[d];
        R fx = s.fx[d];
        s.w[0][d] = (R)0.5 * ((R)1.5 - fx) * ((R)1.5 - fx);
        s.w[1][d] = (R)0.75 - (fx - 1) * (fx - 1);
        s.w[2][d] = (R)0.5 * (fx - (R)0.5) * (fx - (R)0.5);
        s.dw[0][d] = -((R)1.5 - fx);
        s.dw[1][d] = -2 * (fx - 1);
        s.dw[2][d] = fx - (R)0.5;
    }
}

inline bool stencil_in_grid(const Stencil& s, int n) {
    for (int d = 0; d < 3; d++) if (s.base[d] < 0 || s.base[d] > n - 3) return false;
    return true;
}

/* ------------------------------------------------------------------ forward kernels */

/* mpm:219-223 */
void reset_grid_and_grad(FeEngine* h) {
    std::fill(h->g_vin.begin(), h->g_vin.end(), (R)0);
    std::fill(h->g_mass.begin(), h->g_mass.end(), (R)0);
    std::fill(h->g_vout.begin(), h->g_vout.end(), (R)0);
    std::fill(h->gg_vin.begin(), h->gg_vin.end(), (R)0);
    std::fill(h->gg_mass.begin(), h->gg_mass.end(), (R)0);
    std::fill(h->gg_vout.begin(), h->gg_vout.end(), (R)0);
}

/* mpm:304-307 */
void advect_used(FeEngine* h, int f) {
    std::memcpy(h->Us(f + 1), h->Us(f), sizeof(int) * h->N);
}

/* mpm:309-316 */
void process_unused_particles(FeEngine* h, int f) {
    const int N = h->N;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (int p = 0; p < N; p++) {
        if (h->Us(f)[p] == 0) {
            for (int d = 0; d < 3; d++) { h->Vv(f + 1)[p * 3 + d] = h->Vv(f)[p * 3 + d]; h->X(f + 1)[p * 3 + d] = h->X(f)[p * 3 + d]; }
            for (int d = 0; d < 9; d++) { h->Cc(f + 1)[p * 9 + d] = h->Cc(f)[p * 9 + d]; h->Ff(f + 1)[p * 9 + d] = h->Ff(f)[p * 9 + d]; }
        }
    }
}

/* Injector.act: injector.py:80-105 (called from AgentInjector.act_kernel, agent_injector.py:30-32) */
int injector_act(FeEngine* h, Effector& e, int f, int f_global) {
    const int flux = e.d.flux;
    if (e.act_range.empty()) { h->err = "injector has no act_range"; return 1; }
    const int row = e.d.locally_random ? f : f_global;
    if (row < 0 || row >= e.d.random_length) { h->err = "injector random_vector row out of range"; return 1; }
    /* AgentInjector.check_act_range, agent_injector.py:38-39 */
    if (e.act_id[f] + flux > (int)e.act_range.size()) { h->err = "too many particles added"; return 1; }
    const R* q = &e.quat[f * 4];
    for (int i = 0; i < flux; i++) {
        int pid = e.act_range[e.act_id[f] + i];
        const R* rv = &e.random_vector[((size_t)row * flux + i) * 3];
        R ip[3], iv[3];
        transform_by_quat(e.d.inject_p, q, ip);
        transform_by_quat(e.d.inject_v, q, iv);
        R vnorm = std::sqrt(e.d.inject_v[0] * e.d.inject_v[0] + e.d.inject_v[1] * e.d.inject_v[1] + e.d.inject_v[2] * e.d.inject_v[2]);
        for (int d = 0; d < 3; d++) {
            R offset = (rv[d] * 2 - 1) * e.d.radius;
            h->X(f + 1)[pid * 3 + d] = offset + e.pos[f * 3 + d] + ip[d];
            if (e.d.randomize_inject_v) h->Vv(f + 1)[pid * 3 + d] = iv[d] + (rv[d] * 2 - 1) * vnorm * (R)2.0;
            else h->Vv(f + 1)[pid * 3 + d] = iv[d];
        }
        h->Us(f + 1)[pid] = 1;
    }
    e.act_id[f + 1] = e.act_id[f] + flux;
    return 0;
}

/* mpm:254-264: compute_F_tmp + svd */
void compute_F_tmp_svd(FeEngine* h, int f) {
    const int N = h->N;
    const R dt = h->cfg.dt;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        M3 Cm = m_load(&h->Cc(f)[p * 9]), Fm = m_load(&h->Ff(f)[p * 9]);
        M3 Ft = m_mul(m_add(m_ident(), m_scale(Cm, dt)), Fm);           /* mpm:258 */
        M3 U, S, V;
        svd3(Ft, U, S, V);                                               /* mpm:264 */
        m_store(&h->Ftmp[p * 9], Ft); m_store(&h->U[p * 9], U); m_store(&h->S[p * 9], S); m_store(&h->V[p * 9], V);
    }
}

/* the per-particle quantities p2g needs (mpm:339-344), shared by forward and adjoint */
struct P2GLocal { M3 Ft, U, S, V, r, stress_raw, affine; R J, scale; };

inline void p2g_local(FeEngine* h, int f, int p, P2GLocal& l) {
    l.Ft = m_load(&h->Ftmp[p * 9]); l.U = m_load(&h->U[p * 9]); l.S = m_load(&h->S[p * 9]); l.V = m_load(&h->V[p * 9]);
    l.J = m_det(l.S);                                                                   /* mpm:339 */
    l.r = m_mul(l.U, m_T(l.V));                                                         /* mpm:341 */
    M3 st = m_scale(m_mul(m_sub(l.Ft, l.r), m_T(l.Ft)), 2 * h->mu[p]);
    R iso = h->lam[p] * l.J * (l.J - 1);
    st.m[0][0] += iso; st.m[1][1] += iso; st.m[2][2] += iso;                            /* mpm:342 */
    l.stress_raw = st;
    l.scale = -h->cfg.dt * h->cfg.p_vol * 4 * h->inv_dx * h->inv_dx;                    /* mpm:343 */
    l.affine = m_add(m_scale(st, l.scale), m_scale(m_load(&h->Cc(f)[p * 9]), h->mass[p])); /* mpm:344 */
}

/* mpm:331-378 */

/* Scatter without atomics (option "scatter" = 1; bench.py's CPU baseline): the used particles are bucketed by the 4x4x4-cell block of
 * their stencil base; a block's particles write nodes [4B, 4B + 5] per axis, so blocks whose indices have the same parity along every
 * axis never touch the same node -- eight colours, one after the other, the blocks of a colour in parallel with plain adds.  With
 * `#pragma omp atomic` (the default, which the parity tests use) more than ~16 threads made the scatter loops SLOWER. */
struct ColourLists { std::vector<int> pid; std::vector<int> blk_start; std::vector<int> blocks[8]; };
static void build_colour_lists(FeEngine* h, int f, ColourLists& L) {
    const int N = h->N, n = h->n, nb = (n + 3) / 4;
    std::vector<int> key(N, -1), cnt((size_t)nb * nb * nb + 1, 0);
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s);
        if (!stencil_in_grid(s, n)) continue;
        key[p] = ((s.base[0] >> 2) * nb + (s.base[1] >> 2)) * nb + (s.base[2] >> 2);
        cnt[key[p] + 1]++;
    }
    for (size_t b = 0; b + 1 < cnt.size(); b++) cnt[b + 1] += cnt[b];
    L.blk_start = cnt;
    L.pid.assign(cnt.back(), 0);
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (int p = 0; p < N; p++) if (key[p] >= 0) L.pid[fill[key[p]]++] = p;
    for (int c = 0; c < 8; c++) L.blocks[c].clear();
    for (int b = 0; b + 1 < (int)cnt.size(); b++) {
        if (cnt[b + 1] == cnt[b]) continue;
        const int bi = b / (nb * nb), bj = (b / nb) % nb, bk = b % nb;
        L.blocks[(bi & 1) * 4 + (bj & 1) * 2 + (bk & 1)].push_back(b);
    }
}
/* run one(p, atomic) over every particle: in particle order with atomics, or colour by colour without */
template <typename F>
static void scatter_loop(FeEngine* h, int f, F&& one) {
    const int N = h->N;
    if (h->scatter_coloured && h->threads > 1) {
        ColourLists L; build_colour_lists(h, f, L);
        for (int c = 0; c < 8; c++) {
            const std::vector<int>& bl = L.blocks[c];
#pragma omp parallel for num_threads(h->threads) schedule(dynamic, 4)
            for (long long i = 0; i < (long long)bl.size(); i++)
                for (int q = L.blk_start[bl[i]]; q < L.blk_start[bl[i] + 1]; q++) one(L.pid[q], false);
        }
        /* particles without a bucket (unused / stencil off the grid) take the plain road: they scatter nothing */
#pragma omp parallel for num_threads(h->threads) schedule(static)
        for (int p = 0; p < N; p++) {
            bool skip = false;
            if (h->Us(f)[p]) { Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s); skip = stencil_in_grid(s, h->n); }
            if (!skip) one(p, false);
        }
    } else {
        const bool par = h->threads > 1;
#pragma omp parallel for num_threads(h->threads) schedule(static)
        for (int p = 0; p < N; p++) one(p, par);
    }
}

int p2g(FeEngine* h, int f, bool write_F) {
    const int n = h->n;
    std::atomic<int> bad{0};
    scatter_loop(h, f, [&](int p, bool par) {
        if (!h->Us(f)[p]) return;
        Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s);
        if (!stencil_in_grid(s, n)) { bad++; return; }
        P2GLocal l; p2g_local(h, f, p, l);
        const R m = h->mass[p];
        const R* vp = &h->Vv(f)[p * 3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
            R dpos[3] = {(i - s.fx[0]) * h->dx, (j - s.fx[1]) * h->dx, (k - s.fx[2]) * h->dx};  /* mpm:347 */
            R weight = (R)1.0; weight *= s.w[i][0]; weight *= s.w[j][1]; weight *= s.w[k][2];
            size_t c = cell_index(n, s.base[0] + i, s.base[1] + j, s.base[2] + k);
            for (int a = 0; a < 3; a++) {
                R add = weight * (m * vp[a] + (l.affine.m[a][0] * dpos[0] + l.affine.m[a][1] * dpos[1] + l.affine.m[a][2] * dpos[2]));
                if (par) {
#pragma omp atomic
                    h->g_vin[c * 3 + a] += add;                                              /* mpm:352 */
                } else h->g_vin[c * 3 + a] += add;
            }
            if (par) {
#pragma omp atomic
                h->g_mass[c] += weight * m;                                                  /* mpm:353 */
            } else h->g_mass[c] += weight * m;
        }
        if (!write_F) return;
        /* mpm:355-378 */
        M3 Fn = m_zero();
        int cls = h->mat_cls[p];
        if (cls == FE_MAT_LIQUID) {
            R c = std::pow(l.J, (R)(1.0 / 3.0));
            Fn = m_scale(m_ident(), c);
        } else if (cls == FE_MAT_ELASTIC || cls == FE_MAT_RIGID) {
            Fn = l.Ft;
        } else if (cls == FE_MAT_PLASTO_ELASTIC || cls == FE_MAT_PLASTO_ELASTIC_DEMO) {
            M3 Sn = m_zero();
            for (int d = 0; d < 3; d++) Sn.m[d][d] = std::min(std::max(l.S.m[d][d], (R)(1 - 2e-3)), (R)(1 + 3e-3));
            Fn = m_mul(m_mul(l.U, Sn), m_T(l.V));
        }
        m_store(&h->Ff(f + 1)[p * 9], Fn);
    });
    if (bad) { h->err = "particle stencil left the grid (p2g)"; return 1; }
    return 0;
}

/* Effector.move_kernel: effector.py:157-161 */
void effector_move(Effector& e, int f) {
    R xin[3] = {e.pos[f * 3] + e.v[f * 3], e.pos[f * 3 + 1] + e.v[f * 3 + 1], e.pos[f * 3 + 2] + e.v[f * 3 + 2]};
    R xn[3], J[3][3];
    impose_x(e.d.boundary, xin, xn, J);
    for (int d = 0; d < 3; d++) e.pos[(f + 1) * 3 + d] = xn[d];
    R qw[4]; w2quat(&e.w[f * 3], qw);
    qmul(qw, &e.quat[f * 4], &e.quat[(f + 1) * 4]);
}


/* ------------------------------------------------------------------ SDF colliders */

/* Static.sdf_ (static.py:34-49): trilinear sample in voxel coordinates, 1.0 outside the voxel box */
R sdf_sample(const Sdf& s, const R pv[3]) {
    int base[3];
    for (int d = 0; d < 3; d++) {
        base[d] = (int)std::floor(pv[d]);
        if (base[d] >= s.res - 1 || base[d] < 0) return (R)1.0;
    }
    R sd = 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        const int vp[3] = {base[0] + i, base[1] + j, base[2] + k};
        R w = 1;
        for (int d = 0; d < 3; d++) w *= 1 - std::fabs(pv[d] - vp[d]);
        sd += w * s.vox[((size_t)vp[0] * s.res + vp[1]) * s.res + vp[2]];
    }
    return sd;
}
void sdf_to_voxels(const Sdf& s, const R pm[3], R pv[3]) {                     /* geom.transform_by_T_ti */
    for (int d = 0; d < 3; d++) pv[d] = s.T[d * 4] * pm[0] + s.T[d * 4 + 1] * pm[1] + s.T[d * 4 + 2] * pm[2] + s.T[d * 4 + 3];
}
/* Static.normal (static.py:52-80): central differences with delta = 1e-2 in voxel space, normalised, rotated back by
 * inverse(T[:3,:3]), normalised again; normalize(v) = v / sqrt(|v|^2 + EPS) (geom.py:93-94) */
void sdf_normal(const Sdf& s, const R pv[3], R n[3]) {
    const R delta = (R)1e-2;
    R g[3];
    for (int d = 0; d < 3; d++) {
        R inc[3] = {pv[0], pv[1], pv[2]}, dec[3] = {pv[0], pv[1], pv[2]};
        inc[d] += delta; dec[d] -= delta;
        g[d] = (sdf_sample(s, inc) - sdf_sample(s, dec)) / (2 * delta);
    }
    R nn = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + EPS);
    for (int d = 0; d < 3; d++) g[d] /= nn;
    for (int d = 0; d < 3; d++) n[d] = s.Rinv[d * 3] * g[0] + s.Rinv[d * 3 + 1] * g[1] + s.Rinv[d * 3 + 2] * g[2];
    nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + EPS);
    for (int d = 0; d < 3; d++) n[d] /= nn;
}
/* The contact law shared by Static.collide (static.py:82-103) and Dynamic.collide (dynamic.py:104-122), on the
 * velocity relative to the collider: remove the inward normal part, Coulomb friction on the tangential part.
 * Conscious fix: the reference evaluates rel_v_t / |rel_v_t| unconditionally and multiplies by flag = 0 when
 * |rel_v_t| <= EPS, which is NaN * 0 = NaN for an exactly normal impact; here the quotient is only formed when flag = 1.
 * If `g` is given it is pulled back through the law (adjoint w.r.t. rel_v with the normal held fixed; Taichi's
 * min/max pass the gradient to the selected operand). */
void contact_law(const R n[3], R friction, const R rv[3], R out[3], R* g) {
    const R nc = rv[0] * n[0] + rv[1] * n[1] + rv[2] * n[2];
    const R a = std::min(nc, (R)0);
    R vt[3] = {rv[0] - a * n[0], rv[1] - a * n[1], rv[2] - a * n[2]};
    const R vtn = std::sqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
    const bool flag = nc < 0 && vtn > EPS;
    const R t = vtn + nc * friction;
    if (flag) {
        const R sc = std::max((R)0, t) / vtn;
        for (int d = 0; d < 3; d++) out[d] = vt[d] * sc;
    } else {
        for (int d = 0; d < 3; d++) out[d] = vt[d];
    }
    if (!g) return;
    R gvt[3], gnc = 0;
    if (flag && t > 0) {
        R u[3] = {vt[0] / vtn, vt[1] / vtn, vt[2] / vtn};
        const R ug = u[0] * g[0] + u[1] * g[1] + u[2] * g[2];
        /* out = vt + friction * nc * u */
        for (int d = 0; d < 3; d++) gvt[d] = g[d] + friction * nc * (g[d] - u[d] * ug) / vtn;
        gnc = friction * ug;
    } else if (flag) {
        gvt[0] = gvt[1] = gvt[2] = 0;
    } else {
        for (int d = 0; d < 3; d++) gvt[d] = g[d];
    }
    const R ga = -(n[0] * gvt[0] + n[1] * gvt[1] + n[2] * gvt[2]);
    if (nc < 0) gnc += ga;
    for (int d = 0; d < 3; d++) g[d] = gvt[d] + gnc * n[d];
}
/* Static.collide (static.py:82-103); with g != nullptr also pulls g back (in place) */
void static_collide(const Sdf& s, const R pos[3], R v[3], R* g) {
    R pv[3];
    sdf_to_voxels(s, pos, pv);
    if (sdf_sample(s, pv) > 0) return;
    R n[3], out[3];
    sdf_normal(s, pv, n);
    contact_law(n, s.friction, v, out, g);
    for (int d = 0; d < 3; d++) v[d] = out[d];
}
/* ------------------------------------------------------------------ dynamic colliders (Rigid effector)
 * Dynamic.collide (dynamic.py:29-122) is differentiated by Taichi through everything it touches: the material velocity,
 * the query position, and the effector pose at f and f+1 (sdf, finite-difference normal, quaternion transforms, contact
 * law, softness).  The restatement is written once over a scalar type T; with T = Dual (value + one tangent) a call
 * returns one column of the Jacobian, and the adjoint is assembled column by column (20 inputs).  Branch semantics:
 * floor/comparisons act on values; min/max pass the tangent of the selected operand (Taichi's rule). */
struct Dual { R v, d; Dual() : v(0), d(0) {} Dual(R v_) : v(v_), d(0) {} Dual(R v_, R d_) : v(v_), d(d_) {} };
inline Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
inline Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
inline Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
inline Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
inline Dual operator/(Dual a, Dual b) { return Dual(a.v / b.v, (a.d * b.v - a.v * b.d) / (b.v * b.v)); }
inline Dual tsqrt(Dual a) { R r = std::sqrt(a.v); return Dual(r, a.d / (2 * r)); }
inline Dual texp(Dual a) { R r = std::exp(a.v); return Dual(r, a.d * r); }
inline Dual tsin(Dual a) { return Dual(std::sin(a.v), a.d * std::cos(a.v)); }
inline Dual tcos(Dual a) { return Dual(std::cos(a.v), -a.d * std::sin(a.v)); }
inline Dual tabs(Dual a) { return a.v >= 0 ? a : -a; }
inline R tsqrt(R a) { return std::sqrt(a); }
inline R texp(R a) { return std::exp(a); }
inline R tsin(R a) { return std::sin(a); }
inline R tcos(R a) { return std::cos(a); }
inline R tabs(R a) { return std::fabs(a); }
inline R val(R a) { return a; }
inline R val(Dual a) { return a.v; }

template <class T> void t_quat_rot(const T v[3], const T q[4], T out[3]) {                   /* geom.py:97-102 */
    T uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
    T uuv[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
    for (int i = 0; i < 3; i++) out[i] = v[i] + T((R)2) * (q[0] * uv[i] + uuv[i]);
}
template <class T> void t_inv_quat(const T q[4], T out[4]) {                                  /* geom.py:30-32, .normalized() */
    T nn = tsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    out[0] = q[0] / nn; out[1] = -q[1] / nn; out[2] = -q[2] / nn; out[3] = -q[3] / nn;
}
template <class T> T t_sdf_sample(const Sdf& s, const T pv[3]) {                              /* dynamic.py:39-54 */
    int base[3];
    for (int d = 0; d < 3; d++) {
        base[d] = (int)std::floor(val(pv[d]));
        if (base[d] >= s.res - 1 || base[d] < 0) return T((R)1.0);
    }
    T sd((R)0);
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        const int vp[3] = {base[0] + i, base[1] + j, base[2] + k};
        T w((R)1);
        for (int d = 0; d < 3; d++) w = w * (T((R)1) - tabs(pv[d] - T((R)vp[d])));
        sd = sd + w * T(s.vox[((size_t)vp[0] * s.res + vp[1]) * s.res + vp[2]]);
    }
    return sd;
}
template <class T> void t_normalize(T v[3]) {                                                  /* geom.py:93-94 */
    T nn = tsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + T(EPS));
    for (int d = 0; d < 3; d++) v[d] = v[d] / nn;
}
template <class T> void t_sdf_normal_voxels(const Sdf& s, const T pv[3], T g[3]) {             /* dynamic.py:73-88 */
    const R delta = (R)1e-2;
    for (int d = 0; d < 3; d++) {
        T inc[3] = {pv[0], pv[1], pv[2]}, dec[3] = {pv[0], pv[1], pv[2]};
        inc[d] = inc[d] + T(delta); dec[d] = dec[d] - T(delta);
        g[d] = (t_sdf_sample(s, inc) - t_sdf_sample(s, dec)) / T(2 * delta);
    }
    t_normalize(g);
}
/* Dynamic.collide (dynamic.py:96-122) with the pose (p0,q0) = (pos,quat)[f], (p1,q1) = (pos,quat)[f+1].  Returns whether
 * the contact branch was taken. */
template <class T> bool t_dynamic_collide(const Sdf& s, const T p0[3], const T q0[4], const T p1[3], const T q1[4],
                                          const T pos[3], const T mv[3], R dt, T out[3]) {
    T rel0[3] = {pos[0] - p0[0], pos[1] - p0[1], pos[2] - p0[2]}, qi[4], pm[3], pv[3];
    t_inv_quat(q0, qi);
    t_quat_rot(rel0, qi, pm);                                                                  /* inv_transform_by_trans_quat */
    for (int d = 0; d < 3; d++) pv[d] = T(s.T[d * 4]) * pm[0] + T(s.T[d * 4 + 1]) * pm[1] + T(s.T[d * 4 + 2]) * pm[2] + T(s.T[d * 4 + 3]);
    T sd = t_sdf_sample(s, pv);
    T infl = texp(-sd * T(s.softness));
    if (val(infl) > 1) infl = T((R)1);                                                          /* min(exp(..), 1) */
    for (int d = 0; d < 3; d++) out[d] = mv[d];
    if (!(val(sd) <= 0 || (s.softness > 0 && val(infl) > (R)0.1))) return false;
    T pn[3], cv[3];
    t_quat_rot(pm, q1, pn);
    for (int d = 0; d < 3; d++) cv[d] = (pn[d] + p1[d] - pos[d]) / T(dt);                      /* collider_v, dynamic.py:90-94 */
    if (s.friction > (R)10.0) { for (int d = 0; d < 3; d++) out[d] = cv[d]; return true; }
    T rel[3] = {mv[0] - cv[0], mv[1] - cv[1], mv[2] - cv[2]};
    T gvx[3], nm[3], n[3];
    t_sdf_normal_voxels(s, pv, gvx);
    for (int d = 0; d < 3; d++) nm[d] = T(s.Rinv[d * 3]) * gvx[0] + T(s.Rinv[d * 3 + 1]) * gvx[1] + T(s.Rinv[d * 3 + 2]) * gvx[2];
    t_quat_rot(nm, q0, n);
    t_normalize(n);
    T nc = rel[0] * n[0] + rel[1] * n[1] + rel[2] * n[2];
    T a = val(nc) < 0 ? nc : T((R)0);
    T vt[3] = {rel[0] - a * n[0], rel[1] - a * n[1], rel[2] - a * n[2]};
    const R vtn_v = std::sqrt(val(vt[0]) * val(vt[0]) + val(vt[1]) * val(vt[1]) + val(vt[2]) * val(vt[2]));
    if (val(nc) < 0 && vtn_v > EPS) {                                                           /* flag; quotient only formed here */
        T vtn = tsqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
        T t = vtn + nc * T(s.friction);
        T sc = val(t) > 0 ? t / vtn : T((R)0);
        for (int d = 0; d < 3; d++) vt[d] = vt[d] * sc;
    }
    for (int d = 0; d < 3; d++) out[d] = cv[d] + vt[d] * infl + rel[d] * (T((R)1) - infl);
    return true;
}

/* agent.collide at particle level (mpm:419-422; AgentRigid.collide, agent_rigid.py:21-23): every effector that carries
 * a mesh, in order.  x_tmp = x + dt * new_v is re-formed from the current velocity before each collider. */
void agent_collide_particle(FeEngine* h, int f, const R x[3], R nv[3], bool at_node = false) {
    const R dt = h->cfg.dt;
    const R sdt = at_node ? (R)0 : dt;           /* at a grid node (mpm:393-395) the position is the node's, not x + dt v */
    for (const Effector& e : h->effs) {
        if (e.mesh < 0) continue;
        R pos[3] = {x[0] + sdt * nv[0], x[1] + sdt * nv[1], x[2] + sdt * nv[2]}, out[3];
        if (!(pos[1] > h->collide_min_y)) continue;
        t_dynamic_collide<R>(h->meshes[e.mesh], &e.pos[f * 3], &e.quat[f * 4], &e.pos[(f + 1) * 3], &e.quat[(f + 1) * 4], pos, nv, dt, out);
        for (int d = 0; d < 3; d++) nv[d] = out[d];
    }
}
/* its adjoint: g (d/d new_v after the colliders) is pulled back to d/d new_v before them; the x and pose parts are
 * accumulated.  Effector pose adjoints are shared by all particles, hence the critical section. */
void agent_collide_particle_grad(FeEngine* h, int f, const R x[3], const R nv0[3], R g[3], R gx[3], bool at_node = false) {
    const R dt = h->cfg.dt;
    const R sdt = at_node ? (R)0 : dt;
    std::vector<std::array<R, 3>> vin;
    R nv[3] = {nv0[0], nv0[1], nv0[2]};
    for (const Effector& e : h->effs) {
        if (e.mesh < 0) continue;
        vin.push_back({nv[0], nv[1], nv[2]});
        R pos[3] = {x[0] + sdt * nv[0], x[1] + sdt * nv[1], x[2] + sdt * nv[2]}, out[3];
        if (!(pos[1] > h->collide_min_y)) continue;
        t_dynamic_collide<R>(h->meshes[e.mesh], &e.pos[f * 3], &e.quat[f * 4], &e.pos[(f + 1) * 3], &e.quat[(f + 1) * 4], pos, nv, dt, out);
        for (int d = 0; d < 3; d++) nv[d] = out[d];
    }
    int k = (int)vin.size();
    for (int ei = (int)h->effs.size() - 1; ei >= 0; ei--) {
        Effector& e = h->effs[ei];
        if (e.mesh < 0) continue;
        k--;
        const R* v = vin[k].data();
        const Sdf& s = h->meshes[e.mesh];
        if (!(x[1] + sdt * v[1] > h->collide_min_y)) continue;
        R gin[3] = {0, 0, 0}, gpose[14];
        /* inputs: 0-2 new_v (enters as mat_v and, times dt, in the position), 3-5 x, 6-8 pos[f], 9-12 quat[f],
         *         13-15 pos[f+1], 16-19 quat[f+1] */
        bool hit = true;
        for (int dir = 0; dir < 20 && hit; dir++) {
            Dual p0[3], q0[4], p1[3], q1[4], pos[3], mv[3], out[3];
            for (int d = 0; d < 3; d++) {
                p0[d] = Dual(e.pos[f * 3 + d], dir == 6 + d ? 1 : 0);
                p1[d] = Dual(e.pos[(f + 1) * 3 + d], dir == 13 + d ? 1 : 0);
                mv[d] = Dual(v[d], dir == d ? 1 : 0);
                pos[d] = Dual(x[d] + sdt * v[d], dir == d ? sdt : (dir == 3 + d ? 1 : 0));
            }
            for (int d = 0; d < 4; d++) {
                q0[d] = Dual(e.quat[f * 4 + d], dir == 9 + d ? 1 : 0);
                q1[d] = Dual(e.quat[(f + 1) * 4 + d], dir == 16 + d ? 1 : 0);
            }
            hit = t_dynamic_collide<Dual>(s, p0, q0, p1, q1, pos, mv, dt, out);
            const R c = g[0] * out[0].d + g[1] * out[1].d + g[2] * out[2].d;
            if (dir < 3) gin[dir] = c; else if (dir < 6) { if (gx) gx[dir - 3] += c; } else gpose[dir - 6] = c;
        }
        if (!hit) continue;                       /* identity: g passes through unchanged */
        for (int d = 0; d < 3; d++) g[d] = gin[d];
#pragma omp critical(fe_effector_grad)
        {
            for (int d = 0; d < 3; d++) { e.gpos[f * 3 + d] += gpose[d]; e.gpos[(f + 1) * 3 + d] += gpose[7 + d]; }
            for (int d = 0; d < 4; d++) { e.gquat[f * 4 + d] += gpose[3 + d]; e.gquat[(f + 1) * 4 + d] += gpose[10 + d]; }
        }
    }
}
/* quaternion part of move_kernel (effector.py:161): quat[f+1] = qmul(w2quat(w[f]), quat[f]), and its adjoint */
template <class T> void t_move_quat(const T w[3], const T q[4], T out[4]) {
    T wn = tsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + T(EPS));                            /* geom.py:18-28 */
    T sh = tsin(wn / T((R)2)), a[4] = {tcos(wn / T((R)2)), w[0] / wn * sh, w[1] / wn * sh, w[2] / wn * sh};
    T t[4][4];                                                                                   /* qmul(a, q), geom.py:8-16 */
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t[i][j] = q[i] * a[j];
    T o[4] = {t[0][0] - t[1][1] - t[2][2] - t[3][3], t[0][1] + t[1][0] - t[2][3] + t[3][2],
              t[0][2] + t[1][3] + t[2][0] - t[3][1], t[0][3] - t[1][2] + t[2][1] + t[3][0]};
    T nn = tsqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int i = 0; i < 4; i++) out[i] = o[i] / nn;
}
void effector_move_quat_grad(Effector& e, int f) {
    const R* gq = &e.gquat[(f + 1) * 4];
    if (gq[0] == 0 && gq[1] == 0 && gq[2] == 0 && gq[3] == 0) return;
    for (int dir = 0; dir < 7; dir++) {
        Dual w[3], q[4], out[4];
        for (int d = 0; d < 3; d++) w[d] = Dual(e.w[f * 3 + d], dir == d ? 1 : 0);
        for (int d = 0; d < 4; d++) q[d] = Dual(e.quat[f * 4 + d], dir == 3 + d ? 1 : 0);
        t_move_quat<Dual>(w, q, out);
        const R c = gq[0] * out[0].d + gq[1] * out[1].d + gq[2] * out[2].d + gq[3] * out[3].d;
        if (dir < 3) e.gw[f * 3 + dir] += c; else e.gquat[f * 4 + dir - 3] += c;
    }
}

/* the collider chain of grid_op (mpm:386-390) for one node; returns the velocities before each static */
void statics_forward(FeEngine* h, const R xn[3], R vo[3], std::vector<R>* trace) {
    for (const Sdf& s : h->statics) {
        if (trace) { trace->push_back(vo[0]); trace->push_back(vo[1]); trace->push_back(vo[2]); }
        static_collide(s, xn, vo, nullptr);
    }
}

void grid_op(FeEngine* h, int f) {
    const int n = h->n;
    const bool dyn = h->has_mesh_effector && (h->collide_type & 2);
    const size_t n3 = (size_t)n * n * n;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)n3; c++) {
        R m = h->g_mass[c];
        if (m > EPS) {
            R vo[3];
            R inv = 1 / m;
            for (int a = 0; a < 3; a++) vo[a] = inv * h->g_vin[c * 3 + a] + h->cfg.dt * h->cfg.gravity[a];
            int i = (int)(c / ((size_t)n * n)), j = (int)((c / n) % n), k = (int)(c % n);
            R xn[3] = {i * h->dx, j * h->dx, k * h->dx};
            R kk[3];
            if (!h->statics.empty()) statics_forward(h, xn, vo, nullptr);                   /* mpm:386-390 */
            if (dyn) agent_collide_particle(h, f, xn, vo, true);                            /* mpm:393-395 */
            impose_v(h->cfg.boundary, xn, vo, kk);
            for (int a = 0; a < 3; a++) h->g_vout[c * 3 + a] = vo[a];
        }
    }
}

/* mpm:400-426 */
int g2p(FeEngine* h, int f) {
    const int N = h->N, n = h->n;
    int bad = 0;
#pragma omp parallel for num_threads(h->threads) schedule(static) reduction(+:bad)
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s);
        if (!stencil_in_grid(s, n)) { bad++; continue; }
        R nv[3] = {0, 0, 0}; M3 nC = m_zero();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
            R dpos[3] = {i - s.fx[0], j - s.fx[1], k - s.fx[2]};                            /* mpm:410 */
            size_t c = cell_index(n, s.base[0] + i, s.base[1] + j, s.base[2] + k);
            const R* gv = &h->g_vout[c * 3];
            R weight = (R)1.0; weight *= s.w[i][0]; weight *= s.w[j][1]; weight *= s.w[k][2];
            for (int a = 0; a < 3; a++) {
                nv[a] += weight * gv[a];
                for (int b = 0; b < 3; b++) nC.m[a][b] += 4 * h->inv_dx * weight * gv[a] * dpos[b];
            }
        }
        if (h->has_mesh_effector && (h->collide_type & 1)) agent_collide_particle(h, f, &h->X(f)[p * 3], nv);             /* mpm:418-422 */
        for (int a = 0; a < 3; a++) h->Vv(f + 1)[p * 3 + a] = nv[a];
        m_store(&h->Cc(f + 1)[p * 9], nC);
    }
    if (bad) { h->err = "particle stencil left the grid (g2p)"; return 1; }
    return 0;
}

/* MAT_RIGID shape matching, forward part shared by advect and advect_grad (mpm:428-434 / 436-441):
 * reset_bodies_and_grad (mpm:449-454), compute_COM (456-462), compute_H (464-478), compute_H_svd (480-483),
 * compute_R (491-495).  Body accumulation runs serially in particle order. */
void rigid_forward(FeEngine* h, int f) {
    const int N = h->N;
    const R dt = h->cfg.dt;
    for (int b = 0; b < h->n_bodies; b++) if (h->body_cls[b] == FE_MAT_RIGID) std::memset(&h->bodies[b], 0, sizeof(Body));
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p] || h->mat_cls[p] != FE_MAT_RIGID) continue;
        Body& B = h->bodies[h->body_id[p]];
        const R nb = (R)h->body_n[h->body_id[p]];
        for (int d = 0; d < 3; d++) {
            B.com0[d] += h->X(f)[p * 3 + d] / nb;
            B.com1[d] += (h->X(f)[p * 3 + d] + dt * h->Vv(f + 1)[p * 3 + d]) / nb;
        }
    }
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p] || h->mat_cls[p] != FE_MAT_RIGID) continue;
        Body& B = h->bodies[h->body_id[p]];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            B.H.m[i][j] += (h->X(f)[p * 3 + i] - B.com0[i]) * (h->X(f)[p * 3 + j] + dt * h->Vv(f + 1)[p * 3 + j] - B.com1[j]);
    }
    for (int b = 0; b < h->n_bodies; b++) {
        if (h->body_cls[b] != FE_MAT_RIGID) continue;
        Body& B = h->bodies[b];
        svd3(B.H, B.U, B.S, B.V);                       /* mpm:483 */
        B.Rm = m_mul(B.V, m_T(B.U));                    /* mpm:495 */
    }
}

/* advect, mpm:428-434 + advect_kernel mpm:497-505 */
void advect(FeEngine* h, int f) {
    const int N = h->N;
    if (h->has_rigid) rigid_forward(h, f);
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        if (h->mat_cls[p] == FE_MAT_RIGID) {
            const Body& B = h->bodies[h->body_id[p]];
            for (int i = 0; i < 3; i++) {
                R a = B.com1[i];
                for (int j = 0; j < 3; j++) a += B.Rm.m[i][j] * (h->X(f)[p * 3 + j] - B.com0[j]);
                h->X(f + 1)[p * 3 + i] = a;
            }
        } else {
            for (int d = 0; d < 3; d++) h->X(f + 1)[p * 3 + d] = h->X(f)[p * 3 + d] + h->cfg.dt * h->Vv(f + 1)[p * 3 + d];
        }
    }
}

/* ------------------------------------------------------------------ adjoint kernels */

/* advect_grad, mpm:436-447: advect_kernel.grad, compute_R.grad, compute_H_svd_grad (mpm:485-489), compute_H.grad,
 * compute_COM.grad -- what Taichi's autodiff derives from mpm:456-505, written out by hand */
void advect_grad(FeEngine* h, int f) {
    const int N = h->N;
    const R dt = h->cfg.dt;
    if (h->has_rigid) rigid_forward(h, f);
    /* advect_kernel.grad */
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        if (h->mat_cls[p] == FE_MAT_RIGID) {
            Body& B = h->bodies[h->body_id[p]];
            const R* g = &h->GX(f + 1)[p * 3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B.gR.m[i][j] += g[i] * (h->X(f)[p * 3 + j] - B.com0[j]);
            for (int j = 0; j < 3; j++) {
                R rtg = 0;
                for (int i = 0; i < 3; i++) rtg += B.Rm.m[i][j] * g[i];
                h->GX(f)[p * 3 + j] += rtg;
                B.gcom0[j] -= rtg;
                B.gcom1[j] += g[j];
            }
        } else {
            for (int d = 0; d < 3; d++) {
                R g = h->GX(f + 1)[p * 3 + d];
                h->GX(f)[p * 3 + d] += g;
                h->GV(f + 1)[p * 3 + d] += dt * g;
            }
        }
    }
    if (!h->has_rigid) return;
    for (int b = 0; b < h->n_bodies; b++) {
        if (h->body_cls[b] != FE_MAT_RIGID) continue;
        Body& B = h->bodies[b];
        B.gV = m_add(B.gV, m_mul(B.gR, B.U));            /* R = V U^T */
        B.gU = m_add(B.gU, m_mul(m_T(B.gR), B.V));
        B.gH = backward_svd(B.gU, B.gS, B.gV, B.U, B.S, B.V);   /* mpm:489 */
    }
    /* compute_H.grad */
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p] || h->mat_cls[p] != FE_MAT_RIGID) continue;
        Body& B = h->bodies[h->body_id[p]];
        R a[3], bb[3];
        for (int d = 0; d < 3; d++) {
            a[d] = h->X(f)[p * 3 + d] - B.com0[d];
            bb[d] = h->X(f)[p * 3 + d] + dt * h->Vv(f + 1)[p * 3 + d] - B.com1[d];
        }
        for (int d = 0; d < 3; d++) {
            R ga = 0, gb = 0;
            for (int e = 0; e < 3; e++) { ga += B.gH.m[d][e] * bb[e]; gb += B.gH.m[e][d] * a[e]; }
            h->GX(f)[p * 3 + d] += ga + gb;
            h->GV(f + 1)[p * 3 + d] += dt * gb;
            B.gcom0[d] -= ga;
            B.gcom1[d] -= gb;
        }
    }
    /* compute_COM.grad */
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p] || h->mat_cls[p] != FE_MAT_RIGID) continue;
        const Body& B = h->bodies[h->body_id[p]];
        const R nb = (R)h->body_n[h->body_id[p]];
        for (int d = 0; d < 3; d++) {
            h->GX(f)[p * 3 + d] += (B.gcom0[d] + B.gcom1[d]) / nb;
            h->GV(f + 1)[p * 3 + d] += dt * B.gcom1[d] / nb;
        }
    }
}

/* g2p.grad (mpm:538) */
void g2p_grad(FeEngine* h, int f) {
    const int n = h->n;
    scatter_loop(h, f, [&](int p, bool par) {
        if (!h->Us(f)[p]) return;
        Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s);
        if (!stencil_in_grid(s, n)) return;
        R gvn[3] = {h->GV(f + 1)[p * 3], h->GV(f + 1)[p * 3 + 1], h->GV(f + 1)[p * 3 + 2]};
        M3 gCn = m_load(&h->GC(f + 1)[p * 9]);
        R gfx[3] = {0, 0, 0};
        const R c4 = 4 * h->inv_dx;
        if (h->has_mesh_effector && (h->collide_type & 1)) {
            /* agent.collide's adjoint (mpm:418-422) comes first in reverse order: it needs the gathered velocity again */
            R nv[3] = {0, 0, 0}, gxc[3] = {0, 0, 0};
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
                size_t c = cell_index(n, s.base[0] + i, s.base[1] + j, s.base[2] + k);
                R weight = (R)1.0; weight *= s.w[i][0]; weight *= s.w[j][1]; weight *= s.w[k][2];
                for (int a = 0; a < 3; a++) nv[a] += weight * h->g_vout[c * 3 + a];
            }
            agent_collide_particle_grad(h, f, &h->X(f)[p * 3], nv, gvn, gxc);
            for (int d = 0; d < 3; d++) h->GX(f)[p * 3 + d] += gxc[d];
        }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
            int o[3] = {i, j, k};
            R dpos[3] = {i - s.fx[0], j - s.fx[1], k - s.fx[2]};
            size_t c = cell_index(n, s.base[0] + i, s.base[1] + j, s.base[2] + k);
            const R* gv = &h->g_vout[c * 3];
            R weight = s.w[i][0] * s.w[j][1] * s.w[k][2];
            /* q[a] = gv_next[a] + 4 inv_dx (gC_next dpos)[a] : adjoint of g_v[a] per unit weight */
            R q[3];
            for (int a = 0; a < 3; a++) q[a] = gvn[a] + c4 * (gCn.m[a][0] * dpos[0] + gCn.m[a][1] * dpos[1] + gCn.m[a][2] * dpos[2]);
            R sdot = gv[0] * q[0] + gv[1] * q[1] + gv[2] * q[2];
            for (int a = 0; a < 3; a++) {
                if (par) {
#pragma omp atomic
                    h->gg_vout[c * 3 + a] += weight * q[a];
                } else h->gg_vout[c * 3 + a] += weight * q[a];
            }
            /* weight path */
            for (int d = 0; d < 3; d++) {
                R dW = 1;
                for (int e = 0; e < 3; e++) dW *= (e == d) ? s.dw[o[e]][e] : s.w[o[e]][e];
                gfx[d] += dW * sdot;
            }
            /* dpos path: dpos_b = o_b - fx_b */
            for (int b = 0; b < 3; b++) gfx[b] -= c4 * weight * (gv[0] * gCn.m[0][b] + gv[1] * gCn.m[1][b] + gv[2] * gCn.m[2][b]);
        }
        for (int d = 0; d < 3; d++) h->GX(f)[p * 3 + d] += h->inv_dx * gfx[d];
    });
}

/* grid_op.grad (mpm:539) */
void grid_op_grad(FeEngine* h, int f) {
    const int n = h->n;
    const bool dyn = h->has_mesh_effector && (h->collide_type & 2);
    const size_t n3 = (size_t)n * n * n;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)n3; c++) {
        R m = h->g_mass[c];
        if (m > EPS) {
            R vo[3];
            R inv = 1 / m;
            for (int a = 0; a < 3; a++) vo[a] = inv * h->g_vin[c * 3 + a] + h->cfg.dt * h->cfg.gravity[a];
            int i = (int)(c / ((size_t)n * n)), j = (int)((c / n) % n), k = (int)(c % n);
            R xn[3] = {i * h->dx, j * h->dx, k * h->dx};
            R kk[3];
            std::vector<R> trace;
            if (!h->statics.empty()) statics_forward(h, xn, vo, &trace);
            const R vdyn[3] = {vo[0], vo[1], vo[2]};
            if (dyn) agent_collide_particle(h, f, xn, vo, true);
            impose_v(h->cfg.boundary, xn, vo, kk);
            R gv[3];
            for (int a = 0; a < 3; a++) gv[a] = h->gg_vout[c * 3 + a] * kk[a];
            if (dyn && (gv[0] != 0 || gv[1] != 0 || gv[2] != 0)) agent_collide_particle_grad(h, f, xn, vdyn, gv, nullptr, true);
            for (int si = (int)h->statics.size() - 1; si >= 0; si--) {                      /* colliders, in reverse */
                R vin[3] = {trace[si * 3], trace[si * 3 + 1], trace[si * 3 + 2]};
                static_collide(h->statics[si], xn, vin, gv);
            }
            R gm = 0;
            for (int a = 0; a < 3; a++) {
                R g = gv[a];
                h->gg_vin[c * 3 + a] += g * inv;
                gm += -h->g_vin[c * 3 + a] * g * inv * inv;
            }
            h->gg_mass[c] += gm;
        }
    }
}

/* Effector.move_kernel.grad (effector.py:154-155).  The quaternion branch
 * (quat[f+1] = qmul(w2quat(w[f]), quat[f])) carries no gradient to a 3-dim action
 * (effector.py:258-260 only writes w when action_dim>3); its adjoint is not restated. */
void effector_move_grad(Effector& e, int f) {
    R xin[3] = {e.pos[f * 3] + e.v[f * 3], e.pos[f * 3 + 1] + e.v[f * 3 + 1], e.pos[f * 3 + 2] + e.v[f * 3 + 2]};
    R xn[3], J[3][3];
    impose_x(e.d.boundary, xin, xn, J);
    for (int d = 0; d < 3; d++) {
        R g = 0;
        for (int i = 0; i < 3; i++) g += J[i][d] * e.gpos[(f + 1) * 3 + i];
        e.gpos[f * 3 + d] += g;
        e.gv[f * 3 + d] += g;
    }
    effector_move_quat_grad(e, f);
}

/* p2g.grad + svd_grad + compute_F_tmp.grad (mpm:544-546) */
void p2g_grad(FeEngine* h, int f) {
    const int N = h->N, n = h->n;
    const R dt = h->cfg.dt;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s);
        if (!stencil_in_grid(s, n)) continue;
        P2GLocal l; p2g_local(h, f, p, l);
        const R m = h->mass[p];
        const R* vp = &h->Vv(f)[p * 3];
        R Gv[3] = {0, 0, 0}; M3 GA = m_zero(); R gfx[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
            int o[3] = {i, j, k};
            R dpos[3] = {(i - s.fx[0]) * h->dx, (j - s.fx[1]) * h->dx, (k - s.fx[2]) * h->dx};
            size_t c = cell_index(n, s.base[0] + i, s.base[1] + j, s.base[2] + k);
            const R* gvin = &h->gg_vin[c * 3];
            R gms = h->gg_mass[c];
            R weight = s.w[i][0] * s.w[j][1] * s.w[k][2];
            R mom[3];
            for (int a = 0; a < 3; a++) mom[a] = m * vp[a] + (l.affine.m[a][0] * dpos[0] + l.affine.m[a][1] * dpos[1] + l.affine.m[a][2] * dpos[2]);
            R sdot = gvin[0] * mom[0] + gvin[1] * mom[1] + gvin[2] * mom[2] + gms * m;
            for (int a = 0; a < 3; a++) {
                Gv[a] += weight * gvin[a];
                for (int b = 0; b < 3; b++) GA.m[a][b] += weight * gvin[a] * dpos[b];
            }
            for (int d = 0; d < 3; d++) {
                R dW = 1;
                for (int e = 0; e < 3; e++) dW *= (e == d) ? s.dw[o[e]][e] : s.w[o[e]][e];
                gfx[d] += dW * sdot;
            }
            for (int b = 0; b < 3; b++) gfx[b] -= h->dx * weight * (gvin[0] * l.affine.m[0][b] + gvin[1] * l.affine.m[1][b] + gvin[2] * l.affine.m[2][b]);
        }
        for (int d = 0; d < 3; d++) {
            h->GX(f)[p * 3 + d] += h->inv_dx * gfx[d];
            h->GV(f)[p * 3 + d] += m * Gv[d];
        }
        m_accum(&h->GC(f)[p * 9], m_scale(GA, m));                     /* affine = stress + m C */
        /* stress adjoint */
        M3 gs = m_scale(GA, l.scale);                                    /* adjoint of stress_raw */
        M3 P = m_sub(l.Ft, l.r);
        R mu2 = 2 * h->mu[p];
        M3 gFt = m_scale(m_add(m_mul(gs, l.Ft), m_mul(m_T(gs), P)), mu2);
        M3 gr = m_scale(m_mul(gs, l.Ft), -mu2);
        M3 gU = m_mul(gr, l.V);
        M3 gV = m_mul(m_T(gr), l.U);
        R gJ = h->lam[p] * (2 * l.J - 1) * m_trace(gs);
        M3 gS = m_zero();
        /* F_new adjoint (mpm:355-378) */
        M3 Fg = m_load(&h->GF(f + 1)[p * 9]);
        int cls = h->mat_cls[p];
        if (cls == FE_MAT_LIQUID) {
            gJ += ((R)(1.0 / 3.0)) * std::pow(l.J, (R)(1.0 / 3.0 - 1.0)) * m_trace(Fg);
        } else if (cls == FE_MAT_ELASTIC || cls == FE_MAT_RIGID) {
            gFt = m_add(gFt, Fg);
        } else if (cls == FE_MAT_PLASTO_ELASTIC || cls == FE_MAT_PLASTO_ELASTIC_DEMO) {
            const R lo = (R)(1 - 2e-3), hi = (R)(1 + 3e-3);
            M3 Sn = m_zero();
            for (int d = 0; d < 3; d++) Sn.m[d][d] = std::min(std::max(l.S.m[d][d], lo), hi);
            M3 UtFgV = m_mul(m_mul(m_T(l.U), Fg), l.V);
            for (int d = 0; d < 3; d++) {
                R sd = l.S.m[d][d];
                R mx = std::max(sd, lo);
                bool pass = (sd > lo) && (mx < hi);     /* Taichi max/min adjoint tie rules */
                if (pass) gS.m[d][d] += UtFgV.m[d][d];
            }
            gU = m_add(gU, m_mul(m_mul(Fg, l.V), Sn));
            gV = m_add(gV, m_mul(m_mul(m_T(Fg), l.U), Sn));
        }
        /* J = det(S) (mpm:339) */
        gS.m[0][0] += gJ * l.S.m[1][1] * l.S.m[2][2];
        gS.m[1][1] += gJ * l.S.m[0][0] * l.S.m[2][2];
        gS.m[2][2] += gJ * l.S.m[0][0] * l.S.m[1][1];
        /* svd_grad (mpm:266-270) */
        gFt = m_add(gFt, backward_svd(gU, gS, gV, l.U, l.S, l.V));
        /* compute_F_tmp.grad: F_tmp = (I + dt C) F (mpm:258) */
        M3 Cm = m_load(&h->Cc(f)[p * 9]), Fm = m_load(&h->Ff(f)[p * 9]);
        m_accum(&h->GC(f)[p * 9], m_scale(m_mul(gFt, m_T(Fm)), dt));
        m_accum(&h->GF(f)[p * 9], m_mul(m_T(m_add(m_ident(), m_scale(Cm, dt))), gFt));
    }
}

/* AgentInjector.act_kernel.grad (agent_injector.py:27-28): x[f+1,pid] = offset + pos[f] + R(q) inject_p,
 * v[f+1,pid] = R(q) inject_v.  Position gradient only (see effector_move_grad). */
void injector_act_grad(FeEngine* h, Effector& e, int f) {
    const int flux = e.d.flux;
    for (int i = 0; i < flux; i++) {
        int idx = e.act_id[f] + i;
        if (idx >= (int)e.act_range.size()) break;
        int pid = e.act_range[idx];
        for (int d = 0; d < 3; d++) e.gpos[f * 3 + d] += h->GX(f + 1)[pid * 3 + d];
        /* inject_p and inject_v are rotated by quat[f] (injector.py:92-96): one forward-mode pass per quaternion component */
        for (int k = 0; k < 4; k++) {
            Dual q[4], ip[3], iv[3], rp[3], rv[3];
            for (int d = 0; d < 4; d++) q[d] = Dual(e.quat[f * 4 + d], d == k ? 1 : 0);
            for (int d = 0; d < 3; d++) { ip[d] = Dual((R)e.d.inject_p[d]); iv[d] = Dual((R)e.d.inject_v[d]); }
            t_quat_rot(ip, q, rp); t_quat_rot(iv, q, rv);
            R c = 0;
            for (int d = 0; d < 3; d++) c += h->GX(f + 1)[pid * 3 + d] * rp[d].d + h->GV(f + 1)[pid * 3 + d] * rv[d].d;
            e.gquat[f * 4 + k] += c;
        }
    }
}

/* process_unused_particles.grad (mpm:551) */
void process_unused_particles_grad(FeEngine* h, int f) {
    const int N = h->N;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (int p = 0; p < N; p++) {
        if (h->Us(f)[p] == 0) {
            for (int d = 0; d < 3; d++) { h->GV(f)[p * 3 + d] += h->GV(f + 1)[p * 3 + d]; h->GX(f)[p * 3 + d] += h->GX(f + 1)[p * 3 + d]; }
            for (int d = 0; d < 9; d++) { h->GC(f)[p * 9 + d] += h->GC(f + 1)[p * 9 + d]; h->GF(f)[p * 9 + d] += h->GF(f + 1)[p * 9 + d]; }
        }
    }
}

/* Boundary.is_out (boundaries.py:80-93 cylinder, 127-134 cube) */
bool boundary_is_out(const FeBoundary& b, const R x[3]) {
    if (b.type == FE_BOUNDARY_CYLINDER) {
        bool out = x[1] > (R)b.upper[1] || x[1] < (R)b.lower[1];
        const R rx = x[0] - (R)b.xz_center[0], rz = x[2] - (R)b.xz_center[1];
        if (std::sqrt(rx * rx + rz * rz + EPS) > (R)b.xz_radius) out = true;
        return out;
    }
    for (int d = 0; d < 3; d++) if (x[d] > (R)b.upper[d] || x[d] < (R)b.lower[d]) return true;
    return false;
}

/* collector_act_kernel (agent_pouring.py:30-41, agent_jetbot.py:33-43): a used particle outside the collector boundary is
 * parked at NOWHERE in frame f+1 and marked unused in BOTH frames, so the rest of substep f skips it.  The reference leaves
 * v, C, F of frame f+1 as they were (stale); here they are the copies process_unused_particles would have made, so that
 * the state is defined.  Its .grad contributes nothing (x[f+1] is a constant store, `used` carries no gradient); the
 * backward pass then sees used[f] == 0 and process_unused_particles.grad passes the adjoint straight through. */
void collector_act(FeEngine* h, int f) {
    const int N = h->N;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (int p = 0; p < N; p++) {
        if (!h->Us(f)[p]) continue;
        if (h->collector_mat >= 0 && h->mat[p] != h->collector_mat) continue;
        if (!boundary_is_out(h->collector, &h->X(f)[p * 3])) continue;
        h->Us(f + 1)[p] = 0; h->Us(f)[p] = 0;
        for (int d = 0; d < 3; d++) { h->X(f + 1)[p * 3 + d] = (R)-100; h->Vv(f + 1)[p * 3 + d] = h->Vv(f)[p * 3 + d]; }
        for (int d = 0; d < 9; d++) { h->Cc(f + 1)[p * 9 + d] = h->Cc(f)[p * 9 + d]; h->Ff(f + 1)[p * 9 + d] = h->Ff(f)[p * 9 + d]; }
    }
}

int substep(FeEngine* h, int f, int f_global, int act) {
    /* mpm:515-533 */
    reset_grid_and_grad(h);
    advect_used(h, f);
    process_unused_particles(h, f);
    const bool inject = act && !(h->inject_till >= 0 && f_global >= h->inject_till);
    for (auto& e : h->effs) if (e.d.type == FE_EFF_INJECTOR) {
        if (inject) { if (injector_act(h, e, f, f_global)) return 1; }
        else e.act_id[f + 1] = e.act_id[f];
    }
    if (act && h->has_collector) collector_act(h, f);
    compute_F_tmp_svd(h, f);
    if (p2g(h, f, true)) return 1;
    if (act) for (auto& e : h->effs) effector_move(e, f);
    grid_op(h, f);
    if (g2p(h, f)) return 1;
    advect(h, f);
    return 0;
}

int substep_grad(FeEngine* h, int f, int f_global, int act) {
    /* The reference keeps grid[f] and F_tmp/U/S/V[f] of every frame (mpm:106,117); this
     * restatement keeps one grid and recomputes those forward values of frame f first. */
    reset_grid_and_grad(h);
    compute_F_tmp_svd(h, f);
    if (p2g(h, f, false)) return 1;
    grid_op(h, f);
    /* mpm:535-552 */
    advect_grad(h, f);
    g2p_grad(h, f);
    grid_op_grad(h, f);
    if (act) for (int i = (int)h->effs.size() - 1; i >= 0; i--) effector_move_grad(h->effs[i], f);
    p2g_grad(h, f);
    if (act && !(h->inject_till >= 0 && f_global >= h->inject_till)) for (auto& e : h->effs) if (e.d.type == FE_EFF_INJECTOR) injector_act_grad(h, e, f);
    process_unused_particles_grad(h, f);
    return 0;
}

/* ====================================================================== smoke field
 * fluidlab/fluidengine/simulators/smoke_field.py restated kernel by kernel, with the adjoint Taichi's autodiff derives
 * from it written out by hand.  Positions are in cell units (cell centre = index + 0.5), velocities in cells per unit
 * time.  Conscious fix: compute_location (smoke_field.py:298-306) falls back to the *unclamped* index when the clamped
 * cell is not free, which reads out of bounds for indices outside the grid; here the fallback is the clamped index. */

static inline int sm_clampi(int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); }
/* compute_location, smoke_field.py:298-306 */
static size_t sm_loc(Smoke& m, const unsigned char* fr, int u, int v, int w, int du, int dv, int dw) {
    int I[3] = {sm_clampi(u + du, m.n), sm_clampi(v + dv, m.n), sm_clampi(w + dw, m.n)};
    if (!fr[m.idx(I[0], I[1], I[2])]) { I[0] = sm_clampi(u, m.n); I[1] = sm_clampi(v, m.n); I[2] = sm_clampi(w, m.n); }
    return m.idx(I[0], I[1], I[2]);
}
/* is_free, smoke_field.py:309-320 */
static int sm_isfree(Smoke& m, const unsigned char* fr, int u, int v, int w, int du, int dv, int dw) {
    int I[3] = {u + du, v + dv, w + dw};
    for (int d = 0; d < 3; d++) if (I[d] < 0 || I[d] > m.n - 1) return 0;
    return fr[m.idx(I[0], I[1], I[2])] ? 1 : 0;
}
/* trilerp, smoke_field.py:322-343: value of a `comps`-component field at p, the 8 cells/weights used, and dw/dp */
struct SmTri { size_t cell[8]; R w[8]; R dw[8][3]; };
static void sm_trilerp(Smoke& m, const unsigned char* fr, const R* field, int comps, const R p[3], R* out, SmTri* t) {
    int base[3]; R fr_[3];
    for (int d = 0; d < 3; d++) { base[d] = (int)std::floor(p[d] - (R)0.5); fr_[d] = p[d] - (R)0.5 - base[d]; }
    for (int c = 0; c < comps; c++) out[c] = 0;
    R wt = 0;
    int o = 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++, o++) {
        const int off[3] = {i, j, k};
        R wd[3], dwd[3];
        for (int d = 0; d < 3; d++) { wd[d] = 1 - std::fabs(fr_[d] - off[d]); dwd[d] = off[d] ? (R)1 : (R)-1; }   /* fr_ in [0,1) */
        const R w = wd[0] * wd[1] * wd[2];
        const size_t cell = sm_loc(m, fr, base[0] + i, base[1] + j, base[2] + k, 0, 0, 0);
        for (int c = 0; c < comps; c++) out[c] += w * field[cell * comps + c];
        wt += w;
        if (t) { t->cell[o] = cell; t->w[o] = w; t->dw[o][0] = dwd[0] * wd[1] * wd[2]; t->dw[o][1] = wd[0] * dwd[1] * wd[2]; t->dw[o][2] = wd[0] * wd[1] * dwd[2]; }
    }
    for (int c = 0; c < comps; c++) out[c] /= wt;              /* wt == 1 up to rounding: the two weights per axis sum to 1 */
}
/* adjoint of one trilerp: scatters g (d/d value) into gfield, returns d/dp */
static void sm_trilerp_grad(const SmTri& t, const R* field, R* gfield, int comps, const R* g, R gp[3]) {
    gp[0] = gp[1] = gp[2] = 0;
    for (int o = 0; o < 8; o++) {
        R dot = 0;
        for (int c = 0; c < comps; c++) {
            dot += field[t.cell[o] * comps + c] * g[c];
            if (gfield) {
#pragma omp atomic
                gfield[t.cell[o] * comps + c] += t.w[o] * g[c];
            }
        }
        for (int d = 0; d < 3; d++) gp[d] += t.dw[o][d] * dot;
    }
}

/* compute_free_space, smoke_field.py:191-201 */
static void smoke_free_space(FeEngine* h, Smoke& m, int s) {
    unsigned char* fr = m.FR(s);
    const R dx = (R)1 / m.n;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)m.n3; c++) {
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        unsigned char f = (m.c.lower_y < j && j < m.c.higher_y) ? 1 : 0;
        if (f) {
            const R pw[3] = {(i + (R)0.5) * dx, (j + (R)0.5) * dx, (k + (R)0.5) * dx};
            for (const Sdf& st : h->statics) { R pv[3]; sdf_to_voxels(st, pw, pv); if (sdf_sample(st, pv) <= 0) f = 0; }   /* is_collide, static.py:105-113 */
        }
        fr[c] = f;
    }
}
static Effector* smoke_aircon(FeEngine* h) {
    for (auto& e : h->effs) if (e.d.type == FE_EFF_AIRCON) return &e;
    return nullptr;
}
/* per-cell pieces of advect_and_impulse shared by forward and adjoint */
struct SmAdv { R p0[3], v1[3], p1[3], v2[3], p2[3], v3[3], pf[3], vf[3]; SmTri t1, t2, t3, tv, tq; R dist, factor, dir[3]; };
static void smoke_advect_cell(Smoke& m, Effector& a, int s, int f, int i, int j, int k, SmAdv& A, R* qf) {
    const unsigned char* fr = m.FR(s);
    const R* vfield = m.F(m.v, s, 3);
    const R dt = m.c.dt;
    A.p0[0] = i + (R)0.5; A.p0[1] = j + (R)0.5; A.p0[2] = k + (R)0.5;
    sm_trilerp(m, fr, vfield, 3, A.p0, A.v1, &A.t1);                                          /* backtrace, RK3: 347-360 */
    for (int d = 0; d < 3; d++) A.p1[d] = A.p0[d] - (R)0.5 * dt * A.v1[d];
    sm_trilerp(m, fr, vfield, 3, A.p1, A.v2, &A.t2);
    for (int d = 0; d < 3; d++) A.p2[d] = A.p0[d] - (R)0.75 * dt * A.v2[d];
    sm_trilerp(m, fr, vfield, 3, A.p2, A.v3, &A.t3);
    for (int d = 0; d < 3; d++) A.pf[d] = A.p0[d] - dt * (((R)2 / 9) * A.v1[d] + ((R)1 / 3) * A.v2[d] + ((R)4 / 9) * A.v3[d]);
    sm_trilerp(m, fr, vfield, 3, A.pf, A.vf, &A.tv);
    sm_trilerp(m, fr, m.F(m.q, s, m.c.q_dim), m.c.q_dim, A.pf, qf, &A.tq);
    /* agent impulse, 213-218 */
    const R dx = (R)1 / m.n;
    R dd[3] = {i - a.pos[f * 3] / dx, j - a.pos[f * 3 + 1] / dx, k - a.pos[f * 3 + 2] / dx};
    A.dist = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] + EPS);
    A.factor = std::exp(-A.dist / a.ra[f]);
    transform_by_quat(a.d.inject_v, &a.quat[f * 4], A.dir);
}
/* advect_and_impulse, smoke_field.py:203-232 */
static void smoke_advect(FeEngine* h, Smoke& m, Effector& a, int s, int f) {
    const unsigned char* fr = m.FR(s);
    const int qd = m.c.q_dim;
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)m.n3; c++) {
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        R* vt = &m.F(m.vt, s, 3)[c * 3];
        R* qn = &m.F(m.q, s + 1, qd)[c * qd];
        if (fr[c]) {
            SmAdv A; R qf[8];
            smoke_advect_cell(m, a, s, f, i, j, k, A, qf);
            for (int d = 0; d < 3; d++) vt[d] = A.vf[d] + A.dir[d] * a.sa[f] * A.factor * m.c.dt;
            for (int d = 0; d < qd; d++) qn[d] = (1 - A.factor) * qf[d] + A.factor * m.c.low_T;     /* ti.Vector([low_T]) broadcasts */
        } else {
            vt[0] = vt[1] = vt[2] = 0;
            for (int d = 0; d < qd; d++) qn[d] = m.F(m.q, s, qd)[c * qd + d];
        }
    }
}
static void smoke_advect_grad(FeEngine* h, Smoke& m, Effector& a, int s, int f) {
    const unsigned char* fr = m.FR(s);
    const int qd = m.c.q_dim;
    const R dt = m.c.dt, dx = (R)1 / m.n;
    R* gvfield = m.F(m.gv, s, 3);
    R* gqfield = m.F(m.gq, s, qd);
    const R* vfield = m.F(m.v, s, 3);
    const R* qfield = m.F(m.q, s, qd);
    R gpos[3] = {0, 0, 0}, gquat[4] = {0, 0, 0, 0}, gs = 0, gr = 0;
#pragma omp parallel for num_threads(h->threads) schedule(static) reduction(+:gpos[:3], gquat[:4], gs, gr)
    for (long long c = 0; c < (long long)m.n3; c++) {
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        const R* gvt = &m.F(m.gvt, s, 3)[c * 3];
        const R* gqn = &m.F(m.gq, s + 1, qd)[c * qd];
        if (!fr[c]) {
            for (int d = 0; d < qd; d++) {
#pragma omp atomic
                gqfield[c * qd + d] += gqn[d];
            }
            continue;
        }
        SmAdv A; R qf[8];
        smoke_advect_cell(m, a, s, f, i, j, k, A, qf);
        /* v_tmp = v_f + dir s factor dt ; q' = (1 - factor) q_f + factor low_T */
        R gvf[3] = {gvt[0], gvt[1], gvt[2]}, gqf[8], gfac = 0, gdir[3];
        for (int d = 0; d < qd; d++) { gqf[d] = (1 - A.factor) * gqn[d]; gfac += gqn[d] * (m.c.low_T - qf[d]); }
        for (int d = 0; d < 3; d++) { gfac += gvt[d] * A.dir[d] * a.sa[f] * dt; gdir[d] = gvt[d] * a.sa[f] * A.factor * dt; gs += gvt[d] * A.dir[d] * A.factor * dt; }
        /* factor = exp(-dist / r) */
        const R gdist = gfac * A.factor * (-1 / a.ra[f]);
        gr += gfac * A.factor * A.dist / (a.ra[f] * a.ra[f]);
        R dd[3] = {i - a.pos[f * 3] / dx, j - a.pos[f * 3 + 1] / dx, k - a.pos[f * 3 + 2] / dx};
        for (int d = 0; d < 3; d++) gpos[d] += gdist * (dd[d] / A.dist) * (-1 / dx);
        /* dir = R(quat) inject_v: one forward-mode column per quaternion component */
        for (int qc = 0; qc < 4; qc++) {
            Dual q[4], vin[3], out[3];
            for (int d = 0; d < 4; d++) q[d] = Dual(a.quat[f * 4 + d], d == qc ? 1 : 0);
            for (int d = 0; d < 3; d++) vin[d] = Dual(a.d.inject_v[d]);
            t_quat_rot(vin, q, out);
            gquat[qc] += gdir[0] * out[0].d + gdir[1] * out[1].d + gdir[2] * out[2].d;
        }
        /* the two interpolations at the back-traced point, then the RK3 chain in reverse */
        R gpf[3], t[3], gp2[3], gp1[3], gv1[3], gv2[3], gv3[3];
        sm_trilerp_grad(A.tv, vfield, gvfield, 3, gvf, gpf);
        sm_trilerp_grad(A.tq, qfield, gqfield, qd, gqf, t);
        for (int d = 0; d < 3; d++) gpf[d] += t[d];
        for (int d = 0; d < 3; d++) { gv1[d] = -dt * ((R)2 / 9) * gpf[d]; gv2[d] = -dt * ((R)1 / 3) * gpf[d]; gv3[d] = -dt * ((R)4 / 9) * gpf[d]; }
        sm_trilerp_grad(A.t3, vfield, gvfield, 3, gv3, gp2);
        for (int d = 0; d < 3; d++) gv2[d] += -(R)0.75 * dt * gp2[d];
        sm_trilerp_grad(A.t2, vfield, gvfield, 3, gv2, gp1);
        for (int d = 0; d < 3; d++) gv1[d] += -(R)0.5 * dt * gp1[d];
        sm_trilerp_grad(A.t1, vfield, gvfield, 3, gv1, t);           /* p0 is a constant: d/dp dropped */
    }
    for (int d = 0; d < 3; d++) a.gpos[f * 3 + d] += gpos[d];
    for (int d = 0; d < 4; d++) a.gquat[f * 4 + d] += gquat[d];
    a.gsa[f] += gs; a.gra[f] += gr;
}
/* divergence, smoke_field.py:234-258, and its adjoint */
static void smoke_divergence(FeEngine* h, Smoke& m, int s, bool grad) {
    const unsigned char* fr = m.FR(s);
    const R* vt = m.F(m.vt, s, 3);
    R* gvt = m.F(m.gvt, s, 3);
    static const int NB[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)m.n3; c++) {
        if (!fr[c]) continue;
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        if (!grad) {
            R val[6];
            for (int b = 0; b < 6; b++) {
                const int ax = b / 2;
                if (!sm_isfree(m, fr, i, j, k, NB[b][0], NB[b][1], NB[b][2])) val[b] = -vt[c * 3 + ax];
                else val[b] = vt[sm_loc(m, fr, i, j, k, NB[b][0], NB[b][1], NB[b][2]) * 3 + ax];
            }
            m.F(m.dv, s, 1)[c] = (val[1] - val[0] + val[3] - val[2] + val[5] - val[4]) * (R)0.5;
        } else {
            const R g = m.F(m.gdv, s, 1)[c] * (R)0.5;
            for (int b = 0; b < 6; b++) {
                const int ax = b / 2;
                const R sg = (b & 1) ? g : -g;
                if (!sm_isfree(m, fr, i, j, k, NB[b][0], NB[b][1], NB[b][2])) {
#pragma omp atomic
                    gvt[c * 3 + ax] += -sg;
                } else {
                    const size_t cn = sm_loc(m, fr, i, j, k, NB[b][0], NB[b][1], NB[b][2]);
#pragma omp atomic
                    gvt[cn * 3 + ax] += sg;
                }
            }
        }
    }
}
/* pressure_jacobi, smoke_field.py:130-143 (new_pf only written on free cells) and its adjoint (autodiff of the same) */
static void smoke_jacobi(FeEngine* h, Smoke& m, int s, const std::vector<R>& pf, std::vector<R>& npf) {
    const unsigned char* fr = m.FR(s);
    const R* dv = m.F(m.dv, s, 1);
    static const int NB[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)m.n3; c++) {
        if (!fr[c]) continue;
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        R sum = 0;
        for (int b = 0; b < 6; b++) sum += pf[sm_loc(m, fr, i, j, k, NB[b][0], NB[b][1], NB[b][2])];
        npf[c] = (sum - dv[c]) / (R)6.0;
    }
}
static void smoke_jacobi_grad(FeEngine* h, Smoke& m, int s, std::vector<R>& gpf, const std::vector<R>& gnpf) {
    const unsigned char* fr = m.FR(s);
    R* gdv = m.F(m.gdv, s, 1);
    static const int NB[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)m.n3; c++) {
        if (!fr[c]) continue;
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        const R g = gnpf[c] / (R)6.0;
        gdv[c] += -g;
        for (int b = 0; b < 6; b++) {
            const size_t cn = sm_loc(m, fr, i, j, k, NB[b][0], NB[b][1], NB[b][2]);
#pragma omp atomic
            gpf[cn] += g;
        }
    }
}
/* subtract_gradient, smoke_field.py:273-288 */
static void smoke_subtract(FeEngine* h, Smoke& m, int s, bool grad) {
    const unsigned char* fr = m.FR(s);
    static const int NB[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    const R* pn = m.F(m.p, s + 1, 1);
    R* gpn = m.F(m.gp, s + 1, 1);
#pragma omp parallel for num_threads(h->threads) schedule(static)
    for (long long c = 0; c < (long long)m.n3; c++) {
        const int i = (int)(c / ((size_t)m.n * m.n)), j = (int)((c / m.n) % m.n), k = (int)(c % m.n);
        const R* vt = &m.F(m.vt, s, 3)[c * 3];
        if (!grad) {
            R* vn = &m.F(m.v, s + 1, 3)[c * 3];
            for (int d = 0; d < 3; d++) vn[d] = vt[d];
            if (fr[c]) for (int d = 0; d < 3; d++)
                vn[d] -= (R)0.5 * (pn[sm_loc(m, fr, i, j, k, NB[2 * d + 1][0], NB[2 * d + 1][1], NB[2 * d + 1][2])] -
                                   pn[sm_loc(m, fr, i, j, k, NB[2 * d][0], NB[2 * d][1], NB[2 * d][2])]);
        } else {
            const R* gvn = &m.F(m.gv, s + 1, 3)[c * 3];
            R* gvt = &m.F(m.gvt, s, 3)[c * 3];
            for (int d = 0; d < 3; d++) {
#pragma omp atomic
                gvt[d] += gvn[d];
            }
            if (fr[c]) for (int d = 0; d < 3; d++) {
                const size_t cr = sm_loc(m, fr, i, j, k, NB[2 * d + 1][0], NB[2 * d + 1][1], NB[2 * d + 1][2]);
                const size_t cl = sm_loc(m, fr, i, j, k, NB[2 * d][0], NB[2 * d][1], NB[2 * d][2]);
#pragma omp atomic
                gpn[cr] += -(R)0.5 * gvn[d];
#pragma omp atomic
                gpn[cl] += (R)0.5 * gvn[d];
            }
        }
    }
}
/* SmokeField.step, smoke_field.py:95-111 */
static int smoke_step(FeEngine* h, int s, int f) {
    Smoke& m = *h->smoke;
    Effector* a = smoke_aircon(h);
    if (!a) FE_FAIL(h, "smoke_step needs an AirCon effector (agent.aircon, smoke_field.py:213)");
    smoke_free_space(h, m, s);
    smoke_advect(h, m, *a, s, f);
    smoke_divergence(h, m, s, false);
    const unsigned char* fr = m.FR(s);
    std::fill(m.pc.begin(), m.pc.end(), (R)0); std::fill(m.pn.begin(), m.pn.end(), (R)0);          /* reset_swap_and_grad */
    for (size_t c = 0; c < m.n3; c++) if (fr[c]) m.pc[c] = m.F(m.p, s, 1)[c];                     /* pressure_to_swap */
    for (int it = 0; it < m.c.solver_iters; it++) { smoke_jacobi(h, m, s, m.pc, m.pn); std::swap(m.pc, m.pn); }
    for (size_t c = 0; c < m.n3; c++) if (fr[c]) m.F(m.p, s + 1, 1)[c] = m.pc[c];                 /* pressure_from_swap */
    smoke_subtract(h, m, s, false);
    return 0;
}
/* SmokeField.step_grad, smoke_field.py:113-128 */
static int smoke_step_grad(FeEngine* h, int s, int f) {
    Smoke& m = *h->smoke;
    Effector* a = smoke_aircon(h);
    if (!a) FE_FAIL(h, "smoke_step_grad needs an AirCon effector");
    smoke_free_space(h, m, s);
    smoke_subtract(h, m, s, true);
    const unsigned char* fr = m.FR(s);
    std::fill(m.gpc.begin(), m.gpc.end(), (R)0); std::fill(m.gpn.begin(), m.gpn.end(), (R)0);
    for (size_t c = 0; c < m.n3; c++) if (fr[c]) m.gpc[c] += m.F(m.gp, s + 1, 1)[c];              /* pressure_from_swap.grad */
    for (int it = m.c.solver_iters - 1; it >= 0; it--) {
        std::swap(m.gpc, m.gpn);                                                                    /* p_swap.swap(); cur.grad.fill(0) */
        std::fill(m.gpc.begin(), m.gpc.end(), (R)0);
        smoke_jacobi_grad(h, m, s, m.gpc, m.gpn);
    }
    for (size_t c = 0; c < m.n3; c++) if (fr[c]) m.F(m.gp, s, 1)[c] += m.gpc[c];                  /* pressure_to_swap.grad */
    smoke_divergence(h, m, s, true);
    smoke_advect_grad(h, m, *a, s, f);
    return 0;
}
static void smoke_reset_grad(FeEngine* h) {
    Smoke& m = *h->smoke;
    for (auto* t : {&m.gv, &m.gvt, &m.gdv, &m.gp, &m.gq, &m.gpc, &m.gpn}) std::fill(t->begin(), t->end(), (R)0);
}

double now_ms(FeEngine* h) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h->t0).count();
}

} // namespace

/* ====================================================================== C ABI */
extern "C" {

const char* fe_backend(void) { return sizeof(R) == 4 ? "oracle-f32" : "oracle-f64"; }
int fe_real_size(void) { return (int)sizeof(R); }

FeEngine* fe_create(const FeConfig* cfg) {
    if (!cfg || cfg->struct_size != (int)sizeof(FeConfig)) { g_create_err = "FeConfig size mismatch"; return nullptr; }
    if (cfg->n_grid < 4 || cfg->n_particles < 0 || cfg->max_substeps_local < 1 || cfg->n_substeps < 1) { g_create_err = "invalid FeConfig"; return nullptr; }
    FeEngine* h = new FeEngine();
    h->cfg = *cfg; h->N = cfg->n_particles; h->L = cfg->max_substeps_local; h->n = cfg->n_grid;
    h->dx = (R)1 / cfg->n_grid; h->inv_dx = (R)cfg->n_grid;
    size_t N = h->N, Fm = h->L + 1, n3 = (size_t)h->n * h->n * h->n;
    try {
        h->x.assign(Fm * N * 3, 0); h->v.assign(Fm * N * 3, 0); h->C.assign(Fm * N * 9, 0); h->F.assign(Fm * N * 9, 0);
        h->gx.assign(Fm * N * 3, 0); h->gv.assign(Fm * N * 3, 0); h->gC.assign(Fm * N * 9, 0); h->gF.assign(Fm * N * 9, 0);
        h->used.assign(Fm * N, 0);
        for (auto* t : {&h->Ftmp, &h->U, &h->V, &h->S, &h->gFtmp, &h->gU, &h->gV, &h->gS}) t->assign(N * 9, 0);
        h->mu.assign(N, 0); h->lam.assign(N, 0); h->mass.assign(N, 0);
        h->mat.assign(N, 0); h->mat_cls.assign(N, 0); h->body_id.assign(N, 0);
        h->g_vin.assign(n3 * 3, 0); h->g_mass.assign(n3, 0); h->g_vout.assign(n3 * 3, 0);
        h->gg_vin.assign(n3 * 3, 0); h->gg_mass.assign(n3, 0); h->gg_vout.assign(n3 * 3, 0);
    } catch (...) { g_create_err = "out of memory"; delete h; return nullptr; }
    h->t0 = std::chrono::steady_clock::now();
    return h;
}

void fe_destroy(FeEngine* h) { if (h) delete h->smoke; delete h; }
const char* fe_last_error(FeEngine* h) { return h ? h->err.c_str() : g_create_err.c_str(); }
int fe_sync(FeEngine*) { return 0; }

int fe_get_option(FeEngine* h, const char* name, double* value) {      // (the oracle has four options of its own; the HIP engine's tunables read as 0)
    if (!value) { h->err = "fe_get_option: null output"; return 1; }
    if (!std::strcmp(name, "threads")) { *value = h->threads; return 0; }
    if (!std::strcmp(name, "scatter")) { *value = h->scatter_coloured; return 0; }
    if (!std::strcmp(name, "inject_till")) { *value = h->inject_till; return 0; }
    if (!std::strcmp(name, "collide_min_y")) { *value = (double)h->collide_min_y; return 0; }
    if (!std::strcmp(name, "collide_type")) { *value = h->collide_type; return 0; }
    *value = 0.0;
    return 0;
}
int fe_set_option(FeEngine* h, const char* name, double value) {
    if (!std::strcmp(name, "threads")) {
        int t = (int)value;
#ifdef _OPENMP
        if (t <= 0) t = omp_get_max_threads();
#else
        t = 1;
#endif
        h->threads = t; return 0;
    }
    if (!std::strcmp(name, "scatter")) { h->scatter_coloured = value != 0; return 0; }
    if (!std::strcmp(name, "inject_till")) { h->inject_till = (int)value; return 0; }
    if (!std::strcmp(name, "collide_min_y")) { h->collide_min_y = (R)value; return 0; }
    if (!std::strcmp(name, "collide_type")) {
        const int t = (int)value;
        if (t < 1 || t > 3) { h->err = "collide_type must be 1 (particle), 2 (grid) or 3 (both)"; return 1; }
        h->collide_type = t; return 0;
    }
    /* HIP-engine tunables are accepted and ignored so the same host code drives both */
    return 0;
}

int fe_init_particles(FeEngine* h, const fe_real* x, const int* used, const int* mat, const int* mat_cls,
                      const fe_real* mu, const fe_real* lam, const fe_real* rho, const int* body_id) {
    const int N = h->N;
    for (int i = 0; i < N; i++) {
        for (int d = 0; d < 3; d++) { h->X(0)[i * 3 + d] = x[i * 3 + d]; h->Vv(0)[i * 3 + d] = 0; }   /* mpm:163-165 */
        for (int d = 0; d < 9; d++) { h->Ff(0)[i * 9 + d] = (d % 4 == 0) ? 1 : 0; h->Cc(0)[i * 9 + d] = 0; }
        h->Us(0)[i] = used[i];
        h->mat[i] = mat[i]; h->mat_cls[i] = mat_cls[i]; h->mu[i] = mu[i]; h->lam[i] = lam[i];
        h->mass[i] = h->cfg.p_vol * rho[i];                                                            /* mpm:174 */
        h->body_id[i] = body_id ? body_id[i] : 0;
    }
    /* init_bodies, mpm:176-201 */
    h->n_bodies = 0;
    for (int i = 0; i < N; i++) {
        if (h->body_id[i] < 0) FE_FAIL(h, "negative body_id");
        h->n_bodies = std::max(h->n_bodies, h->body_id[i] + 1);
    }
    h->body_n.assign(h->n_bodies, 0); h->body_cls.assign(h->n_bodies, -1); h->bodies.assign(h->n_bodies, Body());
    h->has_rigid = false;
    for (int i = 0; i < N; i++) {
        int b = h->body_id[i];
        if (h->body_n[b]++ == 0) h->body_cls[b] = mat_cls[i];            /* mat_cls[body_id == i][0], mpm:201 */
        if (mat_cls[i] == FE_MAT_RIGID) h->has_rigid = true;
    }
    h->initialized = true;
    return 0;
}

int fe_substep(FeEngine* h, int f, int f_global, int act) {
    if (f < 0 || f >= h->L) FE_FAIL(h, "substep frame out of range");
    double t = h->prof_on ? now_ms(h) : 0;
    int rc = substep(h, f, f_global, act);
    if (h->prof_on) { h->prof_ms[0] += now_ms(h) - t; h->prof_n[0]++; }
    return rc;
}
int fe_substep_grad(FeEngine* h, int f, int f_global, int act) {
    if (f < 0 || f >= h->L) FE_FAIL(h, "substep frame out of range");
    double t = h->prof_on ? now_ms(h) : 0;
    int rc = substep_grad(h, f, f_global, act);
    if (h->prof_on) { h->prof_ms[1] += now_ms(h) - t; h->prof_n[1]++; }
    return rc;
}
int fe_step(FeEngine* h, int f0, int f_global0, int n, int act) {
    for (int i = 0; i < n; i++) if (fe_substep(h, f0 + i, f_global0 + i, act)) return 1;
    return 0;
}
int fe_step_grad(FeEngine* h, int f0, int f_global0, int n, int act) {
    for (int i = n - 1; i >= 0; i--) if (fe_substep_grad(h, f0 + i, f_global0 + i, act)) return 1;
    return 0;
}
int fe_step_batch(FeEngine** hs, int n_env, int f0, int f_global0, int n, int act) {      /* include/fluidengine.h: lockstep envs, one after the other here */
    for (int e = 0; e < n_env; e++) if (fe_step(hs[e], f0, f_global0, n, act)) return 1;
    return 0;
}
int fe_step_grad_batch(FeEngine** hs, int n_env, int f0, int f_global0, int n, int act) {
    for (int e = 0; e < n_env; e++) if (fe_step_grad(hs[e], f0, f_global0, n, act)) return 1;
    return 0;
}

int fe_get_frame(FeEngine* h, int f, fe_real* x, fe_real* v, fe_real* C, fe_real* F, int* used) {
    CHECK_FRAME(h, f);
    size_t N = h->N;
    if (x) std::memcpy(x, h->X(f), sizeof(R) * N * 3);
    if (v) std::memcpy(v, h->Vv(f), sizeof(R) * N * 3);
    if (C) std::memcpy(C, h->Cc(f), sizeof(R) * N * 9);
    if (F) std::memcpy(F, h->Ff(f), sizeof(R) * N * 9);
    if (used) std::memcpy(used, h->Us(f), sizeof(int) * N);
    return 0;
}
int fe_set_frame(FeEngine* h, int f, const fe_real* x, const fe_real* v, const fe_real* C, const fe_real* F, const int* used) {
    CHECK_FRAME(h, f);
    size_t N = h->N;
    if (x) std::memcpy(h->X(f), x, sizeof(R) * N * 3);
    if (v) std::memcpy(h->Vv(f), v, sizeof(R) * N * 3);
    if (C) std::memcpy(h->Cc(f), C, sizeof(R) * N * 9);
    if (F) std::memcpy(h->Ff(f), F, sizeof(R) * N * 9);
    if (used) std::memcpy(h->Us(f), used, sizeof(int) * N);
    return 0;
}
int fe_get_frame_dev(FeEngine* h, int f, fe_real* x, fe_real* v, fe_real* C, fe_real* F, int* used) { return fe_get_frame(h, f, x, v, C, F, used); }
int fe_set_frame_dev(FeEngine* h, int f, const fe_real* x, const fe_real* v, const fe_real* C, const fe_real* F, const int* used) {
    return fe_set_frame(h, f, x, v, C, F, used);
}
int fe_copy_frame(FeEngine* h, int src, int dst) {
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    return fe_set_frame(h, dst, h->X(src), h->Vv(src), h->Cc(src), h->Ff(src), h->Us(src));
}
int fe_copy_grad(FeEngine* h, int src, int dst) {
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    size_t N = h->N;
    std::memcpy(h->GX(dst), h->GX(src), sizeof(R) * N * 3); std::memcpy(h->GV(dst), h->GV(src), sizeof(R) * N * 3);
    std::memcpy(h->GC(dst), h->GC(src), sizeof(R) * N * 9); std::memcpy(h->GF(dst), h->GF(src), sizeof(R) * N * 9);
    std::memcpy(h->Us(dst), h->Us(src), sizeof(int) * N);    /* mpm:604 copies `used` too */
    return 0;
}
int fe_reset_grad(FeEngine* h) {
    for (auto* t : {&h->gx, &h->gv, &h->gC, &h->gF, &h->gg_vin, &h->gg_mass, &h->gg_vout}) std::fill(t->begin(), t->end(), (R)0);
    for (auto& e : h->effs) for (auto* t : {&e.gpos, &e.gquat, &e.gv, &e.gw, &e.gabuf, &e.gabuf_p, &e.gsa, &e.gra}) std::fill(t->begin(), t->end(), (R)0);
    if (h->smoke) smoke_reset_grad(h);
    return 0;
}
int fe_reset_grad_till_frame(FeEngine* h, int f) {
    CHECK_FRAME(h, f);
    size_t N = h->N;
    std::fill(h->gx.begin(), h->gx.begin() + (size_t)f * N * 3, (R)0); std::fill(h->gv.begin(), h->gv.begin() + (size_t)f * N * 3, (R)0);
    std::fill(h->gC.begin(), h->gC.begin() + (size_t)f * N * 9, (R)0); std::fill(h->gF.begin(), h->gF.begin() + (size_t)f * N * 9, (R)0);
    return 0;
}
int fe_agent_set_collector(FeEngine* h, const FeBoundary* b, int mat) {
    h->has_collector = b != nullptr;
    if (b) h->collector = *b;
    h->collector_mat = mat;
    return 0;
}
int fe_agent_reset_grad_till_frame(FeEngine* h, int f) {
    CHECK_FRAME(h, f);
    for (auto& e : h->effs) {
        std::fill(e.gpos.begin(), e.gpos.begin() + f * 3, (R)0); std::fill(e.gquat.begin(), e.gquat.begin() + f * 4, (R)0);
        std::fill(e.gv.begin(), e.gv.begin() + f * 3, (R)0); std::fill(e.gw.begin(), e.gw.begin() + f * 3, (R)0);
        std::fill(e.gsa.begin(), e.gsa.begin() + f, (R)0); std::fill(e.gra.begin(), e.gra.begin() + f, (R)0);
    }
    return 0;
}
int fe_get_grad(FeEngine* h, int f, fe_real* gx, fe_real* gv, fe_real* gC, fe_real* gF) {
    CHECK_FRAME(h, f);
    size_t N = h->N;
    if (gx) std::memcpy(gx, h->GX(f), sizeof(R) * N * 3);
    if (gv) std::memcpy(gv, h->GV(f), sizeof(R) * N * 3);
    if (gC) std::memcpy(gC, h->GC(f), sizeof(R) * N * 9);
    if (gF) std::memcpy(gF, h->GF(f), sizeof(R) * N * 9);
    return 0;
}
int fe_add_grad(FeEngine* h, int f, const fe_real* gx, const fe_real* gv, const fe_real* gC, const fe_real* gF) {
    CHECK_FRAME(h, f);
    size_t N = h->N;
    if (gx) for (size_t i = 0; i < N * 3; i++) h->GX(f)[i] += gx[i];
    if (gv) for (size_t i = 0; i < N * 3; i++) h->GV(f)[i] += gv[i];
    if (gC) for (size_t i = 0; i < N * 9; i++) h->GC(f)[i] += gC[i];
    if (gF) for (size_t i = 0; i < N * 9; i++) h->GF(f)[i] += gF[i];
    return 0;
}
int fe_add_grad_dev(FeEngine* h, int f, const fe_real* gx, const fe_real* gv, const fe_real* gC, const fe_real* gF) { return fe_add_grad(h, f, gx, gv, gC, gF); }   /* host == device here */
int fe_get_mat(FeEngine* h, int* mat) { std::memcpy(mat, h->mat.data(), sizeof(int) * h->N); return 0; }

/* ---- effectors */
int fe_add_effector(FeEngine* h, const FeEffectorDesc* d, const fe_real* random_vector) {
    if (!d || d->struct_size != (int)sizeof(FeEffectorDesc)) { h->err = "FeEffectorDesc size mismatch"; return -1; }
    if (!(d->action_dim == 0 || d->action_dim == 3 || d->action_dim == 6 || (d->type == FE_EFF_AIRCON && d->action_dim == 8))) {
        h->err = "action_dim must be 0, 3 or 6 (8 for an AirCon)"; return -1;
    }
    Effector e; e.d = *d;
    int Fm = h->L + 1;
    e.sa.assign(Fm, 0); e.ra.assign(Fm, 0); e.gsa.assign(Fm, 0); e.gra.assign(Fm, 0);
    e.pos.assign(Fm * 3, 0); e.quat.assign(Fm * 4, 0); e.v.assign(Fm * 3, 0); e.w.assign(Fm * 3, 0);
    e.gpos.assign(Fm * 3, 0); e.gquat.assign(Fm * 4, 0); e.gv.assign(Fm * 3, 0); e.gw.assign(Fm * 3, 0);
    int ad = std::max(d->action_dim, 1);
    e.abuf.assign((size_t)h->cfg.max_action_steps * ad, 0); e.gabuf.assign((size_t)h->cfg.max_action_steps * ad, 0);
    e.abuf_p.assign(ad, 0); e.gabuf_p.assign(ad, 0);
    e.act_id.assign(Fm, 0);
    if (d->type == FE_EFF_INJECTOR) {
        if (!random_vector || d->random_length <= 0 || d->flux <= 0) { h->err = "injector needs random_vector, random_length, flux"; return -1; }
        e.random_vector.assign(random_vector, random_vector + (size_t)d->random_length * d->flux * 3);
    }
    h->effs.push_back(e);
    return (int)h->effs.size() - 1;
}
int fe_eff_set_act_range(FeEngine* h, int e, const int* act_range, int n) {
    CHECK_EFF(h, e);
    h->effs[e].act_range.assign(act_range, act_range + n);
    if (n > 0) h->effs[e].act_id[0] = act_range[0];        /* injector.py:68 (sic: the first pool id, not 0) */
    return 0;
}
int fe_eff_get_state(FeEngine* h, int e, int f, fe_real* s) {
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    Effector& E = h->effs[e];
    for (int j = 0; j < 3; j++) s[j] = E.pos[f * 3 + j];
    for (int j = 0; j < 4; j++) s[3 + j] = E.quat[f * 4 + j];
    s[7] = (R)E.act_id[f];
    return 0;
}
int fe_eff_set_state(FeEngine* h, int e, int f, const fe_real* s) {
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    Effector& E = h->effs[e];
    for (int j = 0; j < 3; j++) E.pos[f * 3 + j] = s[j];
    for (int j = 0; j < 4; j++) E.quat[f * 4 + j] = s[3 + j];
    E.act_id[f] = (int)s[7];
    return 0;
}
int fe_eff_get_vw(FeEngine* h, int e, int f, fe_real* v3, fe_real* w3) {
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    for (int j = 0; j < 3; j++) { v3[j] = h->effs[e].v[f * 3 + j]; w3[j] = h->effs[e].w[f * 3 + j]; }
    return 0;
}
int fe_eff_set_vw(FeEngine* h, int e, int f, const fe_real* v3, const fe_real* w3) {
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    for (int j = 0; j < 3; j++) { h->effs[e].v[f * 3 + j] = v3[j]; h->effs[e].w[f * 3 + j] = w3[j]; }
    return 0;
}
int fe_eff_set_action(FeEngine* h, int e, int s, int s_global, int n_substeps, const fe_real* action) {
    CHECK_EFF(h, e);
    Effector& E = h->effs[e];
    const int ad = E.d.action_dim;
    if (ad == 0) return 0;
    if (s_global < 0 || s_global >= h->cfg.max_action_steps) FE_FAIL(h, "s_global out of range");    /* effector.py:263 */
    if (s < 0 || (s + 1) * n_substeps > h->L + 1) FE_FAIL(h, "s out of range");                         /* effector.py:264 */
    for (int j = 0; j < ad; j++) E.abuf[(size_t)s_global * ad + j] = action[j];                        /* effector.py:218-221 */
    for (int j = s * n_substeps; j < (s + 1) * n_substeps; j++) {                                      /* effector.py:252-260 */
        R nf = (R)n_substeps;
        for (int k = 0; k < 3; k++) E.v[j * 3 + k] = E.abuf[(size_t)s_global * ad + k] * E.d.action_scale_v[k] / nf;
        if (ad > 3) for (int k = 0; k < 3; k++) E.w[j * 3 + k] = E.abuf[(size_t)s_global * ad + k + 3] * E.d.action_scale_v[k + 3] / nf;
        if (ad > 6) {                                                                                  /* aircon.py:211-213 */
            E.sa[j] = E.abuf[(size_t)s_global * ad + 6] * E.d.action_scale_v[6];
            E.ra[j] = E.abuf[(size_t)s_global * ad + 7] * E.d.action_scale_v[7];
        }
    }
    return 0;
}
int fe_eff_set_action_grad(FeEngine* h, int e, int s, int s_global, int n_substeps) {
    CHECK_EFF(h, e);
    Effector& E = h->effs[e];
    const int ad = E.d.action_dim;
    if (ad == 0) return 0;
    if (s_global < 0 || s_global >= h->cfg.max_action_steps) FE_FAIL(h, "s_global out of range");
    for (int j = s * n_substeps; j < (s + 1) * n_substeps; j++) {
        R nf = (R)n_substeps;
        for (int k = 0; k < 3; k++) E.gabuf[(size_t)s_global * ad + k] += E.gv[j * 3 + k] * E.d.action_scale_v[k] / nf;
        if (ad > 3) for (int k = 0; k < 3; k++) E.gabuf[(size_t)s_global * ad + k + 3] += E.gw[j * 3 + k] * E.d.action_scale_v[k + 3] / nf;
        if (ad > 6) {
            E.gabuf[(size_t)s_global * ad + 6] += E.gsa[j] * E.d.action_scale_v[6];
            E.gabuf[(size_t)s_global * ad + 7] += E.gra[j] * E.d.action_scale_v[7];
        }
    }
    return 0;
}
int fe_eff_apply_action_p(FeEngine* h, int e, const fe_real* action_p) {
    CHECK_EFF(h, e);
    Effector& E = h->effs[e];
    if (E.d.action_dim == 0) return 0;
    for (int j = 0; j < E.d.action_dim; j++) E.abuf_p[j] = action_p[j];
    R xin[3], xn[3], J[3][3];
    for (int d = 0; d < 3; d++) xin[d] = E.abuf_p[d] * E.d.action_scale_p[d];                         /* effector.py:223-225 */
    impose_x(E.d.boundary, xin, xn, J);
    for (int d = 0; d < 3; d++) E.pos[d] = xn[d];
    return 0;
}
int fe_eff_apply_action_p_grad(FeEngine* h, int e) {
    CHECK_EFF(h, e);
    Effector& E = h->effs[e];
    if (E.d.action_dim == 0) return 0;
    R xin[3], xn[3], J[3][3];
    for (int d = 0; d < 3; d++) xin[d] = E.abuf_p[d] * E.d.action_scale_p[d];
    impose_x(E.d.boundary, xin, xn, J);
    for (int d = 0; d < 3; d++) {
        R g = 0; for (int i = 0; i < 3; i++) g += J[i][d] * E.gpos[i];
        E.gabuf_p[d] += g * E.d.action_scale_p[d];
    }
    return 0;
}
int fe_eff_get_action_grad(FeEngine* h, int e, int s, int n, fe_real* grad) {
    CHECK_EFF(h, e);
    Effector& E = h->effs[e];
    const int ad = E.d.action_dim;
    if (s < 0 || s + n > h->cfg.max_action_steps) FE_FAIL(h, "action grad range out of bounds");
    for (int i = 0; i < n; i++) for (int j = 0; j < ad; j++) grad[i * ad + j] = E.gabuf[(size_t)(s + i) * ad + j];
    for (int j = 0; j < ad; j++) grad[n * ad + j] = E.gabuf_p[j];
    return 0;
}
int fe_agent_copy_frame(FeEngine* h, int src, int dst) {
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    for (auto& E : h->effs) {
        for (int j = 0; j < 3; j++) { E.pos[dst * 3 + j] = E.pos[src * 3 + j]; E.v[dst * 3 + j] = E.v[src * 3 + j]; E.w[dst * 3 + j] = E.w[src * 3 + j]; }
        for (int j = 0; j < 4; j++) E.quat[dst * 4 + j] = E.quat[src * 4 + j];
        E.sa[dst] = E.sa[src]; E.ra[dst] = E.ra[src];                          /* aircon.py:148-155 */
        if (E.d.type == FE_EFF_INJECTOR) E.act_id[dst] = E.act_id[src];       /* injector.py:174-179 */
    }
    return 0;
}
int fe_agent_copy_grad(FeEngine* h, int src, int dst) {
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    for (auto& E : h->effs) {
        for (int j = 0; j < 3; j++) { E.gpos[dst * 3 + j] = E.gpos[src * 3 + j]; E.gv[dst * 3 + j] = E.gv[src * 3 + j]; E.gw[dst * 3 + j] = E.gw[src * 3 + j]; }
        for (int j = 0; j < 4; j++) E.gquat[dst * 4 + j] = E.gquat[src * 4 + j];
        E.gsa[dst] = E.gsa[src]; E.gra[dst] = E.gra[src];
    }
    return 0;
}

/* ---- AirCon strength / radius (aircon.py:178-191) */
int fe_eff_get_sr(FeEngine* h, int e, int f, fe_real* s, fe_real* r) {
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    *s = h->effs[e].sa[f]; *r = h->effs[e].ra[f];
    return 0;
}
int fe_eff_set_sr(FeEngine* h, int e, int f, fe_real s, fe_real r) {
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    h->effs[e].sa[f] = s; h->effs[e].ra[f] = r;
    return 0;
}

/* ---- smoke field (smoke_field.py) */
#define CHECK_SMOKE(h, s) do { if (!(h)->smoke) FE_FAIL(h, "no smoke field"); if ((s) < 0 || (s) > (h)->smoke->S) FE_FAIL(h, "smoke frame out of range"); } while (0)
int fe_smoke_create(FeEngine* h, const FeSmokeConfig* c) {
    if (!c || c->struct_size != (int)sizeof(FeSmokeConfig)) FE_FAIL(h, "FeSmokeConfig size mismatch");
    if (c->res < 4 || c->q_dim < 1 || c->q_dim > 8 || c->max_steps_local < 1 || c->solver_iters < 0) FE_FAIL(h, "bad smoke configuration");
    delete h->smoke;
    Smoke* m = new Smoke();
    m->c = *c; m->n = c->res; m->S = c->max_steps_local; m->n3 = (size_t)c->res * c->res * c->res;
    const size_t F = (size_t)(m->S + 1) * m->n3;
    m->v.assign(F * 3, 0); m->vt.assign(F * 3, 0); m->dv.assign(F, 0); m->p.assign(F, 0); m->q.assign(F * c->q_dim, 0);
    m->gv.assign(F * 3, 0); m->gvt.assign(F * 3, 0); m->gdv.assign(F, 0); m->gp.assign(F, 0); m->gq.assign(F * c->q_dim, 0);
    m->fr.assign(F, 0);
    m->pc.assign(m->n3, 0); m->pn.assign(m->n3, 0); m->gpc.assign(m->n3, 0); m->gpn.assign(m->n3, 0);
    /* init_fields, smoke_field.py:86-93: q[0] = high_T in the slab (ti.Vector([high_T]) broadcasts over q_dim) */
    for (int i = 0; i < m->n; i++) for (int j = 0; j < m->n; j++) for (int k = 0; k < m->n; k++)
        if (c->lower_y < j && j < c->higher_y) for (int d = 0; d < c->q_dim; d++) m->q[m->idx(i, j, k) * c->q_dim + d] = c->high_T;
    h->smoke = m;
    return 0;
}
int fe_smoke_step(FeEngine* h, int s, int f) { CHECK_SMOKE(h, s); if (s >= h->smoke->S) FE_FAIL(h, "smoke step frame out of range"); CHECK_FRAME(h, f); return smoke_step(h, s, f); }
int fe_smoke_step_grad(FeEngine* h, int s, int f) { CHECK_SMOKE(h, s); if (s >= h->smoke->S) FE_FAIL(h, "smoke step frame out of range"); CHECK_FRAME(h, f); return smoke_step_grad(h, s, f); }
int fe_smoke_get_frame(FeEngine* h, int s, fe_real* v, fe_real* v_tmp, fe_real* div, fe_real* p, fe_real* q) {
    CHECK_SMOKE(h, s);
    Smoke& m = *h->smoke;
    if (v) std::memcpy(v, m.F(m.v, s, 3), sizeof(R) * m.n3 * 3);
    if (v_tmp) std::memcpy(v_tmp, m.F(m.vt, s, 3), sizeof(R) * m.n3 * 3);
    if (div) std::memcpy(div, m.F(m.dv, s, 1), sizeof(R) * m.n3);
    if (p) std::memcpy(p, m.F(m.p, s, 1), sizeof(R) * m.n3);
    if (q) std::memcpy(q, m.F(m.q, s, m.c.q_dim), sizeof(R) * m.n3 * m.c.q_dim);
    return 0;
}
int fe_smoke_set_frame(FeEngine* h, int s, const fe_real* v, const fe_real* v_tmp, const fe_real* div, const fe_real* p, const fe_real* q) {
    CHECK_SMOKE(h, s);
    Smoke& m = *h->smoke;
    if (v) std::memcpy(m.F(m.v, s, 3), v, sizeof(R) * m.n3 * 3);
    if (v_tmp) std::memcpy(m.F(m.vt, s, 3), v_tmp, sizeof(R) * m.n3 * 3);
    if (div) std::memcpy(m.F(m.dv, s, 1), div, sizeof(R) * m.n3);
    if (p) std::memcpy(m.F(m.p, s, 1), p, sizeof(R) * m.n3);
    if (q) std::memcpy(m.F(m.q, s, m.c.q_dim), q, sizeof(R) * m.n3 * m.c.q_dim);
    return 0;
}
int fe_smoke_get_grad(FeEngine* h, int s, fe_real* gv, fe_real* gq) {
    CHECK_SMOKE(h, s);
    Smoke& m = *h->smoke;
    if (gv) std::memcpy(gv, m.F(m.gv, s, 3), sizeof(R) * m.n3 * 3);
    if (gq) std::memcpy(gq, m.F(m.gq, s, m.c.q_dim), sizeof(R) * m.n3 * m.c.q_dim);
    return 0;
}
int fe_smoke_add_grad(FeEngine* h, int s, const fe_real* gv, const fe_real* gq) {
    CHECK_SMOKE(h, s);
    Smoke& m = *h->smoke;
    if (gv) for (size_t i = 0; i < m.n3 * 3; i++) m.F(m.gv, s, 3)[i] += gv[i];
    if (gq) for (size_t i = 0; i < m.n3 * m.c.q_dim; i++) m.F(m.gq, s, m.c.q_dim)[i] += gq[i];
    return 0;
}
int fe_smoke_copy_frame(FeEngine* h, int src, int dst) {
    CHECK_SMOKE(h, src); CHECK_SMOKE(h, dst);
    Smoke& m = *h->smoke;
    const int qd = m.c.q_dim;
    std::memcpy(m.F(m.v, dst, 3), m.F(m.v, src, 3), sizeof(R) * m.n3 * 3); std::memcpy(m.F(m.vt, dst, 3), m.F(m.vt, src, 3), sizeof(R) * m.n3 * 3);
    std::memcpy(m.F(m.dv, dst, 1), m.F(m.dv, src, 1), sizeof(R) * m.n3); std::memcpy(m.F(m.p, dst, 1), m.F(m.p, src, 1), sizeof(R) * m.n3);
    std::memcpy(m.F(m.q, dst, qd), m.F(m.q, src, qd), sizeof(R) * m.n3 * qd);
    return 0;
}
int fe_smoke_copy_grad(FeEngine* h, int src, int dst) {
    CHECK_SMOKE(h, src); CHECK_SMOKE(h, dst);
    Smoke& m = *h->smoke;
    const int qd = m.c.q_dim;
    std::memcpy(m.F(m.gv, dst, 3), m.F(m.gv, src, 3), sizeof(R) * m.n3 * 3); std::memcpy(m.F(m.gvt, dst, 3), m.F(m.gvt, src, 3), sizeof(R) * m.n3 * 3);
    std::memcpy(m.F(m.gdv, dst, 1), m.F(m.gdv, src, 1), sizeof(R) * m.n3); std::memcpy(m.F(m.gp, dst, 1), m.F(m.gp, src, 1), sizeof(R) * m.n3);
    std::memcpy(m.F(m.gq, dst, qd), m.F(m.gq, src, qd), sizeof(R) * m.n3 * qd);
    return 0;
}
int fe_smoke_reset_grad(FeEngine* h) { if (!h->smoke) FE_FAIL(h, "no smoke field"); smoke_reset_grad(h); return 0; }
int fe_smoke_reset_grad_till_frame(FeEngine* h, int s) {
    CHECK_SMOKE(h, s);
    Smoke& m = *h->smoke;
    const size_t F = (size_t)s * m.n3;
    std::fill(m.gv.begin(), m.gv.begin() + F * 3, (R)0); std::fill(m.gvt.begin(), m.gvt.begin() + F * 3, (R)0);
    std::fill(m.gdv.begin(), m.gdv.begin() + F, (R)0); std::fill(m.gp.begin(), m.gp.begin() + F, (R)0);
    std::fill(m.gq.begin(), m.gq.begin() + F * m.c.q_dim, (R)0);
    return 0;
}

static int make_sdf(FeEngine* h, const FeSdfDesc* d, const fe_real* voxels, Sdf& s) {
    if (!d || d->struct_size != (int)sizeof(FeSdfDesc) || d->res < 2 || !voxels) { h->err = "bad FeSdfDesc"; return 1; }
    s.res = d->res; s.friction = d->friction; s.softness = d->softness;
    s.vox.assign(voxels, voxels + (size_t)d->res * d->res * d->res);
    for (int i = 0; i < 16; i++) s.T[i] = d->T_mesh_to_voxels[i];
    M3 A;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A.m[i][j] = s.T[i * 4 + j];
    const R det = A.m[0][0] * (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) - A.m[0][1] * (A.m[1][0] * A.m[2][2] - A.m[1][2] * A.m[2][0]) +
                  A.m[0][2] * (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]);
    if (det == 0) { h->err = "singular T_mesh_to_voxels"; return 1; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {       /* inverse = adj / det */
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        s.Rinv[j * 3 + i] = (A.m[i1][j1] * A.m[i2][j2] - A.m[i1][j2] * A.m[i2][j1]) / det;
    }
    return 0;
}
int fe_eff_set_mesh(FeEngine* h, int e, const FeSdfDesc* d, const fe_real* voxels) {
    CHECK_EFF(h, e);
    Sdf s;
    if (make_sdf(h, d, voxels, s)) return 1;
    h->meshes.push_back(std::move(s));
    h->effs[e].mesh = (int)h->meshes.size() - 1;
    h->has_mesh_effector = true;
    return 0;
}
int fe_add_static(FeEngine* h, const FeSdfDesc* d, const fe_real* voxels) {
    Sdf s;
    if (make_sdf(h, d, voxels, s)) return -1;
    h->statics.push_back(std::move(s));
    return (int)h->statics.size() - 1;
}

/* ---- loss */
int fe_loss_alloc(FeEngine* h, int max_loss_steps) {
    h->loss_steps = max_loss_steps;
    h->tgt.assign((size_t)max_loss_steps * h->N * 3, 0);
    h->chamfer.assign(max_loss_steps, 0); h->step_loss.assign(max_loss_steps, 0);
    return 0;
}
int fe_loss_set_target(FeEngine* h, int s, const fe_real* x) {
    if (s < 0 || s >= h->loss_steps) FE_FAIL(h, "loss step out of range");
    std::memcpy(&h->tgt[(size_t)s * h->N * 3], x, sizeof(R) * h->N * 3);
    return 0;
}
int fe_loss_clear(FeEngine* h) {
    std::fill(h->chamfer.begin(), h->chamfer.end(), (R)0); std::fill(h->step_loss.begin(), h->step_loss.end(), (R)0);
    return 0;
}
int fe_loss_step(FeEngine* h, int s, int f, int matching_mat, fe_real weight) {
    if (s < 0 || s >= h->loss_steps) FE_FAIL(h, "loss step out of range");
    CHECK_FRAME(h, f);
    const R* t = &h->tgt[(size_t)s * h->N * 3];
    R acc = 0;
    for (int p = 0; p < h->N; p++) {
        if (h->Us(f)[p] && (matching_mat < 0 || h->mat[p] == matching_mat)) {              /* shapematching_loss.py:83; < 0: every material (latteartstir_loss.py:62-68) */
            R d0 = h->X(f)[p * 3] - t[p * 3], d1 = h->X(f)[p * 3 + 1] - t[p * 3 + 1], d2 = h->X(f)[p * 3 + 2] - t[p * 3 + 2];
            acc += d0 * d0 + d1 * d1 + d2 * d2;
        }
    }
    h->chamfer[s] += acc;
    h->step_loss[s] += h->chamfer[s] * weight;                                            /* shapematching_loss.py:88 */
    return 0;
}
int fe_loss_step_grad(FeEngine* h, int s, int f, int matching_mat, fe_real weight, fe_real step_loss_grad) {
    if (s < 0 || s >= h->loss_steps) FE_FAIL(h, "loss step out of range");
    CHECK_FRAME(h, f);
    const R* t = &h->tgt[(size_t)s * h->N * 3];
    R g = weight * step_loss_grad;
    for (int p = 0; p < h->N; p++) {
        if (h->Us(f)[p] && (matching_mat < 0 || h->mat[p] == matching_mat))
            for (int d = 0; d < 3; d++) h->GX(f)[p * 3 + d] += 2 * (h->X(f)[p * 3 + d] - t[p * 3 + d]) * g;
    }
    return 0;
}
int fe_loss_get(FeEngine* h, fe_real* step_loss, int n) {
    if (n > h->loss_steps) FE_FAIL(h, "loss_get: n too large");
    for (int i = 0; i < n; i++) step_loss[i] = h->step_loss[i];
    return 0;
}

/* ---- measurement */
/* utils/mesh.py:63-96 -- what mesh_to_sdf approximates, restated exactly: distance to the closest point of the closest triangle
 * (region walk of Ericson, Real-Time Collision Detection 5.1.5), sign from the generalized winding number (sum of the triangles'
 * solid angles, van Oosterom & Strackee 1983, over 4 pi) > 1/2.  Computed in double whatever R is. */
static double mesh_tri_dist2(const double a[3], const double b[3], const double c[3]) {
    double ab[3], ac[3], bc[3];
    for (int d = 0; d < 3; d++) { ab[d] = b[d] - a[d]; ac[d] = c[d] - a[d]; bc[d] = c[d] - b[d]; }
    auto dot = [](const double* u, const double* v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
    const double d1 = -dot(ab, a), d2 = -dot(ac, a), d3 = -dot(ab, b), d4 = -dot(ac, b), d5 = -dot(ab, c), d6 = -dot(ac, c);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    double q[3];
    if (d1 <= 0 && d2 <= 0) { for (int d = 0; d < 3; d++) q[d] = a[d]; }
    else if (d3 >= 0 && d4 <= d3) { for (int d = 0; d < 3; d++) q[d] = b[d]; }
    else if (d6 >= 0 && d5 <= d6) { for (int d = 0; d < 3; d++) q[d] = c[d]; }
    else if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double t = d1 / (d1 - d3); for (int d = 0; d < 3; d++) q[d] = a[d] + t * ab[d]; }
    else if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double t = d2 / (d2 - d6); for (int d = 0; d < 3; d++) q[d] = a[d] + t * ac[d]; }
    else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int d = 0; d < 3; d++) q[d] = b[d] + t * bc[d]; }
    else { const double den = 1 / (va + vb + vc), v = vb * den, w = vc * den; for (int d = 0; d < 3; d++) q[d] = a[d] + ab[d] * v + ac[d] * w; }
    return dot(q, q);
}
int fe_mesh_sdf(int device, const float* verts, int nv, const int* faces, int nf, const float* points, long long n_points, float* sdf) {
    (void)device;
    if (nv <= 0 || nf <= 0 || n_points < 0 || !verts || !faces || (n_points > 0 && (!points || !sdf))) { g_create_err = "fe_mesh_sdf: invalid arguments"; return 1; }
    for (long long i = 0; i < (long long)nf * 3; i++) if (faces[i] < 0 || faces[i] >= nv) { g_create_err = "fe_mesh_sdf: face index out of range"; return 1; }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n_points; i++) {
        double best = 1e300, omega = 0;
        for (int t = 0; t < nf; t++) {
            double tri[3][3];
            for (int k = 0; k < 3; k++) for (int d = 0; d < 3; d++) tri[k][d] = (double)verts[(size_t)faces[(size_t)t * 3 + k] * 3 + d] - (double)points[i * 3 + d];
            best = std::min(best, mesh_tri_dist2(tri[0], tri[1], tri[2]));
            const double* a = tri[0]; const double* b = tri[1]; const double* c = tri[2];
            const double la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]), lc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            const double det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
            const double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2], bc = b[0] * c[0] + b[1] * c[1] + b[2] * c[2], ca = c[0] * a[0] + c[1] * a[1] + c[2] * a[2];
            omega += 2 * std::atan2(det, la * lb * lc + ab * lc + bc * la + ca * lb);
        }
        const double d = std::sqrt(best);
        sdf[i] = (float)(std::fabs(omega) > 2 * 3.14159265358979323846 ? -d : d);       /* either face orientation */
    }
    return 0;
}

int fe_get_work_stats(FeEngine*, int, long long out[24]) { for (int i = 0; i < 24; i++) out[i] = 0; return 0; }   // no work lists here
int fe_get_work_stats_n(FeEngine*, int, long long* out, int n) { for (int i = 0; i < n && i < 24; i++) out[i] = 0; return 0; }
int fe_get_stats(FeEngine* h, int f, FeStats* out) {
    CHECK_FRAME(h, f);
    const int n = h->n, nb = (n + 3) / 4;
    std::vector<unsigned char> touched((size_t)n * n * n, 0), blk((size_t)nb * nb * nb, 0);
    long long used = 0;
    for (int p = 0; p < h->N; p++) {
        if (!h->Us(f)[p]) continue;
        used++;
        Stencil s; make_stencil(&h->X(f)[p * 3], h->inv_dx, s);
        if (!stencil_in_grid(s, n)) continue;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
            int a = s.base[0] + i, b = s.base[1] + j, c = s.base[2] + k;
            touched[cell_index(n, a, b, c)] = 1;
            blk[((size_t)(a >> 2) * nb + (b >> 2)) * nb + (c >> 2)] = 1;
        }
    }
    long long nc = 0, nbk = 0;
    for (unsigned char t : touched) nc += t;
    for (unsigned char t : blk) nbk += t;
    out->n_used = used; out->n_cells_touched = nc; out->n_blocks_active = nbk; out->n_slow_path = 0;
    out->bytes_state = (long long)((h->x.size() + h->v.size() + h->C.size() + h->F.size()) * 2 * sizeof(R) + h->used.size() * sizeof(int));
    return 0;
}
static double g_timer_start_ms = 0;
int fe_timer_start(FeEngine* h) { g_timer_start_ms = now_ms(h); return 0; }
double fe_timer_stop_ms(FeEngine* h) { return now_ms(h) - g_timer_start_ms; }
int fe_profile_enable(FeEngine* h, int on) {
    h->prof_on = on != 0;
    if (on) { h->prof_ms[0] = h->prof_ms[1] = 0; h->prof_n[0] = h->prof_n[1] = 0; }
    return 0;
}
int fe_profile_read(FeEngine* h, char* buf, int buf_len, double* ms_total, long long* launches, int cap) {
    std::snprintf(buf, buf_len, "substep\nsubstep_grad");
    for (int i = 0; i < 2 && i < cap; i++) { ms_total[i] = h->prof_ms[i]; launches[i] = h->prof_n[i]; }
    return 2;
}

} // extern "C"
