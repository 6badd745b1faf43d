// fe_math.h — per-particle / per-node fp32 math of the MLS-MPM substep for gfx950.
//
// Everything here is straight-line register math (3x3 matrices live in VGPRs, loops are
// fully unrolled); the kernels in fe_engine.hip own all memory traffic.  Functions are
// __host__ __device__ so tests/csrc_math_test.cpp can exercise them on the build host.
// Reference semantics are cited as mpm:NNN = fluidlab/fluidengine/simulators/mpm_simulator.py.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define FE_HD __host__ __device__ __forceinline__

// `real` is float in the product; tests/csrc/ builds this header with -DFE_T=double on the host
// to check the hand-derived adjoints against finite differences at full precision.
#ifndef FE_T
#define FE_T float
#endif
typedef FE_T real;
#define R_(x) ((real)(x))

#define FE_EPS R_(1e-12)                 // fluidlab/configs/macros.py:213
#define FE_MAT_LIQUID_ 200
#define FE_MAT_PLASTO_ELASTIC_ 201
#define FE_MAT_ELASTIC_ 202
#define FE_MAT_RIGID_ 203
#define FE_MAT_PLASTO_ELASTIC_DEMO_ 204

struct m3 { real a[3][3]; };
struct v3 { real a[3]; };

FE_HD m3 m3_zero() { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = R_(0.0);
    return r; }
FE_HD m3 m3_ident() { m3 r = m3_zero(); r.a[0][0] = r.a[1][1] = r.a[2][2] = R_(1.0); return r; }
FE_HD m3 m3_mul(const m3& x, const m3& y) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j];
    return r; }
// x * y^T
FE_HD m3 m3_mul_nt(const m3& x, const m3& y) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][0] * y.a[j][0] + x.a[i][1] * y.a[j][1] + x.a[i][2] * y.a[j][2];
    return r; }
// x^T * y
FE_HD m3 m3_mul_tn(const m3& x, const m3& y) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[0][i] * y.a[0][j] + x.a[1][i] * y.a[1][j] + x.a[2][i] * y.a[2][j];
    return r; }
FE_HD m3 m3_T(const m3& x) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[j][i];
    return r; }
FE_HD m3 m3_add(const m3& x, const m3& y) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][j] + y.a[i][j];
    return r; }
FE_HD m3 m3_sub(const m3& x, const m3& y) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][j] - y.a[i][j];
    return r; }
FE_HD m3 m3_scale(const m3& x, real s) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][j] * s;
    return r; }
FE_HD m3 m3_had(const m3& x, const m3& y) { m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][j] * y.a[i][j];
    return r; }
FE_HD real m3_det(const m3& x) {
    return x.a[0][0] * (x.a[1][1] * x.a[2][2] - x.a[1][2] * x.a[2][1])
         - x.a[0][1] * (x.a[1][0] * x.a[2][2] - x.a[1][2] * x.a[2][0])
         + x.a[0][2] * (x.a[1][0] * x.a[2][1] - x.a[1][1] * x.a[2][0]);
}
FE_HD real m3_trace(const m3& x) { return x.a[0][0] + x.a[1][1] + x.a[2][2]; }
// cofactor matrix: d det(x) / d x
FE_HD m3 m3_cof(const m3& x) { m3 r;
    r.a[0][0] = x.a[1][1] * x.a[2][2] - x.a[1][2] * x.a[2][1];
    r.a[0][1] = x.a[1][2] * x.a[2][0] - x.a[1][0] * x.a[2][2];
    r.a[0][2] = x.a[1][0] * x.a[2][1] - x.a[1][1] * x.a[2][0];
    r.a[1][0] = x.a[0][2] * x.a[2][1] - x.a[0][1] * x.a[2][2];
    r.a[1][1] = x.a[0][0] * x.a[2][2] - x.a[0][2] * x.a[2][0];
    r.a[1][2] = x.a[0][1] * x.a[2][0] - x.a[0][0] * x.a[2][1];
    r.a[2][0] = x.a[0][1] * x.a[1][2] - x.a[0][2] * x.a[1][1];
    r.a[2][1] = x.a[0][2] * x.a[1][0] - x.a[0][0] * x.a[1][2];
    r.a[2][2] = x.a[0][0] * x.a[1][1] - x.a[0][1] * x.a[1][0];
    return r; }

// ---------------------------------------------------------------------------------------
// 3x3 SVD, contract of taichi 1.1.0 ti.svd (call site mpm:264): F = U diag(sig) V^T with
// U, V proper rotations, |sig| descending, a negative determinant carried by sig[2].
// One-sided (Hestenes) Jacobi on the columns of F: works on F itself, not F^T F, so fp32
// keeps full relative accuracy for the near-identity F of fluids.  At most 5 sweeps; the loop
// ends early only when a whole sweep rotated nothing in ANY lane of the wave (a wave-uniform
// exit: no divergence).  Such a sweep leaves A and V bit-for-bit unchanged, so every later
// sweep would do the same: the result is that of the fixed 5 sweeps.  (F within 0.3 % of a
// rotation -- the plastic clamp of ICECREAM -- is done after two sweeps; the third finds that out.)
// ---------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define FE_WAVE_ANY(x) (__any(x) != 0)
#else
#define FE_WAVE_ANY(x) (x)
#endif
FE_HD bool svd3_rot(m3& A, m3& V, const int p, const int q) {
    real alpha = A.a[0][p] * A.a[0][p] + A.a[1][p] * A.a[1][p] + A.a[2][p] * A.a[2][p];
    real beta  = A.a[0][q] * A.a[0][q] + A.a[1][q] * A.a[1][q] + A.a[2][q] * A.a[2][q];
    real gamma = A.a[0][p] * A.a[0][q] + A.a[1][p] * A.a[1][q] + A.a[2][p] * A.a[2][q];
    // skip when the columns are already orthogonal to fp32 precision (also avoids 0/0)
    const real tol2 = sizeof(real) == 4 ? R_(1e-15) : R_(1e-31);   // (fp64 only in the host tests)
    bool live = gamma * gamma > tol2 * alpha * beta && fabs(gamma) > R_(1e-30);
    if (!FE_WAVE_ANY(live)) return false;                          // (the identity rotation in every lane: skipped as a whole)
    real g = live ? gamma : R_(1.0);
    real zeta = (beta - alpha) / (R_(2.0) * g);
    real t = copysign(R_(1.0), zeta) / (fabs(zeta) + sqrt(R_(1.0) + zeta * zeta));
    real c = R_(1.0) / sqrt(R_(1.0) + t * t);
    real s = c * t;
    c = live ? c : R_(1.0);
    s = live ? s : R_(0.0);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        real ap = A.a[i][p], aq = A.a[i][q];
        A.a[i][p] = c * ap - s * aq; A.a[i][q] = s * ap + c * aq;
        real vp = V.a[i][p], vq = V.a[i][q];
        V.a[i][p] = c * vp - s * vq; V.a[i][q] = s * vp + c * vq;
    }
    return live;
}

FE_HD void m3_swap_cols(m3& A, int p, int q) {
#pragma unroll
    for (int i = 0; i < 3; i++) { real t = A.a[i][p]; A.a[i][p] = A.a[i][q]; A.a[i][q] = t; }
}

FE_HD void svd3(const m3& F, m3& U, real sig[3], m3& V) {
    m3 A = F;
    V = m3_ident();
#pragma unroll 1
    for (int sweep = 0; sweep < (sizeof(real) == 4 ? 5 : 12); sweep++) {
        bool moved = svd3_rot(A, V, 0, 1);
        moved |= svd3_rot(A, V, 0, 2);
        moved |= svd3_rot(A, V, 1, 2);
        if (!FE_WAVE_ANY(moved)) break;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) sig[j] = sqrt(A.a[0][j] * A.a[0][j] + A.a[1][j] * A.a[1][j] + A.a[2][j] * A.a[2][j]);
    // sort descending: 3-element network, swapping the columns of A and V together
    if (sig[0] < sig[1]) { real t = sig[0]; sig[0] = sig[1]; sig[1] = t; m3_swap_cols(A, 0, 1); m3_swap_cols(V, 0, 1); }
    if (sig[1] < sig[2]) { real t = sig[1]; sig[1] = sig[2]; sig[2] = t; m3_swap_cols(A, 1, 2); m3_swap_cols(V, 1, 2); }
    if (sig[0] < sig[1]) { real t = sig[0]; sig[0] = sig[1]; sig[1] = t; m3_swap_cols(A, 0, 1); m3_swap_cols(V, 0, 1); }
    const real rel = fmax(sig[0] * R_(1e-6), R_(1e-30));
    // column 0
    if (sig[0] > R_(1e-30)) {
        real inv = R_(1.0) / sig[0];
#pragma unroll
        for (int i = 0; i < 3; i++) U.a[i][0] = A.a[i][0] * inv;
    } else { U.a[0][0] = R_(1.0); U.a[1][0] = R_(0.0); U.a[2][0] = R_(0.0); }
    // column 1
    if (sig[1] > rel) {
        real inv = R_(1.0) / sig[1];
#pragma unroll
        for (int i = 0; i < 3; i++) U.a[i][1] = A.a[i][1] * inv;
    } else {
        // any unit vector orthogonal to column 0: e_k - (e_k.u0) u0 with k = argmin |u0_k|
        // (selects only: a runtime-indexed register array would spill to scratch)
        real a0 = fabs(U.a[0][0]), a1 = fabs(U.a[1][0]), a2 = fabs(U.a[2][0]);
        bool k0 = a0 <= a1 && a0 <= a2;
        bool k1 = !k0 && a1 <= a2;
        bool k2 = !k0 && !k1;
        // (arithmetic blend, not ?: — clang folds a select of loads into a dynamically indexed load,
        //  which pins the whole Constitutive struct in scratch)
        real e0 = k0 ? R_(1.0) : R_(0.0), e1 = k1 ? R_(1.0) : R_(0.0), e2 = k2 ? R_(1.0) : R_(0.0);
        real d = e0 * U.a[0][0] + e1 * U.a[1][0] + e2 * U.a[2][0];
        real w0 = e0 - d * U.a[0][0];
        real w1 = e1 - d * U.a[1][0];
        real w2 = e2 - d * U.a[2][0];
        real inv = R_(1.0) / sqrt(w0 * w0 + w1 * w1 + w2 * w2);
        U.a[0][1] = w0 * inv; U.a[1][1] = w1 * inv; U.a[2][1] = w2 * inv;
    }
    // column 2
    if (sig[2] > rel) {
        real inv = R_(1.0) / sig[2];
#pragma unroll
        for (int i = 0; i < 3; i++) U.a[i][2] = A.a[i][2] * inv;
    } else {
        U.a[0][2] = U.a[1][0] * U.a[2][1] - U.a[2][0] * U.a[1][1];
        U.a[1][2] = U.a[2][0] * U.a[0][1] - U.a[0][0] * U.a[2][1];
        U.a[2][2] = U.a[0][0] * U.a[1][1] - U.a[1][0] * U.a[0][1];
    }
    if (m3_det(U) < R_(0.0)) { U.a[0][2] = -U.a[0][2]; U.a[1][2] = -U.a[1][2]; U.a[2][2] = -U.a[2][2]; sig[2] = -sig[2]; }
    if (m3_det(V) < R_(0.0)) { V.a[0][2] = -V.a[0][2]; V.a[1][2] = -V.a[1][2]; V.a[2][2] = -V.a[2][2]; sig[2] = -sig[2]; }
}

// mpm:294-302
FE_HD real svd_clamp(real a) { return a >= R_(0.0) ? fmax(a, R_(1e-8)) : fmin(a, -R_(1e-8)); }

// mpm:272-292 with a diagonal grad_S (the only form p2g's adjoint produces)
FE_HD m3 backward_svd(const m3& gU, const real gS[3], const m3& gV, const m3& U, const real sig[3], const m3& V) {
    real s2[3] = {sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2]};
    m3 a = m3_mul_tn(U, gU);      // U^T gU
    m3 b = m3_mul_tn(V, gV);      // V^T gV
    // inner = (Fm o (a - a^T)) S  +  S (Fm o (b - b^T))  +  diag(gS)
    m3 inner;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (i == j) inner.a[i][j] = gS[i];
            else {
                real fm = R_(1.0) / svd_clamp(s2[j] - s2[i]);
                inner.a[i][j] = fm * (a.a[i][j] - a.a[j][i]) * sig[j] + sig[i] * fm * (b.a[i][j] - b.a[j][i]);
            }
        }
    return m3_mul_nt(m3_mul(U, inner), V);
}

// ---------------------------------------------------------------------------------------
// boundaries (fluidlab/fluidengine/boundaries/boundaries.py)
// ---------------------------------------------------------------------------------------
struct BoundaryP {
    int   type;          // 0 cube, 1 cylinder
    real lower[3], upper[3];
    real cx, cz, radius, restitution;
    int   lock_dims;
};

// ---- SDF colliders (fluidlab/fluidengine/meshes/static.py:25-104, dynamic.py:29-122) --------------------------------
struct SdfP {
    int res;                 // voxels [res]^3, C order
    const real* vox;
    real T[12];              // rows 0..2 of T_mesh_to_voxels (mesh.py:121)
    real Rinv[9];            // inverse(T[:3,:3]) (static.py:58)
    real friction, softness;
};
// Static.sdf_ (static.py:34-49): trilinear sample in voxel coordinates, 1 outside the voxel box
FE_HD real sdf_sample(const SdfP& s, const real pv[3]) {
    int b[3];
    bool outside = false;
#pragma unroll
    for (int d = 0; d < 3; d++) { b[d] = (int)floor(pv[d]); outside = outside || b[d] >= s.res - 1 || b[d] < 0; }
    if (outside) return R_(1.0);
    real f[3] = {pv[0] - (real)b[0], pv[1] - (real)b[1], pv[2] - (real)b[2]};
    const real* v = s.vox + ((size_t)b[0] * s.res + b[1]) * s.res + b[2];
    const size_t sy = s.res, sx = (size_t)s.res * s.res;
    real sd = R_(0.0);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                real w = (i ? f[0] : R_(1.0) - f[0]) * (j ? f[1] : R_(1.0) - f[1]) * (k ? f[2] : R_(1.0) - f[2]);
                sd += w * v[i * sx + j * sy + k];
            }
    return sd;
}
FE_HD void sdf_to_voxels(const SdfP& s, const real p[3], real pv[3]) {
    for (int d = 0; d < 3; d++) pv[d] = s.T[d * 4] * p[0] + s.T[d * 4 + 1] * p[1] + s.T[d * 4 + 2] * p[2] + s.T[d * 4 + 3];
}
// Static.normal (static.py:52-80): central differences (delta = 1e-2 voxels), normalised, rotated back, normalised
FE_HD void sdf_normal(const SdfP& s, const real pv[3], real n[3]) {
    const real delta = R_(1e-2);
    real g[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {                    // (selects, not a runtime-indexed array: that would live in scratch)
        const real inc[3] = {pv[0] + (d == 0 ? delta : R_(0.0)), pv[1] + (d == 1 ? delta : R_(0.0)), pv[2] + (d == 2 ? delta : R_(0.0))};
        const real dec[3] = {pv[0] - (d == 0 ? delta : R_(0.0)), pv[1] - (d == 1 ? delta : R_(0.0)), pv[2] - (d == 2 ? delta : R_(0.0))};
        g[d] = (sdf_sample(s, inc) - sdf_sample(s, dec)) / (R_(2.0) * delta);
    }
    real nn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + FE_EPS);
    for (int d = 0; d < 3; d++) g[d] /= nn;
    for (int d = 0; d < 3; d++) n[d] = s.Rinv[d * 3] * g[0] + s.Rinv[d * 3 + 1] * g[1] + s.Rinv[d * 3 + 2] * g[2];
    nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + FE_EPS);
    for (int d = 0; d < 3; d++) n[d] /= nn;
}
// Contact law of Static.collide / Dynamic.collide (static.py:88-101) on the velocity relative to the collider.
// `g`, if given, is pulled back through it in place (normal held fixed; min/max pass the gradient to the selected
// operand).  The quotient vt/|vt| is only formed when flag = 1 (the reference's NaN * 0 for an exactly normal impact
// is not reproduced).
FE_HD void contact_law(const real n[3], real friction, const real rv[3], real out[3], real* g) {
    const real nc = rv[0] * n[0] + rv[1] * n[1] + rv[2] * n[2];
    const real a = fmin(nc, R_(0.0));
    real vt[3] = {rv[0] - a * n[0], rv[1] - a * n[1], rv[2] - a * n[2]};
    const real vtn = sqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
    const bool flag = nc < R_(0.0) && vtn > FE_EPS;
    const real t = vtn + nc * friction;
    const real sc = flag ? fmax(R_(0.0), t) / vtn : R_(1.0);
    for (int d = 0; d < 3; d++) out[d] = vt[d] * sc;
    if (!g) return;
    real gvt[3], gnc = R_(0.0);
    if (flag && t > R_(0.0)) {
        const real u[3] = {vt[0] / vtn, vt[1] / vtn, vt[2] / vtn};
        const real ug = u[0] * g[0] + u[1] * g[1] + u[2] * g[2];
        for (int d = 0; d < 3; d++) gvt[d] = g[d] + friction * nc * (g[d] - u[d] * ug) / vtn;
        gnc = friction * ug;
    } else if (flag) {
        gvt[0] = gvt[1] = gvt[2] = R_(0.0);
    } else {
        for (int d = 0; d < 3; d++) gvt[d] = g[d];
    }
    const real ga = -(n[0] * gvt[0] + n[1] * gvt[1] + n[2] * gvt[2]);
    if (nc < R_(0.0)) gnc += ga;
    for (int d = 0; d < 3; d++) g[d] = gvt[d] + gnc * n[d];
}
// Static.collide (static.py:82-103) at world position `pos`; with g != nullptr also pulls g back
FE_HD void static_collide(const SdfP& s, const real pos[3], real v[3], real* g) {
    real pv[3];
    sdf_to_voxels(s, pos, pv);
    if (sdf_sample(s, pv) > R_(0.0)) return;
    real n[3], out[3];
    sdf_normal(s, pv, n);
    contact_law(n, s.friction, v, out, g);
    for (int d = 0; d < 3; d++) v[d] = out[d];
}

// impose_x_v velocity part (boundaries.py:40-63, 107-121): multiplies v in place, returns multipliers
FE_HD void boundary_v(const BoundaryP& b, const real x[3], real v[3], real k[3]) {
    k[0] = k[1] = k[2] = R_(1.0);
    if (b.type == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (x[i] >= b.upper[i] && v[i] >= R_(0.0)) k[i] = -b.restitution;
            else if (x[i] <= b.lower[i] && v[i] <= R_(0.0)) k[i] = -b.restitution;
        }
    } else {
        if (x[1] > b.upper[1] && v[1] > R_(0.0)) k[1] = -b.restitution;
        else if (x[1] < b.lower[1] && v[1] < R_(0.0)) k[1] = -b.restitution;
        real rx = x[0] - b.cx, rz = x[2] - b.cz;
        real nrm = sqrt(rx * rx + rz * rz + FE_EPS);
        if (nrm > b.radius) { k[0] = R_(0.0); k[2] = R_(0.0); }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (b.lock_dims & (1 << i)) k[i] = R_(0.0);
        v[i] = (k[i] == R_(1.0)) ? v[i] : ((k[i] == R_(0.0)) ? R_(0.0) : v[i] * k[i]);
    }
}

// Boundary.is_out (boundaries.py:80-93 cylinder, 127-134 cube)
FE_HD bool boundary_is_out(const BoundaryP& b, const real x[3]) {
    if (b.type == 0) {
        bool out = false;
#pragma unroll
        for (int i = 0; i < 3; i++) out = out || x[i] > b.upper[i] || x[i] < b.lower[i];
        return out;
    }
    const real rx = x[0] - b.cx, rz = x[2] - b.cz;
    return x[1] > b.upper[1] || x[1] < b.lower[1] || sqrt(rx * rx + rz * rz + FE_EPS) > b.radius;
}

// impose_x (boundaries.py:66-78, 123-126) and its Jacobian under Taichi's min/max adjoint rules
FE_HD void boundary_x(const BoundaryP& b, const real x[3], real xn[3], real J[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) J[i][j] = R_(0.0);
        real m = fmin(x[i], b.upper[i]);
        xn[i] = fmax(m, b.lower[i]);
        J[i][i] = (x[i] < b.upper[i] && m > b.lower[i]) ? R_(1.0) : R_(0.0);
    }
    if (b.type == 1) {
        real rx = x[0] - b.cx, rz = x[2] - b.cz;
        real nrm = sqrt(rx * rx + rz * rz + FE_EPS);
        if (nrm > b.radius) {
            xn[0] = rx / nrm * b.radius + b.cx;
            xn[2] = rz / nrm * b.radius + b.cz;
            real k = b.radius / nrm, k3 = b.radius / (nrm * nrm * nrm);
            J[0][0] = k - k3 * rx * rx; J[0][2] = -k3 * rx * rz;
            J[2][0] = -k3 * rz * rx;    J[2][2] = k - k3 * rz * rz;
        }
    }
}

// ---------------------------------------------------------------------------------------
// quaternions (fluidlab/utils/geom.py:8-28, 97-102)
// ---------------------------------------------------------------------------------------
FE_HD void quat_mul(const real q[4], const real r[4], real out[4]) {
    real w = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    real x = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    real y = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    real z = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
    real inv = R_(1.0) / sqrt(w * w + x * x + y * y + z * z);
    out[0] = w * inv; out[1] = x * inv; out[2] = y * inv; out[3] = z * inv;
}
FE_HD void quat_from_w(const real aa[3], real out[4]) {
    real w = sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2] + FE_EPS);
    real sh = sin(w * R_(0.5));
    out[0] = cos(w * R_(0.5)); out[1] = aa[0] / w * sh; out[2] = aa[1] / w * sh; out[3] = aa[2] / w * sh;
}
FE_HD void quat_rotate(const real v[3], const real q[4], real out[3]) {
    real ux = q[2] * v[2] - q[3] * v[1], uy = q[3] * v[0] - q[1] * v[2], uz = q[1] * v[1] - q[2] * v[0];
    real wx = q[2] * uz - q[3] * uy, wy = q[3] * ux - q[1] * uz, wz = q[1] * uy - q[2] * ux;
    out[0] = v[0] + R_(2.0) * (q[0] * ux + wx);
    out[1] = v[1] + R_(2.0) * (q[0] * uy + wy);
    out[2] = v[2] + R_(2.0) * (q[0] * uz + wz);
}

// ---------------------------------------------------------------------------------------
// Dynamic (moving) SDF collider of a Rigid effector -- dynamic.py:29-122 -- and the quaternion part of move_kernel
// (effector.py:161), written over a scalar type T.  T = real evaluates; T = Dual carries one tangent, so a call yields
// one Jacobian column and an adjoint is assembled column by column (forward mode: the collide adjoint has 20 inputs and
// only the few particles in contact pay for it).  floor/comparisons act on values; min/max pass the tangent of the
// selected operand (Taichi's rule); the quotient vt/|vt| is only formed when the friction branch is taken.
// ---------------------------------------------------------------------------------------
struct Dual {
    real v, d;
    FE_HD Dual() : v(R_(0.0)), d(R_(0.0)) {}
    FE_HD Dual(real v_) : v(v_), d(R_(0.0)) {}
    FE_HD Dual(real v_, real d_) : v(v_), d(d_) {}
};
FE_HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
FE_HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
FE_HD Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
FE_HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
FE_HD Dual operator/(Dual a, Dual b) { real i = R_(1.0) / b.v; return Dual(a.v * i, (a.d - a.v * i * b.d) * i); }
FE_HD Dual t_sqrt(Dual a) { real r = sqrt(a.v); return Dual(r, a.d / (R_(2.0) * r)); }
FE_HD Dual t_exp(Dual a) { real r = exp(a.v); return Dual(r, a.d * r); }
FE_HD Dual t_sin(Dual a) { return Dual(sin(a.v), a.d * cos(a.v)); }
FE_HD Dual t_cos(Dual a) { return Dual(cos(a.v), -a.d * sin(a.v)); }
FE_HD Dual t_abs(Dual a) { return a.v >= R_(0.0) ? a : -a; }
FE_HD real t_sqrt(real a) { return sqrt(a); }
FE_HD real t_exp(real a) { return exp(a); }
FE_HD real t_sin(real a) { return sin(a); }
FE_HD real t_cos(real a) { return cos(a); }
FE_HD real t_abs(real a) { return fabs(a); }
FE_HD real t_val(real a) { return a; }
FE_HD real t_val(Dual a) { return a.v; }

template <class T> FE_HD void t_quat_rotate(const T v[3], const T q[4], T out[3]) {              // geom.py:97-102
    T ux = q[2] * v[2] - q[3] * v[1], uy = q[3] * v[0] - q[1] * v[2], uz = q[1] * v[1] - q[2] * v[0];
    T wx = q[2] * uz - q[3] * uy, wy = q[3] * ux - q[1] * uz, wz = q[1] * uy - q[2] * ux;
    out[0] = v[0] + T(R_(2.0)) * (q[0] * ux + wx);
    out[1] = v[1] + T(R_(2.0)) * (q[0] * uy + wy);
    out[2] = v[2] + T(R_(2.0)) * (q[0] * uz + wz);
}
template <class T> FE_HD T t_sdf_sample(const SdfP& s, const T pv[3]) {                          // dynamic.py:39-54
    int b[3];
    bool outside = false;
#pragma unroll
    for (int d = 0; d < 3; d++) { b[d] = (int)floor(t_val(pv[d])); outside = outside || b[d] >= s.res - 1 || b[d] < 0; }
    if (outside) return T(R_(1.0));
    const T f0 = pv[0] - T((real)b[0]), f1 = pv[1] - T((real)b[1]), f2 = pv[2] - T((real)b[2]);   // in [0,1): |.| is the identity
    const T g0 = T(R_(1.0)) - f0, g1 = T(R_(1.0)) - f1, g2 = T(R_(1.0)) - f2;
    const real* v = s.vox + ((size_t)b[0] * s.res + b[1]) * s.res + b[2];
    const size_t sy = s.res, sx = (size_t)s.res * s.res;
    T sd = g0 * g1 * g2 * T(v[0]);
    sd = sd + g0 * g1 * f2 * T(v[1]);
    sd = sd + g0 * f1 * g2 * T(v[sy]);
    sd = sd + g0 * f1 * f2 * T(v[sy + 1]);
    sd = sd + f0 * g1 * g2 * T(v[sx]);
    sd = sd + f0 * g1 * f2 * T(v[sx + 1]);
    sd = sd + f0 * f1 * g2 * T(v[sx + sy]);
    sd = sd + f0 * f1 * f2 * T(v[sx + sy + 1]);
    return sd;
}
template <class T> FE_HD void t_normalize3(T v[3]) {                                             // geom.py:93-94
    T inv = T(R_(1.0)) / t_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + T(FE_EPS));
    v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv;
}
// returns whether the contact branch was taken (otherwise out = mv)
template <class T> FE_HD bool t_dynamic_collide(const SdfP& s, const T p0[3], const T q0[4], const T p1[3], const T q1[4],
                                                const T pos[3], const T mv[3], real dt, T out[3]) {
    const T rel0[3] = {pos[0] - p0[0], pos[1] - p0[1], pos[2] - p0[2]};
    const T qn = T(R_(1.0)) / t_sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);     // inv_quat(...).normalized()
    const T qi[4] = {q0[0] * qn, -(q0[1] * qn), -(q0[2] * qn), -(q0[3] * qn)};
    T pm[3], pv[3];
    t_quat_rotate(rel0, qi, pm);
#pragma unroll
    for (int d = 0; d < 3; d++) pv[d] = T(s.T[d * 4]) * pm[0] + T(s.T[d * 4 + 1]) * pm[1] + T(s.T[d * 4 + 2]) * pm[2] + T(s.T[d * 4 + 3]);
    const T sd = t_sdf_sample(s, pv);
    T infl = t_exp(-sd * T(s.softness));
    if (t_val(infl) > R_(1.0)) infl = T(R_(1.0));
    out[0] = mv[0]; out[1] = mv[1]; out[2] = mv[2];
    if (!(t_val(sd) <= R_(0.0) || (s.softness > R_(0.0) && t_val(infl) > R_(0.1)))) return false;
    T pn[3], cv[3];
    t_quat_rotate(pm, q1, pn);
    const T idt = T(R_(1.0) / dt);
#pragma unroll
    for (int d = 0; d < 3; d++) cv[d] = (pn[d] + p1[d] - pos[d]) * idt;                          // collider_v, dynamic.py:90-94
    if (s.friction > R_(10.0)) { out[0] = cv[0]; out[1] = cv[1]; out[2] = cv[2]; return true; }
    const T rel[3] = {mv[0] - cv[0], mv[1] - cv[1], mv[2] - cv[2]};
    const real delta = R_(1e-2);
    T g[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {                                                                // normal_, dynamic.py:73-88
        const T inc[3] = {pv[0] + T(d == 0 ? delta : R_(0.0)), pv[1] + T(d == 1 ? delta : R_(0.0)), pv[2] + T(d == 2 ? delta : R_(0.0))};
        const T dec[3] = {pv[0] - T(d == 0 ? delta : R_(0.0)), pv[1] - T(d == 1 ? delta : R_(0.0)), pv[2] - T(d == 2 ? delta : R_(0.0))};
        g[d] = (t_sdf_sample(s, inc) - t_sdf_sample(s, dec)) * T(R_(1.0) / (R_(2.0) * delta));
    }
    t_normalize3(g);
    T nm[3], n[3];
#pragma unroll
    for (int d = 0; d < 3; d++) nm[d] = T(s.Rinv[d * 3]) * g[0] + T(s.Rinv[d * 3 + 1]) * g[1] + T(s.Rinv[d * 3 + 2]) * g[2];
    t_quat_rotate(nm, q0, n);
    t_normalize3(n);
    const T nc = rel[0] * n[0] + rel[1] * n[1] + rel[2] * n[2];
    const T a = t_val(nc) < R_(0.0) ? nc : T(R_(0.0));
    T vt[3] = {rel[0] - a * n[0], rel[1] - a * n[1], rel[2] - a * n[2]};
    const real vtn_v = sqrt(t_val(vt[0]) * t_val(vt[0]) + t_val(vt[1]) * t_val(vt[1]) + t_val(vt[2]) * t_val(vt[2]));
    if (t_val(nc) < R_(0.0) && vtn_v > FE_EPS) {
        const T vtn = t_sqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
        const T t = vtn + nc * T(s.friction);
        const T sc = t_val(t) > R_(0.0) ? t / vtn : T(R_(0.0));
        vt[0] = vt[0] * sc; vt[1] = vt[1] * sc; vt[2] = vt[2] * sc;
    }
    const T keep = T(R_(1.0)) - infl;
#pragma unroll
    for (int d = 0; d < 3; d++) out[d] = cv[d] + vt[d] * infl + rel[d] * keep;
    return true;
}
// quat[f+1] = qmul(w2quat(w[f]), quat[f])  (effector.py:161, geom.py:8-28)
template <class T> FE_HD void t_move_quat(const T w[3], const T q[4], T out[4]) {
    const T wn = t_sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + T(FE_EPS));
    const T sh = t_sin(wn * T(R_(0.5))) / wn;
    const T a[4] = {t_cos(wn * T(R_(0.5))), w[0] * sh, w[1] * sh, w[2] * sh};
    T o[4];
    o[0] = q[0] * a[0] - q[1] * a[1] - q[2] * a[2] - q[3] * a[3];
    o[1] = q[0] * a[1] + q[1] * a[0] - q[2] * a[3] + q[3] * a[2];
    o[2] = q[0] * a[2] + q[1] * a[3] + q[2] * a[0] - q[3] * a[1];
    o[3] = q[0] * a[3] - q[1] * a[2] + q[2] * a[1] + q[3] * a[0];
    const T inv = T(R_(1.0)) / t_sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int i = 0; i < 4; i++) out[i] = o[i] * inv;
}

// ---------------------------------------------------------------------------------------
// quadratic B-spline stencil (mpm:335-337)
// ---------------------------------------------------------------------------------------
struct Stencil {
    int   base[3];
    real fx[3];
    real w[3][3];    // w[i][d]: weight of offset i along dimension d
};
FE_HD void stencil_make(const real x[3], real inv_dx, Stencil& s) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
        s.base[d] = (int)(x[d] * inv_dx - R_(0.5));     // truncation toward zero == Taichi cast(int)
        real fx = x[d] * inv_dx - (real)s.base[d];
        s.fx[d] = fx;
        s.w[0][d] = R_(0.5) * (R_(1.5) - fx) * (R_(1.5) - fx);
        s.w[1][d] = R_(0.75) - (fx - R_(1.0)) * (fx - R_(1.0));
        s.w[2][d] = R_(0.5) * (fx - R_(0.5)) * (fx - R_(0.5));
    }
}
FE_HD real stencil_dw(const Stencil& s, int i, int d) {
    real fx = s.fx[d];
    return i == 0 ? -(R_(1.5) - fx) : (i == 1 ? -R_(2.0) * (fx - R_(1.0)) : fx - R_(0.5));
}
FE_HD bool stencil_inside(const Stencil& s, int n) {
    // compare against n - 3, not base + 2 < n: v_cvt_i32_f32 saturates an exploded coordinate to INT_MAX and +2 would wrap
    return s.base[0] >= 0 && s.base[1] >= 0 && s.base[2] >= 0 && s.base[0] <= n - 3 && s.base[1] <= n - 3 && s.base[2] <= n - 3;
}

// ---------------------------------------------------------------------------------------
// constitutive model: everything p2g needs from (C, F) of one particle (mpm:254-264, 339-378)
// ---------------------------------------------------------------------------------------
struct Constitutive {
    m3    Ft;         // F_tmp = (I + dt C) F
    m3    U, V;       // only valid when `full`
    real sig[3];     // only valid when `full`
    real J;          // det S
    m3    affine;     // stress*scale + mass*C   (mpm:342-344)
    m3    Fnew;       // F[f+1]                  (mpm:355-378)
    bool  full;       // SVD was needed (mu != 0 or a non-liquid class)
};

// J^(1/3) of the liquid's F update (mpm:359: ti.pow(J, 1/3), NaN for J < 0) and J^(-2/3) of its adjoint.  powf() is ~130 VALU
// instructions of special-casing per call on gfx950 -- 7 % of what a wave of k_p2g issues per unit, 10 % of k_p2g_grad's
// (scripts/valu_profile.py) -- for an argument that is within a few per cent of 1 in any state worth simulating.  In fp32 the root
// comes from the hardware's log2 / exp2 (v_log_f32, v_exp_f32: ~1e-6 relative together) and ONE Newton step on c^3 = J, which squares
// that error: the result is within an ulp of the correctly rounded root (tests/csrc/math_test.cpp checks the same code against pow in
// fp64 on the host).  log2 of a negative J is NaN and stays NaN, J = 0 gives 0, like pow.  fp64 (host checks) keeps pow.
FE_HD real fe_cbrt_pos(real J) {
    if (sizeof(real) == 8) return pow(J, R_(1.0) / R_(3.0));
#if defined(__HIP_DEVICE_COMPILE__)
    const float c0 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)J) * (1.f / 3.f));
    const float c2 = c0 * c0;
    const float c1 = c0 - __builtin_fmaf(c2, c0, -(float)J) * __builtin_amdgcn_rcpf(3.f * c2);
#else
    const float c0 = exp2f(log2f((float)J) * (1.f / 3.f));
    const float c2 = c0 * c0;
    const float c1 = c0 - fmaf(c2, c0, -(float)J) / (3.f * c2);
#endif
    // (the Newton step needs a finite, non-zero first guess: a denormal J -- log2 flushes it: c0 = 0 -- and J = inf -- c0 = inf -- keep c0, as pow would
    //  give ~0 and inf there; the states of a collapse or a blow-up, but no NaN of the engine's own making.  ADVICE r5)
    return (real)((float)J == 0.f ? 0.f : ((c0 > 0.f && c0 < 3.0e38f) ? c1 : c0));
}
// J^(-2/3) = J^(1/3) / J
FE_HD real fe_pow_m23(real J) {
    if (sizeof(real) == 8) return pow(J, R_(1.0) / R_(3.0) - R_(1.0));
#if defined(__HIP_DEVICE_COMPILE__)
    return (real)((float)fe_cbrt_pos(J) * __builtin_amdgcn_rcpf((float)J));
#else
    return fe_cbrt_pos(J) / J;
#endif
}

// `scale` = -dt * p_vol * 4 * inv_dx^2 (mpm:343).  GENERAL=false is the specialisation the engine launches when
// every particle of the scene is an inviscid liquid: the SVD path is compiled out (fewer registers, more waves).
template <bool GENERAL>
FE_HD void constitutive_eval_t(const m3& C, const m3& F, real dt, real mu, real lam, real mass, int cls_, real scale,
                               Constitutive& k) {
    const int cls = GENERAL ? cls_ : FE_MAT_LIQUID_;
    m3 IdtC = m3_scale(C, dt);
    IdtC.a[0][0] += R_(1.0); IdtC.a[1][1] += R_(1.0); IdtC.a[2][2] += R_(1.0);
    k.Ft = m3_mul(IdtC, F);
    // An inviscid liquid (mu == 0, MAT_LIQUID: WATER/MILK/COFFEE) consumes only J = det S = det F_tmp
    // (U, V proper rotations), so the SVD is skipped: stress = lam J (J-1) I and F_new = J^(1/3) I.
    k.full = GENERAL && !(mu == R_(0.0) && cls == FE_MAT_LIQUID_);
    m3 stress = m3_zero();
    if (k.full) {
        svd3(k.Ft, k.U, k.sig, k.V);
        k.J = k.sig[0] * k.sig[1] * k.sig[2];
        m3 r = m3_mul_nt(k.U, k.V);
        stress = m3_scale(m3_mul_nt(m3_sub(k.Ft, r), k.Ft), R_(2.0) * mu);
    } else {
        k.J = m3_det(k.Ft);
    }
    real iso = lam * k.J * (k.J - R_(1.0));
    stress.a[0][0] += iso; stress.a[1][1] += iso; stress.a[2][2] += iso;
    k.affine = m3_add(m3_scale(stress, scale), m3_scale(C, mass));
    if (cls == FE_MAT_LIQUID_) {
        real c = fe_cbrt_pos(k.J);                       // pow(J, 1/3): NaN for J < 0, like ti.pow (mpm:359)
        k.Fnew = m3_zero(); k.Fnew.a[0][0] = k.Fnew.a[1][1] = k.Fnew.a[2][2] = c;
    } else if (cls == FE_MAT_ELASTIC_ || cls == FE_MAT_RIGID_) {
        k.Fnew = k.Ft;
    } else {    // PLASTO_ELASTIC and PLASTO_ELASTIC_DEMO are identical (mpm:367-376)
        m3 US = k.U;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            real sn = fmin(fmax(k.sig[d], R_(1.0) - R_(2e-3)), R_(1.0) + R_(3e-3));
            US.a[0][d] *= sn; US.a[1][d] *= sn; US.a[2][d] *= sn;
        }
        k.Fnew = m3_mul_nt(US, k.V);
    }
}

FE_HD void constitutive_eval(const m3& C, const m3& F, real dt, real mu, real lam, real mass, int cls, real scale,
                             Constitutive& k) {
    constitutive_eval_t<true>(C, F, dt, mu, lam, mass, cls, scale, k);
}

// Adjoint of constitutive_eval: given GA = d/d(affine) (already including the sum over nodes) and
// Fg = d/d(F[f+1]), accumulate gC, gF (the adjoints of C[f], F[f]).  Follows p2g.grad, svd_grad and
// compute_F_tmp.grad (mpm:544-546); closed forms in SURVEY.md Appendix A.
template <bool GENERAL>
FE_HD void constitutive_grad_t(const m3& C, const m3& F, real dt, real mu, real lam, real mass, int cls_, real scale,
                               const Constitutive& k, const m3& GA, const m3& Fg, m3& gC, m3& gF) {
    const int cls = GENERAL ? cls_ : FE_MAT_LIQUID_;
    gC = m3_scale(GA, mass);                       // affine = stress + m C
    m3 gs = m3_scale(GA, scale);                   // adjoint of the unscaled stress
    real gJ = lam * (R_(2.0) * k.J - R_(1.0)) * m3_trace(gs);
    m3 gFt;
    if (cls == FE_MAT_LIQUID_) gJ += (R_(1.0) / R_(3.0)) * fe_pow_m23(k.J) * m3_trace(Fg);
    if (!GENERAL || !k.full) {
        // J = det F_tmp  =>  d J / d F_tmp = cof(F_tmp)
        gFt = m3_scale(m3_cof(k.Ft), gJ);
    } else {
        m3 r = m3_mul_nt(k.U, k.V);
        m3 P = m3_sub(k.Ft, r);
        real mu2 = R_(2.0) * mu;
        gFt = m3_scale(m3_add(m3_mul(gs, k.Ft), m3_mul_tn(gs, P)), mu2);
        m3 gr = m3_scale(m3_mul(gs, k.Ft), -mu2);
        m3 gU = m3_mul(gr, k.V);
        m3 gV = m3_mul_tn(gr, k.U);
        real gS[3] = {R_(0.0), R_(0.0), R_(0.0)};
        if (cls == FE_MAT_ELASTIC_ || cls == FE_MAT_RIGID_) {
            gFt = m3_add(gFt, Fg);
        } else if (cls != FE_MAT_LIQUID_) {
            const real lo = R_(1.0) - R_(2e-3), hi = R_(1.0) + R_(3e-3);
            m3 UtFgV = m3_mul(m3_mul_tn(k.U, Fg), k.V);
            m3 FgV = m3_mul(Fg, k.V), FgtU = m3_mul_tn(Fg, k.U);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                real sd = k.sig[d];
                real sn = fmin(fmax(sd, lo), hi);
                bool pass = (sd > lo) && (fmax(sd, lo) < hi);      // Taichi max/min adjoint tie rules
                if (pass) gS[d] += UtFgV.a[d][d];
#pragma unroll
                for (int i = 0; i < 3; i++) { gU.a[i][d] += FgV.a[i][d] * sn; gV.a[i][d] += FgtU.a[i][d] * sn; }
            }
        }
        gS[0] += gJ * k.sig[1] * k.sig[2];
        gS[1] += gJ * k.sig[0] * k.sig[2];
        gS[2] += gJ * k.sig[0] * k.sig[1];
        gFt = m3_add(gFt, backward_svd(gU, gS, gV, k.U, k.sig, k.V));
    }
    // F_tmp = (I + dt C) F
    gC = m3_add(gC, m3_scale(m3_mul_nt(gFt, F), dt));
    m3 IdtC = m3_scale(C, dt);
    IdtC.a[0][0] += R_(1.0); IdtC.a[1][1] += R_(1.0); IdtC.a[2][2] += R_(1.0);
    gF = m3_mul_tn(IdtC, gFt);
}

FE_HD void constitutive_grad(const m3& C, const m3& F, real dt, real mu, real lam, real mass, int cls, real scale,
                             const Constitutive& k, const m3& GA, const m3& Fg, m3& gC, m3& gF) {
    constitutive_grad_t<true>(C, F, dt, mu, lam, mass, cls, scale, k, GA, Fg, gC, gF);
}
