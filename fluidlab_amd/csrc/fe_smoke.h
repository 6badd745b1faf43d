// fe_smoke.h — SmokeField on the MI355X (fluidlab/fluidengine/simulators/smoke_field.py), included by fe_engine.hip.
//
// An Eulerian smoke / temperature solver on its own res^3 grid, stepped once per *step* (mpm:745-747, 765-767).  Only a
// y-slab of the grid is free space (lower_y < j < higher_y minus the static colliders): every launch covers the slab's
// bounding box only (7 of 128 layers in Circulation-v0) instead of the reference's full-grid sweeps.
//
//   forward  : free space -> RK3 back-trace advection + AirCon impulse -> divergence -> `solver_iters` Jacobi sweeps ->
//              pressure-gradient subtraction                                            (smoke_field.py:95-111)
//   backward : the hand-derived adjoint of each of those, in reverse                    (smoke_field.py:113-128)
//
// Adjoint formulation: every stencil adjoint (subtract_gradient, Jacobi, divergence) is written as a *gather* over the
// cells that read the target — the neighbour rule of compute_location (298-306) is invertible: cell c reads t as its
// `dir` neighbour iff (c = t - dir, both free) or (c = t, free, t + dir out of range or not free) — so those kernels need
// no atomics and are deterministic.  Only the advection adjoint scatters (5 trilinear samples x 8 cells) with float atomics.
// The Jacobi sweeps are latency-bound launches of ~100k cells; they are captured once into a HIP graph per direction
// (forward / adjoint) and replayed per step, the per-frame pointers being read through a small device-side table.
//
// Layout: frames s in [0, max_steps_local]; per frame v, v_tmp [n3][3], div, p [n3], q [n3][q_dim] (+ the same for the
// adjoints) and the free mask [n3] (u8).  Positions are in cell units (cell centre = index + 0.5).
// Conscious fix (as in the oracle): compute_location falls back to the *clamped* index when the clamped cell is not free;
// the reference falls back to the unclamped one, which is out of bounds outside the grid.

struct SmokeFrameP {                 // what the graph-captured Jacobi kernels need of the current frame
    const unsigned char* fr;
    const float* dv;
    float* gdv;
};

struct SmokeP {
    int n, S, qd, iters, ly, hy;     // res, max_steps_local, q_dim, solver_iters, slab: ly < j < hy
    float dt, low_T;
    size_t n3;
    float *v, *vt, *dv, *p, *q, *gv, *gvt, *gdv, *gp, *gq;
    unsigned char* fr;
    float *pc, *pn, *gpc, *gpn;      // p_swap.cur / nxt + grads
    SmokeFrameP* cur;                // device copy of the current frame's pointers
};

struct SmokeState {
    SmokeP P;
    FeSmokeConfig cfg;
    hipGraphExec_t g_fwd = nullptr, g_bwd = nullptr;
    float* red = nullptr;            // 9 floats: AirCon adjoint reduction (pos 3, quat 4, s, r)
};

__device__ __forceinline__ size_t sm_idx(const SmokeP& P, int i, int j, int k) { return ((size_t)i * P.n + j) * P.n + k; }
__device__ __forceinline__ int sm_clampi(int a, int n) { return a < 0 ? 0 : (a > n - 1 ? n - 1 : a); }
// slab-bounded launch: thread -> (i, j, k) with ly < j < hy
__device__ __forceinline__ bool sm_cell(const SmokeP& P, int& i, int& j, int& k) {
    const int nj = P.hy - P.ly - 1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (nj <= 0 || t >= (long long)P.n * nj * P.n) return false;
    k = (int)(t % P.n); j = P.ly + 1 + (int)((t / P.n) % nj); i = (int)(t / ((long long)P.n * nj));
    return j < P.n && j >= 0;
}
// compute_location, smoke_field.py:298-306 (with the clamped fallback)
__device__ __forceinline__ size_t sm_loc(const SmokeP& P, const unsigned char* fr, int u, int v, int w, int du, int dv, int dw) {
    int a = sm_clampi(u + du, P.n), b = sm_clampi(v + dv, P.n), c = sm_clampi(w + dw, P.n);
    if (!fr[sm_idx(P, a, b, c)]) { a = sm_clampi(u, P.n); b = sm_clampi(v, P.n); c = sm_clampi(w, P.n); }
    return sm_idx(P, a, b, c);
}
// is_free, smoke_field.py:309-320
__device__ __forceinline__ bool sm_isfree(const SmokeP& P, const unsigned char* fr, int u, int v, int w, int du, int dv, int dw) {
    const int a = u + du, b = v + dv, c = w + dw;
    if (a < 0 || b < 0 || c < 0 || a > P.n - 1 || b > P.n - 1 || c > P.n - 1) return false;
    return fr[sm_idx(P, a, b, c)] != 0;
}
__device__ __constant__ const int SM_NB[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};

// compute_free_space, smoke_field.py:191-201
__global__ __launch_bounds__(256) void k_smoke_free(SmokeP P, int s, StaticsP ST) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const float dx = 1.f / (float)P.n;
    const float pw[3] = {((float)i + 0.5f) * dx, ((float)j + 0.5f) * dx, ((float)k + 0.5f) * dx};
    unsigned char f = 1;
    for (int si = 0; si < ST.n; si++) { float pv[3]; sdf_to_voxels(ST.s[si], pw, pv); if (sdf_sample(ST.s[si], pv) <= 0.f) f = 0; }   // is_collide, static.py:105-113
    P.fr[(size_t)s * P.n3 + sm_idx(P, i, j, k)] = f;
}

// trilerp, smoke_field.py:322-343: 8 cells, their weights and d weight / d p
struct SmTri { size_t cell[8]; float w[8]; float f[3]; };
template <int C>
__device__ __forceinline__ void sm_trilerp(const SmokeP& P, const unsigned char* fr, const float* field, const float p[3], float* out, SmTri& t) {
    int base[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { base[d] = (int)floorf(p[d] - 0.5f); t.f[d] = p[d] - 0.5f - (float)base[d]; }
    for (int c = 0; c < C; c++) out[c] = 0.f;
    float wt = 0.f;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int oi = o >> 2, oj = (o >> 1) & 1, ok = o & 1;
        const float w = (oi ? t.f[0] : 1.f - t.f[0]) * (oj ? t.f[1] : 1.f - t.f[1]) * (ok ? t.f[2] : 1.f - t.f[2]);
        const size_t cell = sm_loc(P, fr, base[0] + oi, base[1] + oj, base[2] + ok, 0, 0, 0);
        for (int c = 0; c < C; c++) out[c] += w * field[cell * C + c];
        wt += w;
        t.cell[o] = cell; t.w[o] = w;
    }
    for (int c = 0; c < C; c++) out[c] /= wt;          // wt == 1 up to rounding (the two weights of an axis sum to 1)
}
// adjoint of one trilerp: scatters g into gfield (atomics), returns d/dp
template <int C>
__device__ __forceinline__ void sm_trilerp_grad(const SmTri& t, const float* field, float* gfield, const float* g, float gp[3]) {
    gp[0] = gp[1] = gp[2] = 0.f;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int oi = o >> 2, oj = (o >> 1) & 1, ok = o & 1;
        float dot = 0.f;
        for (int c = 0; c < C; c++) { dot += field[t.cell[o] * C + c] * g[c]; atomicAdd(&gfield[t.cell[o] * C + c], t.w[o] * g[c]); }
        const float w0 = oi ? t.f[0] : 1.f - t.f[0], w1 = oj ? t.f[1] : 1.f - t.f[1], w2 = ok ? t.f[2] : 1.f - t.f[2];
        gp[0] += (oi ? 1.f : -1.f) * w1 * w2 * dot;
        gp[1] += w0 * (oj ? 1.f : -1.f) * w2 * dot;
        gp[2] += w0 * w1 * (ok ? 1.f : -1.f) * dot;
    }
}

struct SmAdv { float v1[3], v2[3], v3[3], vf[3], pf[3]; SmTri t1, t2, t3, tv, tq; float dist, factor, dir[3], dd[3]; };
// the per-cell forward pieces of advect_and_impulse (smoke_field.py:203-232) shared with the adjoint
template <int QD>
__device__ __forceinline__ void sm_advect_cell(const SmokeP& P, const EffP& a, int s, int f, int i, int j, int k, SmAdv& A, float* qf) {
    const unsigned char* fr = P.fr + (size_t)s * P.n3;
    const float* vfield = P.v + (size_t)s * P.n3 * 3;
    const float p0[3] = {(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
    float p1[3], p2[3];
    sm_trilerp<3>(P, fr, vfield, p0, A.v1, A.t1);                                                 // backtrace (RK3), 347-360
    for (int d = 0; d < 3; d++) p1[d] = p0[d] - 0.5f * P.dt * A.v1[d];
    sm_trilerp<3>(P, fr, vfield, p1, A.v2, A.t2);
    for (int d = 0; d < 3; d++) p2[d] = p0[d] - 0.75f * P.dt * A.v2[d];
    sm_trilerp<3>(P, fr, vfield, p2, A.v3, A.t3);
    for (int d = 0; d < 3; d++) A.pf[d] = p0[d] - P.dt * ((2.f / 9.f) * A.v1[d] + (1.f / 3.f) * A.v2[d] + (4.f / 9.f) * A.v3[d]);
    sm_trilerp<3>(P, fr, vfield, A.pf, A.vf, A.tv);
    sm_trilerp<QD>(P, fr, P.q + (size_t)s * P.n3 * QD, A.pf, qf, A.tq);
    const float n = (float)P.n;                                                                    // 1 / dx
    A.dd[0] = (float)i - a.pos[f * 3] * n; A.dd[1] = (float)j - a.pos[f * 3 + 1] * n; A.dd[2] = (float)k - a.pos[f * 3 + 2] * n;
    A.dist = sqrtf(A.dd[0] * A.dd[0] + A.dd[1] * A.dd[1] + A.dd[2] * A.dd[2] + FE_EPS);
    A.factor = expf(-A.dist / a.ra[f]);
    quat_rotate(a.inject_v, a.quat + f * 4, A.dir);
}
// advect_and_impulse, smoke_field.py:203-232.  Cells outside the slab are never free: v_tmp = 0 there and q is carried
// over by k_smoke_carry_q (one full-grid copy).
template <int QD>
__global__ __launch_bounds__(256) void k_smoke_advect(SmokeP P, const EffP* ap, int s, int f) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const size_t c = sm_idx(P, i, j, k);
    float* vt = P.vt + ((size_t)s * P.n3 + c) * 3;
    float* qn = P.q + ((size_t)(s + 1) * P.n3 + c) * QD;
    if (P.fr[(size_t)s * P.n3 + c]) {
        const EffP& a = *ap;
        SmAdv A; float qf[QD];
        sm_advect_cell<QD>(P, a, s, f, i, j, k, A, qf);
        const float m = a.sa[f] * A.factor * P.dt;
        for (int d = 0; d < 3; d++) vt[d] = A.vf[d] + A.dir[d] * m;
        for (int d = 0; d < QD; d++) qn[d] = (1.f - A.factor) * qf[d] + A.factor * P.low_T;
    } else {
        vt[0] = vt[1] = vt[2] = 0.f;
        for (int d = 0; d < QD; d++) qn[d] = P.q[((size_t)s * P.n3 + c) * QD + d];
    }
}
// adjoint of advect_and_impulse; red = {pos.grad[3], quat.grad[4], s.grad, r.grad} of the AirCon at f
template <int QD>
__global__ __launch_bounds__(256) void k_smoke_advect_grad(SmokeP P, const EffP* ap, int s, int f, float* red) {
    __shared__ float s_red[9];
    if (threadIdx.x < 9) s_red[threadIdx.x] = 0.f;
    __syncthreads();
    int i, j, k;
    const bool in = sm_cell(P, i, j, k);
    float part[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (in) {
        const size_t c = sm_idx(P, i, j, k);
        const float* gvt = P.gvt + ((size_t)s * P.n3 + c) * 3;
        const float* gqn = P.gq + ((size_t)(s + 1) * P.n3 + c) * QD;
        float* gqfield = P.gq + (size_t)s * P.n3 * QD;
        if (!P.fr[(size_t)s * P.n3 + c]) {
            for (int d = 0; d < QD; d++) atomicAdd(&gqfield[c * QD + d], gqn[d]);
        } else {
            const EffP& a = *ap;
            SmAdv A; float qf[QD];
            sm_advect_cell<QD>(P, a, s, f, i, j, k, A, qf);
            float* gvfield = P.gv + (size_t)s * P.n3 * 3;
            const float* vfield = P.v + (size_t)s * P.n3 * 3;
            const float* qfield = P.q + (size_t)s * P.n3 * QD;
            // v_tmp = v_f + dir s factor dt ; q' = (1 - factor) q_f + factor low_T
            float gqf[QD], gfac = 0.f, gdir[3];
            for (int d = 0; d < QD; d++) { gqf[d] = (1.f - A.factor) * gqn[d]; gfac += gqn[d] * (P.low_T - qf[d]); }
            for (int d = 0; d < 3; d++) {
                gfac += gvt[d] * A.dir[d] * a.sa[f] * P.dt;
                gdir[d] = gvt[d] * a.sa[f] * A.factor * P.dt;
                part[7] += gvt[d] * A.dir[d] * A.factor * P.dt;
            }
            const float r = a.ra[f];
            const float gdist = gfac * A.factor * (-1.f / r);
            part[8] = gfac * A.factor * A.dist / (r * r);
            for (int d = 0; d < 3; d++) part[d] = gdist * (A.dd[d] / A.dist) * (-(float)P.n);
            for (int qc = 0; qc < 4; qc++) {                                       // dir = R(quat) inject_v, one column per component
                Dual q[4], vin[3], out[3];
                for (int d = 0; d < 4; d++) q[d] = Dual(a.quat[f * 4 + d], d == qc ? 1.f : 0.f);
                for (int d = 0; d < 3; d++) vin[d] = Dual(a.inject_v[d]);
                t_quat_rotate(vin, q, out);
                part[3 + qc] = gdir[0] * out[0].d + gdir[1] * out[1].d + gdir[2] * out[2].d;
            }
            // the two interpolations at the back-traced point, then the RK3 chain in reverse
            const float gvf[3] = {gvt[0], gvt[1], gvt[2]};
            float gpf[3], t[3], gp2[3], gp1[3], gv1[3], gv2[3], gv3[3];
            sm_trilerp_grad<3>(A.tv, vfield, gvfield, gvf, gpf);
            sm_trilerp_grad<QD>(A.tq, qfield, gqfield, gqf, t);
            for (int d = 0; d < 3; d++) gpf[d] += t[d];
            for (int d = 0; d < 3; d++) { gv1[d] = -P.dt * (2.f / 9.f) * gpf[d]; gv2[d] = -P.dt * (1.f / 3.f) * gpf[d]; gv3[d] = -P.dt * (4.f / 9.f) * gpf[d]; }
            sm_trilerp_grad<3>(A.t3, vfield, gvfield, gv3, gp2);
            for (int d = 0; d < 3; d++) gv2[d] += -0.75f * P.dt * gp2[d];
            sm_trilerp_grad<3>(A.t2, vfield, gvfield, gv2, gp1);
            for (int d = 0; d < 3; d++) gv1[d] += -0.5f * P.dt * gp1[d];
            sm_trilerp_grad<3>(A.t1, vfield, gvfield, gv1, t);                   // p0 is a constant
        }
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
        float v = part[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(&s_red[r], v);
    }
    __syncthreads();
    if (threadIdx.x < 9 && s_red[threadIdx.x] != 0.f) atomicAdd(&red[threadIdx.x], s_red[threadIdx.x]);
}
__global__ void k_smoke_red_apply(EffP a, int f, float* red) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int d = 0; d < 3; d++) a.gpos[f * 3 + d] += red[d];
    for (int d = 0; d < 4; d++) a.gquat[f * 4 + d] += red[3 + d];
    a.gsa[f] += red[7]; a.gra[f] += red[8];
    for (int d = 0; d < 9; d++) red[d] = 0.f;
}
// q[s+1] = q[s] / q.grad[s] += q.grad[s+1] for the cells outside the slab (never free; `else` branches of 229-232)
template <bool GRAD>
__global__ __launch_bounds__(256) void k_smoke_carry_q(SmokeP P, int s) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.n3) return;
    const int j = (int)((t / P.n) % P.n);
    if (j > P.ly && j < P.hy) return;
    for (int d = 0; d < P.qd; d++) {
        if (!GRAD) P.q[((size_t)(s + 1) * P.n3 + t) * P.qd + d] = P.q[((size_t)s * P.n3 + t) * P.qd + d];
        else P.gq[((size_t)s * P.n3 + t) * P.qd + d] += P.gq[((size_t)(s + 1) * P.n3 + t) * P.qd + d];
    }
}
// divergence, smoke_field.py:234-258
__global__ __launch_bounds__(256) void k_smoke_div(SmokeP P, int s) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const unsigned char* fr = P.fr + (size_t)s * P.n3;
    const size_t c = sm_idx(P, i, j, k);
    if (!fr[c]) return;
    const float* vt = P.vt + (size_t)s * P.n3 * 3;
    float val[6];
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const int ax = b >> 1;
        if (!sm_isfree(P, fr, i, j, k, SM_NB[b][0], SM_NB[b][1], SM_NB[b][2])) val[b] = -vt[c * 3 + ax];
        else val[b] = vt[sm_loc(P, fr, i, j, k, SM_NB[b][0], SM_NB[b][1], SM_NB[b][2]) * 3 + ax];
    }
    P.dv[(size_t)s * P.n3 + c] = (val[1] - val[0] + val[3] - val[2] + val[5] - val[4]) * 0.5f;
}
// divergence adjoint as a gather: v_tmp.grad[t][ax] += 0.5 (div.grad[t - e] - div.grad[t + e]) over free neighbours that see
// t as free, + the mirrored-wall terms of t itself
__global__ __launch_bounds__(256) void k_smoke_div_grad(SmokeP P, int s) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const unsigned char* fr = P.fr + (size_t)s * P.n3;
    const size_t c = sm_idx(P, i, j, k);
    if (!fr[c]) return;                                         // only free cells are read by the divergence stencil
    const float* gdv = P.gdv + (size_t)s * P.n3;
    float* gvt = P.gvt + ((size_t)s * P.n3 + c) * 3;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        const int* em = SM_NB[2 * ax]; const int* ep = SM_NB[2 * ax + 1];
        float g = 0.f;
        if (sm_isfree(P, fr, i, j, k, em[0], em[1], em[2])) g += 0.5f * gdv[sm_idx(P, i + em[0], j + em[1], k + em[2])];   // t is its + neighbour
        else g += 0.5f * gdv[c];                                   // t's own - side is a wall: -val_l = +v_c
        if (sm_isfree(P, fr, i, j, k, ep[0], ep[1], ep[2])) g -= 0.5f * gdv[sm_idx(P, i + ep[0], j + ep[1], k + ep[2])];   // t is its - neighbour
        else g -= 0.5f * gdv[c];                                   // + side wall: val_r = -v_c
        gvt[ax] += g;
    }
}
// pressure_to_swap / pressure_from_swap (smoke_field.py:261-271) and their adjoints
template <int MODE>      // 0: pc = p[s] ; 1: p[s+1] = pc ; 2: gpc = gp[s+1] (pn/gpn cleared) ; 3: gp[s] += gpc
__global__ __launch_bounds__(256) void k_smoke_swap(SmokeP P, int s) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const size_t c = sm_idx(P, i, j, k);
    const bool fr = P.fr[(size_t)s * P.n3 + c] != 0;
    if (MODE == 0) { P.pc[c] = fr ? P.p[(size_t)s * P.n3 + c] : 0.f; P.pn[c] = 0.f; }
    if (MODE == 1) { if (fr) P.p[(size_t)(s + 1) * P.n3 + c] = P.pc[c]; }
    if (MODE == 2) { P.gpc[c] = fr ? P.gp[(size_t)(s + 1) * P.n3 + c] : 0.f; P.gpn[c] = 0.f; }
    if (MODE == 3) { if (fr) P.gp[(size_t)s * P.n3 + c] += P.gpc[c]; }
}
// pressure_jacobi, smoke_field.py:130-143 (frame pointers through the device table: graph-capturable for any s)
__global__ __launch_bounds__(256) void k_smoke_jacobi(SmokeP P, const SmokeFrameP* cur, const float* pf, float* npf) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const unsigned char* fr = cur->fr;
    const size_t c = sm_idx(P, i, j, k);
    if (!fr[c]) return;
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < 6; b++) sum += pf[sm_loc(P, fr, i, j, k, SM_NB[b][0], SM_NB[b][1], SM_NB[b][2])];
    npf[c] = (sum - cur->dv[c]) * (1.f / 6.f);
}
// its adjoint as a gather: gpf[t] = (1/6) sum over readers of t; div.grad[t] -= gnpf[t] / 6.  gpf is fully overwritten
// (`cur.grad.fill(0)` before each reverse sweep, smoke_field.py:121-124).
__global__ __launch_bounds__(256) void k_smoke_jacobi_grad(SmokeP P, const SmokeFrameP* cur, float* gpf, const float* gnpf) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const unsigned char* fr = cur->fr;
    const size_t c = sm_idx(P, i, j, k);
    if (!fr[c]) { gpf[c] = 0.f; return; }
    const float g = gnpf[c];
    cur->gdv[c] -= g * (1.f / 6.f);
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        // reader c' = t - dir sees t as its dir-neighbour when c' is free; t itself does when t + dir is a wall
        if (sm_isfree(P, fr, i, j, k, -SM_NB[b][0], -SM_NB[b][1], -SM_NB[b][2])) sum += gnpf[sm_idx(P, i - SM_NB[b][0], j - SM_NB[b][1], k - SM_NB[b][2])];
        if (!sm_isfree(P, fr, i, j, k, SM_NB[b][0], SM_NB[b][1], SM_NB[b][2])) sum += g;
    }
    gpf[c] = sum * (1.f / 6.f);
}
// subtract_gradient, smoke_field.py:273-288
__global__ __launch_bounds__(256) void k_smoke_subtract(SmokeP P, int s) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const unsigned char* fr = P.fr + (size_t)s * P.n3;
    const size_t c = sm_idx(P, i, j, k);
    const float* vt = P.vt + ((size_t)s * P.n3 + c) * 3;
    float* vn = P.v + ((size_t)(s + 1) * P.n3 + c) * 3;
    const float* pn = P.p + (size_t)(s + 1) * P.n3;
    float o[3] = {vt[0], vt[1], vt[2]};
    if (fr[c]) {
#pragma unroll
        for (int d = 0; d < 3; d++)
            o[d] -= 0.5f * (pn[sm_loc(P, fr, i, j, k, SM_NB[2 * d + 1][0], SM_NB[2 * d + 1][1], SM_NB[2 * d + 1][2])] -
                            pn[sm_loc(P, fr, i, j, k, SM_NB[2 * d][0], SM_NB[2 * d][1], SM_NB[2 * d][2])]);
    }
    vn[0] = o[0]; vn[1] = o[1]; vn[2] = o[2];
}
// adjoint: v_tmp.grad[s] += v.grad[s+1] everywhere in the slab; p.grad[s+1] gathered from its readers
__global__ __launch_bounds__(256) void k_smoke_subtract_grad(SmokeP P, int s) {
    int i, j, k;
    if (!sm_cell(P, i, j, k)) return;
    const unsigned char* fr = P.fr + (size_t)s * P.n3;
    const size_t c = sm_idx(P, i, j, k);
    const float* gvn_all = P.gv + (size_t)(s + 1) * P.n3 * 3;
    float* gvt = P.gvt + ((size_t)s * P.n3 + c) * 3;
    for (int d = 0; d < 3; d++) gvt[d] += gvn_all[c * 3 + d];
    if (!fr[c]) return;
    float g = 0.f;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const int d = b >> 1;
        const float coef = (b & 1) ? -0.5f : 0.5f;              // a reader's `b` neighbour enters v[d] with -0.5 (right) / +0.5 (left)
        if (sm_isfree(P, fr, i, j, k, -SM_NB[b][0], -SM_NB[b][1], -SM_NB[b][2]))
            g += coef * gvn_all[sm_idx(P, i - SM_NB[b][0], j - SM_NB[b][1], k - SM_NB[b][2]) * 3 + d];
        if (!sm_isfree(P, fr, i, j, k, SM_NB[b][0], SM_NB[b][1], SM_NB[b][2])) g += coef * gvn_all[c * 3 + d];
    }
    P.gp[(size_t)(s + 1) * P.n3 + c] += g;
}
// cells outside the slab: v[s+1] = v_tmp[s] = 0 (never free) and v_tmp.grad[s] += v.grad[s+1] is irrelevant (v_tmp is a
// constant 0 there); nothing to launch.

// ---------------------------------------------------------------------------------------------------------------- host
static dim3 smoke_grid(const SmokeP& P) {
    const long long cells = (long long)P.n * (P.hy - P.ly - 1 > 0 ? P.hy - P.ly - 1 : 0) * P.n;
    return dim3((unsigned)((cells + 255) / 256 > 0 ? (cells + 255) / 256 : 1));
}
static int smoke_find_aircon(FeEngine* h) {
    for (size_t i = 0; i < h->effs.size(); i++) if (h->effs[i].p.type == FE_EFF_AIRCON) return (int)i;
    return -1;
}
static int smoke_set_cur(FeEngine* h, int s) {
    SmokeP& P = h->smoke->P;
    SmokeFrameP c; c.fr = P.fr + (size_t)s * P.n3; c.dv = P.dv + (size_t)s * P.n3; c.gdv = P.gdv + (size_t)s * P.n3;
    HIPCK(h, hipMemcpyAsync(P.cur, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
    return 0;
}
// the `solver_iters` sweeps, captured once (pointers ping-pong with the sweep parity, the frame comes from P.cur)
static int smoke_build_graphs(FeEngine* h) {
    SmokeState& st = *h->smoke;
    SmokeP& P = st.P;
    if (P.iters == 0) return 0;
    for (int dir = 0; dir < 2; dir++) {
        hipGraph_t g = nullptr;
        HIPCK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        if (dir == 0) {
            float *a = P.pc, *b = P.pn;
            for (int it = 0; it < P.iters; it++) { hipLaunchKernelGGL(k_smoke_jacobi, smoke_grid(P), dim3(256), 0, h->stream, P, P.cur, a, b); std::swap(a, b); }
        } else {
            // reverse: before sweep `it` the roles are swapped back (p_swap.swap()), cur.grad is rebuilt from nxt.grad
            float *a = P.gpc, *b = P.gpn;
            for (int it = P.iters - 1; it >= 0; it--) { std::swap(a, b); hipLaunchKernelGGL(k_smoke_jacobi_grad, smoke_grid(P), dim3(256), 0, h->stream, P, P.cur, a, b); }
        }
        HIPCK(h, hipStreamEndCapture(h->stream, &g));
        hipGraphExec_t ex = nullptr;
        HIPCK(h, hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        (dir == 0 ? st.g_fwd : st.g_bwd) = ex;
    }
    return 0;
}
static void smoke_destroy(FeEngine* h) {
    if (!h->smoke) return;
    SmokeP& P = h->smoke->P;
    for (void* q : {(void*)P.v, (void*)P.vt, (void*)P.dv, (void*)P.p, (void*)P.q, (void*)P.gv, (void*)P.gvt, (void*)P.gdv, (void*)P.gp, (void*)P.gq,
                    (void*)P.fr, (void*)P.pc, (void*)P.pn, (void*)P.gpc, (void*)P.gpn, (void*)P.cur, (void*)h->smoke->red}) if (q) (void)hipFree(q);
    if (h->smoke->g_fwd) (void)hipGraphExecDestroy(h->smoke->g_fwd);
    if (h->smoke->g_bwd) (void)hipGraphExecDestroy(h->smoke->g_bwd);
    delete h->smoke;
    h->smoke = nullptr;
}
static int smoke_reset_grad_impl(FeEngine* h) {
    if (!h->smoke) return 0;
    SmokeP& P = h->smoke->P;
    const size_t F = (size_t)(P.S + 1) * P.n3;
    HIPCK(h, hipMemsetAsync(P.gv, 0, sizeof(float) * F * 3, h->stream)); HIPCK(h, hipMemsetAsync(P.gvt, 0, sizeof(float) * F * 3, h->stream));
    HIPCK(h, hipMemsetAsync(P.gdv, 0, sizeof(float) * F, h->stream)); HIPCK(h, hipMemsetAsync(P.gp, 0, sizeof(float) * F, h->stream));
    HIPCK(h, hipMemsetAsync(P.gq, 0, sizeof(float) * F * P.qd, h->stream));
    HIPCK(h, hipMemsetAsync(P.gpc, 0, sizeof(float) * P.n3, h->stream)); HIPCK(h, hipMemsetAsync(P.gpn, 0, sizeof(float) * P.n3, h->stream));
    return 0;
}

#define CHECK_SMOKE(h, s) do { if (!(h)->smoke) FAIL(h, "no smoke field"); if ((s) < 0 || (s) > (h)->smoke->P.S) FAIL(h, "smoke frame out of range"); } while (0)
#define SMOKE_QD_DISPATCH(QD, CALL1, CALL2, CALL3) do { if ((QD) == 1) { CALL1; } else if ((QD) == 2) { CALL2; } else { CALL3; } } while (0)

extern "C" {

__global__ void k_smoke_init_q(SmokeP P, float high_T) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.n3) return;
    const int j = (int)((t / P.n) % P.n);
    if (j > P.ly && j < P.hy) for (int d = 0; d < P.qd; d++) P.q[t * P.qd + d] = high_T;        // init_fields, smoke_field.py:86-93
}

int fe_smoke_create(FeEngine* h, const FeSmokeConfig* c) {
    FE_ENTRY(h);
    if (hipSetDevice(h->device) != hipSuccess) FAIL(h, "hipSetDevice failed");
    if (!c || c->struct_size != (int)sizeof(FeSmokeConfig)) FAIL(h, "FeSmokeConfig size mismatch");
    if (c->res < 4 || c->q_dim < 1 || c->q_dim > 3 || c->max_steps_local < 1 || c->solver_iters < 0) FAIL(h, "bad smoke configuration (q_dim 1..3)");
    smoke_destroy(h);
    SmokeState* st = new SmokeState();
    h->smoke = st;
    st->cfg = *c;
    SmokeP& P = st->P;
    std::memset(&P, 0, sizeof(P));
    P.n = c->res; P.S = c->max_steps_local; P.qd = c->q_dim; P.iters = c->solver_iters; P.ly = c->lower_y; P.hy = c->higher_y;
    P.dt = c->dt; P.low_T = c->low_T; P.n3 = (size_t)c->res * c->res * c->res;
    const size_t F = (size_t)(P.S + 1) * P.n3;
    if (dev_alloc(h, &P.v, F * 3) || dev_alloc(h, &P.vt, F * 3) || dev_alloc(h, &P.dv, F) || dev_alloc(h, &P.p, F) || dev_alloc(h, &P.q, F * P.qd) ||
        dev_alloc(h, &P.gv, F * 3) || dev_alloc(h, &P.gvt, F * 3) || dev_alloc(h, &P.gdv, F) || dev_alloc(h, &P.gp, F) || dev_alloc(h, &P.gq, F * P.qd) ||
        dev_alloc(h, &P.fr, F) || dev_alloc(h, &P.pc, P.n3) || dev_alloc(h, &P.pn, P.n3) || dev_alloc(h, &P.gpc, P.n3) || dev_alloc(h, &P.gpn, P.n3) ||
        dev_alloc(h, &P.cur, 1) || dev_alloc(h, &st->red, 9)) return 1;
    hipLaunchKernelGGL(k_smoke_init_q, dim3((unsigned)((P.n3 + 255) / 256)), dim3(256), 0, h->stream, P, c->high_T);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (smoke_build_graphs(h)) return 1;
    return check_async(h);
}

int fe_smoke_step(FeEngine* h, int s, int f) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s); CHECK_FRAME(h, f);
    SmokeP& P = h->smoke->P;
    if (s >= P.S) FAIL(h, "smoke step frame out of range");
    const int ia = smoke_find_aircon(h);
    if (ia < 0) FAIL(h, "smoke_step needs an AirCon effector (agent.aircon, smoke_field.py:213)");
    const EffP* ap = h->effs_dev + ia;
    const dim3 g = smoke_grid(P), full((unsigned)((P.n3 + 255) / 256));
    hipLaunchKernelGGL(k_smoke_free, g, dim3(256), 0, h->stream, P, s, statics_p(h));
    hipLaunchKernelGGL(k_smoke_carry_q<false>, full, dim3(256), 0, h->stream, P, s);
    SMOKE_QD_DISPATCH(P.qd, hipLaunchKernelGGL(k_smoke_advect<1>, g, dim3(256), 0, h->stream, P, ap, s, f),
                      hipLaunchKernelGGL(k_smoke_advect<2>, g, dim3(256), 0, h->stream, P, ap, s, f),
                      hipLaunchKernelGGL(k_smoke_advect<3>, g, dim3(256), 0, h->stream, P, ap, s, f));
    hipLaunchKernelGGL(k_smoke_div, g, dim3(256), 0, h->stream, P, s);
    hipLaunchKernelGGL(k_smoke_swap<0>, g, dim3(256), 0, h->stream, P, s);
    if (P.iters > 0) {
        if (smoke_set_cur(h, s)) return 1;
        HIPCK(h, hipGraphLaunch(h->smoke->g_fwd, h->stream));
        if (P.iters & 1) std::swap(P.pc, P.pn);               // the result of the last sweep is "cur" (p_swap.swap())
    }
    hipLaunchKernelGGL(k_smoke_swap<1>, g, dim3(256), 0, h->stream, P, s);
    if (P.iters & 1) std::swap(P.pc, P.pn);                   // keep the captured pointer roles
    hipLaunchKernelGGL(k_smoke_subtract, g, dim3(256), 0, h->stream, P, s);
    return check_async(h);
}

int fe_smoke_step_grad(FeEngine* h, int s, int f) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s); CHECK_FRAME(h, f);
    SmokeP& P = h->smoke->P;
    if (s >= P.S) FAIL(h, "smoke step frame out of range");
    const int ia = smoke_find_aircon(h);
    if (ia < 0) FAIL(h, "smoke_step_grad needs an AirCon effector");
    const EffP* ap = h->effs_dev + ia;
    const dim3 g = smoke_grid(P), full((unsigned)((P.n3 + 255) / 256));
    hipLaunchKernelGGL(k_smoke_free, g, dim3(256), 0, h->stream, P, s, statics_p(h));
    hipLaunchKernelGGL(k_smoke_subtract_grad, g, dim3(256), 0, h->stream, P, s);
    hipLaunchKernelGGL(k_smoke_swap<2>, g, dim3(256), 0, h->stream, P, s);
    if (P.iters > 0) {
        // reverse sweeps (smoke_field.py:121-124): seed in gpc; sweep 1 builds gpn = J^T gpc, sweep 2 gpc = J^T gpn, ...
        if (smoke_set_cur(h, s)) return 1;
        HIPCK(h, hipGraphLaunch(h->smoke->g_bwd, h->stream));
    }
    if (P.iters & 1) std::swap(P.gpc, P.gpn);               // an odd sweep count leaves the result in the other buffer
    hipLaunchKernelGGL(k_smoke_swap<3>, g, dim3(256), 0, h->stream, P, s);
    if (P.iters & 1) std::swap(P.gpc, P.gpn);
    hipLaunchKernelGGL(k_smoke_div_grad, g, dim3(256), 0, h->stream, P, s);
    hipLaunchKernelGGL(k_smoke_carry_q<true>, full, dim3(256), 0, h->stream, P, s);
    SMOKE_QD_DISPATCH(P.qd, hipLaunchKernelGGL(k_smoke_advect_grad<1>, g, dim3(256), 0, h->stream, P, ap, s, f, h->smoke->red),
                      hipLaunchKernelGGL(k_smoke_advect_grad<2>, g, dim3(256), 0, h->stream, P, ap, s, f, h->smoke->red),
                      hipLaunchKernelGGL(k_smoke_advect_grad<3>, g, dim3(256), 0, h->stream, P, ap, s, f, h->smoke->red));
    hipLaunchKernelGGL(k_smoke_red_apply, dim3(1), dim3(64), 0, h->stream, h->effs[ia].p, f, h->smoke->red);
    return check_async(h);
}

static int smoke_io(FeEngine* h, int s, float* base, int comps, void* host, bool to_host) {
    if (!host) return 0;
    SmokeP& P = h->smoke->P;
    float* dev = base + (size_t)s * P.n3 * comps;
    if (to_host) HIPCK(h, hipMemcpyAsync(host, dev, sizeof(float) * P.n3 * comps, hipMemcpyDeviceToHost, h->stream));
    else HIPCK(h, hipMemcpyAsync(dev, host, sizeof(float) * P.n3 * comps, hipMemcpyHostToDevice, h->stream));
    return 0;
}
int fe_smoke_get_frame(FeEngine* h, int s, fe_real* v, fe_real* v_tmp, fe_real* div, fe_real* p, fe_real* q) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s);
    SmokeP& P = h->smoke->P;
    if (smoke_io(h, s, P.v, 3, v, true) || smoke_io(h, s, P.vt, 3, v_tmp, true) || smoke_io(h, s, P.dv, 1, div, true) ||
        smoke_io(h, s, P.p, 1, p, true) || smoke_io(h, s, P.q, P.qd, q, true)) return 1;
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_smoke_set_frame(FeEngine* h, int s, const fe_real* v, const fe_real* v_tmp, const fe_real* div, const fe_real* p, const fe_real* q) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s);
    SmokeP& P = h->smoke->P;
    if (smoke_io(h, s, P.v, 3, (void*)v, false) || smoke_io(h, s, P.vt, 3, (void*)v_tmp, false) || smoke_io(h, s, P.dv, 1, (void*)div, false) ||
        smoke_io(h, s, P.p, 1, (void*)p, false) || smoke_io(h, s, P.q, P.qd, (void*)q, false)) return 1;
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_smoke_get_grad(FeEngine* h, int s, fe_real* gv, fe_real* gq) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s);
    SmokeP& P = h->smoke->P;
    if (smoke_io(h, s, P.gv, 3, gv, true) || smoke_io(h, s, P.gq, P.qd, gq, true)) return 1;
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
__global__ void k_smoke_axpy(float* dst, const float* src, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] += src[t];
}
int fe_smoke_add_grad(FeEngine* h, int s, const fe_real* gv, const fe_real* gq) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s);
    SmokeP& P = h->smoke->P;
    // staged through the scratch v_tmp / q adjoint-free buffers is not possible (all live): use a temporary allocation
    for (int which = 0; which < 2; which++) {
        const float* src = which == 0 ? gv : gq;
        if (!src) continue;
        const int comps = which == 0 ? 3 : P.qd;
        const size_t cnt = P.n3 * comps;
        float* tmp = nullptr;
        HIPCK(h, hipMalloc((void**)&tmp, sizeof(float) * cnt));
        HIPCK(h, hipMemcpyAsync(tmp, src, sizeof(float) * cnt, hipMemcpyHostToDevice, h->stream));
        float* dst = (which == 0 ? P.gv : P.gq) + (size_t)s * cnt;
        hipLaunchKernelGGL(k_smoke_axpy, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, dst, tmp, cnt);
        HIPCK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(tmp);
    }
    return check_async(h);
}
static int smoke_copy(FeEngine* h, int src, int dst, bool grad) {
    SmokeP& P = h->smoke->P;
    float* arr[5] = {grad ? P.gv : P.v, grad ? P.gvt : P.vt, grad ? P.gdv : P.dv, grad ? P.gp : P.p, grad ? P.gq : P.q};
    const int comps[5] = {3, 3, 1, 1, P.qd};
    for (int a = 0; a < 5; a++)
        HIPCK(h, hipMemcpyAsync(arr[a] + (size_t)dst * P.n3 * comps[a], arr[a] + (size_t)src * P.n3 * comps[a], sizeof(float) * P.n3 * comps[a],
                                hipMemcpyDeviceToDevice, h->stream));
    return 0;
}
int fe_smoke_copy_frame(FeEngine* h, int src, int dst) { FE_ENTRY(h); CHECK_SMOKE(h, src); CHECK_SMOKE(h, dst); return src == dst ? 0 : smoke_copy(h, src, dst, false); }
int fe_smoke_copy_grad(FeEngine* h, int src, int dst) { FE_ENTRY(h); CHECK_SMOKE(h, src); CHECK_SMOKE(h, dst); return src == dst ? 0 : smoke_copy(h, src, dst, true); }
int fe_smoke_reset_grad(FeEngine* h) { FE_ENTRY(h); if (!h->smoke) FAIL(h, "no smoke field"); return smoke_reset_grad_impl(h); }
int fe_smoke_reset_grad_till_frame(FeEngine* h, int s) {
    FE_ENTRY(h);
    CHECK_SMOKE(h, s);
    SmokeP& P = h->smoke->P;
    const size_t F = (size_t)s * P.n3;
    if (F == 0) return 0;
    HIPCK(h, hipMemsetAsync(P.gv, 0, sizeof(float) * F * 3, h->stream)); HIPCK(h, hipMemsetAsync(P.gvt, 0, sizeof(float) * F * 3, h->stream));
    HIPCK(h, hipMemsetAsync(P.gdv, 0, sizeof(float) * F, h->stream)); HIPCK(h, hipMemsetAsync(P.gp, 0, sizeof(float) * F, h->stream));
    HIPCK(h, hipMemsetAsync(P.gq, 0, sizeof(float) * F * P.qd, h->stream));
    return 0;
}

}  // extern "C"
