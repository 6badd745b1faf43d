// fe_engine.hip — MI355X (gfx950) FluidEngine MLS-MPM core: device state, HIP kernels, C ABI.
//
// Replaces the Taichi kernels of fluidlab/fluidengine/simulators/mpm_simulator.py (cited as
// mpm:NNN) behind include/fluidengine.h.  Written for CDNA4 only: wave64, float4-plane SoA
// particle frames (16 B/lane coalesced loads), a 4x4x4-blocked grid with an active-block list
// (no dense n^3 sweeps), hardware fp32 atomics.  See DESIGN.md for the data layout and the
// per-kernel roofline accounting.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fluidengine.h"
#include "fe_math.h"

static_assert(sizeof(fe_real) == 4, "the HIP engine is fp32 (macros.py:207-211)");

// =========================================================================================
// device-side layout
// =========================================================================================
// One particle frame = 24 floats + `used`, stored as float4 / float planes of Np (= N padded to
// 64) entries, grouped by the kernel that produces them:
//   A0 = (x0 x1 x2 v0)  A1 = (v1 v2 C00 C01)  A2 = (C02 C10 C11 C12)  a3 a4 a5 = C20 C21 C22   <- g2p
//   B0 = (F00 F01 F02 F10)  B1 = (F11 F12 F20 F21)  b2 = F22                                    <- p2g
// The adjoint frames use the same layout.
#define FR_WORDS 25          // 24 state words + used
#define GR_WORDS 24

// A plane is addressed as (uniform 64-bit frame base) + (32-bit byte offset: uniform plane offset + slot * element size), so a
// lane's address is one VGPR next to an SGPR pair (global_load ... v_off, s[base:base+1]) and cheap to form again.  With ten 64-bit
// pointers per frame view the compiler kept ~25 precomputed 64-bit lane addresses alive across the stencil loops (k_p2g and
// k_p2g_grad spilled them to scratch).  A frame is 100 B per slot, so 32 bits cover 40M particles (checked in fe_create).
template <typename T>
struct Plane {
    char* base; unsigned off;
    __host__ __device__ __forceinline__ T& operator[](int s) const { return *(T*)(base + (size_t)(off + (unsigned)s * (unsigned)sizeof(T))); }
    __host__ __device__ __forceinline__ T* ptr() const { return (T*)(base + (size_t)off); }
};
// Write-through (sc1) stores, option "write_through".  A kernel boundary costs its ~1.8 us plus (dirty bytes left in the per-XCD
// L2s) / ~6 TB/s (MI355X_MICROARCH.md, `boundary`): the substep kernels leave 11-19 MB each.  Bulk outputs that no later part of the
// same kernel reads -- the new particle state, the adjoint frame, the slabs -- can go through to memory while the kernel is still
// computing.  A buffer store through a descriptor over the plane's base, with the plane offset + slot as the 32-bit offset.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000); }
__device__ __forceinline__ void wt_store16(void* base, unsigned byte_off, float4 v) {
    const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, wt_rsrc(base), (int)byte_off, 0, 16);              // aux 16 = sc1
}
__device__ __forceinline__ void wt_store4(void* base, unsigned byte_off, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, wt_rsrc(base), (int)byte_off, 0, 16); }
struct FrameV {
    Plane<float4> A0, A1, A2; Plane<float> a3, a4, a5; Plane<float4> B0, B1; Plane<float> b2; Plane<int> used;
    int wt;                                  // stores into this frame go through to memory (kernels that produce a whole frame)
    // F stored compactly (round 5, option "compact_F"): of the nine words only b2 is valid.  An inviscid liquid's F is c I from its first substep on
    // (mpm:359: F = J^(1/3) I) and the adjoint of such an F is isotropic too (IdtC^T cof(F_tmp) = det(IdtC) c^2 I), of which the next substep
    // consumes the trace -- the SVD-free kernels read and write 4 bytes where the layout has 36: 64 bytes per particle less in k_p2g, up to 96 in
    // k_p2g_grad, 32 in the sort.  The values are the ones the full planes held (c itself; the trace, summed where it used to be summed on reading):
    // results are bit-identical.  1 = a state frame (b2 = c, F = c I); 2 = an adjoint frame (b2 = trace of the adjoint).  The host keeps the flags per
    // frame / ring slot, API calls that hand F out expand the planes first (k_expand_F).
    int iso;
};
__device__ __forceinline__ void pstore(const FrameV& fr, const Plane<float4>& p, int s, float4 v) {
    if (fr.wt) wt_store16(p.base, p.off + (unsigned)s * 16u, v); else p[s] = v;
}
__device__ __forceinline__ void pstore(const FrameV& fr, const Plane<float>& p, int s, float v) {
    if (fr.wt) wt_store4(p.base, p.off + (unsigned)s * 4u, __float_as_uint(v)); else p[s] = v;
}
__device__ __forceinline__ void pstore(const FrameV& fr, const Plane<int>& p, int s, int v) {
    if (fr.wt) wt_store4(p.base, p.off + (unsigned)s * 4u, (unsigned)v); else p[s] = v;
}
__host__ __device__ inline FrameV frame_view(float* base_, size_t Np_, int wt = 0, int iso = 0) {
    FrameV v;
    v.wt = wt; v.iso = iso;
    char* base = (char*)base_;
    const unsigned Np = (unsigned)Np_;
    v.A0 = {base, 0u}; v.A1 = {base, 16u * Np}; v.A2 = {base, 32u * Np};
    v.a3 = {base, 48u * Np}; v.a4 = {base, 52u * Np}; v.a5 = {base, 56u * Np};
    v.B0 = {base, 60u * Np}; v.B1 = {base, 76u * Np}; v.b2 = {base, 92u * Np};
    v.used = {base, 96u * Np};
    return v;
}

struct PState { float x[3], v[3]; m3 C, F; };
// (field by field: the adjoint k_pgg_g2pg hands from its p2g_grad part to its g2p_grad part has no F -- a whole-struct copy would move 36 undefined bytes
//  and kept the struct in scratch)
__device__ __forceinline__ void copy_xvC(PState& d, const PState& s) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        d.x[a] = s.x[a]; d.v[a] = s.v[a];
#pragma unroll
        for (int b = 0; b < 3; b++) d.C.a[a][b] = s.C.a[a][b];
    }
}

__device__ __forceinline__ void load_xvC(const FrameV& fr, int s, PState& p) {
    float4 a0 = fr.A0[s], a1 = fr.A1[s], a2 = fr.A2[s];
    p.x[0] = a0.x; p.x[1] = a0.y; p.x[2] = a0.z; p.v[0] = a0.w; p.v[1] = a1.x; p.v[2] = a1.y;
    p.C.a[0][0] = a1.z; p.C.a[0][1] = a1.w; p.C.a[0][2] = a2.x; p.C.a[1][0] = a2.y; p.C.a[1][1] = a2.z; p.C.a[1][2] = a2.w;
    p.C.a[2][0] = fr.a3[s]; p.C.a[2][1] = fr.a4[s]; p.C.a[2][2] = fr.a5[s];
}
__device__ __forceinline__ void load_F(const FrameV& fr, int s, m3& F) {
    if (fr.iso) {                            // (uniform) compact: c I of a state frame, or an adjoint's trace carried in the last entry
        const float c = fr.b2[s];
        const float d = fr.iso == 1 ? c : 0.f;
        F.a[0][0] = d; F.a[0][1] = 0.f; F.a[0][2] = 0.f; F.a[1][0] = 0.f; F.a[1][1] = d; F.a[1][2] = 0.f; F.a[2][0] = 0.f; F.a[2][1] = 0.f; F.a[2][2] = c;
        return;
    }
    float4 b0 = fr.B0[s], b1 = fr.B1[s];
    F.a[0][0] = b0.x; F.a[0][1] = b0.y; F.a[0][2] = b0.z; F.a[1][0] = b0.w;
    F.a[1][1] = b1.x; F.a[1][2] = b1.y; F.a[2][0] = b1.z; F.a[2][1] = b1.w; F.a[2][2] = fr.b2[s];
}
__device__ __forceinline__ void store_xvC(const FrameV& fr, int s, const float x[3], const float v[3], const m3& C) {
    const float4 a0 = make_float4(x[0], x[1], x[2], v[0]), a1 = make_float4(v[1], v[2], C.a[0][0], C.a[0][1]), a2 = make_float4(C.a[0][2], C.a[1][0], C.a[1][1], C.a[1][2]);
    if (fr.wt) {                                             // (one uniform branch around the whole group of stores)
        const unsigned o16 = (unsigned)s * 16u, o4 = (unsigned)s * 4u;
        wt_store16(fr.A0.base, fr.A0.off + o16, a0); wt_store16(fr.A0.base, fr.A1.off + o16, a1); wt_store16(fr.A0.base, fr.A2.off + o16, a2);
        wt_store4(fr.A0.base, fr.a3.off + o4, __float_as_uint(C.a[2][0])); wt_store4(fr.A0.base, fr.a4.off + o4, __float_as_uint(C.a[2][1]));
        wt_store4(fr.A0.base, fr.a5.off + o4, __float_as_uint(C.a[2][2]));
    } else {
        fr.A0[s] = a0; fr.A1[s] = a1; fr.A2[s] = a2;
        fr.a3[s] = C.a[2][0]; fr.a4[s] = C.a[2][1]; fr.a5[s] = C.a[2][2];
    }
}
__device__ __forceinline__ void store_F(const FrameV& fr, int s, const m3& F) {
    if (fr.iso) {                            // (uniform) compact: c of F = c I (the caller's business that it is), or the adjoint's trace
        const float c = fr.iso == 1 ? F.a[2][2] : (F.a[0][0] + F.a[1][1]) + F.a[2][2];
        if (fr.wt) wt_store4(fr.A0.base, fr.b2.off + (unsigned)s * 4u, __float_as_uint(c)); else fr.b2[s] = c;
        return;
    }
    const float4 b0 = make_float4(F.a[0][0], F.a[0][1], F.a[0][2], F.a[1][0]), b1 = make_float4(F.a[1][1], F.a[1][2], F.a[2][0], F.a[2][1]);
    if (fr.wt) {
        wt_store16(fr.A0.base, fr.B0.off + (unsigned)s * 16u, b0); wt_store16(fr.A0.base, fr.B1.off + (unsigned)s * 16u, b1);
        wt_store4(fr.A0.base, fr.b2.off + (unsigned)s * 4u, __float_as_uint(F.a[2][2]));
    } else { fr.B0[s] = b0; fr.B1[s] = b1; fr.b2[s] = F.a[2][2]; }
}
// F and the `used` flag together (p2g: one branch for the whole group)
__device__ __forceinline__ void store_F_used(const FrameV& fr, int s, const m3& F, int used) {
    if (fr.iso) {
        if (fr.wt) { wt_store4(fr.A0.base, fr.b2.off + (unsigned)s * 4u, __float_as_uint(F.a[2][2])); wt_store4(fr.A0.base, fr.used.off + (unsigned)s * 4u, (unsigned)used); }
        else { fr.b2[s] = F.a[2][2]; fr.used[s] = used; }
        return;
    }
    const float4 b0 = make_float4(F.a[0][0], F.a[0][1], F.a[0][2], F.a[1][0]), b1 = make_float4(F.a[1][1], F.a[1][2], F.a[2][0], F.a[2][1]);
    if (fr.wt) {
        wt_store16(fr.A0.base, fr.B0.off + (unsigned)s * 16u, b0); wt_store16(fr.A0.base, fr.B1.off + (unsigned)s * 16u, b1);
        wt_store4(fr.A0.base, fr.b2.off + (unsigned)s * 4u, __float_as_uint(F.a[2][2])); wt_store4(fr.A0.base, fr.used.off + (unsigned)s * 4u, (unsigned)used);
    } else { fr.B0[s] = b0; fr.B1[s] = b1; fr.b2[s] = F.a[2][2]; fr.used[s] = used; }
}

// grid: float4 per node, nodes grouped in 4x4x4 blocks of 64 contiguous float4 (1 KiB)
__device__ __forceinline__ int cell_addr(int i, int j, int k, int nb) {
    return (((((i >> 2) * nb) + (j >> 2)) * nb + (k >> 2)) << 6) | ((i & 3) << 4) | ((j & 3) << 2) | (k & 3);
}

// Profiling builds only (-DFE_TIMELINE -> libfluidengine_hip_tl.so, scripts/timeline.py): thread 0 of every workgroup stamps
// s_memrealtime (100 MHz) at the phase boundaries of the six substep kernels.  Not compiled into the product library.
#ifdef FE_TIMELINE
#define TL_WGS 2048
#define TL(S_, k) do { if (threadIdx.x == 0 && blockIdx.x < TL_WGS) { (S_).tl[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        if ((k) == 0) (S_).tl[TL_WGS * 8 + blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#else
#define TL(S_, k) do { } while (0)
#endif

struct SimP {
#ifdef FE_TIMELINE
    unsigned long long* tl;
#endif
    int N, Np, n, nb;
    int ncell;                               // nb^3 * 64: plane stride of the SoA accumulator grids
    int xcd;                                 // option "xcd_map": consecutive work items on the same XCD (shared L2)
    int wsort;                               // option "wave_sort": the scatter kernels regroup their lanes by stencil base before the scan (wave_sort_dest)
    int lsplit;                              // option "lane_split": waves with at most 21 / 7 particles give every particle 3 / 9 lanes (lane_split); bit 0 k_p2g, 1 k_g2p_grad2, 2 k_p2g_grad
    int wt;                                  // option "write_through": bulk outputs as sc1 stores (see wt_store16); bit 0 p2g, 1 g2p, 2 g2p_grad, 3 p2g_grad
    float dx, inv_dx, dt, stress_scale;     // stress_scale = -dt * p_vol * 4 * inv_dx^2 (mpm:343)
    int uni; float uinfo[4];                 // every particle has the same material record (mu, lam, mass, class | material): it travels here instead of 16 bytes per particle and kernel
    float g[3];
    BoundaryP bnd;
    unsigned fg_base;                        // option "fuse_grid": what the `started` counters of the fused grid pass read before this launch (FgDev)
    int fg_nowait;                           // ... its waves never wait (tests: every entry takes the skipped road)
};

struct EffP {
    int type, action_dim;
    float scale_v[8], scale_p[8];
    BoundaryP bnd;
    int flux; float radius; float inject_v[3], inject_p[3];
    int locally_random, randomize_inject_v, random_length;
    // device arrays
    float *pos, *quat, *v, *w, *gpos, *gquat, *gv, *gw;      // [L+1] x {3,4,3,3}
    float *sa, *ra, *gsa, *gra;                              // AirCon strength s[f], radius r[f] + grads (aircon.py:20-21)
    float *abuf, *gabuf, *abuf_p, *gabuf_p;                  // [max_action_steps, adim], [adim]
    float* random_vector;                                    // [random_length, flux, 3]
    int has_mesh; SdfP mesh;                                 // Rigid.setup_mesh (rigid.py:19-24): a moving SDF collider
};
struct AgentP { int n; int inj; const EffP* e; float collide_min_y; const BoundaryP* collector; int collector_mat;
                unsigned char* hit; };   // hit[f * Np + slot]: the particle met a collider in g2p of frame f (steers k_collide_grad)       // effector parameter blocks live in device memory (1 KiB as kernarg spilled SGPRs)
struct InjectP { int on, act_id, row, flux; };                     // per-substep injection parameters (host-known)

struct PInfo { float mu, lam, mass; int cls, mat; };
__device__ __forceinline__ PInfo unpack_info(const float4 t) {
    PInfo r; r.mu = t.x; r.lam = t.y; r.mass = t.z;
    int bits = __float_as_int(t.w); r.cls = bits & 0xffff; r.mat = (bits >> 16) & 0xffff;
    return r;
}
__device__ __forceinline__ PInfo load_info(const float4* info, int i) { return unpack_info(info[i]); }
// the record of slot i of an order -- or the scene's one record (uniform branch): a single-material scene reads 16 bytes per particle less
// in k_p2g and k_p2g_grad, and its sorts move 16 bytes per particle less
__device__ __forceinline__ PInfo load_info(const SimP& S, const float4* info, int i) {
    if (S.uni) return unpack_info(make_float4(S.uinfo[0], S.uinfo[1], S.uinfo[2], S.uinfo[3]));
    return unpack_info(info[i]);
}

// A block is on the active list of the order a substep runs in exactly when that order's blk_slot holds an entry for it (>= 0):
// the orders keep their own tables, so forward and backward substeps of any frame agree without re-flagging anything (round 2
// kept the value 2 in blk_flag on exactly one order's list and switched it with two launches whenever backward crossed a sort).
// blk_flag only says whether a block OUTSIDE that list is on the substep's dynamic list already (1).
__device__ __forceinline__ bool mark_dynamic(int b, int* blk_flag, int* blk_list, int* blk_count) {      // true: this call put the block on the list
    const int fl = blk_flag[b];
    if (fl != 1) {
        if (atomicCAS(&blk_flag[b], fl, 1) == fl) { int i = atomicAdd(blk_count, 1); __hip_atomic_store(blk_list + i, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    }
    return false;
}

// -----------------------------------------------------------------------------------------
// Particle order and work decomposition.
// Every K substeps the particles of a frame are counting-sorted by the 4x4x4 grid block of
// their stencil base (k_sort_*).  A sorted order ("table") carries a work list: one item per
// occupied block (split at ITEM_MAX particles) = a contiguous slot range.  One workgroup
// processes one item with the block's stencil footprint staged in LDS as an 8^3-node tile
// (block + 2 halo nodes + 1 node of drift margin on each side), so the 27-node APIC scatter /
// gather runs on LDS (ds_add_f64 / ds_read) instead of global atomics / L2 gathers.  Slots behind the
// last item ("tail": unused pool particles, particles injected since the last sort) and
// particles that drifted out of their tile take the global path, which is always correct.
// -----------------------------------------------------------------------------------------
#define TILE_T 8
#define TILE_N 512
// A tile is handed over as a SLAB of its inner 6^3 nodes (tile indices 1..6 per axis = the block and its two halo nodes per side:
// everything an item deposits right after a sort).  The outer shell -- the node of drift margin on each side -- is rarely
// touched; what does land there goes to the slow-path accumulator with global atomics and marks its block dirty.  Round 2 stored
// all 8^3 nodes: 8 KiB per item where 3.4 KiB carry data, written by the scatter kernels (left dirty in L2 at the kernel boundary)
// and gathered by the grid kernels from 16 candidate slabs per node, of which a fresh order fills 3.4.
#define SLAB_T 6
#define SLAB_N 216
#define ITEM_MAX_CAP 128      // upper bound (and default) of the runtime `item_max` option: an item is one pass of one half workgroup
#define WG 256
#define HALF 128
// Work items and workgroups.  An item holds <= 128 particles of one block; a workgroup (4 waves) takes TWO consecutive items, one
// per pair of waves.  When both belong to the same block (the usual case in a dense region: a block of ~400 particles is 4 items,
// and blocks with several items are padded to an even number) the two halves share one LDS tile and hand over one slab -- exactly
// what a 256-particle item did in round 1.  When they belong to different blocks (blocks with few particles: a splash, a thin
// layer, an injected jet) each half has its own tile.  Round 1 gave every block its own workgroup: a block that has spread into
// 1,600 sparsely filled blocks needed two rounds of the chip's 1,024 resident workgroups per kernel (scripts/timeline.py).
// The LDS tiles (SoA planes of TILE_N floats, two tiles per workgroup).  File scope, so every access is a known-LDS ds_*
// instruction (a `float*` parameter that may also be null degrades to flat_*).
__shared__ float  s_tile[2 * 4 * TILE_N];   // gathered node values (d v_in, d m): k_p2g_grad (4 planes)
__shared__ float  s_gtile[2 * 3 * TILE_N];  // gathered v_out: k_g2p (3 planes)
// Scatter accumulators are fp64: measured on MI355X ds_add_f32 sustains ~0.2 T lane-ops/s chip-wide, ds_add_f64
// ~1.6 T and ds_add_u64 ~2.8 T in this access pattern (profiles/r01_ubench_lds_types.txt); fp64 also makes the
// in-tile sum insensitive to the order of the atomics.
__shared__ double s_acc[2 * 4 * TILE_N];    // k_p2g
__shared__ float  s_tile3[2 * 3 * TILE_N];  // k_g2p_grad2: v_out ...
__shared__ double s_acc3[2 * 3 * TILE_N];   // ... and d v_out (3 planes each: 36 KB per workgroup, four workgroups per CU)
// k_p2g_grad's per-thread stash columns (used_particle_p2g_grad)
#define STASH_GENERAL 36
#define STASH_LIQUID 18
__shared__ float s_stash_g[STASH_GENERAL * WG];
__shared__ float s_stash_l[STASH_LIQUID * WG];
// k_pgg_g2pg (substep f's p2g_grad, then substep f - 1's g2p_grad in one launch) works through both kernels' LDS one after the other: ONE arena of 36 KB that
// the four arrays above are views of (AR = true), so that four workgroups per CU stay resident -- side by side they would be 70 KB.
//   p2g_grad part, pair units: tiles [0, 8 TILE_N) floats, stash behind them (18 x 256 floats);  quad units: four 4-plane tiles, two of them where the stash is
//   g2p_grad part, pair units: v_out tiles [0, 6 TILE_N) floats, the fp64 accumulators behind them; quad units: the wave's words in the accumulators' bytes
// AR: 0 = the kernels' own arrays, 1 = the arena of the SVD-free k_pgg_g2pg, 2 = the SVD build's (tiles + its 36 KB stash: 52 KB, three workgroups per CU as k_p2g_grad<true>)
__shared__ __attribute__((aligned(16))) float s_bw[18 * TILE_N];
__shared__ __attribute__((aligned(16))) float s_bwg[26 * TILE_N];
template <int AR> __device__ __forceinline__ float* lds_tile4() { if constexpr (AR == 1) return s_bw; else if constexpr (AR == 2) return s_bwg; else return s_tile; }
template <int AR> __device__ __forceinline__ float* lds_stash_l() { if constexpr (AR == 1) return s_bw + 8 * TILE_N; else return s_stash_l; }
template <int AR> __device__ __forceinline__ float* lds_stash_g() { if constexpr (AR == 2) return s_bwg + 8 * TILE_N; else return s_stash_g; }
template <int AR> __device__ __forceinline__ float* lds_tile3() { if constexpr (AR == 1) return s_bw; else if constexpr (AR == 2) return s_bwg; else return s_tile3; }
template <int AR> __device__ __forceinline__ double* lds_acc3() { if constexpr (AR == 1) return (double*)(s_bw + 6 * TILE_N); else if constexpr (AR == 2) return (double*)(s_bwg + 6 * TILE_N); else return s_acc3; }
template <bool GENERAL, int AR = 0> struct Stash { static __device__ __forceinline__ float* at() { if constexpr (GENERAL) return lds_stash_g<AR>(); else return lds_stash_l<AR>(); } };
// Effector pose adjoints of the workgroup's particles in contact (agent.collide's adjoint): summed here first -- every
// contact particle adds to the same 14 numbers per effector, and same-address global atomics serialise.
#define FE_MAX_EFF 4
__shared__ float s_pose[FE_MAX_EFF * 14];
// Keeps the fully unrolled 27-node stencil loops from being software-pipelined into one giant basic
// block (which drove k_p2g_grad to 512 registers + scratch): nothing is scheduled across the fence.
#define NODE_FENCE() __builtin_amdgcn_sched_barrier(0)
// The 27-node stencil loops are rolled as 9 (i,j) iterations x 3 unrolled k nodes: a fully unrolled
// 27-node body gets software-pipelined into one basic block that needs >512 registers.  Per-iteration
// weights come from register selects (a runtime-indexed array would live in scratch).
__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }
// Workgroups are dealt round-robin to the 8 XCDs (workgroup id % 8), each with its own L2.  Work items are sorted by
// block, and neighbouring blocks share tile halos and slabs, so workgroup `wg` takes item (wg % 8) * per + wg / 8:
// every XCD gets one contiguous eighth of the list.  A bijection on [0, 8 * per); ids >= n_work are skipped.
// `mode` (option "xcd_map"): 0 = none (unit = slot: consecutive units on consecutive XCDs), 1 = contiguous eighths, k >= 2 = blocked-
// cyclic: runs of k consecutive units per XCD, the runs dealt round-robin.  Contiguous eighths keep the most neighbours together but
// hand whole regions -- the dense core of a splash, all the two-item pairs at the head of the list -- to single XCDs.
__device__ __forceinline__ int xcd_item(int wg, int per, int mode) {
    if (mode == 0) return wg;
    const int x = wg & 7, j = wg >> 3;
    if (mode == 1) return x * per + j;
    return ((j / mode) * 8 + x) * mode + j % mode;
}
#define STW(st, i, d) sel3((i), (st).w[0][d], (st).w[1][d], (st).w[2][d])

// -----------------------------------------------------------------------------------------
// Wavefront aggregation of scatter contributions.  After the cell-level sort, lanes holding particles of one
// cell are adjacent and write the same 27 nodes.  A segmented inclusive scan over each 16-lane DPP row (row_shr
// 1/2/4/8: pure VALU, no LDS traffic) sums each run of equal keys; only the last lane of a run issues the LDS
// atomic.  Measured motivation (profiles/r01e): ~30 us of a 44 us P2G launch were ds_add_f64.
// Correct for ANY lane order: only *adjacent* equal keys are merged.
// Runs are cut at the row boundaries: carrying them across with row_bcast:15 / :31 costs two more DPP steps per
// value and saves at most 3 atomics per wave and node -- measured (round 2, A/B on the same box): evolving block
// 5,786 -> 5,824 pairs/s, 1M-particle block 3,167 -> 3,246, p2g 74.2 -> 71.6 us, g2p_grad 70.4 -> 67.1 us.
// (Cutting at 8-lane groups as well was measured slower in round 1: p2g 21.0 -> 25.3 us.)
// -----------------------------------------------------------------------------------------
struct SegScan { float f1, f2, f4, f8; bool tail; };

// must be executed by all 64 lanes of the wave
__device__ __forceinline__ SegScan seg_setup(int key) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // (opaque: what is derived from it below must not become a loop invariant held -- spilled -- across the kernel)
    const int prev = __shfl_up(key, 1, 64);
    const bool is_head = (lane & 15) == 0 || key != prev;
    const unsigned long long mask = __ballot(is_head);
    const unsigned long long lower = mask & ((2ull << lane) - 1ull);       // run heads at or before this lane
    const int head = 63 - __clzll((long long)lower);
    const int dist = lane - head;
    SegScan sc;
    sc.f1 = dist >= 1 ? 1.f : 0.f; sc.f2 = dist >= 2 ? 1.f : 0.f; sc.f4 = dist >= 4 ? 1.f : 0.f; sc.f8 = dist >= 8 ? 1.f : 0.f;
    sc.tail = lane == 63 || ((mask >> (lane + 1)) & 1ull);                 // last lane of its run
    return sc;
}
// Four independent values at once, one v_fmac_f32_dpp per step and value (the compiler's own lowering is
// v_mov_b32_dpp + v_fma_f32).  A DPP read of a VGPR written by the previous VALU needs 2 wait states, which
// inline asm has to provide itself.
// All four steps of the four chains in ONE asm statement: between separate statements the compiler pads a wait state of its own
// (three s_nop per node and scan, 81 issue slots per particle).  Inside, a value written in one step is read by DPP four
// instructions later (the three other chains sit in between): no wait states needed.
#define SEG_ROW(r, ctrl, flag) "v_fmac_f32_dpp " r ", " r ", " flag " " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
__device__ __forceinline__ void seg_scan4(const SegScan& sc, float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t"                           // the inputs were just produced by VALU
                 SEG_ROW("%0", "row_shr:1", "%4") SEG_ROW("%1", "row_shr:1", "%4") SEG_ROW("%2", "row_shr:1", "%4") SEG_ROW("%3", "row_shr:1", "%4")
                 SEG_ROW("%0", "row_shr:2", "%5") SEG_ROW("%1", "row_shr:2", "%5") SEG_ROW("%2", "row_shr:2", "%5") SEG_ROW("%3", "row_shr:2", "%5")
                 SEG_ROW("%0", "row_shr:4", "%6") SEG_ROW("%1", "row_shr:4", "%6") SEG_ROW("%2", "row_shr:4", "%6") SEG_ROW("%3", "row_shr:4", "%6")
                 SEG_ROW("%0", "row_shr:8", "%7") SEG_ROW("%1", "row_shr:8", "%7") SEG_ROW("%2", "row_shr:8", "%7") SEG_ROW("%3", "row_shr:8", "%7")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(sc.f1), "v"(sc.f2), "v"(sc.f4), "v"(sc.f8));
}
// three values (the adjoint scatter of g2p has no mass component): two other VALUs between a write and its DPP read, so one
// wait state per step
__device__ __forceinline__ void seg_scan3(const SegScan& sc, float& a, float& b, float& c) {
    asm volatile("s_nop 1\n\t"
                 SEG_ROW("%0", "row_shr:1", "%3") SEG_ROW("%1", "row_shr:1", "%3") SEG_ROW("%2", "row_shr:1", "%3") "s_nop 0\n\t"
                 SEG_ROW("%0", "row_shr:2", "%4") SEG_ROW("%1", "row_shr:2", "%4") SEG_ROW("%2", "row_shr:2", "%4") "s_nop 0\n\t"
                 SEG_ROW("%0", "row_shr:4", "%5") SEG_ROW("%1", "row_shr:4", "%5") SEG_ROW("%2", "row_shr:4", "%5") "s_nop 0\n\t"
                 SEG_ROW("%0", "row_shr:8", "%6") SEG_ROW("%1", "row_shr:8", "%6") SEG_ROW("%2", "row_shr:8", "%6")
                 : "+v"(a), "+v"(b), "+v"(c) : "v"(sc.f1), "v"(sc.f2), "v"(sc.f4), "v"(sc.f8));
}

// -----------------------------------------------------------------------------------------
// Lanes regrouped by stencil base inside the wave (option "wave_sort", round 4).
// The scan merges ADJACENT lanes with equal keys, and adjacent they are right after a sort.  Then the particles move: a few substeps
// later a cell's particles sit in the wave as  a a b a b b a b  -- host-side count on the benchmark block (scripts/run_stats.py, runs
// = LDS atomics per node and value): 40.6k runs on a fresh order, 111k five substeps later, 129k after nine, where the block hits the
// floor; 38k / 85k / 108k while it falls.  ds_add_f64 goes at ~2.6 lane-operations per clock and CU, and with every second lane a
// run of its own the scatter kernels are bound by it: k_p2g 32.5 us with a sort every 10 substeps, 22.3 us with one every substep
// (k_g2p_grad 33.5 / 23.4; on the falling block 23.4 / 19.7) -- but a sort is 60 us.  Regrouping the 64 lanes of the wave by their
// CURRENT key before the scan brings the runs back to 60k / 64k (48k / 51k falling) for ~150 instructions per wave: the rank of
// every lane among the wave's (key, lane) pairs (wave_sort_dest), then one ds_permute per value the scatter loop consumes.
// A wave whose keys are still in order (a fresh sort, an item of one cell) skips all of it.
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ bool wave_needs_sort(int key) {       // all 64 lanes; wave-uniform result
    const int prev = __shfl_up(key, 1, 64);
    return __any((threadIdx.x & 63) != 0 && prev > key);
}
// byte address (for ds_permute) of the lane this lane's values go to: its rank among the wave's (key, lane) pairs.  0 <= key < 1024.
// Bit-serial, most significant bit first: every lane keeps the set of lanes whose keys equal its own so far (`eq`, a 64-bit mask) and
// counts those that fell below it: a ballot per bit, two and-nots, two population counts -- ~12 instructions for each of the 10 bits,
// where comparing against all 64 lanes one cross-lane read at a time took ~220 (and an SGPR hazard stall after each read).
#ifndef WSORT_BITS
#define WSORT_BITS 1
#endif
__device__ __forceinline__ int wave_sort_dest(int key) {
#if WSORT_BITS
    unsigned eq_lo = ~0u, eq_hi = ~0u;                         // (two halves: the 64-bit form kept a lane mask alive -- in scratch -- across the kernel)
    int lt = 0;
#pragma unroll
    for (int b = 9; b >= 0; b--) {
        const bool mine = (key >> b) & 1;
        const unsigned long long m = __ballot(mine);
        const unsigned z_lo = eq_lo & ~(unsigned)m, z_hi = eq_hi & ~(unsigned)(m >> 32);     // equal so far, with a 0 in this bit
        lt += mine ? __popc(z_lo) + __popc(z_hi) : 0;          // ... smaller than a key with a 1 there
        eq_lo = mine ? eq_lo ^ z_lo : z_lo; eq_hi = mine ? eq_hi ^ z_hi : z_hi;
    }
    return (lt + (int)__builtin_amdgcn_mbcnt_hi(eq_hi, __builtin_amdgcn_mbcnt_lo(eq_lo, 0u))) << 2;      // equal keys keep their lane order
#else
    const unsigned kl = ((unsigned)key << 6) | (threadIdx.x & 63);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 64; j++) rank += ((unsigned)__builtin_amdgcn_readlane((int)kl, j) < kl) ? 1 : 0;
    return rank << 2;
#endif
}
__device__ __forceinline__ void wave_send(int dest, float& v) { v = __int_as_float(__builtin_amdgcn_ds_permute(dest, __float_as_int(v))); }
__device__ __forceinline__ void wave_send(int dest, int& v) { v = __builtin_amdgcn_ds_permute(dest, v); }

struct TableP {
    const int*  pid_of_slot;   // [Np]
    const float4* info;        // [Np] material record of the particle in each slot (pinfo in slot order: a coalesced load, not pinfo[pid])
    // (the sort's item list -- (block, start, count <= 128), sorted by block --, its pair list for blocks with several items and
    //  its list of single-item blocks stay on the host side of the table: the kernels read what k_build_units made of them)
    const int*  meta;          // meta[0] = n_items, [1] = tail_start, [2] = n_active, [3] = n pairs, [4] = n singles, [5] = n unit slots
    const struct UnitRec* units;  // what workgroup w of the scatter kernels works on: ready-made descriptors, XCD order (see Unit); meta[5] slots
    const struct UnitRec* units_p;// the same work as pair units only, for the gather kernels (build_units_dev); meta[9] slots
    int units_cap;
    const int2* nbr;           // [n_active * 27] (first item, item count) of the 27 neighbours of each active-list entry's block
    const int*  active;        // blocks within one block of an occupied block: every block a tile can reach
    const int*  blk_slot;      // [nblk] index of a block in `active`, or -1
    unsigned long long* arrive;// [nblk] arrival word of every active-list entry (fused grid pass, FG_* below), then int expected[nblk]
};

struct TileO { int ox, oy, oz; };
// A work item names its block by its three coordinates, ten bits each (k_sort_blk_final packs them once per sort): the particle kernels used
// to take the block NUMBER apart again in every unit -- three divisions by the run-time nb for the tile's origin and three more for the
// 27 neighbour entries, ~60 VALU and ~100 SALU instructions per unit and kernel (scripts/valu_profile.py; a uniform integer division is a
// v_rcp sequence on this part).
#define BLK_PACK(bi, bj, bk) (((bi) << 20) | ((bj) << 10) | (bk))
#define BLK_I(pk) ((pk) >> 20)
#define BLK_J(pk) (((pk) >> 10) & 1023)
#define BLK_K(pk) ((pk) & 1023)
__device__ __forceinline__ TileO tile_origin(int pk) {
    TileO t; t.ox = BLK_I(pk) * 4 - 1; t.oy = BLK_J(pk) * 4 - 1; t.oz = BLK_K(pk) * 4 - 1; return t;
}
// local index of the stencil base inside the tile, or -1 when the 3^3 stencil does not fit
__device__ __forceinline__ int tile_base(const TileO& t, const Stencil& st) {
    int lx = st.base[0] - t.ox, ly = st.base[1] - t.oy, lz = st.base[2] - t.oz;
    bool in = (unsigned)lx <= TILE_T - 3 && (unsigned)ly <= TILE_T - 3 && (unsigned)lz <= TILE_T - 3;
    return in ? (lx * TILE_T + ly) * TILE_T + lz : -1;
}
__device__ __forceinline__ bool tile_node(const TileO& t, int l, int n, int& i, int& j, int& k) {
    i = t.ox + (l >> 6); j = t.oy + ((l >> 3) & 7); k = t.oz + (l & 7);
    return (unsigned)i < (unsigned)n && (unsigned)j < (unsigned)n && (unsigned)k < (unsigned)n;
}

// What one half of a workgroup (a pair unit) or one wave of it (a quad unit) works on.  Uniform per wave.
struct PairCtx {
    int4 it;         // (block coordinates (BLK_PACK), first slot, count <= 128, 0) of this half's / wave's item; count 0 when it idles
    int  ti;         // LDS tile: 0, or 1 when the two items of a pair belong to different blocks; the wave's number in a quad
    int  slab;       // slab the tile is handed over in (the item's index; a shared tile goes to the first item's)
    int  nth, t0;    // the threads that load / zero / store this tile: all 256 from t0 = tid when shared, this half's 128, a quad's wave
    int  i;          // this thread's particle within the item
    bool live;       // this half's / wave's threads take part in tile loads and stores
    bool quad;
};
// The work of one workgroup of the particle kernels, as the sort leaves it (build_units_dev): item descriptors ready to use.
// The kernels used to walk meta -> pairs[w] / singles[q] -> items[i] before they could ask for their particles: three dependent
// round trips ahead of the first useful load in kernels that are one round of such chains.  Now workgroup w reads units[w] together
// with meta.  The list is stored in XCD order (slot w holds work unit xcd_item(w): every XCD a contiguous eighth of the items).
//   PAIR: a = (block coordinates (BLK_PACK), first slot, count, item index | same << 30), b likewise with b.w = -1 when there is no second item;
//         same: both items belong to one block and share tile and slab
//   QUAD (a.w has QUAD_BIT): a, b, c, d = four items of at most QUAD_MAX particles, each the only item of its block -- one WAVE per
//         item, four tiles per workgroup.  Where the water has come apart most items are of that kind, and a pair unit keeps two
//         waves and a 16 KB fp64 tile waiting on each of them: the particle kernels there are rounds of resident workgroups (measured,
//         profiles/r03_ab_lds_pad_occupancy.txt: splash time = a + b / workgroups per CU with b / 4 = 13-21 us per kernel).
//   a.z == -1: a tail unit (slots from a.y);  a.z == -2: nothing
#define QUAD_BIT (1 << 29)
#define QUAD_MAX 64
#ifndef QH
#define QH 2
#endif
struct UnitRec { int4 a, b, c, d; };         // in memory
struct Unit { int4 a, b, q; };               // as a wave holds it: q = the descriptor of this wave's number (its own item in a quad)
__device__ __forceinline__ PairCtx pair_ctx(const Unit& u) {
    const int tid = threadIdx.x, half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const bool same = (u.a.w >> 30) & 1;
    const int ia = u.a.w & 0x1fffffff, ib = u.b.w;
    const int ih = half ? ib : ia;
    PairCtx c;
    c.live = same || ih >= 0;
    c.it = (half && ib >= 0) ? u.b : u.a;
    c.it.w = 0;
    if (ih < 0) c.it.z = 0;                                  // no second item: the half idles (through the same barriers)
    c.ti = (half && !same) ? 1 : 0;
    c.slab = same ? ia : (ih >= 0 ? ih : ia);
    c.nth = same ? WG : HALF;
    c.t0 = same ? tid : (tid & (HALF - 1));
    c.i = tid & (HALF - 1);
    c.quad = false;
    return c;
}
// (one function, field by field: a ternary over two struct-returning calls puts both structs on the stack)
__device__ __forceinline__ PairCtx unit_ctx(const Unit& u) {
    const int tid = threadIdx.x;
    const bool quad = (u.a.w & QUAD_BIT) != 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = wave >> 1;
    const bool same = (u.a.w >> 30) & 1;
    const int ia = u.a.w & 0x1fffffff;
    // the descriptor this wave works on: its own in a quad, its half's in a pair (the first one's when there is no second)
    // (fields copied out first: a conditional between struct members is a conditional between ADDRESSES to the front end, which keeps
    //  the whole unit on the stack)
    const int ax = u.a.x, ay = u.a.y, az = u.a.z, bx = u.b.x, by = u.b.y, bz = u.b.z, bw = u.b.w, qx = u.q.x, qy = u.q.y, qz = u.q.z, qw = u.q.w;
    const bool second = half && bw >= 0;
    const int mx = quad ? qx : (second ? bx : ax);
    const int my = quad ? qy : (second ? by : ay);
    const int mz = quad ? qz : (second ? bz : az);
    const int ih = quad ? (wave == 0 ? ia : qw) : (half ? bw : ia);          // this wave's / half's item, -1: none
    PairCtx c;
    c.quad = quad;
    c.live = (same && !quad) || ih >= 0;
    c.it = make_int4(mx, my, ih >= 0 ? mz : 0, 0);           // no item: the half / wave idles (through the same barriers)
    c.ti = quad ? wave : ((half && !same) ? 1 : 0);
    c.slab = quad ? (ih >= 0 ? ih : 0) : (same ? ia : (ih >= 0 ? ih : ia));
    c.nth = quad ? 64 : (same ? WG : HALF);
    c.t0 = quad ? (tid & 63) : (same ? tid : (tid & (HALF - 1)));
    c.i = quad ? (tid & 63) : (tid & (HALF - 1));
    return c;
}
// Between the phases of a unit: a pair's tiles are shared by waves (workgroup barrier); a quad's wave owns its tile, its LDS operations
// complete in order, and only the compiler has to be kept from moving accesses across (the tile is read and written as floats
// and as words).  Uniform per workgroup.
__device__ __forceinline__ void unit_sync(bool quad) {
    if (quad) asm volatile("" ::: "memory");
    else __syncthreads();
}
// A quad unit ends without a workgroup barrier (its waves own their tiles), so a wave that is done with its quad may enter the
// workgroup's next unit while the others still scatter into or hand over theirs -- harmless when that unit is a quad again (the
// same wave-owned tiles) or a tail unit (no LDS), a race when it is a PAIR unit, whose tiles are shared by waves and overlap the
// quads' bytes.  With the default blocked-cyclic unit mapping a workgroup's unit numbers only grow (pairs, then quads, then
// tails), but `xcd_map` 0 / 1 and odd grid sizes interleave them.  Uniform per workgroup: every wave walks the same units.
__device__ __forceinline__ void unit_enter(bool quad, bool& prev_quad) {
    if (prev_quad && !quad) __syncthreads();
    prev_quad = quad;
}
// slot w of the unit list (read ahead of meta for the first one: the list is padded to units_cap)
template <bool GATHER = false>
__device__ __forceinline__ Unit unit_load(const TableP& T, int w) {
    const int4* p = (const int4*)((GATHER ? T.units_p : T.units) + (w < T.units_cap ? w : T.units_cap - 1));
    int4 a = p[0], b = p[1], q = p[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    if (w >= T.units_cap) a.z = -2;
    // (wave-uniform: into scalar registers, the particle kernels have no vector registers to spare)
#define UNIT_SCALAR(q) make_int4(__builtin_amdgcn_readfirstlane(q.x), __builtin_amdgcn_readfirstlane(q.y), __builtin_amdgcn_readfirstlane(q.z), __builtin_amdgcn_readfirstlane(q.w))
    Unit u;
    u.a = UNIT_SCALAR(a); u.b = UNIT_SCALAR(b); u.q = UNIT_SCALAR(q);
#undef UNIT_SCALAR
    return u;
}

// -----------------------------------------------------------------------------------------
// Batched environments: B engines of one process (same device, same kernel variants) step together -- one launch per phase with
// gridDim.y = B instead of B launches.  A 200k-particle scene is one round of the chip's resident workgroups and each of its
// launches spends a good part of its time ramping up and draining; B scenes in one launch fill those gaps (fe_step_batch).
// Every substep kernel is a __device__ body with two entry points: k_x(args) and k_x_b(Batch<args>) picking its env by blockIdx.y.
// -----------------------------------------------------------------------------------------
#define FE_MAX_BATCH 8
// The six substep kernels start at multiples of FE_KALIGN_BYTES in the code object: where a kernel sat relative to the instruction
// cache's lines and sets used to move with every edit of an UNRELATED kernel in front of it (k_grid: 12.2 <-> 12.6 us from a change
// in the sort), which made A/B numbers of small changes unreadable.
// (16 KB: the kernels then only move when the code in front of them grows past a 16 KB boundary.  With 1 KB the last edit of the round -- a sort
// kernel 1 KB longer -- put k_p2g at ...e400 instead of ...e000 and cost it 0.4 us per launch, k_grid and k_g2p 0.2 each: profiles/r04_ab_kernel_alignment.txt)
#ifndef FE_KALIGN_BYTES
#define FE_KALIGN_BYTES 16384
#endif
#define FE_KALIGN __attribute__((aligned(FE_KALIGN_BYTES)))
template <typename A> struct Batch { A a[FE_MAX_BATCH]; };

// =========================================================================================
// forward kernels
// =========================================================================================

// Effector.move_kernel (effector.py:157-161), one thread
__device__ void effector_move(const EffP& e, int f) {
    float xin[3] = {e.pos[f * 3] + e.v[f * 3], e.pos[f * 3 + 1] + e.v[f * 3 + 1], e.pos[f * 3 + 2] + e.v[f * 3 + 2]};
    float xn[3], J[3][3];
    boundary_x(e.bnd, xin, xn, J);
    e.pos[(f + 1) * 3] = xn[0]; e.pos[(f + 1) * 3 + 1] = xn[1]; e.pos[(f + 1) * 3 + 2] = xn[2];
    float w3[3] = {e.w[f * 3], e.w[f * 3 + 1], e.w[f * 3 + 2]};
    float q[4] = {e.quat[f * 4], e.quat[f * 4 + 1], e.quat[f * 4 + 2], e.quat[f * 4 + 3]};
    float qw[4], qo[4];
    quat_from_w(w3, qw);
    quat_mul(qw, q, qo);
    e.quat[(f + 1) * 4] = qo[0]; e.quat[(f + 1) * 4 + 1] = qo[1]; e.quat[(f + 1) * 4 + 2] = qo[2]; e.quat[(f + 1) * 4 + 3] = qo[3];
}

// Per-frame store of the forward grid (summed (p, m) and v_out of every static active block), so that the backward
// pass can skip the recompute of P2G + grid_op when the frame had no slow-path particle (gs_flag[f] == 1).
// A block's record in the store: 64 float4 (p, m), then its 64 v_out PACKED at 12 bytes per node (round 3: float4 with w unused --
// a quarter of what k_grid writes there and of what the adjoint's tile load reads back): GS_BLK float4 units = 1,792 bytes.
#define GS_BLK 112
struct GridStore { float4* data; int* flag; int cap;         // data: [(L+1) * cap * GS_BLK] float4
    // Which entries of the order's active list got anything this substep.  The list holds every block a tile may reach (27 per
    // occupied block); what the particles actually reach is 40-60 % of it (water flying apart: 9.6k of 23.4k blocks), and the
    // grid kernels are rounds of a dependent chain per entry.  The scatter kernel marks the entries it deposits into -- every
    // particle ORs the (at most 8) tile regions of its stencil into its wave's set, one lane per region sets touched[entry] (a
    // plain byte store; the entry numbers of the item's 27 neighbours are fetched with the item) -- and slow-path atomics set
    // dirty[entry]: only those blocks have anything in their g_in / gg_out planes to read and re-zero.  A mark is the launch's
    // stamp (1..255, cyclic; nothing is ever cleared: a stale mark that aliases after 255 launches costs one block of zeros).
    // k_grid gives every workgroup a range of entries, works on the marked ones, and records them -- live[f][e] beside the stored
    // grid, cur[e] for the recompute path -- for k_grid_grad: a node without mass has v_out = 0 and passes no adjoint on
    // (mpm:383), and no particle reads such a node.
    unsigned char* touched; unsigned char* dirty; unsigned char* live; unsigned char* cur; unsigned char stamp; };
// tile regions (= neighbour blocks, bit (di+1)*9 + (dj+1)*3 + dk+1) under the 3^3 stencil whose base has tile index lb.
// Tile index 0 = node 4B-1 (block B-1), 1..4 = block B, 5..7 = block B+1; a base index is 0..5.
__device__ __forceinline__ int stencil_regions(int lb) {
    const int lut = 0x26c93;                                    // per base index the 3-bit set {region(t), .., region(t+2)}: 3,2,2,6,6,4
    const int ax = (lut >> (3 * (lb >> 6))) & 7, ay = (lut >> (3 * ((lb >> 3) & 7))) & 7, az = (lut >> (3 * (lb & 7))) & 7;
    const int myz = ((ay & 1) ? az : 0) | ((ay & 2) ? az << 3 : 0) | ((ay & 4) ? az << 6 : 0);
    return ((ax & 1) ? myz : 0) | ((ax & 2) ? myz << 9 : 0) | ((ax & 4) ? myz << 18 : 0);
}
// lane n < 27: the active-list entry of neighbour block n of `block` (-1 outside the grid / not on the list)
__device__ __forceinline__ int neighbour_entry(const int* __restrict__ blk_slot, int nb, int pk) {      // pk: the block's packed coordinates (BLK_PACK)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // (opaque: otherwise the three offsets below are hoisted out of the unit loop and live -- spilled -- across the whole kernel)
    int e = -1;
    if (lane < 27) {
        const int i2 = BLK_I(pk) + lane / 9 - 1, j2 = BLK_J(pk) + (lane / 3) % 3 - 1, k2 = BLK_K(pk) + lane % 3 - 1;
        if ((unsigned)i2 < (unsigned)nb && (unsigned)j2 < (unsigned)nb && (unsigned)k2 < (unsigned)nb) e = blk_slot[(i2 * nb + j2) * nb + k2];
    }
    return e;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or(int x) { return x | __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, true); }
// all lanes of the wave: the union of the lanes' region sets, then one lane per region marks its entry
// (FG kernels: nothing is stored -- the union travels with the tile's arrival, fg_unit_end; returns it)
template <bool FG = false>
__device__ __forceinline__ int touch_regions(int mask, int entry, const GridStore& GS) {
    mask = dpp_or<0x111, 0xf>(mask); mask = dpp_or<0x112, 0xf>(mask); mask = dpp_or<0x114, 0xf>(mask); mask = dpp_or<0x118, 0xf>(mask);   // row_shr 1 2 4 8
    mask = dpp_or<0x142, 0xa>(mask); mask = dpp_or<0x143, 0xc>(mask);                                   // row_bcast 15, 31: lane 63 holds the union
    mask = __builtin_amdgcn_readlane(mask, 63);
    if (!FG && entry >= 0 && ((mask >> (threadIdx.x & 63)) & 1)) GS.touched[entry] = GS.stamp;
    return mask;
}
// a slow-path particle deposited into block b with global atomics
__device__ __forceinline__ void mark_dirty(const GridStore& GS, const int* __restrict__ blk_slot, int b) {
    const int e = blk_slot[b];
    if (e >= 0) GS.dirty[e] = GS.stamp;
}

// -----------------------------------------------------------------------------------------
// The grid pass inside the scatter launches (option "fuse_grid", round 6; kernels with FG = true).
// k_grid / k_grid_grad are a launch boundary, a ramp, a chain of two or three dependent round trips per block and a drain: 20.8 us of a
// substep pair for 7 % of its bytes (a third of the pair where the water has come apart), and nothing in them needs the WHOLE scatter to
// be over -- a block's grid_op needs the slabs of the (at most 27) blocks whose tiles reach it.  So the scatter launch does the grid pass
// itself, as a dataflow step:
//   * every active-list entry has an ARRIVAL WORD (TableP::arrive): a count of the tiles handed over around it so far, and how many of them
//     deposited into it (`touched`) or left something in its slow-path accumulator (`dirty`) -- what GridStore::touched / dirty say in
//     the two-launch form.  A tile's hand-over ends with ONE non-returning 64-bit atomic per neighbour entry (27 lanes of one wave), issued
//     after the slab's write-through (sc1) stores and the shell's atomics have completed (s_waitcnt vmcnt(0); no fence: MI355X_MICROARCH.md,
//     "sc1 payload -> vmcnt(0) -> flag").  How many arrivals an entry waits for is known at sort time (`expected`: the slabs of its
//     27 neighbours, build_units_dev).
//   * every entry has an OWNER: wave static_entry(w) + k * (waves of the launch), the mapping of k_grid.  A wave that is through with its
//     units polls the words of its entries (sc1 loads) and runs grid_op (fg_fwd_block) / its adjoint (fg_bwd_block) on each as it
//     completes -- the same gather in the same fixed order as k_grid (results are bit-identical), outputs written through.  All waves of
//     all workgroups own entries, so the pass is as parallel as the separate launch was, without its boundary, ramp and first round trip.
//   * WAITING is only allowed when it cannot deadlock: a wave waits for an entry only if every workgroup of the launch has started (the
//     `started` counters against SimP::fg_base + gridDim.x): then every unit loop is running or done, and unit loops never wait.  Otherwise
//     (more workgroups than the chip holds at once: a shared GPU) the wave marks its incomplete entries as SKIPPED and goes; they fall to
//     the launch's FINAL wave.
//   * the FINAL wave is the one whose workgroup is the last to finish its unit loops (a sharded returning counter, asked for before the
//     owner's work and read behind it).  It finds out from the same word whether anything RARE happened and, if so, does what needs every
//     unit loop over: skipped entries; blocks outside the order's active list (the dynamic list of the slow path); and LATE deposits -- a
//     slow-path particle whose target is an active entry OUTSIDE the 27 neighbours of its own unit's block (it moved more than a block
//     between two sorts, or sits in a tail unit) is ordered by nobody's arrival, so its deposit goes to accumulator planes of its own
//     (FgDev::late) and the final wave adds it to what the owner stored (forward: the grid store's (p, m) totals; backward: grid_op's
//     adjoint is linear in d v_out).  None of this is on the path of a launch without such particles.
// Because a launch now reads the grid of the substep before (its gather part) while its owners write the one of its own substep, g_out and
// gg_in are double-buffered by the parity of f.  Host side: fuse_grid_ok() says which launches take this form.
// -----------------------------------------------------------------------------------------
#define FG_SH 32                 // shards of the done / finished counters (a returning atomic on ONE word goes at ~88 per us: 1,024 workgroups would queue for 12 us)
#define FG_LINE 32               // a counter per 128-byte line
enum { FGC_STARTED = 0 /* 8 */, FGC_DONE = 8 /* FG_SH */, FGC_FIN = FGC_DONE + FG_SH /* FG_SH */, FGC_TOP = FGC_FIN + FG_SH, FGC_LATE, FGC_SKIP, FGC_ERR, FGC_N };
struct FgDev {                   // device-resident (one pointer in the kernel arguments)
    float* late;                 // [4 * ncell] accumulator planes of the late deposits (the adjoint pass uses three)
    int* late_flag; int* late_list;   // [nblk] entries with late deposits (flag: on the list already)
    unsigned char* skipm;        // [nblk] entries their owner could not wait for (the launch's stamp)
    int* ctr;                    // [FGC_N * FG_LINE]
};
struct FgArgs {                  // what the grid pass of one launch works with (uniform)
    const FgDev* F; const float4* slab; float* acc; float4* out; int* blk_list; int* blk_count; int* blk_flag; int f;
};
#define FG_CNT(a) ((int)((a) & 0x1fffffull))
#define FG_TCH(a) ((int)(((a) >> 21) & 0x1fffffull))
#define FG_DRT(a) ((int)((a) >> 42))
__shared__ int s_fg[10];         // pair units: (touched, dirty) region sets of the workgroup's tiles [0..7]; waves through with their unit loops [8], with their entries [9]
template <bool BWD> __device__ __forceinline__ void fg_owner_phase(const SimP& S, const TableP& T, const GridStore& GS, const FgArgs& A);
__device__ __forceinline__ void fg_rare(const FgDev* F) { (void)__hip_atomic_fetch_or(F->ctr + FGC_TOP * FG_LINE, 1 << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // (the word's low half counts the shards that are done: fg_owner_phase)
// entry e received a deposit that no arrival orders
__device__ __forceinline__ void fg_mark_late(const FgDev* F, int e) {
    int* fl = F->late_flag + e;
    const int v = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != 1 && atomicCAS(fl, v, 1) == v) {
        const int i = atomicAdd(F->ctr + FGC_LATE * FG_LINE, 1);
        __hip_atomic_store(F->late_list + i, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fg_rare(F);
    }
}
// the launch begins: this workgroup has started (thread 0), the workgroup's words are cleared
__device__ __forceinline__ void fg_begin(const FgDev* F) {
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(F->ctr + (FGC_STARTED + (blockIdx.x & 7)) * FG_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 10) s_fg[threadIdx.x] = 0;
    __syncthreads();
}
__device__ __forceinline__ int wave_or(int mask) {             // all 64 lanes; wave-uniform result
    mask = dpp_or<0x111, 0xf>(mask); mask = dpp_or<0x112, 0xf>(mask); mask = dpp_or<0x114, 0xf>(mask); mask = dpp_or<0x118, 0xf>(mask);
    mask = dpp_or<0x142, 0xa>(mask); mask = dpp_or<0x143, 0xc>(mask);
    return __builtin_amdgcn_readlane(mask, 63);
}
// A tile has been handed over (its slab stored write-through, its shell's atomics issued, all of it completed: the caller waited): one lane per
// neighbour entry of the tile's block counts it in, with what the tile did there.  All 64 lanes of the tile's first wave.
__device__ __forceinline__ void fg_arrive(const TableP& T, int nbr_entry, int tmask, int dmask) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    if (lane < 27 && nbr_entry >= 0) {
        const unsigned long long add = 1ull | ((unsigned long long)((tmask >> lane) & 1) << 21) | ((unsigned long long)((dmask >> lane) & 1) << 42);
        (void)__hip_atomic_fetch_add(T.arrive + nbr_entry, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the end of a tile unit of an FG kernel: region sets of the tile's waves merged (pair units: through LDS), everything the tile's waves stored waited for,
// then the arrival by the tile's first wave.  fg_t: the wave's touched set (uniform), fg_d: this lane's dirtied regions.  Contains the unit's closing unit_sync.
__device__ __forceinline__ void fg_unit_end(const TableP& T, const PairCtx& pc, int nbr_entry, int fg_t, int fg_d) {
    int tm = fg_t, dm = wave_or(fg_d);
    if (!pc.quad && (threadIdx.x & 63) == 0) {
        if (tm) atomicOr(&s_fg[2 * pc.ti], tm);
        if (dm) atomicOr(&s_fg[2 * pc.ti + 1], dm);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's slab stores (sc1: through to memory) and atomics have completed
    unit_sync(pc.quad);
    if (pc.live && pc.t0 < 64) {                                // (wave-uniform: the first wave of the tile)
        if (!pc.quad) {
            tm = s_fg[2 * pc.ti]; dm = s_fg[2 * pc.ti + 1];
            if ((threadIdx.x & 63) == 0) { s_fg[2 * pc.ti] = 0; s_fg[2 * pc.ti + 1] = 0; }      // (the next unit's waves add to them behind its first barrier, which this wave takes part in)
        }
        fg_arrive(T, nbr_entry, tm, dm);
    }
}

// one node of a slab (the slab index is wave-uniform: the descriptor of the write-through form is built over the slab itself).
// NPL = 4: (p, m) of k_p2g, one float4 per node.  NPL = 3: the adjoint d v_out of k_g2p_grad, PACKED at 12 bytes per node (2.6 KB per
// slab instead of 3.4: round 3 stored a float4 whose w was padding, a quarter of what the kernel hands over and k_grid_grad gathers).
// Both kinds live in the same buffer at the same slab stride (they are never alive at the same time).
template <int NPL>
__device__ __forceinline__ void slab_store(float4* slab, int item, int l, float4 v, int wt) {
    float4* base = slab + (size_t)item * SLAB_N;
    if (NPL == 4) { if (wt) wt_store16(base, (unsigned)l * 16u, v); else base[l] = v; }
    else {
        const u32x3 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
        if (wt) __builtin_amdgcn_raw_buffer_store_b96(u, wt_rsrc(base), l * 12, 0, 16);               // aux 16 = sc1
        else __builtin_amdgcn_raw_buffer_store_b96(u, wt_rsrc(base), l * 12, 0, 0);
    }
}
__device__ __forceinline__ int tile_region(int t) { return (t + 3) >> 2; }        // tile index 0 -> block B-1, 1..4 -> B, 5..7 -> B+1
// Tile node l of a finished scatter tile: the inner 6^3 go to the item's slab; a shell node that received something is added to
// the slow-path accumulator `acc` (NPL planes of ncell floats) and its block -- active-list entry from the item's 27 neighbour
// entries, lane r of every wave holds neighbour r -- is marked dirty for the grid kernel.  Called by whole waves (shuffle).
// FG (the grid pass rides on this launch): a shell deposit's region is noted in fg_d, for the tile's arrival, instead of a mark in GS.dirty
template <int NPL, bool SHELL_ONLY = false, bool FG = false>
__device__ __forceinline__ void tile_handover(const SimP& S, float4* slab, int item, float* acc, const GridStore& GS, const TileO& to,
                                              int nbr_entry, int l, float4 v, int wt, int& fg_d) {
    asm volatile("" : "+v"(l));             // (opaque, as in neighbour_entry: tz and its region are loop invariants otherwise)
    const int tx = l >> 6, ty = (l >> 3) & 7, tz = l & 7;
    const int reg = tile_region(tx) * 9 + tile_region(ty) * 3 + tile_region(tz);
    const int e = __shfl(nbr_entry, reg, 64);
    // (one condition, not three short-circuited ones: those became nested branches across which the pieces of the slab index were kept
    //  alive as 64-bit values -- and, in k_g2p_grad2, spilled)
    const bool inner = ((unsigned)(tx - 1) < (unsigned)SLAB_T) & ((unsigned)(ty - 1) < (unsigned)SLAB_T) & ((unsigned)(tz - 1) < (unsigned)SLAB_T);
    if (inner) {
        if (!SHELL_ONLY) slab_store<NPL>(slab, item, ((tx - 1) * SLAB_T + (ty - 1)) * SLAB_T + (tz - 1), v, wt);
    } else if (e >= 0 && (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f)) {
        float* dst = acc + cell_addr(to.ox + tx, to.oy + ty, to.oz + tz, S.nb);
        unsafeAtomicAdd(dst, v.x); unsafeAtomicAdd(dst + S.ncell, v.y); unsafeAtomicAdd(dst + 2 * S.ncell, v.z);
        if (NPL > 3) unsafeAtomicAdd(dst + 3 * S.ncell, v.w);
        if (FG) fg_d |= 1 << reg; else GS.dirty[e] = GS.stamp;
    }
}
template <int NPL, bool SHELL_ONLY = false>
__device__ __forceinline__ void tile_handover(const SimP& S, float4* slab, int item, float* acc, const GridStore& GS, const TileO& to,
                                              int nbr_entry, int l, float4 v, int wt) {
    int none = 0;
    tile_handover<NPL, SHELL_ONLY, false>(S, slab, item, acc, GS, to, nbr_entry, l, v, wt, none);
}

#ifndef FE_LEAN_QUADS
#define FE_LEAN_QUADS 0       // (A/B builds: 1 = a quad unit's wave hands over / loads the inner 6^3 nodes of its tile and walks the shell only when a particle sits on it.
                              //  Measured, round 5: nothing in the splash (149.9 vs 149.6 us per pair) and +0.8 us per pair over the timed region, where no quad unit runs --
                              //  the larger kernels sit differently; profiles/r05_ab_lane_split.txt)
#endif
// does the 3^3 stencil with base index lb reach the tile's outer shell (tile index 0 or 7 on some axis: base 0 or 5)?
__device__ __forceinline__ bool stencil_on_shell(int lb) {
    const int bx = lb >> 6, by = (lb >> 3) & 7, bz = lb & 7;
    return (bx == 0) | (bx == 5) | (by == 0) | (by == 5) | (bz == 0) | (bz == 5);
}
// A quad unit's wave hands its fixed-point tile over by itself, eight nodes per lane: round 4 walked all 512 nodes through tile_handover
// (region of the node, cross-lane read of its entry, inner / shell decision: ~52 VALU instructions a node, 416 per wave -- as much as
// half the 27-node loop, scripts/valu_profile.py) although the shell of a small item is rarely touched at all: only a particle whose
// base has left its block since the sort reaches it.  Now the inner 6^3 nodes go straight to the slab (216 nodes: four rounds, no
// decisions), and the walk over the shell only happens in a wave one of whose particles sits on it (`wshell`, wave-uniform).
// acc: the wave's tile (words); inv_p / inv_m: back to floats (fix_scale), NPL planes.
template <int NPL, bool FG = false>
__device__ __forceinline__ void quad_handover(const SimP& S, const int* acc, float inv_p, float inv_m, float4* slab, int item, float* accg,
                                              const GridStore& GS, const TileO& to, int nbr_entry, bool wshell, int wt, int& fg_d) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // (opaque: the node indices below are not to be kept across the unit loop)
#if !FE_LEAN_QUADS
    for (int k = 0; k < TILE_N / 64; k++) {                        // (A/B builds: round 4's walk over all 512 nodes)
        const int l = lane + 64 * k;
        tile_handover<NPL, false, FG>(S, slab, item, accg, GS, to, nbr_entry, l, make_float4((float)acc[l] * inv_p, (float)acc[TILE_N + l] * inv_p, (float)acc[2 * TILE_N + l] * inv_p, NPL > 3 ? (float)acc[3 * TILE_N + l] * inv_m : 0.f), wt, fg_d);
    }
    return;
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int l6 = lane + 64 * k;                               // node of the slab: (tx, ty, tz) in 0..5 = tile indices 1..6
        if (k < 3 || l6 < SLAB_N) {
            const int tx = l6 / 36, r = l6 - 36 * tx, ty = r / 6, tz = r - 6 * ty;
            const int l = ((tx + 1) * TILE_T + ty + 1) * TILE_T + tz + 1;
            const float4 v = make_float4((float)acc[l] * inv_p, (float)acc[TILE_N + l] * inv_p, (float)acc[2 * TILE_N + l] * inv_p, NPL > 3 ? (float)acc[3 * TILE_N + l] * inv_m : 0.f);
            slab_store<NPL>(slab, item, l6, v, wt);
        }
    }
    if (wshell) {
#pragma unroll 2
        for (int k = 0; k < TILE_N / 64; k++) {
            const int l = lane + 64 * k;
            const int tx = l >> 6, ty = (l >> 3) & 7, tz = l & 7;
            const bool inner = ((unsigned)(tx - 1) < (unsigned)SLAB_T) & ((unsigned)(ty - 1) < (unsigned)SLAB_T) & ((unsigned)(tz - 1) < (unsigned)SLAB_T);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!inner) v = make_float4((float)acc[l] * inv_p, (float)acc[TILE_N + l] * inv_p, (float)acc[2 * TILE_N + l] * inv_p, NPL > 3 ? (float)acc[3 * TILE_N + l] * inv_m : 0.f);
            tile_handover<NPL, true, FG>(S, slab, item, accg, GS, to, nbr_entry, l, v, wt, fg_d);
        }
    }
}
template <int NPL>
__device__ __forceinline__ void quad_handover(const SimP& S, const int* acc, float inv_p, float inv_m, float4* slab, int item, float* accg,
                                              const GridStore& GS, const TileO& to, int nbr_entry, bool wshell, int wt) {
    int none = 0;
    quad_handover<NPL, false>(S, acc, inv_p, inv_m, slab, item, accg, GS, to, nbr_entry, wshell, wt, none);
}

struct GridW {            // everything a scattering particle needs of the global grid
    float* g_in; float4* slab; int ncell; int* frame_slow; int* blk_flag; int* blk_list; int* blk_count; int* err; int* slow;
    const FgDev* fg; float4* g_out;      // FG kernels: the grid pass' device block, and where grid_op leaves this substep's v_out
};

// An F that is carried over unchanged (an unused or collected particle) from a frame with full planes into a compact one: of an inviscid liquid's
// F only the determinant is ever consumed (constitutive_eval_t), so c = det(F)^(1/3) stands for it exactly (F = I, the pool's state: c = 1)
__device__ __forceinline__ void carry_F(const FrameV& cur, const FrameV& nxt, m3& F) {
    if (nxt.iso == 1 && cur.iso != 1) { const float c = fe_cbrt_pos(m3_det(F)); F = m3_zero(); F.a[0][0] = F.a[1][1] = F.a[2][2] = c; }
}
// advect_used + process_unused_particles (mpm:304-316) + Injector.act (injector.py:80-105) for one unused slot
__device__ __forceinline__ void unused_particle_fwd(const SimP& S, const FrameV& cur, const FrameV& nxt, int s, int pid,
                                                    const int* __restrict__ pool_idx, const AgentP& agent, const InjectP& inj, int f) {
    PState p;
    load_xvC(cur, s, p);
    load_F(cur, s, p.F);
    int used_next = 0;
    if (inj.on) {
        int j = pool_idx[pid] - inj.act_id;
        if (j >= 0 && j < inj.flux) {                                 // this pool particle is injected now
            const EffP& e = agent.e[agent.inj];
            const float* rv = e.random_vector + ((size_t)inj.row * e.flux + j) * 3;
            float q[4] = {e.quat[f * 4], e.quat[f * 4 + 1], e.quat[f * 4 + 2], e.quat[f * 4 + 3]};
            float ip[3], iv[3];
            quat_rotate(e.inject_p, q, ip);
            quat_rotate(e.inject_v, q, iv);
            float vnorm = sqrtf(e.inject_v[0] * e.inject_v[0] + e.inject_v[1] * e.inject_v[1] + e.inject_v[2] * e.inject_v[2]);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                float offset = (rv[d] * 2.f - 1.f) * e.radius;
                p.x[d] = offset + e.pos[f * 3 + d] + ip[d];
                p.v[d] = e.randomize_inject_v ? iv[d] + (rv[d] * 2.f - 1.f) * vnorm * 2.0f : iv[d];
            }
            used_next = 1;
        }
    }
    store_xvC(nxt, s, p.x, p.v, p.C);
    carry_F(cur, nxt, p.F);
    store_F(nxt, s, p.F);
    pstore(nxt, nxt.used, s, used_next);
}

// collector_act_kernel (agent_pouring.py:30-41, agent_jetbot.py:33-43) for one used slot: outside the collector boundary the
// particle is marked unused in frames f and f+1 and parked at NOWHERE in f+1 (v, C, F carried over, where the reference
// leaves f+1 stale).  Returns true when the particle was taken; the backward pass then finds used[f] == 0.
__device__ __forceinline__ bool collector_takes(const SimP& S, const FrameV& cur, const FrameV& nxt, int s, const float4* __restrict__ info, const AgentP& agent) {
    if (agent.collector_mat >= 0 && load_info(S, info, s).mat != agent.collector_mat) return false;
    PState p;
    load_xvC(cur, s, p);
    if (!boundary_is_out(*agent.collector, p.x)) return false;
    load_F(cur, s, p.F);
    const float nowhere[3] = {-100.f, -100.f, -100.f};                    // macros.py:216
    store_xvC(nxt, s, nowhere, p.v, p.C);
    carry_F(cur, nxt, p.F);
    store_F(nxt, s, p.F);
    cur.used[s] = 0; nxt.used[s] = 0;
    return true;
}

// what a used particle contributes to its 27 nodes (mpm:339-353): momentum at the stencil base + affine increments
struct P2GPrep { Stencil st; float mv[3]; m3 affine; float m; bool inside; };

// compute_F_tmp + svd + stress + F update for one used particle (mpm:254-264, 331-344, 355-378)
struct P2GRaw { PState p; PInfo info; };
__device__ __forceinline__ void p2g_load(const SimP& S, const FrameV& cur, int s, const float4* __restrict__ info_, P2GRaw& r) {
    load_xvC(cur, s, r.p);
    load_F(cur, s, r.p.F);
    r.info = load_info(S, info_, s);
}
// `primary`: this lane is the particle's first one (a split wave has several: lane_split) -- the stores and the error count are its
template <bool WRITE, bool GENERAL>
__device__ __forceinline__ void p2g_compute(const SimP& S, const FrameV& nxt, int s, const P2GRaw& r, const GridW& G, P2GPrep& q, bool primary = true) {
    const PState& p = r.p;
    const PInfo& info = r.info;
    Constitutive k;
    constitutive_eval_t<GENERAL>(p.C, p.F, S.dt, info.mu, info.lam, info.mass, info.cls, S.stress_scale, k);
    if (WRITE && primary) store_F_used(nxt, s, k.Fnew, 1);
    stencil_make(p.x, S.inv_dx, q.st);
    q.inside = stencil_inside(q.st, S.n);
    if (!q.inside && primary) atomicAdd(G.err, 1);
    q.m = info.mass;
    q.affine = k.affine;
    // momentum at the base node, then per-node increments: mom(o) = m v + A (o - fx) dx
#pragma unroll
    for (int a = 0; a < 3; a++)
        q.mv[a] = q.m * p.v[a] - S.dx * (k.affine.a[a][0] * q.st.fx[0] + k.affine.a[a][1] * q.st.fx[1] + k.affine.a[a][2] * q.st.fx[2]);
}
template <bool WRITE, bool GENERAL>
__device__ __forceinline__ void p2g_prepare(const SimP& S, const FrameV& cur, const FrameV& nxt, int s,
                                            const float4* __restrict__ info_, const GridW& G, P2GPrep& q) {
    P2GRaw r;
    p2g_load(S, cur, s, info_, r);
    p2g_compute<WRITE, GENERAL>(S, nxt, s, r, G, q);
}

// global path: 108 scattered global atomics + active-block marking
// FG (the grid pass rides on this launch): `pk` = the block of the particle's own unit (-1: a tail unit).  A deposit into a block among that block's 27
// neighbours is ordered by the unit's arrival (its region goes into fg_d); one into an active entry further away is LATE -- into the planes of its own,
// and the entry on the late list; blocks outside the active list go on the dynamic list as ever (both: the launch's final wave, fg_final).
template <bool FG = false>
__device__ __forceinline__ void p2g_scatter_global(const SimP& S, const P2GPrep& q, const GridW& G, const GridStore& GS, const int* __restrict__ blk_slot, int pk, int& fg_d) {
    const Stencil& st = q.st;
    // the (up to 8) 4^3 blocks this stencil touches: lower and upper block per axis (equal when the three nodes sit in one
    // block).  All flags and list entries are asked for at once (a rolled triple loop made this up to eight rounds of two
    // dependent round trips per particle -- nothing for a few drifted particles, too much for the loose ones).
    const int bl[3] = {st.base[0] >> 2, st.base[1] >> 2, st.base[2] >> 2};
    const int bh[3] = {(st.base[0] + 2) >> 2, (st.base[1] + 2) >> 2, (st.base[2] + 2) >> 2};
    // FG: the corners whose block no arrival of this unit reaches (bit c: further than a block from the unit's own) -- their deposits go to the planes of the late
    // deposits, whether the block is an entry of the active list (-> the late list) or not (-> the dynamic list: a block off the list is never within a block of
    // an occupied one, so everything it ever gets arrives this way).  Geometry only: nothing is asked for ahead of the deposits.
    int late = 0;
    float* acc_late = G.g_in;
    auto region = [&](int c, bool& near) {
        const int d0 = ((c & 1) ? bh[0] : bl[0]) - BLK_I(pk), d1 = ((c & 2) ? bh[1] : bl[1]) - BLK_J(pk), d2 = ((c & 4) ? bh[2] : bl[2]) - BLK_K(pk);
        near = pk >= 0 && (unsigned)(d0 + 1) <= 2u && (unsigned)(d1 + 1) <= 2u && (unsigned)(d2 + 1) <= 2u;
        return (d0 + 1) * 9 + (d1 + 1) * 3 + d2 + 1;
    };
    if (FG) {
        acc_late = G.fg->late;
#pragma unroll
        for (int c = 0; c < 8; c++) { bool near; (void)region(c, near); if (!near) late |= 1 << c; }
    }
#pragma unroll 1
    for (int ij = 0; ij < 9; ij++) {
        const int i = ij / 3, j = ij - 3 * i;
        const float wij = STW(st, i, 0) * STW(st, j, 1);
        const float ox = (float)i * S.dx, oy = (float)j * S.dx;
        const int cij = ((((st.base[0] + i) >> 2) != bl[0]) ? 1 : 0) | ((((st.base[1] + j) >> 2) != bl[1]) ? 2 : 0);
        float mij[3];
#pragma unroll
        for (int a = 0; a < 3; a++) mij[a] = q.mv[a] + q.affine.a[a][0] * ox + q.affine.a[a][1] * oy;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float weight = wij * st.w[kk][2];
            const float oz = (float)kk * S.dx;
            float* acc = G.g_in;
            if (FG) { const int c = cij | ((((st.base[2] + kk) >> 2) != bl[2]) ? 4 : 0); acc = ((late >> c) & 1) ? acc_late : G.g_in; }
            float* dst = acc + cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + kk, S.nb);
#pragma unroll
            for (int a = 0; a < 3; a++) unsafeAtomicAdd(dst + a * S.ncell, weight * (mij[a] + q.affine.a[a][2] * oz));
            unsafeAtomicAdd(dst + 3 * S.ncell, weight * q.m);
        }
    }
    int ent[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int b = (((c & 1) ? bh[0] : bl[0]) * S.nb + ((c & 2) ? bh[1] : bl[1])) * S.nb + ((c & 4) ? bh[2] : bl[2]);
        ent[c] = blk_slot[b];
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
        if (ent[c] >= 0) {                                                                  // a block of the order's active list
            if (!FG) GS.dirty[ent[c]] = GS.stamp;
            else if ((late >> c) & 1) fg_mark_late(G.fg, ent[c]);
            else { bool near; fg_d |= 1 << region(c, near); }
        } else {
            const int b = (((c & 1) ? bh[0] : bl[0]) * S.nb + ((c & 2) ? bh[1] : bl[1])) * S.nb + ((c & 4) ? bh[2] : bl[2]);
            const bool first = mark_dynamic(b, G.blk_flag, G.blk_list, G.blk_count);
            if (FG) { if (first) fg_rare(G.fg); }
            else *G.frame_slow = 1;                                                         // store incomplete for this frame
        }
    }
}
__device__ __forceinline__ void p2g_scatter_global(const SimP& S, const P2GPrep& q, const GridW& G, const GridStore& GS, const int* __restrict__ blk_slot) {
    int none = 0;
    p2g_scatter_global<false>(S, q, G, GS, blk_slot, -1, none);
}

// Fixed-point accumulators of a quad unit's tiles.  Four fp64 tiles (16 KB each) do not fit beside each other at four workgroups per
// CU, and ds_add_f32 is no alternative on this chip (0.2 T lane-ops/s against 5.9 T for ds_add_u32, profiles/r01_ubench_lds_types.txt).
// A quad's tile belongs to ONE wave holding <= 64 particles: the wave takes the largest contribution any of its lanes can make
// (M, a bound), scales by the power of two that puts M at 2^24 -- so that 64 of them stay below 2^30 --, runs the same fp32 segmented
// scan on the scaled values and adds the run totals, rounded to nearest, with ds_add_u32.  The quantum is 2^-24 M, an fp32 ulp of the
// largest contribution (what fp32 atomics into the node would lose, and independent of the order of the adds); the hand-over converts
// back with the exact inverse.  Momentum / adjoint planes and the mass plane have scales of their own.
// fix_nonfinite: fmaxf ignores a NaN and fix_round(NaN) is 0, so a particle state that has blown up would vanish from the sums of a
// quad unit while the fp64 tiles of a pair unit -- and the reference's atomics, mpm:346-353 -- carry it to the grid.  A wave that sees
// a non-finite bound therefore hands its whole tile over as NaN: the divergence stays visible in losses and gradients.
// (DPP: row_shr 1 2 4 8 leave a row's maximum in its last lane, row_bcast 15 / 31 carry it on into lane 63 -- six VALU instructions.  As six
//  __shfl_xor steps it was six dependent ds_bpermute round trips, twice per quad unit and kernel: ~0.6 us of a small wave's chain)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float x) { return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, true))); }
__device__ __forceinline__ float wave_max(float x) {           // all 64 lanes; x >= 0 (a lane without a source reads 0); wave-uniform result
    x = dpp_max<0x111, 0xf>(x); x = dpp_max<0x112, 0xf>(x); x = dpp_max<0x114, 0xf>(x); x = dpp_max<0x118, 0xf>(x);
    x = dpp_max<0x142, 0xa>(x); x = dpp_max<0x143, 0xc>(x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// round to nearest into the fixed-point word.  FIX_RPI: v_cvt_rpi_i32_f32 = floor(x + 0.5) in ONE instruction where rintf + the
// conversion are two (108 / 81 of them per lane in the scatter loops of a quad unit); ties go up instead of to even, which is as
// good a rounding for a sum of ~64 terms.
#ifndef FIX_RPI
#define FIX_RPI 0
#endif
__device__ __forceinline__ int fix_round(float c) {
#if FIX_RPI
    int r; asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(c)); return r;
#else
    return (int)rintf(c);
#endif
}
struct FixScale { float s, inv; };
__device__ __forceinline__ FixScale fix_scale(float M) {       // M wave-uniform, >= 0
    int e = ((__builtin_amdgcn_readfirstlane(__float_as_int(M)) >> 23) & 0xff) - 126;          // M < 2^e (scalar registers from here on)
    e = e < -100 ? -100 : e;
    FixScale f; f.s = __int_as_float((127 + 24 - e) << 23); f.inv = __int_as_float((127 - 24 + e) << 23);
    return f;
}

// -----------------------------------------------------------------------------------------
// Small waves split their stencils over the idle lanes (round 5; option "lane_split").
// What a SIMD pays for is the instruction, not the lane: a wave that holds 5 particles issues the same 27-node loop as one that holds
// 64, and where the water has come apart most waves are of that kind -- in windows 18-35 of the benchmark 40-55 % of the work items hold
// at most 16 particles, 29 % of the timed region's VALU instructions ran on idle lanes (profiles/r04_pmc_issue_counters.txt).  A wave
// whose item (its half of the item) has at most 21 particles therefore gives every particle THREE lanes, one per x offset of the
// stencil: lane L works for particle L % 21 on the nine nodes of plane i = L / 21, all three ask for the particle's state (one address:
// a broadcast) and run the constitutive model (the same instructions the wave issued anyway), and the node loop is nine nodes long
// instead of 27.  At most 7 particles: NINE lanes each, one per (i, j) column of three nodes.  The segmented scan is the same (the key
// carries the group, so lanes of different groups never merge), the sums are the same sums in the same accumulators; stores, slow
// paths and counters belong to the particle's first lane (`primary`).  Gather loops (k_g2p_grad2's first pass) reduce their partial
// sums over the particle's lanes with cross-lane reads.
// -----------------------------------------------------------------------------------------
#ifndef FE_SPLIT3_MAX
#define FE_SPLIT3_MAX 21      // (A/B builds: 0 = never split)
#endif
#ifndef FE_SPLIT9_MAX
#define FE_SPLIT9_MAX 7
#endif
#ifndef FE_SPLIT_PGG
#define FE_SPLIT_PGG 0        // (A/B builds: 1 = k_p2g_grad splits its small waves too, option lane_split bit 2.  Measured, round 5: its fifteen sums cost
#endif                        //  45 / 90 cross-lane reads per wave, which is what the shorter gather loop saves -- splash 26.8 vs 26.5 us, timed region 20.8 vs 21.0 --
                              //  and the two extra instantiations double the kernel's code, 15.6k VALU instructions where it had 7.3k: profiles/r05_ab_lane_split.txt)
struct LaneSplit { int G, p, gofs; bool ok, primary; };       // G: 1, 3, 9 (wave-uniform); p: the lane's particle within the wave; gofs: tile offset of its group's nodes
__device__ __forceinline__ LaneSplit lane_split(int cnt, bool allowed) {      // cnt: particles of this wave (uniform)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // (opaque: p and gofs are not to become loop invariants of the unit loop)
    LaneSplit ls; ls.G = 1; ls.p = lane; ls.gofs = 0; ls.ok = true; ls.primary = true;
    if (allowed && cnt > 0 && cnt <= FE_SPLIT3_MAX) {
        if (cnt <= FE_SPLIT9_MAX) { const int g = lane / 7; ls.G = 9; ls.p = lane - 7 * g; ls.ok = g < 9; ls.primary = g == 0; const int gi = g / 3; ls.gofs = gi * (TILE_T * TILE_T) + (g - 3 * gi) * TILE_T; }
        else { const int g = lane / 21; ls.G = 3; ls.p = lane - 21 * g; ls.ok = g < 3; ls.primary = g == 0; ls.gofs = g * (TILE_T * TILE_T); }
    }
    return ls;
}
// a value of every lane of the particle (lanes p, p + P, ...) summed into all of them; P = 21 (G = 3) or 7 (G = 9: first over the three
// columns of a plane, then over the planes -- six cross-lane reads instead of nine).  All 64 lanes.
template <int G>
__device__ __forceinline__ float split_sum(float v, int gofs) {      // gofs: the lane's group as lane_split encodes it
    if (G == 1) return v;
    constexpr int P = G == 3 ? 21 : 7;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // (opaque: the source lanes below are not to become invariants of the unit loop, held -- spilled -- across the kernel)
    const int gi = gofs >> 6, gj = (gofs >> 3) & 7;
    const int p = lane - P * (G == 3 ? gi : 3 * gi + gj);         // (lane 63 belongs to no particle: whatever it reads is not used)
    if (G == 3) return __shfl(v, p, 64) + __shfl(v, p + P, 64) + __shfl(v, p + 2 * P, 64);
    const float t = __shfl(v, p + P * (3 * gi), 64) + __shfl(v, p + P * (3 * gi + 1), 64) + __shfl(v, p + P * (3 * gi + 2), 64);
    return __shfl(t, p + P * gj, 64) + __shfl(t, p + P * (3 + gj), 64) + __shfl(t, p + P * (6 + gj), 64);
}

// tile path, executed by ALL lanes of the wave: contributions of lanes with `in_tile` are summed over runs of equal
// stencil base (seg_scan) and the last lane of each run adds the total into the LDS accumulators: fp64, or -- QUAD, `q` scaled by
// the caller -- fixed point.  G = 3 / 9: the lane works on the nodes of its group only (lane_split; gofs = their offset in the tile)
template <bool QUAD, int G>
__device__ __forceinline__ void p2g_scatter_tile(const SimP& S, P2GPrep& q, bool in_tile, int lb, int aofs, int gofs) {
    if (G == 1 && S.wsort) {
        int key = in_tile ? lb : 0x3ff;                        // (lanes without a tile particle go to the end: they add nothing)
        if (wave_needs_sort(key)) {
            const int dest = wave_sort_dest(key);
            wave_send(dest, key);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                wave_send(dest, q.mv[a]);
#pragma unroll
                for (int b = 0; b < 3; b++) { wave_send(dest, q.affine.a[a][b]); wave_send(dest, q.st.w[a][b]); }
            }
            wave_send(dest, q.m);
            in_tile = key != 0x3ff; lb = in_tile ? key : 0;
        }
    }
    // (G > 1: the key carries the group's offset -- neighbouring lanes of different groups hold the same particle, i.e. the same base)
    const SegScan sc = seg_setup(in_tile ? lb + (G > 1 ? gofs << 10 : 0) : (0x40000000 | (int)threadIdx.x));
    const bool issue = sc.tail && in_tile;
    const float live = in_tile ? 1.f : 0.f;
    const Stencil& st = q.st;
    const int gi = gofs >> 6, gj = (gofs >> 3) & 7;            // this lane's x (and y) offset when the wave is split
    const float wgi = G > 1 ? sel3(gi, st.w[0][0], st.w[1][0], st.w[2][0]) : 0.f, wgj = G == 9 ? sel3(gj, st.w[0][1], st.w[1][1], st.w[2][1]) : 0.f;
    const int lg = lb + (G > 1 ? gofs : 0);
#pragma unroll
    for (int ij = 0; ij < 9; ij++) {
        const int i = ij / 3, j = ij - 3 * i;
        if ((G > 1 && i > 0) || (G == 9 && j > 0)) continue;      // (compile time: a split wave's lanes have their x (and y) offset from their group)
        const float wij = live * (G > 1 ? wgi : STW(st, i, 0)) * (G == 9 ? wgj : STW(st, j, 1));
        const float ox = (G > 1 ? (float)gi : (float)i) * S.dx, oy = (G == 9 ? (float)gj : (float)j) * S.dx;
        float mij[3];
#pragma unroll
        for (int a = 0; a < 3; a++) mij[a] = q.mv[a] + q.affine.a[a][0] * ox + q.affine.a[a][1] * oy;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float weight = wij * st.w[kk][2];
            const float oz = (float)kk * S.dx;
            const int l = lg + ((G > 1 ? 0 : i) * TILE_T + (G == 9 ? 0 : j)) * TILE_T + kk;
            float c[4];
#pragma unroll
            for (int a = 0; a < 3; a++) c[a] = weight * (mij[a] + q.affine.a[a][2] * oz);
            c[3] = weight * q.m;
            seg_scan4(sc, c[0], c[1], c[2], c[3]);
            if (issue) {
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    if (QUAD) atomicAdd((int*)s_acc + aofs + a * TILE_N + l, fix_round(c[a]));                // ds_add_u32
                    else atomicAdd(&s_acc[aofs + a * TILE_N + l], (double)c[a]);                               // ds_add_f64
                }
            }
        }
    }
}
template <bool QUAD>
__device__ __forceinline__ void p2g_scatter_tile_split(const SimP& S, P2GPrep& q, bool in_tile, int lb, int aofs, const LaneSplit& ls) {
    if (ls.G == 1) p2g_scatter_tile<QUAD, 1>(S, q, in_tile, lb, aofs, 0);      // (wave-uniform choice between three loops, no branch per node)
    else if (ls.G == 3) p2g_scatter_tile<QUAD, 3>(S, q, in_tile, lb, aofs, ls.gofs);
    else p2g_scatter_tile<QUAD, 9>(S, q, in_tile, lb, aofs, ls.gofs);
}

// The gather of g2p (mpm:400-417) for one particle: the new velocity and C from its 27 nodes.  TILE: v_out staged in LDS (3 planes from word
// `tofs`) -- of s_gtile, or (ALIAS: k_g2p_p2g) of the bytes the unit's scatter tile is about to occupy.
// ROLLED: the nine columns one after the other (k_g2p_p2g's path for drifted particles: unrolled, the 27 float4 loads of the global form are asked for
// together -- 108 registers in a kernel that has none to spare; the sums are the same sums in the same order)
// The global form (drifted particles, tail units) with every product and sum spelled out and kept as written (contract off): it is inlined unrolled into
// k_g2p and ROLLED into k_g2p_p2g, and left to itself the compiler fuses multiplies into adds differently in the two -- nothing for a particle among
// others, but an isolated droplet's C' is the rounding residue of M - fx v' times 4 / dx (1e-3 at 64^3), an SVD material's adjoint amplifies it, and the
// fused launch must not change what a trajectory computes.  fma(0, T, M) = M and fma(1, T, M) = M + T exactly: rolled and unrolled give the same bits.
// (UNR: columns in flight at a time on the ROLLED road.  One column at a time a wave with a drifted particle spends nine dependent round trips in here; three at a
//  time -- nine loads in flight -- was measured in round 6 and is no faster anywhere, k_g2p_p2g +0.2 us: profiles/r06_ab_slow_paths.txt.  A/B builds: -DFE_SLOW_UNR=3)
#ifndef FE_SLOW_UNR
#define FE_SLOW_UNR 1
#endif
template <bool ROLLED, int UNR = 1>
__device__ __forceinline__ void g2p_gather_global(const SimP& S, const Stencil& st, const float4* __restrict__ g_out, float nv[3], m3& nC) {
#pragma clang fp contract(off)
    nv[0] = nv[1] = nv[2] = 0.f;
    const float c4 = 4.f * S.inv_dx;
    m3 M = m3_zero();
#pragma unroll ROLLED ? UNR : 9
    for (int ij = 0; ij < 9; ij++) {
        const int i = ij / 3, j = ij - 3 * i;
        const float wij = STW(st, i, 0) * STW(st, j, 1);
        float T[3], Tz[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float weight = wij * st.w[kk][2];
            const float4 gv = g_out[cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + kk, S.nb)];
            const float gw[3] = {weight * gv.x, weight * gv.y, weight * gv.z};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (kk == 0) T[a] = gw[a];
                else { T[a] = T[a] + gw[a]; Tz[a] = kk == 1 ? gw[a] : __builtin_fmaf(2.f, gw[a], Tz[a]); }
            }
        }
        const float fi = (float)i, fj = (float)j;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            nv[a] = nv[a] + T[a];
            M.a[a][0] = __builtin_fmaf(fi, T[a], M.a[a][0]); M.a[a][1] = __builtin_fmaf(fj, T[a], M.a[a][1]); M.a[a][2] = M.a[a][2] + Tz[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) nC.a[a][b] = c4 * __builtin_fmaf(-st.fx[b], nv[a], M.a[a][b]);
}
template <bool TILE, bool ALIAS, bool ROLLED = false>
__device__ __forceinline__ void g2p_gather(const SimP& S, int lb, const Stencil& st, const float4* __restrict__ g_out, int tofs, float nv[3], m3& nC) {
    if (!TILE) { g2p_gather_global<true, ALIAS ? FE_SLOW_UNR : 1>(S, st, g_out, nv, nC); return; }      // (k_g2p_p2g: three columns in flight; k_g2p keeps its 64 registers)      // (rolled everywhere: unrolled, its 27 float4 loads in flight took k_g2p from 80 registers to 123 -- the path of a few drifted particles)
    const float* gt = ALIAS ? (const float*)s_acc : s_gtile;
    nv[0] = nv[1] = nv[2] = 0.f;
    const float c4 = 4.f * S.inv_dx;
    // new_C[a][b] = c4 sum W g[a] (o_b - fx_b) = c4 (M[a][b] - fx_b new_v[a]) with M[a][b] = sum W g[a] o_b; o_0 = i and o_1 = j are
    // constant over the inner k loop, so the per-node work is the three products W g[a] and two partial sums.
    m3 M = m3_zero();
#pragma unroll ROLLED ? 1 : 9
    for (int ij = 0; ij < 9; ij++) {
        const int i = ij / 3, j = ij - 3 * i;
        const float wij = STW(st, i, 0) * STW(st, j, 1);
        float T[3] = {0.f, 0.f, 0.f}, Tz[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float weight = wij * st.w[kk][2];
            float g0, g1, g2;
            if (TILE) {
                const int l = tofs + lb + (i * TILE_T + j) * TILE_T + kk;
                g0 = gt[l]; g1 = gt[TILE_N + l]; g2 = gt[2 * TILE_N + l];
            } else {
                float4 gv = g_out[cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + kk, S.nb)];
                g0 = gv.x; g1 = gv.y; g2 = gv.z;
            }
            const float gw[3] = {weight * g0, weight * g1, weight * g2};
#pragma unroll
            for (int a = 0; a < 3; a++) { T[a] += gw[a]; if (kk > 0) Tz[a] += (float)kk * gw[a]; }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            nv[a] += T[a];
            M.a[a][0] += (float)i * T[a]; M.a[a][1] += (float)j * T[a]; M.a[a][2] += Tz[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) nC.a[a][b] = c4 * (M.a[a][b] - st.fx[b] * nv[a]);
}
// The nodes of a tile that one thread loads: l = t0, t0 + nth, ... -- 2 of them when the whole workgroup loads the tile (nth = 256), 4 for a half, 8 for a quad
// unit's wave.  ALL of them are asked for before the first is waited for (round 6).  Written as `for (l = t0; l < TILE_N; l += nth) { if (in the grid) v = src[..];
// lds[l] = v; }` -- the form every tile load had until then -- the stride is a run-time value, the loop stays rolled, and each round is load -> s_waitcnt vmcnt(0)
// -> ds_write: 2 / 4 / 8 DEPENDENT memory round trips of ~1-2 us each at the head of every unit (the in-kernel timeline showed the units of two small blocks, two
// half tiles, spending 11 us between their unit record and their first arithmetic where the shared tiles took 4.8: profiles/r06_timeline_tail.txt).
// `ld(l)` returns the node's value without branching (clamped address + select); `st(l, v)` writes it to LDS.
template <int NIT, typename LD, typename ST>
__device__ __forceinline__ void tile_nodes_n(int t0, int nth, LD&& ld, ST&& st) {
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) v[it] = ld(t0 + it * nth);
#pragma unroll
    for (int it = 0; it < NIT; it++) st(t0 + it * nth, v[it]);
}
// (MAXIT: at most that many in flight at a time -- where the registers are taken, k_pgg_g2pg's second tile, a thread's nodes come in rounds of MAXIT)
template <bool QUADS = true, int MAXIT = 8, typename LD, typename ST>
__device__ __forceinline__ void tile_nodes(int t0, int nth, LD&& ld, ST&& st) {      // (nth: 256, 128 or -- QUADS: the kernel walks a list with quad units -- 64; uniform; t0 < nth)
    if (nth == WG) tile_nodes_n<2>(t0, nth, ld, st);
    else if (!QUADS || nth == HALF) {
        if constexpr (MAXIT >= 4) tile_nodes_n<4>(t0, nth, ld, st);
        else { tile_nodes_n<2>(t0, nth, ld, st); tile_nodes_n<2>(t0 + 2 * nth, nth, ld, st); }
    } else {
        if constexpr (MAXIT >= 8) tile_nodes_n<8>(t0, nth, ld, st);
        else if constexpr (MAXIT >= 4) { tile_nodes_n<4>(t0, nth, ld, st); tile_nodes_n<4>(t0 + 4 * nth, nth, ld, st); }
        else { for (int r = 0; r < 4; r++) tile_nodes_n<2>(t0 + 2 * r * nth, nth, ld, st); }
    }
}
// node l of the tile at `to`, read from a grid of float4 per node (zero outside the grid)
__device__ __forceinline__ float4 tile_node_load(const TileO& to, const SimP& S, const float4* __restrict__ src, int l) {
    int i, j, k;
    const bool ok = tile_node(to, l, S.n, i, j, k);
    const float4 v = src[ok ? cell_addr(i, j, k, S.nb) : 0];
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}
// (ALIAS: into the bytes of the unit's scatter tile, from word `tofs` -- k_g2p_p2g)
template <bool ALIAS = false>
__device__ __forceinline__ void load_tile3(const TileO& to, const SimP& S, const float4* __restrict__ src, const PairCtx& pc, int tofs_alias = 0) {
    if (!pc.live) return;
    float* gt = ALIAS ? (float*)s_acc : s_gtile;
    const int tofs = ALIAS ? tofs_alias : pc.ti * 3 * TILE_N;
    tile_nodes<ALIAS>(pc.t0, pc.nth, [&](int l) { return tile_node_load(to, S, src, l); },      // (k_g2p walks the pairs-only list)
                      [&](int l, const float4 v) { gt[tofs + l] = v.x; gt[tofs + TILE_N + l] = v.y; gt[tofs + 2 * TILE_N + l] = v.z; });
}

// p2g (mpm:331-378) fused with compute_F_tmp + svd, advect_used + process_unused_particles, Injector.act and,
// on one thread, Effector.move_kernel.  WRITE=false is the backward pass' recompute of grid[f]: scatter only.
// fiso: bit 0 = frame f's F is stored compactly, bit 1 = frame f + 1 is to be (FrameV::iso; the SVD-free kernels only)
// FUSED (k_g2p_p2g, option "fuse_g2p"): the unit first does the g2p of substep f - 1 for its particles -- v_out of that substep gathered into the bytes
// its scatter tile is about to occupy, x' v' C' written to frame f and kept in registers -- and goes on with them: frame f's 60 bytes per particle are
// not read back, and a substep is two launches instead of three.  FU names what the gather needs.  The host fuses where nothing comes between the two:
// no sort at f, no mesh effector acting on particles, no rigid bodies (substep_fwd).
struct FuseP { float* fr_prev; const float4* g_out; int* slow; int* blk_count_prev; };
// FG (option "fuse_grid"): grid_op of substep f rides on this launch (the fused grid pass above): tiles arrive at their neighbour entries, and every wave
// ends with the entries it owns -- no k_grid launch follows.  G.g_out = where v_out of THIS substep goes (FU.g_out: the substep before's, the other buffer).
template <bool WRITE, bool GENERAL, bool FUSED = false, bool FG = false>
__device__ __forceinline__ void p2g_body(SimP S, float* fr_cur, float* fr_next, TableP T,
                                            const int* __restrict__ pool_idx, GridW G, AgentP agent, InjectP inj, int act, int f,
                                            GridStore GS, int fiso, FuseP FU = FuseP{nullptr, nullptr, nullptr, nullptr}) {
    if (!WRITE && GS.cap > 0 && GS.flag[f]) return;      // backward: grid[f] was stored by the forward pass
    const int tid = threadIdx.x;
    if (WRITE && blockIdx.x == 0 && tid == 0 && act) {
        for (int i = 0; i < agent.n; i++) effector_move(agent.e[i], f);
    }
    if (FG) {
        if (blockIdx.x == 0 && tid == 0 && GS.cap > 0) __hip_atomic_store(GS.flag + f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (fg_final takes it back when a block outside the active list got something)
        fg_begin(G.fg);
    }
    if (FUSED && blockIdx.x == 0 && tid == 0) *FU.blk_count_prev = 0;      // (g2p_body: grid_op of substep f - 1 was the last reader of its list; this substep's has a counter of its own)
    FrameV cur = frame_view(fr_cur, S.Np, 0, (!GENERAL && (fiso & 1)) ? 1 : 0);
    FrameV curw = frame_view(fr_cur, S.Np, 0);               // (FUSED: the g2p part's stores; plain ones, as k_g2p's are by default -- option write_through)
    FrameV prev = frame_view(FUSED ? FU.fr_prev : fr_cur, S.Np);
    FrameV nxt = frame_view(fr_next, S.Np, S.wt & 1, (!GENERAL && (fiso & 2)) ? 1 : 0);
    const int slab_wt = FG ? 1 : (S.wt & 1);                  // (FG: the owners of the neighbour entries read the slab in this very launch -- through to memory)
    TL(S, 0);
    Unit un = unit_load(T, blockIdx.x);                       // this workgroup's first unit, asked for together with meta
    const int n_slots = T.meta[5];
    bool prev_quad = false;                                   // (unit_enter)
    // (FG: every unit's record is asked for at the head of its own round, as in k_pgg_g2pg -- with the owners' part behind the loop the allocator
    //  kept the record of the first unit, twelve words, in vector registers across the whole loop: spilled and reloaded per unit)
    for (int wg = blockIdx.x; FG || wg < n_slots; wg += gridDim.x) {
        if (FG || wg != (int)blockIdx.x) un = unit_load(T, wg);
        if (FG && wg >= n_slots) break;
        if (un.a.z == -2) continue;
        if (un.a.z >= 0) {
            const PairCtx pc = unit_ctx(un);
            unit_enter(pc.quad, prev_quad);
            const int4 it = pc.it;
            const TileO to = tile_origin(it.x);
            const int aofs = pc.ti * 4 * TILE_N;                 // (doubles of a pair's tile, words of a quad's)
            int fg_t = 0, fg_d = 0;                              // FG: the regions this wave's particles deposit into (uniform) / this lane left something in the slow-path accumulator of
            TL(S, 1);
            const int nbr_entry = neighbour_entry(T.blk_slot, S.nb, it.x);      // (in flight together with the particle loads)
            auto zero_tile = [&]() {
                if (pc.quad) { if (pc.live) for (int l = pc.t0; l < TILE_N; l += 64) ((int4*)((int*)s_acc + aofs))[l] = make_int4(0, 0, 0, 0); }      // (16 bytes per lane and store)
                else if (pc.live) for (int l = pc.t0; l < 4 * TILE_N; l += pc.nth) s_acc[aofs + l] = 0.0;
            };
            if (!FUSED) { zero_tile(); unit_sync(pc.quad); }
            FixScale fs_p = {1.f, 1.f}, fs_m = {1.f, 1.f};
            bool wshell = false;                                 // (quad units) some particle of the wave reaches its tile's outer shell
            {                                                    // one pass: an item is <= 128 particles, one per lane of the half
                // (a wave with few particles gives each of them three or nine lanes: lane_split.  Not while a collector takes particles
                //  out of the frame: that decision has side effects and belongs to one lane)
                const int wbase = pc.quad ? 0 : (pc.i & 64);         // this wave's first particle within the item
                const int cnt = __builtin_amdgcn_readfirstlane(min(64, max(0, it.z - wbase)));
                const LaneSplit ls = lane_split(cnt, (S.lsplit & 1) != 0 && !(WRITE && !FUSED && act && agent.collector));      // (FUSED: never with a collector, fusable_fwd)
                const int i = wbase + ls.p, s = it.y + i;
                const bool has = ls.ok && i < it.z;
                // (`used` is re-read rather than implied by the work list so host edits of a frame cannot desynchronise it)
                // The state is asked for together with the flag (a slot of an item is valid memory either way): one round trip
                // where `used` -> state were two.
                const int s_ld = has ? s : it.y;
                const int uflag = cur.used[s_ld];
                P2GRaw raw;
                if (FUSED) {
                    // the g2p of substep f - 1 (g2p_body / slot_g2p) for this lane's particle; a split wave's lanes all gather their particle's 27 nodes
                    const int gofs_w = pc.quad ? aofs : 2 * aofs;          // the tile's first word (a pair's tile is counted in doubles)
                    const int uprev = prev.used[s_ld];
                    const float4 a0 = prev.A0[s_ld];
                    load_F(cur, s_ld, raw.p.F);
                    raw.info = load_info(S, T.info, s_ld);
                    load_tile3<true>(to, S, FU.g_out, pc, gofs_w);
                    unit_sync(pc.quad);
                    const float xp[3] = {a0.x, a0.y, a0.z};
                    Stencil stp;
                    stencil_make(xp, S.inv_dx, stp);
                    if (has && uprev != 0 && stencil_inside(stp, S.n)) {      // (outside: counted in err by the p2g of substep f - 1, frame f keeps what it held)
                        const int lbp = tile_base(to, stp);
                        if (lbp >= 0) g2p_gather<true, true>(S, lbp, stp, FU.g_out, gofs_w, raw.p.v, raw.p.C);
                        else { if (ls.primary) atomicAdd(FU.slow, 1); g2p_gather<false, true, true>(S, 0, stp, FU.g_out, 0, raw.p.v, raw.p.C); }
#pragma unroll
                        for (int a = 0; a < 3; a++) raw.p.x[a] = xp[a] + S.dt * raw.p.v[a];
                        if (ls.primary) store_xvC(curw, s, raw.p.x, raw.p.v, raw.p.C);
                    } else load_xvC(cur, s_ld, raw.p);                        // an unused slot, or a particle that entered in substep f - 1 (Injector.act): frame f holds its state
                    unit_sync(pc.quad);                                       // every gather of the tile is through: its bytes become the scatter tile
                    zero_tile();
                    unit_sync(pc.quad);
                } else p2g_load(S, cur, s_ld, T.info, raw);
                bool used = has && uflag != 0;
                bool taken = false;
                if (WRITE && !FUSED && used && act && agent.collector) { taken = collector_takes(S, cur, nxt, s, T.info, agent); used = !taken; }
                P2GPrep q;
                q.inside = false;
                int lb = -1;
                if (used) {
                    p2g_compute<WRITE, GENERAL>(S, nxt, s, raw, G, q, ls.primary);
                    if (q.inside) lb = tile_base(to, q.st);
                } else {
                    q.m = 0.f; q.affine = m3_zero(); q.mv[0] = q.mv[1] = q.mv[2] = 0.f;
                    float zero[3] = {0.f, 0.f, 0.f};
                    stencil_make(zero, S.inv_dx, q.st);
                }
                TL(S, 2);
#ifdef FE_DUMMY_VALU       // (sensitivity probe, scripts/build_variant.sh -DFE_DUMMY_VALU=N: N more VALU instructions per wave and unit -- what does a tenth more arithmetic cost?)
                {
                    float dmy = q.m;
#pragma unroll
                    for (int i_ = 0; i_ < FE_DUMMY_VALU; i_++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dmy));
                    asm volatile("" :: "v"(dmy));
                }
#endif
                const bool in_tile = lb >= 0;
                if (used && q.inside && !in_tile && ls.primary) { atomicAdd(G.slow, 1); p2g_scatter_global<FG>(S, q, G, GS, T.blk_slot, it.x, fg_d); }   // drifted out of the tile (ahead of the tile path, which may pass q on to another lane: wave_sort)
                // a wave without any particle skips the 27-node scan altogether; the branch is wave-uniform, as the DPP scan requires
                if (__any(in_tile)) {
                    fg_t = touch_regions<FG>(in_tile ? stencil_regions(lb) : 0, nbr_entry, GS);
                    if (pc.quad) {
                        wshell = __any(in_tile && stencil_on_shell(lb));
                        float bp = 0.f;
#pragma unroll
                        for (int a = 0; a < 3; a++) bp = fmaxf(bp, fabsf(q.mv[a]) + 2.f * S.dx * (fabsf(q.affine.a[a][0]) + fabsf(q.affine.a[a][1]) + fabsf(q.affine.a[a][2])));
                        fs_p = fix_scale(wave_max(in_tile ? bp : 0.f));
                        fs_m = fix_scale(wave_max(in_tile ? q.m : 0.f));
                        if (__any(in_tile && !(bp <= 3.4e38f))) fs_p.inv = __int_as_float(0x7fc00000);   // a NaN / Inf state: fmaxf and the int conversion would swallow it -- the tile is handed over as NaN (fix_nonfinite)
                        const float sp = in_tile ? fs_p.s : 1.f;          // (in place -- the registers are all taken --, and only where the tile path uses it)
                        q.m *= in_tile ? fs_m.s : 1.f;
#pragma unroll
                        for (int a = 0; a < 3; a++) { q.mv[a] *= sp; q.affine.a[a][0] *= sp; q.affine.a[a][1] *= sp; q.affine.a[a][2] *= sp; }
                        p2g_scatter_tile_split<true>(S, q, in_tile, in_tile ? lb : 0, aofs, ls);
                    } else p2g_scatter_tile_split<false>(S, q, in_tile, in_tile ? lb : 0, aofs, ls);
                }
                if (has && !used && !taken && WRITE && ls.primary) unused_particle_fwd(S, cur, nxt, s, T.pid_of_slot[s], pool_idx, agent, inj, f);
            }
            TL(S, 4);
            unit_sync(pc.quad);
            TL(S, 5);
            // hand the tile over: plain coalesced float4 stores into the item's slab.  No atomics, no waiting:
            // k_grid sums, per node, the slabs of the (at most 8) blocks whose tiles reach it, in a fixed order.
            if (pc.quad) {
                if (pc.live) quad_handover<4, FG>(S, (const int*)s_acc + aofs, fs_p.inv, fs_m.inv, G.slab, pc.slab, G.g_in, GS, to, nbr_entry, wshell, slab_wt, fg_d);
            } else if (pc.live) for (int l = pc.t0; l < TILE_N; l += pc.nth)
                tile_handover<4, false, FG>(S, G.slab, pc.slab, G.g_in, GS, to, nbr_entry, l,
                                 make_float4((float)s_acc[aofs + l], (float)s_acc[aofs + TILE_N + l], (float)s_acc[aofs + 2 * TILE_N + l], (float)s_acc[aofs + 3 * TILE_N + l]), slab_wt, fg_d);
            if (FG) fg_unit_end(T, pc, nbr_entry, fg_t, fg_d);      // (with the unit's closing unit_sync inside)
            else unit_sync(pc.quad);
            TL(S, 6);
        } else {
            const int s = un.a.y + tid;
            if (s < S.N) {
                P2GRaw raw;
                bool have = false;                                       // (FUSED) x v C of frame f are in raw
                if (FUSED && prev.used[s]) {
                    const float4 a0 = prev.A0[s];
                    const float xp[3] = {a0.x, a0.y, a0.z};
                    Stencil stp;
                    stencil_make(xp, S.inv_dx, stp);
                    if (stencil_inside(stp, S.n)) {
                        g2p_gather<false, true, true>(S, 0, stp, FU.g_out, 0, raw.p.v, raw.p.C);
#pragma unroll
                        for (int a = 0; a < 3; a++) raw.p.x[a] = xp[a] + S.dt * raw.p.v[a];
                        store_xvC(curw, s, raw.p.x, raw.p.v, raw.p.C);
                        have = true;
                    }
                }
                if (cur.used[s]) {
                    if (WRITE && !FUSED && act && agent.collector && collector_takes(S, cur, nxt, s, T.info, agent)) continue;
                    P2GPrep q;
                    if (FUSED && have) { load_F(cur, s, raw.p.F); raw.info = load_info(S, T.info, s); p2g_compute<WRITE, GENERAL>(S, nxt, s, raw, G, q); }
                    else p2g_prepare<WRITE, GENERAL>(S, cur, nxt, s, T.info, G, q);
                    int none = 0;
                    if (q.inside) p2g_scatter_global<FG>(S, q, G, GS, T.blk_slot, -1, none);       // (FG: a tail unit has no tile and no arrival -- whatever it deposits into an active entry is late)
                } else if (WRITE) unused_particle_fwd(S, cur, nxt, s, T.pid_of_slot[s], pool_idx, agent, inj, f);
            }
        }
    }
    // (FG: the owners' part follows in the kernel, k_p2g_fg / k_g2p_p2g_fg, with the arguments read afresh)
}
struct P2GArgs { SimP S; float* fr_cur; float* fr_next; TableP T; const int* pool_idx; GridW G; AgentP agent; InjectP inj; int act; int f; GridStore GS; int fiso; FuseP FU; };      // (FU: k_g2p_p2g_b)
template <bool WRITE, bool GENERAL>
__global__ FE_KALIGN __launch_bounds__(WG, GENERAL ? 3 : 4) void k_p2g(SimP S, float* fr_cur, float* fr_next, TableP T, const int* pool_idx, GridW G, AgentP agent, InjectP inj, int act, int f, GridStore GS, int fiso) { p2g_body<WRITE, GENERAL>(S, fr_cur, fr_next, T, pool_idx, G, agent, inj, act, f, GS, fiso); }
// substep f's p2g with the g2p of substep f - 1 in front of it (p2g_body, FUSED)
template <bool GENERAL>
__global__ FE_KALIGN __launch_bounds__(WG, GENERAL ? 3 : 4) void k_g2p_p2g(SimP S, float* fr_cur, float* fr_next, TableP T, const int* pool_idx, GridW G, AgentP agent, InjectP inj, int act, int f, GridStore GS, int fiso, FuseP FU) { p2g_body<true, GENERAL, true>(S, fr_cur, fr_next, T, pool_idx, G, agent, inj, act, f, GS, fiso, FU); }
// ... and with grid_op of substep f behind it (p2g_body, FG): the launches of a forward substep under option "fuse_grid"
// The arguments are ONE struct, and the owners' part reads what it needs of them from the kernel-argument segment again, through a pointer the optimiser cannot see
// through: handed on as values, S / T / GS / G stayed live -- in scalar registers, i.e. spilled to vector lanes and from there to scratch -- across the unit loop
// (first build: 279 spilled scalar registers and 180 bytes of scratch in k_p2g_fg).
template <typename ARGS>
__device__ __forceinline__ const ARGS* fg_args_again() {
    const ARGS* p = (const ARGS*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <bool GENERAL>
__global__ FE_KALIGN __launch_bounds__(WG, GENERAL ? 3 : 4) void k_p2g_fg(P2GArgs A) {
    p2g_body<true, GENERAL, false, true>(A.S, A.fr_cur, A.fr_next, A.T, A.pool_idx, A.G, A.agent, A.inj, A.act, A.f, A.GS, A.fiso);
    const P2GArgs* B = fg_args_again<P2GArgs>();
    TL(B->S, 3);
    fg_owner_phase<false>(B->S, B->T, B->GS, FgArgs{B->G.fg, B->G.slab, B->G.g_in, B->G.g_out, B->G.blk_list, B->G.blk_count, B->G.blk_flag, B->f});
}
template <bool GENERAL>
__global__ FE_KALIGN __launch_bounds__(WG, GENERAL ? 3 : 4) void k_g2p_p2g_fg(P2GArgs A) {
    p2g_body<true, GENERAL, true, true>(A.S, A.fr_cur, A.fr_next, A.T, A.pool_idx, A.G, A.agent, A.inj, A.act, A.f, A.GS, A.fiso, A.FU);
    const P2GArgs* B = fg_args_again<P2GArgs>();
    TL(B->S, 3);
    fg_owner_phase<false>(B->S, B->T, B->GS, FgArgs{B->G.fg, B->G.slab, B->G.g_in, B->G.g_out, B->G.blk_list, B->G.blk_count, B->G.blk_flag, B->f});
}
template <bool GENERAL>
__global__ FE_KALIGN __launch_bounds__(WG, GENERAL ? 3 : 4) void k_g2p_p2g_b(Batch<P2GArgs> B) { const P2GArgs& A = B.a[blockIdx.y]; p2g_body<true, GENERAL, true>(A.S, A.fr_cur, A.fr_next, A.T, A.pool_idx, A.G, A.agent, A.inj, A.act, A.f, A.GS, A.fiso, A.FU); }
template <bool WRITE, bool GENERAL>
__global__ FE_KALIGN __launch_bounds__(WG, GENERAL ? 3 : 4) void k_p2g_b(Batch<P2GArgs> B) { const P2GArgs& A = B.a[blockIdx.y]; p2g_body<WRITE, GENERAL>(A.S, A.fr_cur, A.fr_next, A.T, A.pool_idx, A.G, A.agent, A.inj, A.act, A.f, A.GS, A.fiso); }


// agent.collide at particle level (mpm:418-422; AgentRigid.collide): every effector that carries a mesh, in order,
// x_tmp = x + dt * new_v re-formed before each collider.  NODE: the same chain at a grid node (mpm:393-395,
// Agent.collide_type 'grid' / 'both'), where the position is the node's and does not move with the velocity.
template <bool NODE>
__device__ __forceinline__ bool agent_collide_particle(const SimP& S, const AgentP& agent, int f, const float x[3], float nv[3]) {
    const float sdt = NODE ? 0.f : S.dt;
    bool any = false;
    for (int ei = 0; ei < agent.n; ei++) {
        const EffP& e = agent.e[ei];
        if (!e.has_mesh) continue;
        const float pos[3] = {x[0] + sdt * nv[0], x[1] + sdt * nv[1], x[2] + sdt * nv[2]};
        if (!(pos[1] > agent.collide_min_y)) continue;                           // agent_icecreamdynamic.py:39-43
        float out[3];
        any |= t_dynamic_collide<float>(e.mesh, e.pos + f * 3, e.quat + f * 4, e.pos + (f + 1) * 3, e.quat + (f + 1) * 4, pos, nv, S.dt, out);
        nv[0] = out[0]; nv[1] = out[1]; nv[2] = out[2];
    }
    return any;
}
// velocity of one node after gravity and the domain boundary (mpm:383-398); k[] = boundary multipliers
#define FE_MAX_STATICS 4
struct NodeWork { int c, ni, nj, nk; float4 gi; float go[3]; float pad; };      // a grid node whose collide adjoint is finished by k_grid_collide_grad
struct StaticsP { int n; const SdfP* s; };          // the scene's static SDF colliders (statics.py), parameter blocks in device memory

// STATICS=false keeps the collider-free kernels exactly as lean as before (the SDF code costs ~100 VGPRs)
// DYN: the agent's moving colliders act at the nodes too (mpm:393-395); vdyn = the velocity before them (adjoint input)
template <bool STATICS, bool DYN = false>
__device__ __forceinline__ void node_velocity(const SimP& S, const StaticsP& ST, const float4 gi, int i, int j, int k, float vo[3], float kmul[3],
                                              float (*trace)[3] = nullptr, const AgentP* agent = nullptr, int f = 0, float* vdyn = nullptr,
                                              bool* dyn_hit = nullptr) {
    float inv = 1.f / gi.w;
    vo[0] = inv * gi.x + S.dt * S.g[0];
    vo[1] = inv * gi.y + S.dt * S.g[1];
    vo[2] = inv * gi.z + S.dt * S.g[2];
    float xn[3] = {(float)i * S.dx, (float)j * S.dx, (float)k * S.dx};
    if (STATICS) {
#pragma unroll
        for (int si = 0; si < FE_MAX_STATICS; si++) {                       // collide with statics, mpm:386-390
            if (si < ST.n) {
                if (trace) { trace[si][0] = vo[0]; trace[si][1] = vo[1]; trace[si][2] = vo[2]; }
                static_collide(ST.s[si], xn, vo, nullptr);
            }
        }
    }
    if (DYN) {
        if (vdyn) { vdyn[0] = vo[0]; vdyn[1] = vo[1]; vdyn[2] = vo[2]; }
        const bool hit = agent_collide_particle<true>(S, *agent, f, xn, vo);
        if (dyn_hit) *dyn_hit = hit;
    }
    boundary_v(S.bnd, xn, vo, kmul);
}
// adjoint of the collider chain for one node: g = d/d(v before the boundary) -> d/d(v after gravity)
__device__ __forceinline__ void node_statics_grad(const SimP& S, const StaticsP& ST, int i, int j, int k, const float (*trace)[3], float g[3]) {
    float xn[3] = {(float)i * S.dx, (float)j * S.dx, (float)k * S.dx};
#pragma unroll
    for (int si = FE_MAX_STATICS - 1; si >= 0; si--) {
        if (si < ST.n) {
            float vin[3] = {trace[si][0], trace[si][1], trace[si][2]};
            static_collide(ST.s[si], xn, vin, g);
        }
    }
}

// Sum, for node (oi,oj,ok) of block b, what the work items' slabs hold for it.  The slab of an item of block B' covers the nodes
// [4B', 4B'+6) per axis (tile indices 1..6), so a node with in-block offset o receives from B'=B (slab index o) and, when o <= 1,
// from B'=B-1 (slab index o+4): 1..8 source blocks per node (3.4 on average), visited in a fixed order (deterministic sums).  What
// a tile collected on its outer shell arrives through the slow-path accumulator instead (tile_handover).  The item ranges of the
// 27 neighbour blocks are fetched once per wave (lane n < 27 loads neighbour n) and handed out by cross-lane reads, so a
// lane's critical path is two dependent memory round trips: range -> slab values.
__device__ __forceinline__ int2 nbr_record(const TableP& T, int e, int lane) {      // (laid out by k_build_units: no detour over the block number)
    return lane < 27 ? T.nbr[e * 27 + lane] : make_int2(0, 0);
}
template <int NPL>
__device__ __forceinline__ float4 gather_slabs(const float4* __restrict__ slab, const int2 mine, int lane) {
    const int o[3] = {lane >> 4, (lane >> 2) & 3, lane & 3};
    // Which items of a block own a slab: the items of a block pair up from its first one, and a pair shares the first one's slab:
    // the slabs of a block whose items are [first, first + count) are first, first + 2, ...  The first two of each of the 8 source
    // blocks are loaded unconditionally-shaped (16 independent loads in flight, zero when absent): summing inside a loop made
    // every load wait for the previous one, eight dependent L2/MALL round trips per node.  Blocks with more than two slabs
    // (> 512 particles) finish in the loop below.  The summation order stays fixed (source c ascending, slab ascending).
    int a0[8], n[8];                         // node's element in the source block's first slab; that block's item count (0: none reaches this node)
    float4 v0[8], v1[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        int nbr = 0, si = 0;                        // neighbour code (di+1)*9 + (dj+1)*3 + (dk+1), slab node index
        bool valid = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const bool other = (c >> d) & 1;        // the lower neighbour along d instead of the block itself
            valid = valid && (!other || o[d] <= 1);
            nbr = nbr * 3 + (other ? 0 : 1);
            si = si * SLAB_T + o[d] + (other ? 4 : 0);
        }
        // (both shuffles by ALL lanes, whatever `valid` says: a cross-lane read of a lane that sits out returns 0)
        const int f = __shfl(mine.x, nbr, 64), k = __shfl(mine.y, nbr, 64);
        n[c] = valid ? k : 0;
        a0[c] = valid ? (NPL == 4 ? f * SLAB_N + si : f * (SLAB_N * 16) + si * 12) : 0;      // (element index / byte offset of a packed node)
    }
    // (the loads are unconditional from a clamped, always valid index and masked afterwards: a conditional float4
    // load into an array element ended up in scratch; lanes without a source all read node 0 of slab 0, one broadcast line.
    // Round 4: skipping the source blocks without items by wave-uniform branches around their loads -- where the water has come apart most
    // of the eight are empty -- was slower in every phase, +1.7 us in k_grid and +1.2 in k_grid_grad: profiles/r04_ab_grid_gather_uniform_branches.txt)
    const __amdgpu_buffer_rsrc_t rs = wt_rsrc((void*)slab);
    auto ld = [&](int at) -> float4 {
        if (NPL == 4) return slab[at];
        const u32x3 u = __builtin_amdgcn_raw_buffer_load_b96(rs, at, 0, 0);
        return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), 0.f);
    };
    constexpr int STRIDE = NPL == 4 ? SLAB_N : SLAB_N * 16;       // from one slab to the next, in the units of a0
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const bool h0 = n[c] > 0, h1 = n[c] > 2;
        const float4 a = ld(h0 ? a0[c] : 0);
        const float4 b = ld(h1 ? a0[c] + 2 * STRIDE : 0);
        v0[c] = make_float4(h0 ? a.x : 0.f, h0 ? a.y : 0.f, h0 ? a.z : 0.f, h0 ? a.w : 0.f);
        v1[c] = make_float4(h1 ? b.x : 0.f, h1 ? b.y : 0.f, h1 ? b.z : 0.f, h1 ? b.w : 0.f);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        acc.x += v0[c].x; acc.y += v0[c].y; acc.z += v0[c].z; acc.w += v0[c].w;
        acc.x += v1[c].x; acc.y += v1[c].y; acc.z += v1[c].z; acc.w += v1[c].w;
        for (int k = 4; k < n[c]; k += 2) {
            const float4 v = ld(a0[c] + k * STRIDE);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    return acc;
}

// Short active lists (a block of water that has not come apart: ~1,000 entries) take a shorter road: wave w of the launch owns
// entry static_entry(w) outright, so its block number, marks and neighbour record are asked for at once, together with the
// list's length -- one round trip where the range walk below needs three (length -> marks -> neighbour record), in kernels that
// are nothing but such a chain (k_grid at C2: entry known after 1.9 us, slabs in after 4.9, done at 4.9 of 8.5).  The mapping
// keeps chunks of 64 consecutive entries on one XCD (their blocks are neighbours and share slabs).
__device__ __forceinline__ int static_entry(int wave) {
    const int G = gridDim.x;
    if ((G & 127) != 0) return blockIdx.x * 4 + wave;
    const int j = blockIdx.x >> 3;
    return ((j >> 4) * 8 + (blockIdx.x & 7)) * 64 + (j & 15) * 4 + wave;
}
// grid_op (mpm:380-398) over the 4^3 blocks the scatter reached (GridStore: the marked entries of the order's active list, then
// the dynamic list of slow-path blocks outside it); one wave per block.  STATICS: the scene has SDF colliders.
// KEEP=false (forward): also re-zeroes g_in and the dynamic block flag, so no separate reset_grid pass
// (mpm:219-223) is needed.  KEEP=true (backward recompute): stores the summed (p, m) in g_in for grid_grad.
template <bool KEEP, bool STATICS, bool DYN>
__device__ __forceinline__ void grid_body(SimP S, TableP T, const float4* __restrict__ slab, float* g_in, float4* g_out,
                                              const int* __restrict__ blk_list, const int* __restrict__ blk_count, int* blk_flag,
                                              GridStore GS, int f, int* frame_slow, StaticsP ST, AgentP agent) {
    if (KEEP && GS.cap > 0 && GS.flag[f]) return;        // backward: stored by the forward pass
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TL(S, 0);
    // this wave's entry on the short-list road, asked for before the list's length is known
    const int es = static_entry(wave);
    const bool es_ok = es < S.nb * S.nb * S.nb;
    const int blk_s = es_ok ? T.active[es] : 0;
    const unsigned char tch_s = es_ok ? GS.touched[es] : (unsigned char)0, drt_s = es_ok ? GS.dirty[es] : (unsigned char)0;
    const int2 nbr_s = es_ok ? nbr_record(T, es, lane) : make_int2(0, 0);
    const int n_static = T.meta[2], n_dyn = *blk_count;
    if (!KEEP && blockIdx.x == 0 && threadIdx.x == 0) {
        if (GS.cap > 0) GS.flag[f] = (n_static <= GS.cap && *frame_slow == 0) ? 1 : 0;
        *frame_slow = 0;
    }
    // one block: is_static = an entry of the active list (slab gather, store slot e); dirty = its g_in planes hold slow-path atomics
    auto one_block = [&](int e, int b, bool is_static, bool touched, bool dirty, const int2 nbr) {
        const int c = (b << 6) | lane;
        const int bi = b / (S.nb * S.nb), bj = (b / S.nb) % S.nb, bk = b % S.nb;
        TL(S, 1);
        float4 gi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dirty) gi = make_float4(g_in[c], g_in[S.ncell + c], g_in[2 * S.ncell + c], g_in[3 * S.ncell + c]);
        if (touched) { const float4 t = gather_slabs<4>(slab, nbr, lane); gi.x += t.x; gi.y += t.y; gi.z += t.z; gi.w += t.w; }
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gi.w > FE_EPS) TL(S, 2);
        if (gi.w > FE_EPS) {
            float vo[3], kmul[3];
            node_velocity<STATICS, DYN>(S, ST, gi, bi * 4 + (lane >> 4), bj * 4 + ((lane >> 2) & 3), bk * 4 + (lane & 3), vo, kmul, nullptr, &agent, f);
            out = make_float4(vo[0], vo[1], vo[2], 0.f);
        }
        g_out[c] = out;
        if (!KEEP) {
            if (is_static && e < GS.cap) {
                float4* dst = GS.data + ((size_t)f * GS.cap + e) * GS_BLK;
                dst[lane] = gi;
                const u32x3 u = {__float_as_uint(out.x), __float_as_uint(out.y), __float_as_uint(out.z)};
                __builtin_amdgcn_raw_buffer_store_b96(u, wt_rsrc(dst), 1024 + lane * 12, 0, 0);
            }
            if (dirty) { g_in[c] = 0.f; g_in[S.ncell + c] = 0.f; g_in[2 * S.ncell + c] = 0.f; g_in[3 * S.ncell + c] = 0.f; }
            if (lane == 0 && !is_static) blk_flag[b] = 0;
        } else {
            g_in[c] = gi.x; g_in[S.ncell + c] = gi.y; g_in[2 * S.ncell + c] = gi.z; g_in[3 * S.ncell + c] = gi.w;
        }
        TL(S, 3);
    };
    if (n_static <= 4 * (int)gridDim.x) {                   // short list: every entry has a wave of its own
        if (es < n_static) {
            const bool tm = tch_s == GS.stamp, dm = drt_s == GS.stamp;
            if (lane == 0) {                                 // this launch's entries, recorded for k_grid_grad
                if (KEEP) GS.cur[es] = (tm || dm) ? 1 : 0;
                else if (es < GS.cap) GS.live[(size_t)f * GS.cap + es] = (tm || dm) ? 1 : 0;
            }
            if (tm || dm) one_block(es, blk_s, true, tm, dm, nbr_s);      // (entries without a mark: nothing arrived, and no particle reads those nodes)
        }
    } else {
        // Longer lists: wave w takes the entries w, w + W, w + 2 W, ... (W = the launch's waves), eight at a time.  Marked entries come in
        // clusters (the list is in block order: water in one corner of the box): dealt out in contiguous ranges -- round 2 -- some workgroups
        // had all of their 28 entries to work on and most had none (the splash: 25 us for 10,000 blocks); strided, every wave gets its share.
        // Round 6: the marks, the block numbers AND the neighbour records of all eight are asked for together -- one round trip, then one per marked
        // entry for its slabs (rounds 3-5: the marks first, the first marked entry's block number and record behind them, the next one's while the wave
        // works).  Splash 21.5 -> 20.8 us, early splash 16.6 -> 16.2; the same change cost k_grid_grad 4 us (133 registers: three waves per SIMD) and was
        // not kept there: profiles/r06_ab_grid_long_list_one_trip.txt.
        const int W = 4 * (int)gridDim.x;
        for (int e0 = es; e0 < n_static; e0 += 8 * W) {
            unsigned tm = 0, dm = 0;
            int blk8[8]; int2 nbr8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = e0 + k * W;
                const bool ok = e < n_static;
                const int ec = ok ? e : e0;                   // (a valid entry either way: the loads are unconditional, the marks masked)
                const unsigned char t = GS.touched[ec], d = GS.dirty[ec];
                blk8[k] = T.active[ec]; nbr8[k] = nbr_record(T, ec, lane);
                tm |= ((ok && t == GS.stamp) ? 1u : 0u) << k; dm |= ((ok && d == GS.stamp) ? 1u : 0u) << k;
            }
            tm = __builtin_amdgcn_readfirstlane(tm); dm = __builtin_amdgcn_readfirstlane(dm);
            if (lane < 8 && e0 + lane * W < n_static) {       // this launch's entries, recorded for k_grid_grad
                const int e = e0 + lane * W;
                if (KEEP) GS.cur[e] = ((tm | dm) >> lane) & 1;
                else if (e < GS.cap) GS.live[(size_t)f * GS.cap + e] = ((tm | dm) >> lane) & 1;
            }
            unsigned todo = tm | dm;                          // (entries without a mark: nothing arrived, and no particle reads those nodes)
            while (todo) {
                const int k = __builtin_ctz(todo);
                todo &= todo - 1;
                int blk = blk8[0]; int2 nbr = nbr8[0];
#pragma unroll
                for (int j = 1; j < 8; j++) if (k == j) { blk = blk8[j]; nbr = nbr8[j]; }      // (k is uniform: scalar branches)
                one_block(e0 + k * W, __builtin_amdgcn_readfirstlane(blk), true, (tm >> k) & 1, (dm >> k) & 1, nbr);
            }
        }
    }
    for (int d = blockIdx.x * 4 + wave; d < n_dyn; d += gridDim.x * 4) one_block(-1, blk_list[d], false, false, true, make_int2(0, 0));
}
struct GridArgs { SimP S; TableP T; const float4* slab; float* g_in; float4* g_out; const int* blk_list; const int* blk_count; int* blk_flag; GridStore GS; int f; int* frame_slow; StaticsP ST; AgentP agent; };
template <bool KEEP, bool STATICS, bool DYN>
__global__ FE_KALIGN __launch_bounds__(256, (STATICS || DYN) ? 2 : 4) void k_grid(SimP S, TableP T, const float4* slab, float* g_in, float4* g_out, const int* blk_list, const int* blk_count, int* blk_flag, GridStore GS, int f, int* frame_slow, StaticsP ST, AgentP agent) { grid_body<KEEP, STATICS, DYN>(S, T, slab, g_in, g_out, blk_list, blk_count, blk_flag, GS, f, frame_slow, ST, agent); }
template <bool KEEP, bool STATICS, bool DYN>
__global__ FE_KALIGN __launch_bounds__(256, (STATICS || DYN) ? 2 : 4) void k_grid_b(Batch<GridArgs> B) { const GridArgs& A = B.a[blockIdx.y]; grid_body<KEEP, STATICS, DYN>(A.S, A.T, A.slab, A.g_in, A.g_out, A.blk_list, A.blk_count, A.blk_flag, A.GS, A.f, A.frame_slow, A.ST, A.agent); }

// -----------------------------------------------------------------------------------------
// the fused grid pass (FG kernels): what a wave does with an entry, the owners' loop, the final wave
// -----------------------------------------------------------------------------------------
__device__ __forceinline__ float4 fg_load16(const void* base, unsigned byte_off) {                 // 16 bytes past the vector L1 (sc1)
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc((void*)base), (int)byte_off, 0, 16);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// v_out of a node from its (p, m) totals -- grid_op without colliders (the FG kernels' scenes have none: fuse_grid_ok)
__device__ __forceinline__ float4 fg_node_out(const SimP& S, const float4 gi, int b, int lane, float kmul[3]) {
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    kmul[0] = kmul[1] = kmul[2] = 0.f;
    if (gi.w > FE_EPS) {
        const int bi = b / (S.nb * S.nb), bj = (b / S.nb) % S.nb, bk = b % S.nb;
        float vo[3];
        node_velocity<false, false>(S, StaticsP{0, nullptr}, gi, bi * 4 + (lane >> 4), bj * 4 + ((lane >> 2) & 3), bk * 4 + (lane & 3), vo, kmul);
        out = make_float4(vo[0], vo[1], vo[2], 0.f);
    }
    return out;
}
// what the forward pass leaves of a block: v_out for the gather, and the frame's record in the grid store ((p, m) totals, v_out packed) -- written through
__device__ __forceinline__ void fg_fwd_store(const GridStore& GS, const FgArgs& A, int e, int c, int lane, const float4 gi, const float4 out) {
    wt_store16(A.out, (unsigned)c * 16u, out);
    if (e >= 0 && e < GS.cap) {
        float4* dst = GS.data + ((size_t)A.f * GS.cap + e) * GS_BLK;
        wt_store16(dst, (unsigned)lane * 16u, gi);
        const u32x3 u = {__float_as_uint(out.x), __float_as_uint(out.y), __float_as_uint(out.z)};
        __builtin_amdgcn_raw_buffer_store_b96(u, wt_rsrc(dst), 1024 + lane * 12, 0, 16);
    }
}
// grid_op (mpm:380-398) of active-list entry e, one wave: grid_body's one_block -- the same sums in the same order
// (b: the entry's block; nbr: lane n < 27 holds the item range of neighbour n -- asked for by the caller, ahead of time where it can)
__device__ __forceinline__ void fg_fwd_block(const SimP& S, const GridStore& GS, const FgArgs& A, int e, int b, const int2 nbr, bool touched, bool dirty) {
    const int lane = threadIdx.x & 63;
    const int c = (b << 6) | lane;
    float4 gi = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a returning exchange: the read happens where the depositing atomics did, and leaves the zero the next substep expects)
    if (dirty) gi = make_float4(atomicExch(A.acc + c, 0.f), atomicExch(A.acc + S.ncell + c, 0.f), atomicExch(A.acc + 2 * S.ncell + c, 0.f), atomicExch(A.acc + 3 * S.ncell + c, 0.f));
    if (touched) { const float4 t = gather_slabs<4>(A.slab, nbr, lane); gi.x += t.x; gi.y += t.y; gi.z += t.z; gi.w += t.w; }
    float kmul[3];
    const float4 out = fg_node_out(S, gi, b, lane, kmul);
    fg_fwd_store(GS, A, e, c, lane, gi, out);
}
// grid_op.grad (mpm:539) of entry e for a frame whose grid the forward pass stored: grid_grad_body's one_block
__device__ __forceinline__ float4 fg_grad_out(const float4 gi, const float4 go, const float kmul[3]) {
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gi.w > FE_EPS) {
        const float inv = 1.f / gi.w;
        const float g0 = go.x * kmul[0], g1 = go.y * kmul[1], g2 = go.z * kmul[2];
        out.x = g0 * inv; out.y = g1 * inv; out.z = g2 * inv;
        out.w = -(gi.x * g0 + gi.y * g1 + gi.z * g2) * inv * inv;
    }
    return out;
}
__device__ __forceinline__ void fg_bwd_block(const SimP& S, const GridStore& GS, const FgArgs& A, int e, int b, const int2 nbr, bool dirty) {
    const int lane = threadIdx.x & 63;
    const int c = (b << 6) | lane;
    const float4 gi = GS.data[((size_t)A.f * GS.cap + e) * GS_BLK + lane];
    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dirty) go = make_float4(atomicExch(A.acc + c, 0.f), atomicExch(A.acc + S.ncell + c, 0.f), atomicExch(A.acc + 2 * S.ncell + c, 0.f), 0.f);
    { const float4 t = gather_slabs<3>(A.slab, nbr, lane); go.x += t.x; go.y += t.y; go.z += t.z; }
    float kmul[3];
    (void)fg_node_out(S, gi, b, lane, kmul);
    wt_store16(A.out, (unsigned)c * 16u, fg_grad_out(gi, go, kmul));
}
// (one address for the whole wave; the value into scalar registers)
__device__ __forceinline__ unsigned long long fg_word(const unsigned long long* p) {
    const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
}
__device__ __forceinline__ int fg_int(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int fg_int_u(const int* p) { return __builtin_amdgcn_readfirstlane(fg_int(p)); }      // (one address for the whole wave)
// a wave gives up on a wait that cannot end (a bug, or counters left over from an aborted launch): the error is reported by fe_sync, nothing hangs
#define FG_SPIN_MAX (1 << 22)
__device__ __forceinline__ void fg_give_up(const FgDev* F) { if ((threadIdx.x & 63) == 0) atomicAdd(F->ctr + FGC_ERR * FG_LINE, 1); }
// entry e is complete (its word `a` says what arrived): what the final wave does with a skipped one (the owners: fg_owner_phase)
template <bool BWD>
__device__ __forceinline__ void fg_entry(const SimP& S, const TableP& T, const GridStore& GS, const FgArgs& A, int e, unsigned long long a) {
    const int lane = threadIdx.x & 63;
    const bool touched = FG_TCH(a) != 0, dirty = FG_DRT(a) != 0;
    if (lane == 0) __hip_atomic_store(T.arrive + e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    const int b = T.active[e];
    const int2 nbr = nbr_record(T, e, lane);
    if (!BWD) {
        // (this launch's entries, recorded for the adjoint pass: an entry nothing arrived in has no mass, passes no adjoint on, and no particle reads it)
        if (lane == 0 && e < GS.cap) __hip_atomic_store(GS.live + (size_t)A.f * GS.cap + e, (unsigned char)((touched || dirty) ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (touched || dirty) fg_fwd_block(S, GS, A, e, b, nbr, touched, dirty);
    } else {
        if (GS.live[(size_t)A.f * GS.cap + e] != 0) fg_bwd_block(S, GS, A, e, b, nbr, dirty);
    }
}
// The launch's last workgroup has finished its unit loops -- every deposit of the launch is in memory -- and something rare was registered:
// blocks outside the active list, entries their owners could not wait for, late deposits.  One wave.
template <bool BWD>
__device__ __forceinline__ void fg_final(const SimP& S, const TableP& T, const GridStore& GS, const FgArgs& A) {
    const int lane = threadIdx.x & 63;
    const FgDev* F = A.F;
    int* const ctr = F->ctr;
    const int n_static = T.meta[2];
    const int* __restrict__ expected = (const int*)(T.arrive + S.nb * S.nb * S.nb);
    const int n_dyn = BWD ? 0 : fg_int_u(A.blk_count), n_skip = fg_int_u(ctr + FGC_SKIP * FG_LINE), n_late = fg_int_u(ctr + FGC_LATE * FG_LINE);
    if (!BWD && n_dyn > 0) {                                  // the slow path's blocks outside the order's active list: no slabs, no entry in the store
        for (int d = 0; d < n_dyn; d++) {
            const int b = fg_int_u(A.blk_list + d), c = (b << 6) | lane;         // (a block off the list only ever gets late deposits: p2g_scatter_global)
            const float4 gi = make_float4(atomicExch(F->late + c, 0.f), atomicExch(F->late + S.ncell + c, 0.f), atomicExch(F->late + 2 * S.ncell + c, 0.f), atomicExch(F->late + 3 * S.ncell + c, 0.f));
            float kmul[3];
            fg_fwd_store(GS, A, -1, c, lane, gi, fg_node_out(S, gi, b, lane, kmul));
            if (lane == 0) __hip_atomic_store(A.blk_flag + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            __hip_atomic_store(A.blk_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (GS.cap > 0) __hip_atomic_store(GS.flag + A.f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the frame's store is incomplete: the adjoint pass recomputes its grid
        }
    }
    if (n_skip > 0) {                                         // every unit loop is over: they are complete now
        for (int e0 = 0; e0 < n_static; e0 += 64) {
            const int e = e0 + lane;
            unsigned long long todo = __ballot(e < n_static && __hip_atomic_load(F->skipm + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GS.stamp);
            while (todo) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int ek = e0 + k;
                if (lane == 0) __hip_atomic_store(F->skipm + ek, (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long a = fg_word(T.arrive + ek);
                if (FG_CNT(a) != expected[ek]) fg_give_up(F);
                fg_entry<BWD>(S, T, GS, A, ek, a);
            }
        }
        if (lane == 0) __hip_atomic_store(ctr + FGC_SKIP * FG_LINE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (n_late > 0) {
        // the owners' results are what the late deposits are added to: every workgroup through with its entries first
        for (int it = 0; ; it++) {
            int v = lane < FG_SH ? fg_int(ctr + (FGC_FIN + lane) * FG_LINE) : 0;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((unsigned)__builtin_amdgcn_readfirstlane(v) - S.fg_base == gridDim.x) break;
            if (it > FG_SPIN_MAX) { fg_give_up(F); break; }
            __builtin_amdgcn_s_sleep(8);
        }
        for (int i = 0; i < n_late; i++) {
            const int e = fg_int_u(F->late_list + i);
            const int b = T.active[e], c = (b << 6) | lane;
            if (lane == 0) __hip_atomic_store(F->late_flag + e, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float4 lt = make_float4(atomicExch(F->late + c, 0.f), atomicExch(F->late + S.ncell + c, 0.f), atomicExch(F->late + 2 * S.ncell + c, 0.f), 0.f);
            float kmul[3];
            if (!BWD) {
                lt.w = atomicExch(F->late + 3 * S.ncell + c, 0.f);
                // (p, m) as the owner stored them (nothing, if nothing had arrived), plus the late part: grid_op again
                const bool live = e < GS.cap && __hip_atomic_load(GS.live + (size_t)A.f * GS.cap + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                float4 gi = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) gi = fg_load16(GS.data + ((size_t)A.f * GS.cap + e) * GS_BLK, (unsigned)lane * 16u);
                gi.x += lt.x; gi.y += lt.y; gi.z += lt.z; gi.w += lt.w;
                fg_fwd_store(GS, A, e, c, lane, gi, fg_node_out(S, gi, b, lane, kmul));
                if (lane == 0 && e < GS.cap) __hip_atomic_store(GS.live + (size_t)A.f * GS.cap + e, (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (GS.live[(size_t)A.f * GS.cap + e] != 0) {
                // grid_op's adjoint is linear in d v_out: the late part's image is added to what the owner wrote
                const float4 gi = GS.data[((size_t)A.f * GS.cap + e) * GS_BLK + lane];
                (void)fg_node_out(S, gi, b, lane, kmul);
                const float4 d = fg_grad_out(gi, lt, kmul), o = fg_load16(A.out, (unsigned)c * 16u);
                wt_store16(A.out, (unsigned)c * 16u, make_float4(o.x + d.x, o.y + d.y, o.z + d.z, o.w + d.w));
            }
        }
        if (lane == 0) __hip_atomic_store(ctr + FGC_LATE * FG_LINE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Every wave of an FG kernel, behind its unit loop (all 64 lanes; no workgroup barrier in here: a quad unit's waves come when they come).
// The wave's entries are e_k = static_entry(wave) + k W (W = the launch's waves): lane k looks after entry k -- its word, its expected count and its block
// number are asked for by 64 lanes at once, the words polled with ONE load instruction per round, and an entry nothing arrived in (most of them where the
// water has come apart) is done by its lane alone.  The others are worked on by the whole wave one after the other, in the order they complete, the next
// one's neighbour record asked for ahead; the records of the first eight are on their way before the first poll (first form: poll -> block number and
// record -> slabs -> accumulator, four dependent round trips per entry and five entries per wave in the splash: +40 us on k_g2p_p2g where k_grid took 21).
#define FG_NPF 2
template <bool BWD>
__device__ __forceinline__ void fg_owner_phase(const SimP& S, const TableP& T, const GridStore& GS, const FgArgs& A) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const FgDev* F = A.F;
    int* const ctr = F->ctr;
    const int G_ = gridDim.x, W = 4 * G_;
    const int es = static_entry(wave);
    const int n_static = T.meta[2];
    const int* __restrict__ expected = (const int*)(T.arrive + S.nb * S.nb * S.nb);
    // this lane's entry of the wave's first 64, and what is known of it before anything arrives
    const int e_l = es + lane * W;
    const bool have = e_l < n_static;
    const int exp_l = have ? expected[e_l] : 0, b_l = have ? T.active[e_l] : 0;
    const unsigned char live_l = (BWD && have) ? GS.live[(size_t)A.f * GS.cap + e_l] : (unsigned char)0;
    int2 pf[FG_NPF];                                          // the neighbour records of the first entries: lane n < 27 holds neighbour n
#pragma unroll
    for (int j = 0; j < FG_NPF; j++) pf[j] = (es + j * W < n_static) ? nbr_record(T, es + j * W, lane) : make_int2(0, 0);
    // may this wave wait?  Only when every workgroup of the launch has started: all of them are then running or done, and unit loops wait for nothing
    int st = lane < 8 ? fg_int(ctr + (FGC_STARTED + lane) * FG_LINE) : 0;
    st += __shfl_xor(st, 1, 64); st += __shfl_xor(st, 2, 64); st += __shfl_xor(st, 4, 64);
    const bool wait = (unsigned)__builtin_amdgcn_readfirstlane(st) - S.fg_base == (unsigned)G_ && !S.fg_nowait;
    unsigned long long a_l = have ? __hip_atomic_load(T.arrive + e_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    unsigned long long pending = __ballot(have);
    if (!wait) {                                              // ... else the entries that are not complete by now are the final wave's
        const bool skip = have && FG_CNT(a_l) != exp_l;
        if (skip) __hip_atomic_store(F->skipm + e_l, GS.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int nskip = __popcll(__ballot(skip));
        for (int e = es + 64 * W; e < n_static; e += W) { if (lane == 0) __hip_atomic_store(F->skipm + e, GS.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); nskip++; }      // (beyond the wave's first 64: not looked at)
        if (nskip && lane == 0) { atomicAdd(ctr + FGC_SKIP * FG_LINE, nskip); fg_rare(F); }
        pending &= ~__ballot(skip);
    }
    // this wave's unit loops are over, and what it stored and registered has completed; the workgroup's last wave signs the workgroup off
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int r1 = -1;
    bool signs = false;
    if (lane == 0 && atomicAdd(&s_fg[8], 1) == 3) { signs = true; r1 = atomicAdd(ctr + (FGC_DONE + (int)(blockIdx.x % FG_SH)) * FG_LINE, 1); }      // (asked for now, read behind the entries)
    signs = __builtin_amdgcn_readfirstlane(signs ? 1 : 0) != 0;
    // the wave's first 64 entries, in the order they complete
    for (int it = 0; pending; it++) {
        const bool mine_pending = (pending >> lane) & 1ull;
        const bool done = mine_pending && FG_CNT(a_l) == exp_l;
        const unsigned long long done_now = __ballot(done);
        if (!done_now) {
            if (it > FG_SPIN_MAX) { fg_give_up(F); break; }
            __builtin_amdgcn_s_sleep(2);
            if (mine_pending) a_l = __hip_atomic_load(T.arrive + e_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        TL(S, 5);
        const bool touched = FG_TCH(a_l) != 0, dirty = FG_DRT(a_l) != 0;
        bool heavy = false;
        if (done) {                                           // the lane's own part: the word back to zero for the next launch, the entry's record for the adjoint pass
            __hip_atomic_store(T.arrive + e_l, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!BWD) {
                heavy = touched || dirty;
                if (e_l < GS.cap) __hip_atomic_store(GS.live + (size_t)A.f * GS.cap + e_l, (unsigned char)(heavy ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else heavy = live_l != 0;
        }
        unsigned long long todo = __ballot(heavy);
        const unsigned long long dirty_set = __ballot(dirty), touched_set = __ballot(touched);
        pending &= ~done_now;
        if ((pending >> lane) & 1ull) a_l = __hip_atomic_load(T.arrive + e_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next poll: on its way while the wave works)
        auto record = [&](int kk) -> int2 {                   // (kk uniform) the prefetched ones by register select, the others from memory
            if (kk < FG_NPF) {
                int2 r = pf[0];
#pragma unroll
                for (int j = 1; j < FG_NPF; j++) if (kk == j) r = pf[j];
                return r;
            }
            return nbr_record(T, es + kk * W, lane);
        };
        int k = todo ? __builtin_ctzll(todo) : -1;
        int2 nbr = k >= 0 ? record(k) : make_int2(0, 0);
        while (k >= 0) {                                      // the whole wave on one entry after the other, the next one's neighbour record asked for ahead
            todo &= todo - 1;
            const int k_next = todo ? __builtin_ctzll(todo) : -1;
            const int2 nbr_next = k_next >= 0 ? record(k_next) : make_int2(0, 0);
            const int e = es + k * W, b = __builtin_amdgcn_readlane(b_l, k);
            if (!BWD) fg_fwd_block(S, GS, A, e, b, nbr, (touched_set >> k) & 1ull, (dirty_set >> k) & 1ull);
            else fg_bwd_block(S, GS, A, e, b, nbr, (dirty_set >> k) & 1ull);
            k = k_next; nbr = nbr_next;
        }
    }
    // (more than 64 entries per wave -- a grid of 256^3 and up that is active all over: one after the other)
    if (wait) for (int e = es + 64 * W; e < n_static; e += W) {
        const int exp_e = expected[e];
        unsigned long long a = fg_word(T.arrive + e);
        for (int it = 0; FG_CNT(a) != exp_e; it++) {
            if (it > FG_SPIN_MAX) { fg_give_up(F); break; }
            __builtin_amdgcn_s_sleep(2);
            a = fg_word(T.arrive + e);
        }
        fg_entry<BWD>(S, T, GS, A, e, a);
    }
    // through with its entries (everything stored has completed): the workgroup's last wave counts the workgroup as finished
    TL(S, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL(S, 7);
    if (lane == 0 && atomicAdd(&s_fg[9], 1) == 3) (void)__hip_atomic_fetch_add(ctr + (FGC_FIN + (int)(blockIdx.x % FG_SH)) * FG_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!signs) return;
    // was this the launch's last workgroup to finish its unit loops?  (last of its shard, then last shard)
    r1 = __builtin_amdgcn_readfirstlane(r1);
    const int sh = blockIdx.x % FG_SH, n_sh = (G_ - sh + FG_SH - 1) / FG_SH;
    if (r1 != n_sh - 1) return;
    int r2 = 0;
    if (lane == 0) {
        __hip_atomic_store(ctr + (FGC_DONE + sh) * FG_LINE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the shard is complete: ready for the next launch)
        r2 = atomicAdd(ctr + FGC_TOP * FG_LINE, 1);
    }
    r2 = __builtin_amdgcn_readfirstlane(r2);
    if ((r2 & 0xffff) != (G_ < FG_SH ? G_ : FG_SH) - 1) return;
    if (lane == 0) __hip_atomic_store(ctr + FGC_TOP * FG_LINE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((r2 >> 16) != 0) fg_final<BWD>(S, T, GS, A);
}


// g2p (mpm:400-426) + advect_kernel (mpm:497-505) for one used particle
template <bool TILE, bool COLLIDE>
__device__ __forceinline__ void used_particle_g2p(const SimP& S, const FrameV& cur, const FrameV& nxt, int s,
                                                  int lb, const Stencil& st, const float x[3],
                                                  const float4* __restrict__ g_out, const AgentP& agent, int f, int tofs, float xn[3]) {
    float nv[3];
    m3 nC;
    g2p_gather<TILE, false>(S, lb, st, g_out, tofs, nv, nC);
    if (COLLIDE) agent.hit[(size_t)f * S.Np + s] = agent_collide_particle<false>(S, agent, f, x, nv) ? 1 : 0;      // mpm:418-422; the flag steers the backward pass
    xn[0] = x[0] + S.dt * nv[0]; xn[1] = x[1] + S.dt * nv[1]; xn[2] = x[2] + S.dt * nv[2];
    store_xvC(nxt, s, xn, nv, nC);
}

__device__ __forceinline__ void load_tile4(const TileO& to, const SimP& S, const float4* __restrict__ src, const PairCtx& pc) {
    if (!pc.live) return;
    const int tofs = pc.ti * 4 * TILE_N;
    tile_nodes<false>(pc.t0, pc.nth, [&](int l) { return tile_node_load(to, S, src, l); },      // (pair units only: p2g_grad_body loads a quad's tile itself)
                      [&](int l, const float4 v) { s_tile[tofs + l] = v.x; s_tile[tofs + TILE_N + l] = v.y; s_tile[tofs + 2 * TILE_N + l] = v.z; s_tile[tofs + 3 * TILE_N + l] = v.w; });
}

// (u, a0): the slot's `used` flag and first state plane, loaded by the caller -- in the item path before the tile load and
// its barrier, so that the particle loads do not queue up behind them (a substep kernel at this size is one workgroup's chain of
// dependent HBM round trips: table -> item -> tile -> barrier -> particle -> ...)
// Returns whether the particle was moved: xn = its new position (the sort's key, SORTKEY kernels).
template <bool COLLIDE>
__device__ __forceinline__ bool slot_g2p(const SimP& S, const FrameV& cur, const FrameV& nxt, int s, bool use_tile,
                                         const TileO& to, const float4* __restrict__ g_out, int* slow, const AgentP& agent, int f,
                                         int u, const float4 a0, int tofs, float xn[3]) {
    if (!u) return false;
    float x[3] = {a0.x, a0.y, a0.z};
    Stencil st;
    stencil_make(x, S.inv_dx, st);
    if (!stencil_inside(st, S.n)) return false;             // already counted in err by p2g
    const int lb = use_tile ? tile_base(to, st) : -1;
    if (lb >= 0) used_particle_g2p<true, COLLIDE>(S, cur, nxt, s, lb, st, x, g_out, agent, f, tofs, xn);
    else { if (use_tile) atomicAdd(slow, 1); used_particle_g2p<false, COLLIDE>(S, cur, nxt, s, 0, st, x, g_out, agent, f, 0, xn); }
    return true;
}

// -----------------------------------------------------------------------------------------
// The sort's counting stage inside k_g2p (SORTKEY kernels, round 6).  The frame a sort works on is written by the k_g2p launch in front of it (the head of a sort
// interval is never a fused launch), and k_sort_count did nothing but read that frame again: the stencil base of every slot -> key, the slot's rank among the slots
// of its key (returning atomics on the cell counts), the block counts -- 14 us of kernel and a launch boundary for values k_g2p has in registers.  Here every wave
// groups its lanes by key (a loop over the DISTINCT keys of the wave: a dozen, the order being nearly sorted), one lane per key asks for the key's range with one
// returning atomic -- all of a wave's in flight together --, the same for its blocks; the sentinel key (slots not in use / off the grid: the tail) is counted per
// WORKGROUP in the tail units, where every lane carries it.  What the sort's other stages read (key, rank, cnt, bcnt) is what k_sort_count left.
// (k_sort_count's side job -- forgetting the block slots of the table about to be rebuilt -- is not needed any more: k_sort_blk_partial's active-list stage visits
//  every block of the grid and now writes the slot of the blocks that are NOT on the new list as well.)
// -----------------------------------------------------------------------------------------
#define SORT_CLR_WGS 32
struct SortKeyP { int* key; int* rank; int* cnt; int* bcnt; int* nact; };
__shared__ int s_sk[8];
// the key of a position: the blocked cell address of its stencil base, or the sentinel (off the grid)
__device__ __forceinline__ int sort_key_of(const SimP& S, const float x[3]) {
    Stencil st;
    stencil_make(x, S.inv_dx, st);
    return stencil_inside(st, S.n) ? cell_addr(st.base[0], st.base[1], st.base[2], S.nb) : S.ncell;
}
// all 64 lanes; `has`: this lane has a slot `s` with key `kk`.  WG_TAIL: a tail unit -- all 256 threads are here, most of them with the sentinel key.
template <bool WG_TAIL>
__device__ __forceinline__ void sort_key_rank(const SimP& S, const SortKeyP& K, bool has, int s, int kk) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const unsigned long long below = (1ull << lane) - 1ull;
    int r_tail = -1;
    if (WG_TAIL) {                                            // the sentinel, counted by the workgroup: one atomic on the tail's word instead of one per wave
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned long long mt = __ballot(has && kk == S.ncell);
        if (lane == 0) s_sk[wave] = __popcll(mt);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_sk[0] + s_sk[1] + s_sk[2] + s_sk[3];
            int base = 0;
            if (tot > 0) { base = atomicAdd(K.cnt + S.ncell, tot); atomicAdd(K.bcnt + (S.ncell >> 6), tot); }
            s_sk[4] = base;
        }
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += s_sk[w];
        if (has && kk == S.ncell) r_tail = s_sk[4] + before + __popcll(mt & below);
        __syncthreads();                                      // (the words are the next unit's)
        has = has && kk != S.ncell;
    }
    // the wave's lanes by key: leader (the first lane of the key), index within the key, size of the group
    unsigned long long rem = __ballot(has);
    int leader = lane, idx = 0, size = 0;
    while (rem) {
        const int L = __builtin_ctzll(rem);
        const int k = __builtin_amdgcn_readlane(kk, L);
        const unsigned long long m = __ballot(has && kk == k);
        if ((m >> lane) & 1ull) { leader = L; idx = __popcll(m & below); size = __popcll(m); }
        rem &= ~m;
    }
    int base = 0;
    if (has && lane == leader) base = atomicAdd(K.cnt + kk, size);                // (one returning atomic per distinct key, all of them in flight together)
    // ... and by block, for the block counts
    rem = __ballot(has);
    while (rem) {
        const int L = __builtin_ctzll(rem);
        const int b = __builtin_amdgcn_readlane(kk >> 6, L);
        const unsigned long long m = __ballot(has && (kk >> 6) == b);
        if (lane == L) atomicAdd(K.bcnt + b, __popcll(m));
        rem &= ~m;
    }
    const int r = __shfl(base, leader, 64) + idx;
    if (has) { K.key[s] = kk; K.rank[s] = r; }
    else if (WG_TAIL && r_tail >= 0) { K.key[s] = S.ncell; K.rank[s] = r_tail; }
}

// COLLIDE: some effector carries a mesh (Rigid): agent.collide runs on the gathered velocity
// SORTKEY: frame f + 1 is sorted next -- the kernel leaves the sort's keys, ranks and counts (sort_key_rank, above)
template <bool COLLIDE, bool SORTKEY = false>
__device__ __forceinline__ void g2p_body(SimP S, float* fr_cur, float* fr_next, TableP T, const float4* __restrict__ g_out,
                                            int* blk_count, int* slow, AgentP agent, int f, SortKeyP K = SortKeyP{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) { *blk_count = 0; if (SORTKEY) *K.nact = 0; }          // grid_op was the last reader of the active list
    const int n_wg = gridDim.x;
    FrameV cur = frame_view(fr_cur, S.Np);
    FrameV nxt = frame_view(fr_next, S.Np, S.wt & 2);
    TL(S, 0);
    Unit un = unit_load<true>(T, blockIdx.x);                       // this workgroup's first unit, asked for together with meta
    const int n_slots = T.meta[9];
    for (int wg = blockIdx.x; wg < n_slots; wg += n_wg) {
        if (wg != (int)blockIdx.x) un = unit_load<true>(T, wg);
        if (un.a.z == -2) continue;
        if (un.a.z >= 0) {
            const PairCtx pc = pair_ctx(un);                     // (the gather kernels walk the pairs-only list)
            const int4 it = pc.it;
            const TileO to = tile_origin(it.x);
            const int i = pc.i;
            const int s0 = it.y + (i < it.z ? i : 0);
            TL(S, 1);
            const int u = cur.used[s0];
            const float4 a0 = cur.A0[s0];
            load_tile3(to, S, g_out, pc);
            __syncthreads();
            TL(S, 2);
            float xn[3] = {0.f, 0.f, 0.f};
            bool moved = false;
            if (i < it.z) moved = slot_g2p<COLLIDE>(S, cur, nxt, s0, true, to, g_out, slow, agent, f, u, a0, pc.ti * 3 * TILE_N, xn);
            if (SORTKEY) {
                int kk = S.ncell;                                // (not in use, or off the grid: the tail)
                if (moved) kk = sort_key_of(S, xn);
                else if (i < it.z && !u && nxt.used[s0]) { const float4 n0 = nxt.A0[s0]; const float xi[3] = {n0.x, n0.y, n0.z}; kk = sort_key_of(S, xi); }      // (entered in this substep: Injector.act wrote its state)
                sort_key_rank<false>(S, K, i < it.z, s0, kk);
            }
            TL(S, 3);
            __syncthreads();
            TL(S, 4);
        } else {
            const int s = un.a.y + tid;
            TileO none = {0, 0, 0};
            float xn[3] = {0.f, 0.f, 0.f};
            bool moved = false;
            int u = 0;
            if (s < S.N) { u = cur.used[s]; moved = slot_g2p<COLLIDE>(S, cur, nxt, s, false, none, g_out, slow, agent, f, u, cur.A0[s], 0, xn); }
            if (SORTKEY) {
                int kk = S.ncell;
                if (moved) kk = sort_key_of(S, xn);
                else if (s < S.N && !u && nxt.used[s]) { const float4 n0 = nxt.A0[s]; const float xi[3] = {n0.x, n0.y, n0.z}; kk = sort_key_of(S, xi); }
                sort_key_rank<true>(S, K, s < S.N, s, kk);
            }
        }
    }
}
struct G2PArgs { SimP S; float* fr_cur; float* fr_next; TableP T; const float4* g_out; int* blk_count; int* slow; AgentP agent; int f; };
template <bool COLLIDE>
__global__ FE_KALIGN __launch_bounds__(WG) void k_g2p(SimP S, float* fr_cur, float* fr_next, TableP T, const float4* g_out, int* blk_count, int* slow, AgentP agent, int f) { g2p_body<COLLIDE>(S, fr_cur, fr_next, T, g_out, blk_count, slow, agent, f); }
template <bool COLLIDE>
__global__ FE_KALIGN __launch_bounds__(WG) void k_g2p_sortkey(SimP S, float* fr_cur, float* fr_next, TableP T, const float4* g_out, int* blk_count, int* slow, AgentP agent, int f, SortKeyP K) { g2p_body<COLLIDE, true>(S, fr_cur, fr_next, T, g_out, blk_count, slow, agent, f, K); }
template <bool COLLIDE>
__global__ FE_KALIGN __launch_bounds__(WG) void k_g2p_b(Batch<G2PArgs> B) { const G2PArgs& A = B.a[blockIdx.y]; g2p_body<COLLIDE>(A.S, A.fr_cur, A.fr_next, A.T, A.g_out, A.blk_count, A.slow, A.agent, A.f); }


// =========================================================================================
// adjoint kernels (hand-derived; closed forms in SURVEY.md Appendix A, validated by the oracle's
// finite-difference tests).  The adjoint frames form a ring of two: Gn = grad[f+1], Gc = grad[f];
// every substep_grad overwrites Gc completely.
// =========================================================================================

// v_out of node (i,j,k) of frame f for the backward pass' global path: the working grid g_out is only valid when grid[f]
// was recomputed; with a stored frame it comes from the per-frame store (blocks addressed through the order's blk_slot)
struct VoutSrc { const float4* g_out; const float4* store; const int* blk_slot; };
// v_out of node `node` of the block in slot `slot` of a frame's store (`st` = the frame's first record; uniform)
__device__ __forceinline__ float4 store_vout(const float4* __restrict__ st, int slot, int node) {
    const u32x3 u = __builtin_amdgcn_raw_buffer_load_b96(wt_rsrc((void*)st), slot * (GS_BLK * 16) + 1024 + node * 12, 0, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), 0.f);
}
__device__ __forceinline__ float4 vout_at(const SimP& S, const VoutSrc& V, int i, int j, int k) {
    if (!V.store) return V.g_out[cell_addr(i, j, k, S.nb)];
    const int slot = V.blk_slot[(((i >> 2) * S.nb) + (j >> 2)) * S.nb + (k >> 2)];
    return slot >= 0 ? store_vout(V.store, slot, ((i & 3) << 4) | ((j & 3) << 2) | (k & 3)) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// advect_kernel.grad + g2p.grad (mpm:443, 538) for one used particle on the GLOBAL path (drifted out of its tile, a tail unit, sort_interval = 0): scatters d/d(v_out)
// with global atomics, leaves the position adjoint (so far) in Gc.A0.xyz.  (The tile form of this loop was round 2's k_g2p_grad; the tile kernels are g2p_grad_particle2's.)
// (agent.collide's adjoint has already been folded into Gn's x/v adjoints by k_collide_grad)
template <bool GPRE = false>
__device__ __forceinline__ void used_particle_g2p_grad(const SimP& S, const FrameV& Gn, const FrameV& Gc, int s, const Stencil& st, const VoutSrc& V, float* gg_out,
                                                       const PState* gpre = nullptr) {
    PState g;                                   // adjoints of x', v', C'
    if (GPRE) copy_xvC(g, *gpre);               // (k_pgg_g2pg: in registers, not in Gn)
    else load_xvC(Gn, s, g);
    // x' = x + dt v'  =>  v'_bar += dt x'_bar
    float gv[3] = {g.v[0] + S.dt * g.x[0], g.v[1] + S.dt * g.x[1], g.v[2] + S.dt * g.x[2]};
    const float c4 = 4.f * S.inv_dx;
    float gfx[3] = {0.f, 0.f, 0.f};
    // q(o) = gv + c4 gC (o - fx) is linear in the node offset o: base at o = 0, per-(i,j) part, one fma per node for k.
    // sum W c4 (v^T gC)_b = c4 (nv^T gC)_b with nv = sum W v_out leaves the loop (VALU-issue-bound kernel).
    float qb[3], qz[3], nvw[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        qb[a] = gv[a] - c4 * (g.C.a[a][0] * st.fx[0] + g.C.a[a][1] * st.fx[1] + g.C.a[a][2] * st.fx[2]);
        qz[a] = c4 * g.C.a[a][2];
    }
#pragma unroll 1
    for (int ij = 0; ij < 9; ij++) {
        const int i = ij / 3, j = ij - 3 * i;
        const float wi = STW(st, i, 0), wj = STW(st, j, 1);
        const float wiwj = wi * wj, dwiwj = stencil_dw(st, i, 0) * wj, widwj = wi * stencil_dw(st, j, 1);
        float qij[3];
#pragma unroll
        for (int a = 0; a < 3; a++) qij[a] = qb[a] + c4 * (g.C.a[a][0] * (float)i + g.C.a[a][1] * (float)j);
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float wk = st.w[kk][2];
            const float weight = wiwj * wk;
            float q[3];
#pragma unroll
            for (int a = 0; a < 3; a++) q[a] = kk == 0 ? qij[a] : qij[a] + (float)kk * qz[a];
            const int c = cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + kk, S.nb);
            const float4 vo = vout_at(S, V, st.base[0] + i, st.base[1] + j, st.base[2] + kk);
            const float v0 = vo.x, v1 = vo.y, v2 = vo.z;
            float* dst = gg_out + c;
            unsafeAtomicAdd(dst, weight * q[0]);
            unsafeAtomicAdd(dst + S.ncell, weight * q[1]);
            unsafeAtomicAdd(dst + 2 * S.ncell, weight * q[2]);
            const float sdot = v0 * q[0] + v1 * q[1] + v2 * q[2];
            const float w3 = wiwj * wk;
            nvw[0] += w3 * v0; nvw[1] += w3 * v1; nvw[2] += w3 * v2;
            const float t = wk * sdot;
            gfx[0] += dwiwj * t;                               // d weight / d fx_d
            gfx[1] += widwj * t;
            gfx[2] += wiwj * (stencil_dw(st, kk, 2) * sdot);
        }
    }
#pragma unroll
    for (int b = 0; b < 3; b++) gfx[b] -= c4 * (nvw[0] * g.C.a[0][b] + nvw[1] * g.C.a[1][b] + nvw[2] * g.C.a[2][b]);     // dpos_b = o_b - fx_b
    pstore(Gc, Gc.A0, s, make_float4(g.x[0] + S.inv_dx * gfx[0], g.x[1] + S.inv_dx * gfx[1], g.x[2] + S.inv_dx * gfx[2], 0.f));
}

// one slot on the global path (tail / sort_interval = 0)
template <bool GPRE = false>
__device__ __forceinline__ void g2p_grad_slot_global(const SimP& S, const FrameV& cur, const FrameV& Gn, const FrameV& Gc, int s,
                                                     const VoutSrc& V, float* gg_out, const AgentP& agent, int f, const GridStore& GS, const PState* gpre = nullptr) {
    if (!cur.used[s]) return;
    float4 a0 = cur.A0[s];
    float x[3] = {a0.x, a0.y, a0.z};
    Stencil st;
    stencil_make(x, S.inv_dx, st);
    if (!stencil_inside(st, S.n)) {
        if (GPRE) Gc.A0[s] = make_float4(gpre->x[0], gpre->x[1], gpre->x[2], 0.f);
        else { float4 gx = Gn.A0[s]; Gc.A0[s] = make_float4(gx.x, gx.y, gx.z, 0.f); }
        return;
    }
    used_particle_g2p_grad<GPRE>(S, Gn, Gc, s, st, V, gg_out, gpre);
    for (int bx = st.base[0] >> 2; bx <= (st.base[0] + 2) >> 2; bx++)              // the (up to 8) blocks whose gg_out planes now hold atomics
        for (int by = st.base[1] >> 2; by <= (st.base[1] + 2) >> 2; by++)
            for (int bz = st.base[2] >> 2; bz <= (st.base[2] + 2) >> 2; bz++) mark_dirty(GS, V.blk_slot, (bx * S.nb + by) * S.nb + bz);
}

// A particle that has left its tile since the sort, worked on by the WHOLE wave: lane n < 27 takes node n of its stencil (uniform
// inputs: the slot `s` and the position `x`).  The per-lane road above walks the 27 nodes in a rolled loop, and every v_out comes
// through two dependent loads (block table, then the store): nine rounds of two round trips, ~15 us during which the lane's wave --
// and with it its workgroup's tile hand-over -- waits.  The block hitting the floor has ~60 such particles per substep, enough to
// make their workgroups the launch's tail: k_g2p_grad 28.7 us where a fresh order takes 23.4 (k_p2g's slow path is fire-and-forget
// atomics and costs it 1.4 us).  Here the 27 fetches are in flight together and the sums meet through cross-lane adds.
// (gpre: the particle's adjoint in registers, uniform -- k_pgg_g2pg, where Gn does not hold it)
// (pk, nbr_entry: the block of the particle's unit and -- lane r < 27 -- the active-list entries of its 27 neighbours: a node among them finds its record in the
//  frame's store through a cross-lane read instead of the block table -- one memory round trip per drifted particle instead of two, round 6; a particle that has
//  left its tile is seldom further than a block away)
template <bool GPRE = false>
__device__ __forceinline__ void g2p_grad_drifted(const SimP& S, const FrameV& Gn, const FrameV& Gc, int s, const float x[3],
                                                 const VoutSrc& V, float* gg_out, const GridStore& GS, const PState* gpre = nullptr, int pk = -1, int nbr_entry = -1) {
    const int lane = threadIdx.x & 63;
    Stencil st;
    stencil_make(x, S.inv_dx, st);
    PState g;
    if (GPRE) copy_xvC(g, *gpre);
    else load_xvC(Gn, s, g);                                  // (one address for all lanes)
    const float c4 = 4.f * S.inv_dx;
    float qb[3], qx[3], qy[3], qz[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        qx[a] = c4 * g.C.a[a][0]; qy[a] = c4 * g.C.a[a][1]; qz[a] = c4 * g.C.a[a][2];
        qb[a] = (g.v[a] + S.dt * g.x[a]) - (qx[a] * st.fx[0] + qy[a] * st.fx[1] + qz[a] * st.fx[2]);
    }
    const bool node = lane < 27;
    const int n = node ? lane : 0, i = n / 9, j = (n / 3) % 3, k = n % 3;
    const float wi = STW(st, i, 0), wj = STW(st, j, 1), wk = STW(st, k, 2);
    const float dwi = stencil_dw(st, i, 0), dwj = stencil_dw(st, j, 1), dwk = stencil_dw(st, k, 2);
    float4 vo;
    if (V.store) {                                            // (uniform)
        const int nx = st.base[0] + i, ny = st.base[1] + j, nz = st.base[2] + k;
        const int d0 = (nx >> 2) - BLK_I(pk) + 1, d1 = (ny >> 2) - BLK_J(pk) + 1, d2 = (nz >> 2) - BLK_K(pk) + 1;
        const bool near = pk >= 0 && (unsigned)d0 <= 2u && (unsigned)d1 <= 2u && (unsigned)d2 <= 2u;
        const int e_near = __shfl(nbr_entry, near ? d0 * 9 + d1 * 3 + d2 : 0, 64);
        int slot = e_near;
        if (!near) slot = V.blk_slot[(((nx >> 2) * S.nb) + (ny >> 2)) * S.nb + (nz >> 2)];
        vo = store_vout(V.store, slot >= 0 ? slot : 0, ((nx & 3) << 4) | ((ny & 3) << 2) | (nz & 3));
        if (slot < 0) vo = make_float4(0.f, 0.f, 0.f, 0.f);
    } else vo = vout_at(S, V, st.base[0] + i, st.base[1] + j, st.base[2] + k);
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) q[a] = qb[a] + (float)i * qx[a] + (float)j * qy[a] + (float)k * qz[a];
    const float W = node ? wi * wj * wk : 0.f;
    if (node) {
        float* dst = gg_out + cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + k, S.nb);
        unsafeAtomicAdd(dst, W * q[0]); unsafeAtomicAdd(dst + S.ncell, W * q[1]); unsafeAtomicAdd(dst + 2 * S.ncell, W * q[2]);
    }
    const float sdot = node ? vo.x * q[0] + vo.y * q[1] + vo.z * q[2] : 0.f;
    float r[6] = {dwi * wj * wk * sdot, wi * dwj * wk * sdot, wi * wj * dwk * sdot, W * vo.x, W * vo.y, W * vo.z};
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int c = 0; c < 6; c++) r[c] += __shfl_xor(r[c], o, 64);
    // sum_o W c4 (v_o^T gC)_b enters with dpos_b = o_b - fx_b (g2p_grad_particle2)
    const float gfx[3] = {r[0] - (r[3] * qx[0] + r[4] * qx[1] + r[5] * qx[2]), r[1] - (r[3] * qy[0] + r[4] * qy[1] + r[5] * qy[2]),
                          r[2] - (r[3] * qz[0] + r[4] * qz[1] + r[5] * qz[2])};
    if (lane == 0) pstore(Gc, Gc.A0, s, make_float4(g.x[0] + S.inv_dx * gfx[0], g.x[1] + S.inv_dx * gfx[1], g.x[2] + S.inv_dx * gfx[2], 0.f));
    if (lane < 8) {                                           // the (up to 8) blocks whose gg_out planes now hold atomics
        const int b0 = (lane & 1) ? (st.base[0] + 2) >> 2 : st.base[0] >> 2, b1 = (lane & 2) ? (st.base[1] + 2) >> 2 : st.base[1] >> 2,
                  b2 = (lane & 4) ? (st.base[2] + 2) >> 2 : st.base[2] >> 2;
        mark_dirty(GS, V.blk_slot, (b0 * S.nb + b1) * S.nb + b2);
    }
}
// the lanes of a tile unit's wave whose particles drifted out of the tile (`drifted`: used, stencil on the grid, not on the tile;
// `outside`: used, stencil off the grid -- passed through untouched, as the per-lane road does); a00 = the lane's (x, .) plane, s its slot
// (gpre: the lane's own adjoint of x', v', C' in registers -- k_pgg_g2pg; a drifted lane's is then broadcast to the wave)
template <bool GPRE = false>
__device__ __forceinline__ void g2p_grad_wave_slow(const SimP& S, const FrameV& Gn, const FrameV& Gc, int s, const float4 a00, bool drifted, bool outside,
                                                   const VoutSrc& V, float* gg_out, int* slow, const GridStore& GS, const PState* gpre = nullptr, int pk = -1, int nbr_entry = -1) {
    if (outside) {
        if (GPRE) pstore(Gc, Gc.A0, s, make_float4(gpre->x[0], gpre->x[1], gpre->x[2], 0.f));
        else { const float4 gx = Gn.A0[s]; Gc.A0[s] = make_float4(gx.x, gx.y, gx.z, 0.f); }
    }
    unsigned long long todo = __ballot(drifted);
    if (todo && (threadIdx.x & 63) == 0) atomicAdd(slow, __popcll(todo));
    while (todo) {
        const int L = __builtin_ctzll(todo);
        todo &= todo - 1;
#define LANE_F(v) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L))
        const float x[3] = {LANE_F(a00.x), LANE_F(a00.y), LANE_F(a00.z)};
        if (GPRE) {
            PState gu;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                gu.x[a] = LANE_F(gpre->x[a]); gu.v[a] = LANE_F(gpre->v[a]);
#pragma unroll
                for (int b = 0; b < 3; b++) gu.C.a[a][b] = LANE_F(gpre->C.a[a][b]);
            }
            g2p_grad_drifted<true>(S, Gn, Gc, __builtin_amdgcn_readlane(s, L), x, V, gg_out, GS, &gu, pk, nbr_entry);
        } else g2p_grad_drifted(S, Gn, Gc, __builtin_amdgcn_readlane(s, L), x, V, gg_out, GS, nullptr, pk, nbr_entry);
#undef LANE_F
    }
}

// workgroup-level flush of s_pose into the effectors' adjoint arrays (call with all threads; contains barriers)
__device__ __forceinline__ void pose_flush(const AgentP& agent, int f) {
    __syncthreads();
    const int t = threadIdx.x;
    if (t < FE_MAX_EFF * 14) {
        const float v = s_pose[t];
        const int ei = t / 14, k = t % 14;
        if (v != 0.f && ei < agent.n) {
            const EffP& e = agent.e[ei];
            if (k < 3) atomicAdd(&e.gpos[f * 3 + k], v);
            else if (k < 7) atomicAdd(&e.gquat[f * 4 + k - 3], v);
            else if (k < 10) atomicAdd(&e.gpos[(f + 1) * 3 + k - 7], v);
            else atomicAdd(&e.gquat[(f + 1) * 4 + k - 10], v);
        }
        s_pose[t] = 0.f;
    }
    __syncthreads();
}

template <bool QUADS = true, int MAXIT = 8>
__device__ __forceinline__ void g2p_grad_load_tile2(const TileO& to, const SimP& S, const float4* __restrict__ g_out, const float4* __restrict__ st, int nbr_entry, const PairCtx& pc, float* gt);
// (round 2's one-loop build of this adjoint -- k_g2p_grad, option g2p_grad_v = 1: 109 registers, 3 ... 8 us slower than the two-pass build since round 3 -- was removed in round 6;
//  the option value is still accepted and means the default)
struct G2PGradArgs { SimP S; float* fr_cur; float* Gn_; float* Gc_; TableP T; const float4* g_out; float* gg_out; float4* slab; int* slow; GridStore GS; int f; AgentP agent; };




// -----------------------------------------------------------------------------------------
// k_g2p_grad2 (option "g2p_grad_v", default): the same adjoint with the 27-node loop split in two passes.
// Round 2's kernel gathers (v_out -> the position adjoint) and scatters (three scanned values -> d v_out) in ONE loop; unrolled it
// needs 650 B of scratch per lane, so it stayed a rolled 9 x 3 loop with register selects of the weights -- 7.8 us of the
// workgroup's 11.7 at C2, where k_g2p's gather takes 1.5 and k_p2g's four-value scatter 4.9.  Here the gather pass runs first (tile
// reads, nine partial sums, nothing scanned; the position adjoint is stored as soon as it is done), then the scatter pass (no tile
// reads: w q recomputed from 12 coefficients, three scanned values per node).  Between the passes only the coefficients of
// q(o) = qb + o_x qx + o_y qy + o_z qz and the nine weights are live.
// The tile of v_out comes through the 27 neighbour entries of the item's block, fetched by 27 lanes together with the particle
// state (k_p2g's neighbour_entry): round 2 looked every tile node's block up in blk_slot first (two dependent hops per node).
// -----------------------------------------------------------------------------------------
template <bool QUADS, int MAXIT>
__device__ __forceinline__ void g2p_grad_load_tile2(const TileO& to, const SimP& S, const float4* __restrict__ g_out,
                                                    const float4* __restrict__ st, int nbr_entry, const PairCtx& pc, float* gt) {
    if (!pc.live) return;                                    // (whole waves: the shuffles below see all 64 lanes)
    auto put = [&](int l, const float4 v) { gt[l] = v.x; gt[TILE_N + l] = v.y; gt[2 * TILE_N + l] = v.z; };
    if (st) {                                                 // (uniform) the frame's record in the grid store, through the item's 27 neighbour entries
        tile_nodes<QUADS, MAXIT>(pc.t0, pc.nth, [&](int l) {
            const int tx = l >> 6, ty = (l >> 3) & 7, tz = l & 7;
            const int e = __shfl(nbr_entry, tile_region(tx) * 9 + tile_region(ty) * 3 + tile_region(tz), 64);
            const float4 v = store_vout(st, e >= 0 ? e : 0, (((tx + 3) & 3) << 4) | (((ty + 3) & 3) << 2) | ((tz + 3) & 3));      // (a block off the list: entry 0's record is read and dropped)
            return make_float4(e >= 0 ? v.x : 0.f, e >= 0 ? v.y : 0.f, e >= 0 ? v.z : 0.f, 0.f);
        }, put);
    } else for (int l = pc.t0; l < TILE_N; l += pc.nth) put(l, tile_node_load(to, S, g_out, l));      // (the recompute road -- a frame without a record in the store: rolled, one node after the other)
}

// A quad unit's wave loads its tile of v_out by itself, eight nodes per lane (~58 VALU instructions a round: region, entry, record address).  When
// none of its particles sits on the tile's shell -- known before the loads go out: the positions arrive with the neighbour entries -- only the
// inner 6^3 nodes are ever read, and they take four rounds instead of eight (the shell's words stay whatever they were: the gather pass does
// not touch them, and the accumulators that take the words over are zeroed as a whole).
__device__ __forceinline__ void quad_load_vout_inner(const float4* __restrict__ st, int nbr_entry, float* gt) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));          // (opaque, as in quad_handover)
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int l6 = lane + 64 * k;
        const bool ok = l6 < SLAB_N;
        const int c6 = ok ? l6 : 0;
        const int x6 = c6 / 36, r = c6 - 36 * x6, y6 = r / 6, z6 = r - 6 * y6;        // slab coordinates 0..5 = tile indices 1..6: block B (0..3) or B + 1 (4, 5)
        const int e = __shfl(nbr_entry, (x6 >> 2) * 9 + (y6 >> 2) * 3 + (z6 >> 2) + 13, 64);      // (all lanes: region (1 + x6 / 4, ..) = neighbour 13 + ..)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && e >= 0) v = store_vout(st, e, ((x6 & 3) << 4) | ((y6 & 3) << 2) | (z6 & 3));
        if (ok) { const int l = ((x6 + 1) * TILE_T + y6 + 1) * TILE_T + z6 + 1; gt[l] = v.x; gt[TILE_N + l] = v.y; gt[2 * TILE_N + l] = v.z; }
    }
}

// executed by ALL lanes of the wave (`live` = this lane holds a used particle whose stencil fits the tile)
// QUAD: the wave's own tile -- v_out in `gt`, which the fixed-point accumulators (words, see fix_scale) take over between the passes;
// returns the factor that turns them back into floats.
// G = 3 / 9: a split wave (lane_split) -- this lane works on the nodes of plane i = gi (column (gi, gj)) of its particle's stencil, the gather
// pass' partial sums are added up over the particle's lanes, the position adjoint is stored by its first lane (`primary`).
// AR: the accumulators as k_pgg_g2pg's arena has them; gpre (that kernel): the adjoints of x', v', C' come in registers instead of from Gn.
template <int MINW, bool QUAD, int G, int AR = 0>
__device__ __forceinline__ float g2p_grad_particle2(const SimP& S, const FrameV& Gn, const FrameV& Gc, int s, int lb, const Stencil& st,
                                                    bool live, int tofs, const float* gt, int gofs, bool primary, const PState* gpre = nullptr) {      // (lb, live: this lane's particle; from pass 2 on the particle it scatters for)
    PState g;                                   // adjoints of x', v', C'
    double* const acc3 = lds_acc3<AR>();
    // (AR decides, not a test of the pointer: a comparison of a stack object's address with null is a use the optimiser cannot see through, and the object stays in scratch)
    if (AR && live) copy_xvC(g, *gpre);
    else if (!AR && live) load_xvC(Gn, s, g);
    else { g.x[0] = g.x[1] = g.x[2] = g.v[0] = g.v[1] = g.v[2] = 0.f; g.C = m3_zero(); }
    const float c4 = 4.f * S.inv_dx;
    // q(o) = gv + c4 gC (o - fx), gv = v'_bar + dt x'_bar  (x' = x + dt v')
    float qb[3], qx[3], qy[3], qz[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        qx[a] = c4 * g.C.a[a][0]; qy[a] = c4 * g.C.a[a][1]; qz[a] = c4 * g.C.a[a][2];
        qb[a] = (g.v[a] + S.dt * g.x[a]) - (qx[a] * st.fx[0] + qy[a] * st.fx[1] + qz[a] * st.fx[2]);
    }
    const int gi = gofs >> 6, gj = (gofs >> 3) & 7;           // a split wave: this lane's x (and y) offset
    const int l0 = tofs + lb + (G > 1 ? gofs : 0);
    gt += lb + (G > 1 ? gofs : 0);
    constexpr int UNR_X = MINW >= 4 ? 1 : 3;
    constexpr int NI = G > 1 ? 1 : 3, NJ = G == 9 ? 1 : 3;
    // MINW = 4: the x offset of the stencil stays a rolled loop of three (nine nodes unrolled inside it, the x weights by register
    // select): the completely unrolled passes need 166 registers, this form fits the 128 of four waves per SIMD
    {   // ---- pass 1: gather.  gfx_d = sum_o dW/df_d (v_o . q_o),  nvw = sum_o W v_o
        const float dwz[3] = {stencil_dw(st, 0, 2), stencil_dw(st, 1, 2), stencil_dw(st, 2, 2)};      // (the x and y ones are one VALU each: made where used)
        float gfx[3] = {0.f, 0.f, 0.f}, nvw[3] = {0.f, 0.f, 0.f};
#pragma unroll UNR_X
        for (int ii = 0; ii < NI; ii++) {
            const int i = G > 1 ? gi : ii;
            const float wi = (G > 1 || MINW >= 4) ? STW(st, i, 0) : st.w[ii][0], dwi = stencil_dw(st, i, 0);
#pragma unroll
            for (int jj = 0; jj < NJ; jj++) {
                const int j = G == 9 ? gj : jj;
                const float wj = G == 9 ? STW(st, j, 1) : st.w[jj][1];
                float qij[3];
#pragma unroll
                for (int a = 0; a < 3; a++) qij[a] = qb[a] + (float)i * qx[a] + (float)j * qy[a];
                float T = 0.f, Tz = 0.f, P[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    const int l = ((G > 1 ? 0 : ii) * TILE_T + (G == 9 ? 0 : jj)) * TILE_T + kk;
                    const float v0 = gt[l], v1 = gt[TILE_N + l], v2 = gt[2 * TILE_N + l];
                    const float sdot = v0 * (qij[0] + (float)kk * qz[0]) + v1 * (qij[1] + (float)kk * qz[1]) + v2 * (qij[2] + (float)kk * qz[2]);
                    const float wk = st.w[kk][2];
                    T += wk * sdot; Tz += dwz[kk] * sdot;
                    P[0] += wk * v0; P[1] += wk * v1; P[2] += wk * v2;
                }
                const float wiwj = wi * wj;
                gfx[0] += (dwi * wj) * T;
                gfx[1] += (wi * stencil_dw(st, j, 1)) * T;
                gfx[2] += wiwj * Tz;
#pragma unroll
                for (int a = 0; a < 3; a++) nvw[a] += wiwj * P[a];
                // the running sums are pinned here: otherwise the (pure) arithmetic sinks towards its use behind the loop while the 81 tile
                // reads stay where they are, and every tile value is live at once (650 B of scratch per lane)
                asm volatile("" : "+v"(gfx[0]), "+v"(gfx[1]), "+v"(gfx[2]), "+v"(nvw[0]), "+v"(nvw[1]), "+v"(nvw[2]));
            }
        }
        if (G > 1) {                                          // the particle's lanes each hold a part of the sums
#pragma unroll
            for (int a = 0; a < 3; a++) {                     // (one value after the other: asked for together, the reads keep a register each)
                gfx[a] = split_sum<G>(live ? gfx[a] : 0.f, gofs); NODE_FENCE();
                nvw[a] = split_sum<G>(live ? nvw[a] : 0.f, gofs); NODE_FENCE();
            }
        }
        // sum_o W c4 (v_o^T gC)_b = (nvw^T c4 gC)_b enters with dpos_b = o_b - fx_b
        gfx[0] -= nvw[0] * qx[0] + nvw[1] * qx[1] + nvw[2] * qx[2];
        gfx[1] -= nvw[0] * qy[0] + nvw[1] * qy[1] + nvw[2] * qy[2];
        gfx[2] -= nvw[0] * qz[0] + nvw[1] * qz[1] + nvw[2] * qz[2];
        if (live && primary) pstore(Gc, Gc.A0, s, make_float4(g.x[0] + S.inv_dx * gfx[0], g.x[1] + S.inv_dx * gfx[1], g.x[2] + S.inv_dx * gfx[2], 0.f));
    }
    NODE_FENCE();
    float inv = 1.f;
    if (QUAD) {
        // the tile changes hands: every read of v_out is done (the wave's LDS operations complete in order), the words are zeroed and
        // the coefficients of q scaled for the fixed-point sums
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int* acc = (int*)acc3 + tofs;
        for (int l = threadIdx.x & 63; l < 3 * TILE_N / 4; l += 64) ((int4*)acc)[l] = make_int4(0, 0, 0, 0);
        asm volatile("" ::: "memory");
        float b = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++) b = fmaxf(b, fabsf(qb[a]) + 2.f * (fabsf(qx[a]) + fabsf(qy[a]) + fabsf(qz[a])));
        const FixScale fs = fix_scale(wave_max(live ? b : 0.f));
        inv = __any(live && !(b <= 3.4e38f)) ? __int_as_float(0x7fc00000) : fs.inv;       // (fix_nonfinite: a blown-up adjoint stays NaN on the grid, as in the fp64 tiles)
#pragma unroll
        for (int a = 0; a < 3; a++) { qb[a] *= fs.s; qx[a] *= fs.s; qy[a] *= fs.s; qz[a] *= fs.s; }
    }
    // ---- pass 2: scatter d v_out(o) += W(o) q(o), summed over runs of equal stencil base before the LDS atomics
    Stencil sw = st;                                         // (the weights of the particle this lane scatters for: another lane's after wave_sort)
    int l0s = l0;
    if (G == 1 && S.wsort) {
        int key = live ? lb : 0x3ff;
        if (wave_needs_sort(key)) {
            const int dest = wave_sort_dest(key);
            wave_send(dest, key);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                wave_send(dest, qb[a]); wave_send(dest, qx[a]); wave_send(dest, qy[a]); wave_send(dest, qz[a]);
#pragma unroll
                for (int b = 0; b < 3; b++) wave_send(dest, sw.w[a][b]);
            }
            live = key != 0x3ff; lb = live ? key : 0; l0s = tofs + lb;
        }
    }
    const SegScan sc = seg_setup(live ? lb + (G > 1 ? gofs << 10 : 0) : (0x40000000 | (int)threadIdx.x));       // (only now: five registers less across pass 1; a split wave's key carries the group)
    const bool issue = sc.tail && live;
    const float livef = live ? 1.f : 0.f;
#pragma unroll UNR_X
    for (int ii = 0; ii < NI; ii++) {
        const int i = G > 1 ? gi : ii;
        const float lwi = livef * ((G > 1 || MINW >= 4) ? STW(sw, i, 0) : sw.w[ii][0]);
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
            const int j = G == 9 ? gj : jj;
            const float lw = lwi * (G == 9 ? STW(sw, j, 1) : sw.w[jj][1]);
            float qij[3];
#pragma unroll
            for (int a = 0; a < 3; a++) qij[a] = qb[a] + (float)i * qx[a] + (float)j * qy[a];
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                const float weight = lw * sw.w[kk][2];
                float c0 = weight * (qij[0] + (float)kk * qz[0]), c1 = weight * (qij[1] + (float)kk * qz[1]), c2 = weight * (qij[2] + (float)kk * qz[2]);
                seg_scan3(sc, c0, c1, c2);
                if (issue) {
                    const int l = l0s + ((G > 1 ? 0 : ii) * TILE_T + (G == 9 ? 0 : jj)) * TILE_T + kk;
                    if (QUAD) {
                        int* acc = (int*)acc3;
                        atomicAdd(acc + l, fix_round(c0));                   // ds_add_u32
                        atomicAdd(acc + TILE_N + l, fix_round(c1));
                        atomicAdd(acc + 2 * TILE_N + l, fix_round(c2));
                    } else {
                        atomicAdd(&acc3[l], (double)c0);                      // ds_add_f64
                        atomicAdd(&acc3[TILE_N + l], (double)c1);
                        atomicAdd(&acc3[2 * TILE_N + l], (double)c2);
                    }
                }
            }
        }
    }
    return inv;
}
template <int MINW, bool QUAD, int AR = 0>
__device__ __forceinline__ float g2p_grad_particle2_split(const SimP& S, const FrameV& Gn, const FrameV& Gc, int s, int lb, const Stencil& st,
                                                          bool live, int tofs, const float* gt, const LaneSplit& ls, const PState* gpre = nullptr) {
    if (ls.G == 1) return g2p_grad_particle2<MINW, QUAD, 1, AR>(S, Gn, Gc, s, lb, st, live, tofs, gt, 0, true, gpre);      // (wave-uniform)
    if (ls.G == 3) return g2p_grad_particle2<MINW, QUAD, 3, AR>(S, Gn, Gc, s, lb, st, live, tofs, gt, ls.gofs, ls.primary, gpre);
    return g2p_grad_particle2<MINW, QUAD, 9, AR>(S, Gn, Gc, s, lb, st, live, tofs, gt, ls.gofs, ls.primary, gpre);
}
template <int MINW>
__device__ __forceinline__ void g2p_grad2_body(SimP S, float* fr_cur, float* Gn_, float* Gc_, TableP T,
                                                  const float4* __restrict__ g_out, float* gg_out, float4* slab, int* slow,
                                                  GridStore GS, int f, AgentP agent) {
    const int tid = threadIdx.x;
    const bool stored = GS.cap > 0 && GS.flag[f];
    VoutSrc V; V.g_out = g_out; V.store = stored ? GS.data + (size_t)f * GS.cap * GS_BLK : nullptr; V.blk_slot = T.blk_slot;
    FrameV cur = frame_view(fr_cur, S.Np);
    FrameV Gn = frame_view(Gn_, S.Np), Gc = frame_view(Gc_, S.Np, S.wt & 4);
    TL(S, 0);
    Unit un = unit_load(T, blockIdx.x);                       // this workgroup's first unit, asked for together with meta
    const int n_slots = T.meta[5];
    bool prev_quad = false;                                   // (unit_enter)
    for (int wg = blockIdx.x; wg < n_slots; wg += gridDim.x) {
        if (wg != (int)blockIdx.x) un = unit_load(T, wg);
        if (un.a.z == -2) continue;
        if (un.a.z >= 0) {
            const PairCtx pc = unit_ctx(un);
            unit_enter(pc.quad, prev_quad);
            const int4 it = pc.it;
            const TileO to = tile_origin(it.x);
            const int tofs = pc.ti * 3 * TILE_N;                 // (floats / doubles of a pair's tiles, words of a quad's one)
            const int wbase = pc.quad ? 0 : (pc.i & 64);         // this wave's first particle within the item; few of them: three or nine lanes each (lane_split)
            const int cnt = __builtin_amdgcn_readfirstlane(min(64, max(0, it.z - wbase)));
            const LaneSplit ls = lane_split(cnt, (S.lsplit & 2) != 0);
            const int i = wbase + ls.p;
            const bool has = ls.ok && i < it.z;
            const int s = it.y + (has ? i : 0);
            const int nbr_entry = neighbour_entry(T.blk_slot, S.nb, it.x);      // (with the particle loads: one hop) tile load + shell hand-over
            const int u0 = cur.used[s];
            const float4 a00 = cur.A0[s];
            // a quad's wave keeps v_out where its accumulators will be (g2p_grad_particle2): four tiles of each do not fit side by side
            float* gt = pc.quad ? (float*)s_acc3 + tofs : s_tile3 + tofs;
            float inv = 1.f;
            bool wshell = false;                                 // (quad units) some particle of the wave reaches its tile's outer shell
            if (FE_LEAN_QUADS && pc.quad && V.store) {           // (the positions first: they tell whether the tile's shell is needed at all)
                // (the stencil bases only, and nothing of it kept: the stencil proper is made behind the loads as before)
                const bool u_ = has && u0 != 0;
                const int l0_ = (int)(a00.x * S.inv_dx - 0.5f) - to.ox, l1_ = (int)(a00.y * S.inv_dx - 0.5f) - to.oy, l2_ = (int)(a00.z * S.inv_dx - 0.5f) - to.oz;
                const bool in_ = u_ && (unsigned)l0_ <= TILE_T - 3 && (unsigned)l1_ <= TILE_T - 3 && (unsigned)l2_ <= TILE_T - 3;
                wshell = __any(in_ && ((l0_ == 0) | (l0_ == 5) | (l1_ == 0) | (l1_ == 5) | (l2_ == 0) | (l2_ == 5)));
                if (pc.live) { if (wshell) g2p_grad_load_tile2(to, S, g_out, V.store, nbr_entry, pc, gt); else quad_load_vout_inner(V.store, nbr_entry, gt); }
            } else {
                g2p_grad_load_tile2(to, S, g_out, V.store, nbr_entry, pc, gt);
                if (!pc.quad && pc.live) for (int l = pc.t0; l < 3 * TILE_N; l += pc.nth) s_acc3[tofs + l] = 0.0;
            }
            unit_sync(pc.quad);
            TL(S, 2);
            {
                const bool used = has && u0 != 0;
                float x[3] = {0.f, 0.f, 0.f};
                if (used) { x[0] = a00.x; x[1] = a00.y; x[2] = a00.z; }
                Stencil st;
                stencil_make(x, S.inv_dx, st);
                const bool inside = used && stencil_inside(st, S.n);
                const int lb = inside ? tile_base(to, st) : -1;
                const bool live = lb >= 0;
                if (pc.quad && !(FE_LEAN_QUADS && V.store)) wshell = __any(live && stencil_on_shell(lb));
                if (__any(live)) {                               // wave-uniform: empty waves skip the loops
                    if (pc.quad) inv = g2p_grad_particle2_split<MINW, true>(S, Gn, Gc, s, live ? lb : 0, st, live, tofs, gt, ls);
                    else g2p_grad_particle2_split<MINW, false>(S, Gn, Gc, s, live ? lb : 0, st, live, tofs, gt, ls);
                } else if (pc.quad && pc.live) {                 // nothing scattered: the hand-over must not see v_out as sums
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (int l = pc.t0; l < 3 * TILE_N; l += pc.nth) ((int*)s_acc3)[tofs + l] = 0;
                }
                g2p_grad_wave_slow(S, Gn, Gc, s, a00, inside && !live && ls.primary, used && !inside && ls.primary, V, gg_out, slow, GS, nullptr, it.x, nbr_entry);      // (whole waves: wave-uniform loop; a split wave's particles once each)
            }
            TL(S, 5);
            unit_sync(pc.quad);
            TL(S, 6);
            if (pc.quad) {
                if (pc.live) quad_handover<3>(S, (const int*)s_acc3 + tofs, inv, inv, slab, pc.slab, gg_out, GS, to, nbr_entry, wshell, S.wt & 4);      // (as in k_p2g)
            } else if (pc.live) for (int l = pc.t0; l < TILE_N; l += pc.nth)
                tile_handover<3>(S, slab, pc.slab, gg_out, GS, to, nbr_entry, l,
                                 make_float4((float)s_acc3[tofs + l], (float)s_acc3[tofs + TILE_N + l], (float)s_acc3[tofs + 2 * TILE_N + l], 0.f), S.wt & 4);
            unit_sync(pc.quad);
            TL(S, 7);
        } else {
            const int s = un.a.y + tid;
            if (s < S.N) g2p_grad_slot_global(S, cur, Gn, Gc, s, V, gg_out, agent, f, GS);
        }
    }
}
template <int MINW>
__global__ FE_KALIGN __launch_bounds__(WG, MINW) void k_g2p_grad2(SimP S, float* fr_cur, float* Gn_, float* Gc_, TableP T, const float4* g_out, float* gg_out, float4* slab, int* slow, GridStore GS, int f, AgentP agent) { g2p_grad2_body<MINW>(S, fr_cur, Gn_, Gc_, T, g_out, gg_out, slab, slow, GS, f, agent); }
template <int MINW>
__global__ FE_KALIGN __launch_bounds__(WG, MINW) void k_g2p_grad2_b(Batch<G2PGradArgs> B) { const G2PGradArgs& A = B.a[blockIdx.y]; g2p_grad2_body<MINW>(A.S, A.fr_cur, A.Gn_, A.Gc_, A.T, A.g_out, A.gg_out, A.slab, A.slow, A.GS, A.f, A.agent); }


// agent.collide's adjoint (mpm:418-422 in reverse) as a pass of its own, before k_g2p_grad.  Inlined into k_g2p_grad the
// forward-mode Jacobian passes cost every particle of every scene with a Rigid effector the kernel's occupancy (256 VGPRs +
// scratch), and run serially in one lane they are ~30k instructions: every wave holding a single contact particle took
// 50-80 us.  Here only particles flagged by the forward pass do any work, they are compacted into a list, and each gets a
// row of 16 lanes: lane d < 10 runs Jacobian column d (mv 3, p0 3, q0 4), the 27-node gather of the velocity the colliders
// saw is split over the row, and the results meet through row shuffles.  The pulled-back velocity adjoint and the position
// term are folded into the frame-(f+1) adjoints k_g2p_grad starts from; pose adjoints go through s_pose.
__device__ __forceinline__ float row_sum(float v) {          // sum over the 16 lanes of a row, result in every lane
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}
// The collider chain's adjoint for one position, shared by a row of 16 lanes (lane `sub`): x = the particle position
// (NODE: the node position, which does not move with the velocity), nv = the velocity the first collider saw, g = d/d(velocity
// after the chain) on entry and d/d(nv) on exit, gx += the part that flows into x (particles only).
template <bool NODE>
__device__ __forceinline__ void collide_chain_grad_row(const SimP& S, const AgentP& agent, int f, const float x[3], const float nv0[3],
                                                       float g[3], float gx[3], int sub) {
    const int row0 = (threadIdx.x & 63) & ~15;                 // first lane of this row within the wave
    const float sdt = NODE ? 0.f : S.dt;
    float nv[3] = {nv0[0], nv0[1], nv0[2]};
    // forward through the collider chain (every lane, it is cheap), keeping each collider's input velocity
    float vin[FE_MAX_EFF][3];
#pragma unroll
    for (int ei = 0; ei < FE_MAX_EFF; ei++) {
        if (ei < agent.n && agent.e[ei].has_mesh) {
            const EffP& e = agent.e[ei];
            vin[ei][0] = nv[0]; vin[ei][1] = nv[1]; vin[ei][2] = nv[2];
            const float pos[3] = {x[0] + sdt * nv[0], x[1] + sdt * nv[1], x[2] + sdt * nv[2]};
            if (pos[1] > agent.collide_min_y) {
                float out[3];
                t_dynamic_collide<float>(e.mesh, e.pos + f * 3, e.quat + f * 4, e.pos + (f + 1) * 3, e.quat + (f + 1) * 4, pos, nv, S.dt, out);
                nv[0] = out[0]; nv[1] = out[1]; nv[2] = out[2];
            }
        }
    }
#pragma unroll
    for (int ei = FE_MAX_EFF - 1; ei >= 0; ei--) {
        if (!(ei < agent.n && agent.e[ei].has_mesh)) continue;
        const EffP& e = agent.e[ei];
        const float v[3] = {vin[ei][0], vin[ei][1], vin[ei][2]};
        if (!(x[1] + sdt * v[1] > agent.collide_min_y)) continue;
        // One forward-mode Jacobian column per lane.  out = cv + vt(rel, n) infl + rel (1 - infl), rel = mv - cv,
        // cv = (R(q1) pm + p1 - pos) / dt: the inputs that enter only through cv (p1, q1 and the explicit -pos) have the closed-form
        // Jacobian (I - d out/d mv) d cv/d input, so ten columns are enough -- mv (3), p0 (3: it enters through pm only, like
        // the pm-part of pos), q0 (4).
        const int dir = sub;
        Dual p0[3], q0[4], p1[3], q1[4], pos[3], mv[3], out[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            p0[d] = Dual(e.pos[f * 3 + d], dir == 3 + d ? 1.f : 0.f);
            p1[d] = Dual(e.pos[(f + 1) * 3 + d]);
            mv[d] = Dual(v[d], dir == d ? 1.f : 0.f);
            pos[d] = Dual(x[d] + sdt * v[d]);
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            q0[d] = Dual(e.quat[f * 4 + d], dir == 6 + d ? 1.f : 0.f);
            q1[d] = Dual(e.quat[(f + 1) * 4 + d]);
        }
        const bool hit = t_dynamic_collide<Dual>(e.mesh, p0, q0, p1, q1, pos, mv, S.dt, out);      // same branch in every lane of the row
        if (!hit) continue;
        const float r = dir < 10 ? g[0] * out[0].d + g[1] * out[1].d + g[2] * out[2].d : 0.f;
        float c[3], a[3], gq0[4];
#pragma unroll
        for (int d = 0; d < 3; d++) { c[d] = __shfl(r, row0 + d, 64); a[d] = __shfl(r, row0 + 3 + d, 64); }
#pragma unroll
        for (int d = 0; d < 4; d++) gq0[d] = __shfl(r, row0 + 6 + d, 64);
        const float idt = 1.f / S.dt;
        const float w[3] = {(g[0] - c[0]) * idt, (g[1] - c[1]) * idt, (g[2] - c[2]) * idt};      // (I - J_mv)^T g / dt
        float pm[3];
        {
            const float qn = 1.f / sqrtf(e.quat[f * 4] * e.quat[f * 4] + e.quat[f * 4 + 1] * e.quat[f * 4 + 1] + e.quat[f * 4 + 2] * e.quat[f * 4 + 2] + e.quat[f * 4 + 3] * e.quat[f * 4 + 3]);
            const float qi[4] = {e.quat[f * 4] * qn, -e.quat[f * 4 + 1] * qn, -e.quat[f * 4 + 2] * qn, -e.quat[f * 4 + 3] * qn};
            const float rel0[3] = {x[0] + sdt * v[0] - e.pos[f * 3], x[1] + sdt * v[1] - e.pos[f * 3 + 1], x[2] + sdt * v[2] - e.pos[f * 3 + 2]};
            t_quat_rotate(rel0, qi, pm);
        }
        float gq1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Dual qd[4], pmd[3], rot[3];
            for (int d = 0; d < 4; d++) qd[d] = Dual(e.quat[(f + 1) * 4 + d], d == k ? 1.f : 0.f);
            for (int d = 0; d < 3; d++) pmd[d] = Dual(pm[d]);
            t_quat_rotate(pmd, qd, rot);
            gq1[k] = w[0] * rot[0].d + w[1] * rot[1].d + w[2] * rot[2].d;
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float gpos = -a[d] - w[d];
            if (!NODE) gx[d] += gpos;
            g[d] = c[d] + sdt * gpos;
            if (sub == 0) { atomicAdd(&s_pose[ei * 14 + d], a[d]); atomicAdd(&s_pose[ei * 14 + 7 + d], w[d]); }
        }
        if (sub == 0) {
#pragma unroll
            for (int d = 0; d < 4; d++) { atomicAdd(&s_pose[ei * 14 + 3 + d], gq0[d]); atomicAdd(&s_pose[ei * 14 + 10 + d], gq1[d]); }
        }
    }
}

__device__ void collide_grad_row(const SimP& S, const AgentP& agent, int f, const FrameV& cur, const FrameV& Gn, const VoutSrc& V, int s, int sub) {
    const float4 a0 = cur.A0[s];
    const float x[3] = {a0.x, a0.y, a0.z};
    Stencil st;
    stencil_make(x, S.inv_dx, st);
    const float4 g0 = Gn.A0[s], g1 = Gn.A1[s];
    float g[3] = {g0.w + S.dt * g0.x, g1.x + S.dt * g0.y, g1.y + S.dt * g0.z};       // d/d(v[f+1]) = v_bar' + dt x_bar'
    float gx[3] = {0.f, 0.f, 0.f};
    if (stencil_inside(st, S.n)) {                             // (k_g2p_grad passes a particle outside the grid through untouched)
        float nv[3] = {0.f, 0.f, 0.f};
        for (int n = sub; n < 27; n += 16) {
            const int i = n / 9, j = (n / 3) % 3, k = n % 3;
            const float weight = STW(st, i, 0) * STW(st, j, 1) * STW(st, k, 2);
            const float4 gv = vout_at(S, V, st.base[0] + i, st.base[1] + j, st.base[2] + k);
            nv[0] += weight * gv.x; nv[1] += weight * gv.y; nv[2] += weight * gv.z;
        }
        nv[0] = row_sum(nv[0]); nv[1] = row_sum(nv[1]); nv[2] = row_sum(nv[2]);
        collide_chain_grad_row<false>(S, agent, f, x, nv, g, gx, sub);
    }
    // Fold the result into the adjoints k_g2p_grad reads: it forms d/d(gathered velocity) = v_bar' + dt x_bar' and starts d/dx[f]
    // from x_bar', so x_bar' += gx and v_bar' = g - dt x_bar' make it continue from behind the colliders, unchanged itself.
    if (sub == 0) {
        const float nx[3] = {g0.x + gx[0], g0.y + gx[1], g0.z + gx[2]};
        Gn.A0[s] = make_float4(nx[0], nx[1], nx[2], g[0] - S.dt * nx[0]);
        Gn.A1[s] = make_float4(g[1] - S.dt * nx[1], g[2] - S.dt * nx[2], g1.z, g1.w);
    }
}

// contact particles sit together in the sorted order, so a slot-indexed launch leaves all the work to a few workgroups: first
// gather the flagged slots of the frame into one list ...
__global__ __launch_bounds__(256) void k_collide_list(SimP S, float* fr_cur, int f, AgentP agent, int* list, int* count) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < S.N && frame_view(fr_cur, S.Np).used[s] != 0 && agent.hit[(size_t)f * S.Np + s] != 0) list[atomicAdd(count, 1)] = s;
}
// ... then one row of 16 lanes per list entry, 16 entries per workgroup (the grid covers the worst case, N entries)
__global__ __launch_bounds__(256) void k_collide_grad(SimP S, float* fr_cur, float* Gn_, const float4* __restrict__ g_out, TableP T,
                                                      GridStore GS, int f, AgentP agent, const int* __restrict__ list, const int* __restrict__ count) {
    const int n = *count;
    if ((int)blockIdx.x * 16 >= n) return;                       // (uniform) the grid is sized for the worst case, the list is short
    const int tid = threadIdx.x;
    if (tid < FE_MAX_EFF * 14) s_pose[tid] = 0.f;
    __syncthreads();
    FrameV cur = frame_view(fr_cur, S.Np), Gn = frame_view(Gn_, S.Np);
    const bool stored = GS.cap > 0 && GS.flag[f];
    VoutSrc V; V.g_out = g_out; V.store = stored ? GS.data + (size_t)f * GS.cap * GS_BLK : nullptr; V.blk_slot = T.blk_slot;
    for (int base = blockIdx.x * 16; base < n; base += gridDim.x * 16) {
        const int idx = base + (tid >> 4);
        if (idx < n) collide_grad_row(S, agent, f, cur, Gn, V, list[idx], tid & 15);      // whole rows enter or skip together
    }
    pose_flush(agent, f);
}

// grid_op.grad (mpm:539): d/d v_out (slabs of k_g2p_grad + slow-path atomics in gg_out) -> gg_in (d/d v_in, d/d mass);
// re-zeroes g_in, gg_out and the dynamic flags
template <bool STATICS, bool DYN>
__device__ __forceinline__ void grid_grad_body(SimP S, TableP T, const float4* __restrict__ slab, float* g_in, float* gg_out, float4* gg_in,
                                                   const int* __restrict__ blk_list, const int* __restrict__ blk_count, int* blk_flag,
                                                   GridStore GS, int f, StaticsP ST, AgentP agent, NodeWork* work, int* work_count) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (the short-list road of k_grid: this wave's entry, asked for together with the list's length and the frame's store flag)
    const int es = static_entry(wave);
    const bool es_ok = es < S.nb * S.nb * S.nb;
    const int blk_s = es_ok ? T.active[es] : 0;
    const unsigned char lv_st = (es_ok && es < GS.cap) ? GS.live[(size_t)f * GS.cap + es] : (unsigned char)0, lv_cur = es_ok ? GS.cur[es] : (unsigned char)0;
    const unsigned char drt_s = es_ok ? GS.dirty[es] : (unsigned char)0;
    const int2 nbr_s = es_ok ? nbr_record(T, es, lane) : make_int2(0, 0);
    const bool stored = GS.cap > 0 && GS.flag[f];
    const int n_static = T.meta[2], n_dyn = *blk_count;
    auto one_block = [&](int e, int b, bool is_static, bool dirty, const int2 nbr) {
        const int c = (b << 6) | lane;
        const int bi = b / (S.nb * S.nb), bj = (b / S.nb) % S.nb, bk = b % S.nb;
        // total (p, m): from the forward pass' store, or kept in g_in by k_grid<true>
        const float4 gi = stored ? GS.data[((size_t)f * GS.cap + e) * GS_BLK + lane]
                                 : make_float4(g_in[c], g_in[S.ncell + c], g_in[2 * S.ncell + c], g_in[3 * S.ncell + c]);
        float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dirty) go = make_float4(gg_out[c], gg_out[S.ncell + c], gg_out[2 * S.ncell + c], 0.f);
        if (is_static) { const float4 t = gather_slabs<3>(slab, nbr, lane); go.x += t.x; go.y += t.y; go.z += t.z; }
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gi.w > FE_EPS) {
            float vo[3], kmul[3];
            float trace[FE_MAX_STATICS][3];
            const int ni = bi * 4 + (lane >> 4), nj = bj * 4 + ((lane >> 2) & 3), nk = bk * 4 + (lane & 3);
            bool dyn_hit = false;
            node_velocity<STATICS, DYN>(S, ST, gi, ni, nj, nk, vo, kmul, trace, &agent, f, nullptr, &dyn_hit);
            float inv = 1.f / gi.w;
            float gcol[3] = {go.x * kmul[0], go.y * kmul[1], go.z * kmul[2]};
            if (DYN && dyn_hit && (gcol[0] != 0.f || gcol[1] != 0.f || gcol[2] != 0.f)) {
                // a node inside one of the agent's colliders with a live adjoint: its chain is finished by k_grid_collide_grad
                NodeWork w; w.c = c; w.ni = ni; w.nj = nj; w.nk = nk; w.gi = gi; w.go[0] = go.x; w.go[1] = go.y; w.go[2] = go.z; w.pad = 0.f;
                work[atomicAdd(work_count, 1)] = w;
                gcol[0] = gcol[1] = gcol[2] = 0.f;                    // (gg_in[c] is overwritten there)
            }
            if (STATICS) node_statics_grad(S, ST, ni, nj, nk, trace, gcol);
            float g0 = gcol[0], g1 = gcol[1], g2 = gcol[2];
            out.x = g0 * inv; out.y = g1 * inv; out.z = g2 * inv;
            out.w = -(gi.x * g0 + gi.y * g1 + gi.z * g2) * inv * inv;
        }
        gg_in[c] = out;
        if (!stored) { g_in[c] = 0.f; g_in[S.ncell + c] = 0.f; g_in[2 * S.ncell + c] = 0.f; g_in[3 * S.ncell + c] = 0.f; }      // k_grid<true> kept the totals there
        if (dirty) { gg_out[c] = 0.f; gg_out[S.ncell + c] = 0.f; gg_out[2 * S.ncell + c] = 0.f; }
        if (lane == 0 && !is_static) blk_flag[b] = 0;
    };
    // the entries k_grid worked on in this frame (GridStore): the others have no mass, pass nothing on, and are read by no particle
    const unsigned char* __restrict__ live = stored ? GS.live + (size_t)f * GS.cap : GS.cur;
    if (n_static <= 4 * (int)gridDim.x) {
        if (es < n_static && (stored ? lv_st : lv_cur) != 0) one_block(es, blk_s, true, drt_s == GS.stamp, nbr_s);
    } else {                                                  // (the strided walk of k_grid's longer lists)
        const int W = 4 * (int)gridDim.x;
        for (int e0 = es; e0 < n_static; e0 += 8 * W) {
            unsigned lm = 0, dm = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = e0 + k * W;
                const bool ok = e < n_static;
                const unsigned char l = ok ? live[e] : (unsigned char)0, d = ok ? GS.dirty[e] : (unsigned char)0;
                lm |= (l != 0 ? 1u : 0u) << k; dm |= (d == GS.stamp ? 1u : 0u) << k;
            }
            lm = __builtin_amdgcn_readfirstlane(lm); dm = __builtin_amdgcn_readfirstlane(dm);
            unsigned todo = lm;
            int k = todo ? __builtin_ctz(todo) : -1;
            int blk = k >= 0 ? T.active[e0 + k * W] : 0;
            int2 nbr = k >= 0 ? nbr_record(T, e0 + k * W, lane) : make_int2(0, 0);
            while (k >= 0) {
                todo &= todo - 1;
                const int k_next = todo ? __builtin_ctz(todo) : -1;
                const int blk_next = k_next >= 0 ? T.active[e0 + k_next * W] : 0;
                const int2 nbr_next = k_next >= 0 ? nbr_record(T, e0 + k_next * W, lane) : make_int2(0, 0);
                one_block(e0 + k * W, blk, true, (dm >> k) & 1, nbr);
                k = k_next; blk = blk_next; nbr = nbr_next;
            }
        }
    }
    for (int d = blockIdx.x * 4 + wave; d < n_dyn; d += gridDim.x * 4) one_block(0, blk_list[d], false, true, make_int2(0, 0));
}
struct GridGradArgs { SimP S; TableP T; const float4* slab; float* g_in; float* gg_out; float4* gg_in; const int* blk_list; const int* blk_count; int* blk_flag; GridStore GS; int f; StaticsP ST; AgentP agent; NodeWork* work; int* work_count; };
template <bool STATICS, bool DYN>
__global__ FE_KALIGN __launch_bounds__(256, (STATICS || DYN) ? 2 : 4) void k_grid_grad(SimP S, TableP T, const float4* slab, float* g_in, float* gg_out, float4* gg_in, const int* blk_list, const int* blk_count, int* blk_flag, GridStore GS, int f, StaticsP ST, AgentP agent, NodeWork* work, int* work_count) { grid_grad_body<STATICS, DYN>(S, T, slab, g_in, gg_out, gg_in, blk_list, blk_count, blk_flag, GS, f, ST, agent, work, work_count); }
template <bool STATICS, bool DYN>
__global__ FE_KALIGN __launch_bounds__(256, (STATICS || DYN) ? 2 : 4) void k_grid_grad_b(Batch<GridGradArgs> B) { const GridGradArgs& A = B.a[blockIdx.y]; grid_grad_body<STATICS, DYN>(A.S, A.T, A.slab, A.g_in, A.gg_out, A.gg_in, A.blk_list, A.blk_count, A.blk_flag, A.GS, A.f, A.ST, A.agent, A.work, A.work_count); }


// second half of grid_op.grad for the nodes k_grid_grad<.., DYN> set aside: agent.collide's adjoint at the node (mpm:393-395 in
// reverse), one row of 16 lanes per node like k_collide_grad, then the statics' chain and the division by the mass.
template <bool STATICS>
__global__ __launch_bounds__(256) void k_grid_collide_grad(SimP S, float4* gg_in, int f, StaticsP ST, AgentP agent, const NodeWork* __restrict__ work,
                                                           const int* __restrict__ work_count) {
    const int n = *work_count;
    if ((int)blockIdx.x * 16 >= n) return;                       // (uniform)
    const int tid = threadIdx.x;
    if (tid < FE_MAX_EFF * 14) s_pose[tid] = 0.f;
    __syncthreads();
    const int sub = tid & 15;
    for (int base = blockIdx.x * 16; base < n; base += gridDim.x * 16) {
    const int idx = base + (tid >> 4);
    if (idx < n) {
        const NodeWork w = work[idx];
        float vo[3], kmul[3], vdyn[3];
        float trace[FE_MAX_STATICS][3];
        node_velocity<STATICS, true>(S, ST, w.gi, w.ni, w.nj, w.nk, vo, kmul, trace, &agent, f, vdyn);
        float gcol[3] = {w.go[0] * kmul[0], w.go[1] * kmul[1], w.go[2] * kmul[2]};
        const float xn[3] = {(float)w.ni * S.dx, (float)w.nj * S.dx, (float)w.nk * S.dx};
        collide_chain_grad_row<true>(S, agent, f, xn, vdyn, gcol, nullptr, sub);
        if (STATICS) node_statics_grad(S, ST, w.ni, w.nj, w.nk, trace, gcol);
        const float inv = 1.f / w.gi.w;
        if (sub == 0)
            gg_in[w.c] = make_float4(gcol[0] * inv, gcol[1] * inv, gcol[2] * inv, -(w.gi.x * gcol[0] + w.gi.y * gcol[1] + w.gi.z * gcol[2]) * inv * inv);
    }
    }
    pose_flush(agent, f);
}

// Effector.move_kernel.grad (effector.py:154-155), position chain only
__device__ void effector_move_grad(const EffP& e, int f) {
    float xin[3] = {e.pos[f * 3] + e.v[f * 3], e.pos[f * 3 + 1] + e.v[f * 3 + 1], e.pos[f * 3 + 2] + e.v[f * 3 + 2]};
    float xn[3], J[3][3];
    boundary_x(e.bnd, xin, xn, J);
    for (int d = 0; d < 3; d++) {
        float g = J[0][d] * e.gpos[(f + 1) * 3] + J[1][d] * e.gpos[(f + 1) * 3 + 1] + J[2][d] * e.gpos[(f + 1) * 3 + 2];
        atomicAdd(&e.gpos[f * 3 + d], g);       // injected particles add to the same slot concurrently
        e.gv[f * 3 + d] += g;
    }
    // quat[f+1] = qmul(w2quat(w[f]), quat[f]) (effector.py:161): one forward-mode pass per input (w: 3, quat: 4)
    const float gq[4] = {e.gquat[(f + 1) * 4], e.gquat[(f + 1) * 4 + 1], e.gquat[(f + 1) * 4 + 2], e.gquat[(f + 1) * 4 + 3]};
    if (gq[0] != 0.f || gq[1] != 0.f || gq[2] != 0.f || gq[3] != 0.f) {
#pragma unroll 1
        for (int dir = 0; dir < 7; dir++) {
            Dual w[3], q[4], out[4];
            for (int d = 0; d < 3; d++) w[d] = Dual(e.w[f * 3 + d], dir == d ? 1.f : 0.f);
            for (int d = 0; d < 4; d++) q[d] = Dual(e.quat[f * 4 + d], dir == 3 + d ? 1.f : 0.f);
            t_move_quat<Dual>(w, q, out);
            const float c = gq[0] * out[0].d + gq[1] * out[1].d + gq[2] * out[2].d + gq[3] * out[3].d;
            if (dir < 3) e.gw[f * 3 + dir] += c; else atomicAdd(&e.gquat[f * 4 + dir - 3], c);
        }
    }
}

// p2g.grad + svd_grad + compute_F_tmp.grad (mpm:544-546) for one used particle; TILE: (d v_in, d mass) in LDS (4 planes)
// What the constitutive adjoint needs again after the 27-node loop -- C, F and, in the SVD build, U, V, sigma, J -- waits in LDS
// (s_stash, one column per thread) instead of in registers: round 1 re-read C and F from the frame (72 B per particle of extra
// HBM traffic) and re-ran the constitutive model, and the SVD build kept everything live (256 + 32 VGPRs, one wave per SIMD).
// Where k_p2g_grad leaves the adjoint of frame f.  Normally slot s of Gc.  When the next substep of the same fe_step_grad call works in
// another particle order (a sort lies between frames f - 1 and f), the kernel writes the third adjoint buffer in THAT order instead --
// slot to_slot[pid_of_slot[s]] -- and the separate reorder pass (k_perm_reorder: 12.7 us, once per sort interval) is not launched.
struct GradDst {
    FrameV G; const int* __restrict__ to_slot; const int* __restrict__ pid_of_slot;
    __device__ __forceinline__ int slot(int s) const { return to_slot ? to_slot[pid_of_slot[s]] : s; }
};
// (STASH_GENERAL = 36: C, F, U, V; the singular values and J stay in registers: 36 KB + 16 KB of tile = three workgroups per CU.  Declared with the tiles.)
// NOSTASH (a quad unit's wave, whose tile `tl` lies where the pair units keep their stash): C and F are read from the frame again behind
// the loop -- 72 bytes per particle that the wave's own loads left in the L2 a few microseconds earlier
// G = 3 / 9 (tile path of the SVD-free build): a split wave (lane_split) -- this lane gathers the nodes of plane gi (column (gi, gj)) of its
// particle's stencil, the fifteen sums are added up over the particle's lanes, the adjoint is stored by its first lane (`primary`).
// AR: the LDS views of k_pgg_g2pg's arena.  KEEP (that kernel): the adjoints of x, v and C are handed back in `keep` instead of being stored -- the g2p_grad
// part of the same launch consumes them from registers, and stores them itself for the few particles whose adjoint somebody else reads (p2g_grad_g2p_grad_body).
template <bool TILE, bool GENERAL, bool PRE = false, bool NOSTASH = false, int G = 1, int AR = 0, bool KEEP = false>
__device__ __forceinline__ void used_particle_p2g_grad(const SimP& S, const FrameV& cur, const FrameV& Gn, const FrameV& Gc, int s,
                                                       const float4* __restrict__ info_, const TileO& to,
                                                       const float4* __restrict__ gg_in, int* slow, int tofs, const P2GRaw& pre, const GradDst& D, const float* tl = nullptr,
                                                       int gofs = 0, bool primary = true, PState* keep = nullptr) {
    PState p;
    PInfo info;
    if (PRE) { p = pre.p; info = pre.info; }                  // (asked for ahead of the tile load and its barrier: p2g_grad_body)
    else { load_xvC(cur, s, p); load_F(cur, s, p.F); info = load_info(S, info_, s); }
    Constitutive k;
    constitutive_eval_t<GENERAL>(p.C, p.F, S.dt, info.mu, info.lam, info.mass, info.cls, S.stress_scale, k);
    Stencil st;
    stencil_make(p.x, S.inv_dx, st);
    const bool inside = stencil_inside(st, S.n);
    const int lb = (TILE && inside) ? tile_base(to, st) : -1;
    if (TILE && inside && lb < 0) {            // drifted out of the tile: redo on the global path
        if (!primary && !KEEP) return;         // (a split wave: once per particle -- unless every lane is to hold the result)
        if (primary) atomicAdd(slow, 1);
        // (NOSTASH travels along: a quad unit's wave has no stash column -- two of the quad's tiles lie where the pair units keep theirs
        //  (p2g_grad_body), and a drifted particle that parked C and F there overwrote gathered nodes the other lanes were still reading.
        //  ADVICE r4; tests/test_hip_parity.py::test_quad_units_with_drifted_particles_in_p2g_grad)
        used_particle_p2g_grad<false, GENERAL, false, NOSTASH, 1, AR, KEEP>(S, cur, Gn, Gc, s, info_, to, gg_in, slow, 0, pre, D, nullptr, 0, primary, keep);
        return;
    }
    float* stash = Stash<GENERAL, AR>::at();
    int col = threadIdx.x;
    if (!NOSTASH) {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                stash[(a * 3 + b) * WG + col] = p.C.a[a][b]; stash[(9 + a * 3 + b) * WG + col] = p.F.a[a][b];
                if (GENERAL) { stash[(18 + a * 3 + b) * WG + col] = k.U.a[a][b]; stash[(27 + a * 3 + b) * WG + col] = k.V.a[a][b]; }
            }
    }
    float Gv[3] = {0.f, 0.f, 0.f};
    m3 GA = m3_zero();
    float gxs[3] = {0.f, 0.f, 0.f};             // inv_dx * d/d fx from the stencil
    if (inside) {
        const float m = info.mass;
        float gfx[3] = {0.f, 0.f, 0.f};
        float mv[3];
#pragma unroll
        for (int a = 0; a < 3; a++)
            mv[a] = m * p.v[a] - S.dx * (k.affine.a[a][0] * st.fx[0] + k.affine.a[a][1] * st.fx[1] + k.affine.a[a][2] * st.fx[2]);
        // The kernel is bound by VALU issue, so the sums are organised to do as little per node as possible:
        //   GA[a][b] = sum W gin[a] (o_b - fx_b) dx = dx (M[a][b] - fx_b Gv[a]),  M[a][b] = sum W gin[a] o_b, with o_0 = i and
        //   o_1 = j constant over the inner k loop (per-(i,j) partial sums T, Tz);
        //   sum W dx (A^T gin)_b = dx (A^T Gv)_b leaves the loop altogether.
        m3 M = m3_zero();
        const int gi = gofs >> 6, gj = (gofs >> 3) & 7;          // a split wave: this lane's x (and y) offset
        const int lg = lb + (G > 1 ? gofs : 0);
        constexpr int UNR_IJ = (!TILE && KEEP) ? 1 : 9;
        // (k_pgg_g2pg's road for drifted particles and tail units walks the nine columns one after the other: unrolled, its 27 float4 loads are in flight
        //  together and the kernel, which has no register to spare, spills around them)
#pragma unroll UNR_IJ
        for (int ij = 0; ij < 9; ij++) {
            const int ic = ij / 3, jc = ij - 3 * ic;
            if ((G > 1 && ic > 0) || (G == 9 && jc > 0)) continue;      // (compile time)
            const int i = G > 1 ? gi : ic, j = G == 9 ? gj : jc;
            const float wi = STW(st, i, 0), wj = STW(st, j, 1);
            const float wiwj = wi * wj, dwiwj = stencil_dw(st, i, 0) * wj, widwj = wi * stencil_dw(st, j, 1);
            const float ox = (float)i * S.dx, oy = (float)j * S.dx;
            float mij[3];
#pragma unroll
            for (int a = 0; a < 3; a++) mij[a] = mv[a] + k.affine.a[a][0] * ox + k.affine.a[a][1] * oy;
            float T[3] = {0.f, 0.f, 0.f}, Tz[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                float gin[3], gm;
                if (TILE && NOSTASH) {
                    const int l = lg + ((G > 1 ? 0 : ic) * TILE_T + (G == 9 ? 0 : jc)) * TILE_T + kk;
                    gin[0] = tl[l]; gin[1] = tl[TILE_N + l]; gin[2] = tl[2 * TILE_N + l]; gm = tl[3 * TILE_N + l];
                } else if (TILE) {
                    const int l = tofs + lg + ((G > 1 ? 0 : ic) * TILE_T + (G == 9 ? 0 : jc)) * TILE_T + kk;
                    const float* t4 = lds_tile4<AR>();
                    gin[0] = t4[l]; gin[1] = t4[TILE_N + l]; gin[2] = t4[2 * TILE_N + l]; gm = t4[3 * TILE_N + l];
                } else {
                    float4 gi = gg_in[cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + kk, S.nb)];
                    gin[0] = gi.x; gin[1] = gi.y; gin[2] = gi.z; gm = gi.w;
                }
                const float wk = st.w[kk][2];
                const float weight = wiwj * wk;
                const float oz = (float)kk * S.dx;
                float sdot = gm * m;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    sdot += gin[a] * (mij[a] + k.affine.a[a][2] * oz);
                    const float wg = weight * gin[a];
                    T[a] += wg;
                    if (kk > 0) Tz[a] += (float)kk * wg;
                }
                const float t = wk * sdot;
                gfx[0] += dwiwj * t;
                gfx[1] += widwj * t;
                gfx[2] += wiwj * (stencil_dw(st, kk, 2) * sdot);
            }
#pragma unroll
            for (int a = 0; a < 3; a++) {
                Gv[a] += T[a];
                M.a[a][0] += (float)i * T[a]; M.a[a][1] += (float)j * T[a]; M.a[a][2] += Tz[a];
            }
        }
        if (G > 1) {                                           // the particle's lanes each hold a part of the fifteen sums
#pragma unroll
            for (int a = 0; a < 3; a++) {                      // (one value after the other: asked for together, the cross-lane reads keep a register each)
                Gv[a] = split_sum<G>(Gv[a], gofs); NODE_FENCE();
                gfx[a] = split_sum<G>(gfx[a], gofs); NODE_FENCE();
#pragma unroll
                for (int b = 0; b < 3; b++) { M.a[a][b] = split_sum<G>(M.a[a][b], gofs); NODE_FENCE(); }
            }
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) GA.a[a][b] = S.dx * (M.a[a][b] - st.fx[b] * Gv[a]);
#pragma unroll
        for (int b = 0; b < 3; b++) gfx[b] -= S.dx * (Gv[0] * k.affine.a[0][b] + Gv[1] * k.affine.a[1][b] + Gv[2] * k.affine.a[2][b]);
#pragma unroll
        for (int d = 0; d < 3; d++) gxs[d] = S.inv_dx * gfx[d];
    }
    // (the slot index is laundered too: otherwise the ~25 64-bit addresses of the loads and stores below are formed before the
    // loop and live -- or spill -- across it)
    asm volatile("" : "+v"(s));
    const int sd = D.slot(s);                   // where this particle's adjoint goes (asked for here: two dependent loads behind the constitutive adjoint)
    {
        // back from the stash, behind an index the optimiser cannot see through (it would otherwise forward the stored values,
        // i.e. keep them in registers across the loop)
        asm volatile("" : "+v"(col));
        if (NOSTASH) { PState q2; load_xvC(cur, s, q2); p.C = q2.C; load_F(cur, s, p.F); }
        else {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                p.C.a[a][b] = stash[(a * 3 + b) * WG + col]; p.F.a[a][b] = stash[(9 + a * 3 + b) * WG + col];
                if (GENERAL) { k.U.a[a][b] = stash[(18 + a * 3 + b) * WG + col]; k.V.a[a][b] = stash[(27 + a * 3 + b) * WG + col]; }
            }
        }
        // F_tmp = (I + dt C) F again (27 fma) rather than 9 more stash planes; J of the SVD-free build likewise
        m3 IdtC = m3_scale(p.C, S.dt);
        IdtC.a[0][0] += 1.f; IdtC.a[1][1] += 1.f; IdtC.a[2][2] += 1.f;
        k.Ft = m3_mul(IdtC, p.F);
        if (!GENERAL) k.J = m3_det(k.Ft);
    }
    m3 Fg2; load_F(Gn, s, Fg2);
    const float4 gc0 = Gc.A0[s];                // position adjoint so far (k_g2p_grad)
    float gx[3] = {gc0.x + gxs[0], gc0.y + gxs[1], gc0.z + gxs[2]};
    float gvv[3] = {info.mass * Gv[0], info.mass * Gv[1], info.mass * Gv[2]};
    m3 gC, gF;
    constitutive_grad_t<GENERAL>(p.C, p.F, S.dt, info.mu, info.lam, info.mass, info.cls, S.stress_scale, k, GA, Fg2, gC, gF);
    if (KEEP) {                                 // (every lane of a split wave: the g2p_grad part works with the particle's adjoint in all of them)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            keep->x[a] = gx[a]; keep->v[a] = gvv[a];
#pragma unroll
            for (int b = 0; b < 3; b++) keep->C.a[a][b] = gC.a[a][b];
        }
        if (primary) store_F(D.G, sd, gF);
        return;
    }
    if (G > 1 && !primary) return;              // (a split wave: the particle's first lane stores)
    store_xvC(D.G, sd, gx, gvv, gC);
    store_F(D.G, sd, gF);
}

// Returns whether the slot holds a used particle (KEEP: its adjoint is then in `keep`, not in memory).
// (KEEP && TILE: a slot of a work item that is not in use was taken out of use by the host since the sort -- the sort puts unused particles behind the
//  items, and a particle the Injector is about to use waits there: no Injector.act adjoint on this road, which keeps its code out of k_pgg_g2pg's unit loop)
template <bool TILE, bool GENERAL, bool PRE = false, bool NOSTASH = false, int G = 1, int AR = 0, bool KEEP = false>
__device__ __forceinline__ bool slot_p2g_grad(const SimP& S, const FrameV& cur, const FrameV& Gn, const FrameV& Gc, int s, const TableP& T,
                                              const int* __restrict__ pool_idx,
                                              const TileO& to, const float4* __restrict__ gg_in, int* slow, const AgentP& agent,
                                              const InjectP& inj, int f, int tofs, int used, const P2GRaw& pre, const GradDst& D, const float* tl = nullptr,
                                              int gofs = 0, bool primary = true, PState* keep = nullptr) {
    if (!PRE) used = cur.used[s];
    if (used) { used_particle_p2g_grad<TILE, GENERAL, PRE, NOSTASH, G, AR, KEEP>(S, cur, Gn, Gc, s, T.info, to, gg_in, slow, tofs, pre, D, tl, gofs, primary, keep); return true; }
    if ((G > 1 || KEEP) && !primary) return false;      // (a split wave: once per particle)
    // the copy f -> f+1 of an unused particle passes its adjoint straight through (mpm:551)
    // Compact adjoints (FrameV::iso = 2) are a matter of the USED slots: what an unused particle passes on is whatever was seeded on it -- nothing says
    // that is isotropic --, so the slots of particles unused in an adjoint's frame always hold all nine words.  The incoming slot is such a slot unless
    // this very substep injects the particle (then frame f + 1 has it in use and the used path of substep f + 1 wrote the slot: a third of the trace
    // on the diagonal is that adjoint exactly, the adjoint of an isotropic F being isotropic).
    bool injected = false;
    if (!(KEEP && TILE) && inj.on) { const int j = pool_idx[T.pid_of_slot[s]] - inj.act_id; injected = j >= 0 && j < inj.flux; }
    FrameV Gn_f = Gn, Dg_f = D.G;
    if (!injected) Gn_f.iso = 0;
    Dg_f.iso = 0;
    PState g; load_xvC(Gn, s, g); load_F(Gn_f, s, g.F);
    if (Gn_f.iso == 2) { const float t = g.F.a[2][2] * (1.f / 3.f); g.F.a[0][0] = g.F.a[1][1] = g.F.a[2][2] = t; }
    const int sd = D.slot(s);
    store_xvC(D.G, sd, g.x, g.v, g.C); store_F(Dg_f, sd, g.F);
    if (!(KEEP && TILE) && inj.on) {
        int j = pool_idx[T.pid_of_slot[s]] - inj.act_id;
        if (j >= 0 && j < inj.flux) {                      // x[f+1,pid] = offset + pos[f] + R(q) inject_p
            const EffP& e = agent.e[agent.inj];
            float* gp = e.gpos + f * 3;
            atomicAdd(gp + 0, g.x[0]); atomicAdd(gp + 1, g.x[1]); atomicAdd(gp + 2, g.x[2]);
            // inject_p and inject_v are rotated by quat[f] (injector.py:92-96): one forward-mode pass per quaternion component
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                Dual q[4], ip[3], iv[3], rp[3], rv[3];
                for (int d = 0; d < 4; d++) q[d] = Dual(e.quat[f * 4 + d], d == k ? 1.f : 0.f);
                for (int d = 0; d < 3; d++) { ip[d] = Dual(e.inject_p[d]); iv[d] = Dual(e.inject_v[d]); }
                t_quat_rotate(ip, q, rp); t_quat_rotate(iv, q, rv);
                const float c = g.x[0] * rp[0].d + g.x[1] * rp[1].d + g.x[2] * rp[2].d + g.v[0] * rv[0].d + g.v[1] * rv[1].d + g.v[2] * rv[2].d;
                if (c != 0.f) atomicAdd(&e.gquat[f * 4 + k], c);
            }
        }
    }
    return false;
}

// p2g.grad + svd_grad + compute_F_tmp.grad + AgentInjector.act_kernel.grad + process_unused_particles.grad (mpm:551)
// + Effector.move_kernel.grad on one thread
template <bool GENERAL, int MINW>
__device__ __forceinline__ void p2g_grad_body(SimP S, float* fr_cur, float* Gn_, float* Gc_, TableP T,
                                                 const int* __restrict__ pool_idx,
                                                 const float4* __restrict__ gg_in, int* blk_count, int* slow, AgentP agent,
                                                 InjectP inj, int act, int f, float* Gd_, const int* __restrict__ to_slot, int fiso) {      // fiso: bit 0 frame f's F compact, bit 1 the incoming adjoint's, bit 2 the outgoing one's (FrameV::iso)
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) {
        *blk_count = 0;
        if (act) for (int i = agent.n - 1; i >= 0; i--) effector_move_grad(agent.e[i], f);
    }
    FrameV cur = frame_view(fr_cur, S.Np, 0, (!GENERAL && (fiso & 1)) ? 1 : 0);
    FrameV Gn = frame_view(Gn_, S.Np, 0, (!GENERAL && (fiso & 2)) ? 2 : 0), Gc = frame_view(Gc_, S.Np, S.wt & 8);
    const GradDst D = {frame_view(Gd_, S.Np, S.wt & 8, (!GENERAL && (fiso & 4)) ? 2 : 0), to_slot, T.pid_of_slot};
    TL(S, 0);
    // The SVD-free build walks the scatter kernels' list, quad units included (round 4): a quad's wave has its own 8 KB tile -- the first two
    // in s_tile, the other two where the pair units keep their stash, whose part C and F are read from the frame again instead (NOSTASH) --,
    // so that where the water has come apart the launch is two rounds of workgroups instead of three or four.  The SVD build keeps the
    // pairs-only list (its stash holds U and V as well: 36 KB).
    constexpr bool QLIST = !GENERAL;
    // (which of the two lists: the sort's decision, meta[16] -- the first unit of both is asked for together with it)
    Unit un = unit_load<!QLIST>(T, blockIdx.x);                     // this workgroup's first unit, asked for together with meta
    bool useq = QLIST;
    if (QLIST) {
        const Unit un_p = unit_load<true>(T, blockIdx.x);
        useq = T.meta[16] != 0;
        if (!useq) un = un_p;
    }
    const int n_slots = T.meta[useq ? 5 : 9];
    bool prev_quad = false;                                   // (unit_enter)
    for (int wg = blockIdx.x; wg < n_slots; wg += gridDim.x) {
        if (wg != (int)blockIdx.x) { if (useq) un = unit_load<false>(T, wg); else un = unit_load<true>(T, wg); }
        if (un.a.z == -2) continue;
        if (QLIST && un.a.z >= 0 && (un.a.w & QUAD_BIT)) {
            const PairCtx pc = unit_ctx(un);
            unit_enter(true, prev_quad);
            const int4 it = pc.it;
            const TileO to = tile_origin(it.x);
            float* tl = pc.ti < 2 ? s_tile + pc.ti * 4 * TILE_N : s_stash_l + (pc.ti - 2) * 4 * TILE_N;
            if (pc.live) tile_nodes_n<8>(pc.t0, 64, [&](int l) { return tile_node_load(to, S, gg_in, l); },
                                         [&](int l, const float4 v) { tl[l] = v.x; tl[TILE_N + l] = v.y; tl[2 * TILE_N + l] = v.z; tl[3 * TILE_N + l] = v.w; });
            unit_sync(true);
            P2GRaw no_pre;
            {   // (a wave with few particles gives each of them three or nine lanes: lane_split)
                const LaneSplit ls = lane_split(__builtin_amdgcn_readfirstlane(min(64, it.z)), FE_SPLIT_PGG && (S.lsplit & 4) != 0);
                const int i = ls.p;
                if (ls.ok && i < it.z) {
                    if (ls.G == 1) slot_p2g_grad<true, GENERAL, false, true, 1>(S, cur, Gn, Gc, it.y + i, T, pool_idx, to, gg_in, slow, agent, inj, f, 0, 0, no_pre, D, tl);
                    else if (ls.G == 3) slot_p2g_grad<true, GENERAL, false, true, 3>(S, cur, Gn, Gc, it.y + i, T, pool_idx, to, gg_in, slow, agent, inj, f, 0, 0, no_pre, D, tl, ls.gofs, ls.primary);
                    else slot_p2g_grad<true, GENERAL, false, true, 9>(S, cur, Gn, Gc, it.y + i, T, pool_idx, to, gg_in, slow, agent, inj, f, 0, 0, no_pre, D, tl, ls.gofs, ls.primary);
                }
            }
            unit_sync(true);
            continue;
        }
        if (un.a.z >= 0) {
            const PairCtx pc = QLIST ? unit_ctx(un) : pair_ctx(un);       // (a pair unit either way)
            if (QLIST) unit_enter(false, prev_quad);
            const int4 it = pc.it;
            const TileO to = tile_origin(it.x);
            TL(S, 1);
            load_tile4(to, S, gg_in, pc);
            __syncthreads();
            TL(S, 2);
            // (asking for the particle's state ahead of the tile load and its barrier -- the PRE form of slot_p2g_grad -- was measured:
            // falling 15.8 -> 16.4 us, layer 20.5 -> 21.1, splash 30.0 -> 29.7; the same move pays in k_p2g, where nothing precedes it)
            P2GRaw no_pre;
            {   // (a wave with few particles gives each of them three or nine lanes: lane_split; the SVD build keeps one lane per particle)
                const int wbase = pc.i & 64;                     // this wave's first particle within the item
                const LaneSplit ls = lane_split(__builtin_amdgcn_readfirstlane(min(64, max(0, it.z - wbase))), FE_SPLIT_PGG && !GENERAL && (S.lsplit & 4) != 0);
                const int i = wbase + ls.p, tofs = pc.ti * 4 * TILE_N;
                if (ls.ok && i < it.z) {
                    if (GENERAL || ls.G == 1) slot_p2g_grad<true, GENERAL, false, false, 1>(S, cur, Gn, Gc, it.y + i, T, pool_idx, to, gg_in, slow, agent, inj, f, tofs, 0, no_pre, D);
                    else if (ls.G == 3) slot_p2g_grad<true, GENERAL, false, false, 3>(S, cur, Gn, Gc, it.y + i, T, pool_idx, to, gg_in, slow, agent, inj, f, tofs, 0, no_pre, D, nullptr, ls.gofs, ls.primary);
                    else slot_p2g_grad<true, GENERAL, false, false, 9>(S, cur, Gn, Gc, it.y + i, T, pool_idx, to, gg_in, slow, agent, inj, f, tofs, 0, no_pre, D, nullptr, ls.gofs, ls.primary);
                }
            }
            TL(S, 3);
            __syncthreads();
            TL(S, 4);
        } else {
            // a tail unit of THIS kernel does use LDS -- every used particle parks C and F in its stash column --, and two of a quad unit's tiles
            // lie in the stash: behind a quad unit the workgroup meets at a barrier first, as before a pair unit
            if (QLIST) unit_enter(false, prev_quad);
            const int s = un.a.y + tid;
            TileO none = {0, 0, 0};
            P2GRaw none_pre;
            if (s < S.N) slot_p2g_grad<false, GENERAL, false>(S, cur, Gn, Gc, s, T, pool_idx, none, gg_in, slow, agent, inj, f, 0, 0, none_pre, D);
        }
    }
}
struct P2GGradArgs { SimP S; float* fr_cur; float* Gn_; float* Gc_; TableP T; const int* pool_idx; const float4* gg_in; int* blk_count; int* slow; AgentP agent; InjectP inj; int act; int f; float* Gd_; const int* to_slot; int fiso; };
template <bool GENERAL, int MINW>
__global__ FE_KALIGN __launch_bounds__(WG, MINW) void k_p2g_grad(SimP S, float* fr_cur, float* Gn_, float* Gc_, TableP T, const int* pool_idx, const float4* gg_in, int* blk_count, int* slow, AgentP agent, InjectP inj, int act, int f, float* Gd_, const int* to_slot, int fiso) { p2g_grad_body<GENERAL, MINW>(S, fr_cur, Gn_, Gc_, T, pool_idx, gg_in, blk_count, slow, agent, inj, act, f, Gd_, to_slot, fiso); }
template <bool GENERAL, int MINW>
__global__ FE_KALIGN __launch_bounds__(WG, MINW) void k_p2g_grad_b(Batch<P2GGradArgs> B) { const P2GGradArgs& A = B.a[blockIdx.y]; p2g_grad_body<GENERAL, MINW>(A.S, A.fr_cur, A.Gn_, A.Gc_, A.T, A.pool_idx, A.gg_in, A.blk_count, A.slow, A.agent, A.inj, A.act, A.f, A.Gd_, A.to_slot, A.fiso); }

// -----------------------------------------------------------------------------------------
// k_pgg_g2pg (option "fuse_bwd"): substep f's p2g_grad and, behind it in the same launch, substep f - 1's g2p_grad -- the reverse sweep's mirror of k_g2p_p2g.
// The adjoints of x, v and C of frame f are produced by the first and consumed by the second: they stay in registers (60 bytes per particle not written,
// 60 not read back; F's adjoint is stored, the next launch's p2g_grad part reads it), and a backward substep is two launches instead of three.  Same unit
// list as k_g2p_grad2 (quad units; waves are not split here -- option lane_split's bit 2 is k_g2p_grad2's); the two parts work through
// ONE 36 KB arena of LDS one after the other (s_bw).  The SVD-free build only; the host fuses where nothing lies between the two kernels and the forward
// pass stored grid[f - 1] (substep_bwd).  x v C of frame f's adjoint go to memory only where somebody else reads them: a particle that was not in use in
// frame f - 1 (Injector.act's adjoint reads it in the next launch) or whose stencil there is off the grid.
// -----------------------------------------------------------------------------------------
struct BwdFuseP { float* fr_prev; float* Gp_; const float4* g_out; float* gg_out; float4* slab; };
// GENERAL: the SVD build -- the pairs-only list (its p2g_grad part has no quad units: the stash holds U and V as well), arena 2.
template <int MINW, bool GENERAL = false>
__device__ __forceinline__ void p2g_grad_g2p_grad_body(SimP S, float* fr_cur, float* Gn_, float* Gc_, TableP T, const int* __restrict__ pool_idx,
                                                         const float4* __restrict__ gg_in, int* blk_count, int* slow, AgentP agent, InjectP inj, int act, int f,
                                                         int fiso, BwdFuseP B, GridStore GS) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) {
        *blk_count = 0;                                       // (substep f's list; substep f - 1's has a counter of its own: bcount)
        if (act) for (int i = agent.n - 1; i >= 0; i--) effector_move_grad(agent.e[i], f);
    }
    VoutSrc V; V.g_out = B.g_out; V.store = GS.data + (size_t)(f - 1) * GS.cap * GS_BLK; V.blk_slot = T.blk_slot;
    constexpr int AR = GENERAL ? 2 : 1;
    FrameV cur = frame_view(fr_cur, S.Np, 0, (!GENERAL && (fiso & 1)) ? 1 : 0), prev = frame_view(B.fr_prev, S.Np);
    FrameV Gn = frame_view(Gn_, S.Np, 0, (!GENERAL && (fiso & 2)) ? 2 : 0);
    FrameV Gc = frame_view(Gc_, S.Np, S.wt & 8, (!GENERAL && (fiso & 4)) ? 2 : 0);      // the adjoint of frame f: its position part so far is read, F (and, where needed, x v C) written
    FrameV Gp = frame_view(B.Gp_, S.Np, S.wt & 4);                           // the adjoint of frame f - 1: the position part g2p_grad leaves
    const GradDst D = {Gc, nullptr, T.pid_of_slot};
    float* const t4 = lds_tile4<AR>();
    double* const acc3 = lds_acc3<AR>();
    // (every unit's record is asked for at the head of its own round -- the first one together with the list's length, as in the other kernels, the one behind
    //  the last unit in vain: carried into the loop from in front of it, the record's twelve words lived in vector registers across the whole loop, spilled)
    const int n_slots = T.meta[GENERAL ? 9 : 5];
    bool prev_quad = false, any_tail = false;
    for (int wg = blockIdx.x; ; wg += gridDim.x) {
        const Unit un = unit_load<GENERAL>(T, wg);
        if (wg >= n_slots) break;
        if (un.a.z == -2) continue;
        if (un.a.z >= 0) {
            const PairCtx pc = GENERAL ? pair_ctx(un) : unit_ctx(un);
            unit_enter(pc.quad, prev_quad);
            const int4 it = pc.it;
            const TileO to = tile_origin(it.x);
            const int wbase = pc.quad ? 0 : (pc.i & 64);
            const int cnt = __builtin_amdgcn_readfirstlane(min(64, max(0, it.z - wbase)));
            const LaneSplit ls = lane_split(cnt, false);
            const int i = wbase + ls.p;
            const bool has = ls.ok && i < it.z;
            const int s = it.y + (has ? i : 0);
            const int nbr_entry = neighbour_entry(T.blk_slot, S.nb, it.x);
            // ---- part 1: p2g_grad of substep f (p2g_grad_body), one lane per particle -- all the lanes of a split particle
            float* tl = t4 + pc.ti * 4 * TILE_N;                 // (a quad's third and fourth tile lie where the pair units keep their stash, as in k_p2g_grad)
            if (pc.live) tile_nodes<!GENERAL>(pc.t0, pc.nth, [&](int l) { return tile_node_load(to, S, gg_in, l); },
                                              [&](int l, const float4 v) { tl[l] = v.x; tl[TILE_N + l] = v.y; tl[2 * TILE_N + l] = v.z; tl[3 * TILE_N + l] = v.w; });
            unit_sync(pc.quad);
            PState g;
            bool kept = false;
            P2GRaw no_pre;
            if (has) {
                if (pc.quad) kept = slot_p2g_grad<true, GENERAL, false, true, 1, AR, true>(S, cur, Gn, Gc, s, T, pool_idx, to, gg_in, slow, agent, inj, f, 0, 0, no_pre, D, tl, 0, ls.primary, &g);
                else kept = slot_p2g_grad<true, GENERAL, false, false, 1, AR, true>(S, cur, Gn, Gc, s, T, pool_idx, to, gg_in, slow, agent, inj, f, pc.ti * 4 * TILE_N, 0, no_pre, D, nullptr, 0, ls.primary, &g);
            }
            if (!kept) {                                         // (behind the call, not in front of it: fifteen zeros would otherwise be kept through the whole p2g_grad part)
#pragma unroll
                for (int a = 0; a < 3; a++) { g.x[a] = 0.f; g.v[a] = 0.f; }
                g.C = m3_zero();
            }
            // (pinned: the adjoint is fifteen values HERE -- left alone, the arithmetic that makes them sinks towards its uses in the g2p_grad part and its
            //  forty inputs stay live, i.e. spilled, across the tile load in between)
            asm volatile("" : "+v"(g.x[0]), "+v"(g.x[1]), "+v"(g.x[2]), "+v"(g.v[0]), "+v"(g.v[1]), "+v"(g.v[2]));
            asm volatile("" : "+v"(g.C.a[0][0]), "+v"(g.C.a[0][1]), "+v"(g.C.a[0][2]), "+v"(g.C.a[1][0]), "+v"(g.C.a[1][1]), "+v"(g.C.a[1][2]), "+v"(g.C.a[2][0]), "+v"(g.C.a[2][1]), "+v"(g.C.a[2][2]));
            unit_sync(pc.quad);                                  // the gathered tile and the stash are done with: the arena changes hands
            // ---- part 2: g2p_grad of substep f - 1 (g2p_grad2_body) with frame f's adjoint in `g`
            // (a quad's wave meets no barrier: its words of the g2p_grad part lie INSIDE the tile it gathered from in the p2g_grad part -- floats
            //  [4 ti TILE_N, + 3 TILE_N) of the arena, a negative offset from the accumulators' base for the first two waves -- never in another wave's)
            const int tofs = pc.quad ? pc.ti * 4 * TILE_N - 6 * TILE_N : pc.ti * 3 * TILE_N;      // (floats / doubles of a pair's tiles, words of a quad's one)
            const int u0 = prev.used[s];
            const float4 a00 = prev.A0[s];
            float* gt = pc.quad ? (float*)acc3 + tofs : lds_tile3<AR>() + tofs;
            g2p_grad_load_tile2<!GENERAL, GENERAL ? 2 : 4>(to, S, B.g_out, V.store, nbr_entry, pc, gt);      // (fifteen adjoints are live across this load: fewer nodes in flight at a time)
            if (!pc.quad && pc.live) for (int l = pc.t0; l < 3 * TILE_N; l += pc.nth) acc3[tofs + l] = 0.0;
            unit_sync(pc.quad);
            float inv = 1.f;
            bool wshell = false;
            {
                const bool used = has && u0 != 0;
                float x[3] = {0.f, 0.f, 0.f};
                if (used) { x[0] = a00.x; x[1] = a00.y; x[2] = a00.z; }
                Stencil st;
                stencil_make(x, S.inv_dx, st);
                const bool inside = used && stencil_inside(st, S.n);
                const int lb = inside ? tile_base(to, st) : -1;
                const bool live = lb >= 0;
                if (has && kept && ls.primary && !inside) store_xvC(Gc, s, g.x, g.v, g.C);      // (not in use in frame f - 1, or off the grid there: read from memory by the next launch)
                if (pc.quad) wshell = __any(live && stencil_on_shell(lb));
                if (__any(live)) {
                    if (pc.quad) inv = g2p_grad_particle2<4, true, 1, AR>(S, Gc, Gp, s, live ? lb : 0, st, live, tofs, gt, 0, true, &g);
                    else g2p_grad_particle2<4, false, 1, AR>(S, Gc, Gp, s, live ? lb : 0, st, live, tofs, gt, 0, true, &g);
                } else if (pc.quad && pc.live) {                 // nothing scattered: the hand-over must not see v_out as sums
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (int l = pc.t0; l < 3 * TILE_N; l += pc.nth) ((int*)acc3)[tofs + l] = 0;
                }
                g2p_grad_wave_slow<true>(S, Gc, Gp, s, a00, inside && !live && ls.primary, used && !inside && ls.primary, V, B.gg_out, slow, GS, &g, it.x, nbr_entry);
            }
            unit_sync(pc.quad);
            if (pc.quad) {
                if (pc.live) quad_handover<3>(S, (const int*)acc3 + tofs, inv, inv, B.slab, pc.slab, B.gg_out, GS, to, nbr_entry, wshell, S.wt & 4);
            } else if (pc.live) for (int l = pc.t0; l < TILE_N; l += pc.nth)
                tile_handover<3>(S, B.slab, pc.slab, B.gg_out, GS, to, nbr_entry, l,
                                 make_float4((float)acc3[tofs + l], (float)acc3[tofs + TILE_N + l], (float)acc3[tofs + 2 * TILE_N + l], 0.f), S.wt & 4);
            unit_sync(pc.quad);
        } else any_tail = true;
    }
    // The tail units (slots behind the items: loose blocks' particles, unused slots) in a loop of their own behind the tile units: in ONE loop what the two kinds of
    // unit keep in registers is alive across both, and the kernel has none to spare.  (The list has them last; a workgroup walks its few units twice.)
    if (!any_tail) return;
    __syncthreads();                                          // (the stash columns lie where the last unit's tiles do)
    for (int wg = blockIdx.x; wg < n_slots; wg += gridDim.x) {
        const Unit un = unit_load<GENERAL>(T, wg);
        if (un.a.z != -1) continue;
        {
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int s = un.a.y + t_;
            TileO none = {0, 0, 0};
            P2GRaw none_pre;
            if (s < S.N) {
                PState g;
                const bool kept = slot_p2g_grad<false, GENERAL, false, false, 1, AR, true>(S, cur, Gn, Gc, s, T, pool_idx, none, gg_in, slow, agent, inj, f, 0, 0, none_pre, D, nullptr, 0, true, &g);
                if (kept) {
                    if (!prev.used[s]) store_xvC(Gc, s, g.x, g.v, g.C);
                    else g2p_grad_slot_global<true>(S, prev, Gc, Gp, s, V, B.gg_out, agent, f - 1, GS, &g);
                }
            }
        }
    }
}
struct PggArgs { SimP S; float* fr_cur; float* Gn_; float* Gc_; TableP T; const int* pool_idx; const float4* gg_in; int* blk_count; int* slow; AgentP agent; InjectP inj; int act; int f; int fiso; BwdFuseP B; GridStore GS; };
#ifndef FE_FB_WAVES_LESS
#define FE_FB_WAVES_LESS 0      // (scripts/kres.py -DFE_FB_WAVES_LESS=1: the kernel's register peak when the launch bound leaves it room)
#endif
template <int MINW, bool GENERAL = false>
__global__ FE_KALIGN __launch_bounds__(WG, MINW - FE_FB_WAVES_LESS) void k_pgg_g2pg(SimP S, float* fr_cur, float* Gn_, float* Gc_, TableP T, const int* pool_idx, const float4* gg_in, int* blk_count, int* slow, AgentP agent, InjectP inj, int act, int f, int fiso, BwdFuseP B, GridStore GS) { p2g_grad_g2p_grad_body<MINW, GENERAL>(S, fr_cur, Gn_, Gc_, T, pool_idx, gg_in, blk_count, slow, agent, inj, act, f, fiso, B, GS); }
template <int MINW>
__global__ FE_KALIGN __launch_bounds__(WG, MINW - FE_FB_WAVES_LESS) void k_pgg_g2pg_b(Batch<PggArgs> Bt) { const PggArgs& A = Bt.a[blockIdx.y]; p2g_grad_g2p_grad_body<MINW>(A.S, A.fr_cur, A.Gn_, A.Gc_, A.T, A.pool_idx, A.gg_in, A.blk_count, A.slow, A.agent, A.inj, A.act, A.f, A.fiso, A.B, A.GS); }


// =========================================================================================
// block sort (counting sort by 4^3 block of the stencil base)
// =========================================================================================
#ifndef SORT_YFAST
#define SORT_YFAST 0
#endif
#define SORT_HB 4096      // cells in a workgroup's rank window: two y-neighbour blocks are 32 * 64 = 2048 cells apart at 128^3, and water falls in y
// The sort is four launches (round 2: nine to ten, most of them 4-7 us latency chains; round 3: five):
//   k_sort_count        key + rank of every slot, per-cell and per-block counts       (+ the rebuilt table's old block slots are cleared)
//   k_sort_blk_partial  partial sums of the scan over the blocks                      (+ cell starts inside every occupied block, the active list)
//   k_sort_blk_final    the scan -> first slot of every block, work items, pair / single lists
//   k_sort_apply        the permutation                                               (+ unit descriptors and neighbour records)
// Slot order: [particles of DENSE blocks, by cell] [particles of LOOSE blocks, by cell] [unused / outside the grid].
// A block with at most `loose_max` particles is LOOSE: it gets no work item -- no workgroup half, no 16 KB tile, no 27-node scan
// for a handful of lanes, no slab -- and its particles are worked on by the global path of every kernel, 256 to a workgroup, in
// cell order (meta[1] = first such slot).  Where the water has come apart most occupied blocks are of that kind.
//
// Sort key = blocked cell address of the stencil base (block-major, then the 64 cells of the block): slots of one
// block are contiguous (-> work items) AND particles of one cell are adjacent (-> the wavefront segmented scan in
// the scatter kernels merges their contributions before touching LDS).
// histogram + rank.  Keys of one workgroup's 256 slots are nearly always within a narrow range (the previous
// order was sorted too), so ranks come from an LDS histogram (ds_add_rtn_u32) and only one global atomic per
// distinct key per workgroup is issued; keys outside the window fall back to a global atomic.
// Inclusive integer scans in DPP: over the 16 lanes of a row (row_shr 1, 2, 4, 8) and over the wavefront (+ row_bcast 15 into rows 1 and 3,
// row_bcast 31 into rows 2 and 3) -- six VALU instructions.  (__shfl_up is ds_bpermute_b32 + s_waitcnt lgkmcnt(0): the 9 x 6 x 3 of
// them in k_sort_blk_final came out as 162 LDS round trips one after the other, 9 of that kernel's 12.3 us.)
__device__ __forceinline__ int row_iscan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    return x;
}
__device__ __forceinline__ int wave_iscan(int x) {
    x = row_iscan(x);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
__global__ __launch_bounds__(256) void k_sort_count(SimP S, float* fr, int n_pwg, int* key, int* rank, int* cnt, int* bcnt,
                                                    const int* __restrict__ clr_active, const int* __restrict__ clr_meta, int* clr_slot, int* nact) {
    __shared__ int hist[SORT_HB + 1];       // + the sentinel's slot
    __shared__ int kmin;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *nact = 0;              // length of the new order's active list (k_sort_blk_partial's launch counts it)
    if ((int)blockIdx.x >= n_pwg) {          // the table about to be rebuilt: forget the block slots of its previous life
        const int n = clr_meta[2];
        for (int i = (blockIdx.x - n_pwg) * 256 + tid; i < n; i += SORT_CLR_WGS * 256) clr_slot[clr_active[i]] = -1;
        return;
    }
    const int s = blockIdx.x * blockDim.x + tid;
    const bool valid = s < S.N;
    int kk = S.ncell;                                        // sentinel: unused / outside -> tail
    if (valid) {
        FrameV cur = frame_view(fr, S.Np);
        const float4 a0 = cur.A0[s];                         // (asked for together with the flag, not behind it)
        if (cur.used[s]) {
            float x[3] = {a0.x, a0.y, a0.z};
            Stencil st;
            stencil_make(x, S.inv_dx, st);
            if (stencil_inside(st, S.n)) {
                kk = cell_addr(st.base[0], st.base[1], st.base[2], S.nb);
#if SORT_YFAST
                // the 64 cells of a block in y-fastest order (the grid's own layout is z-fastest): gravity is along y in every scene, and a
                // particle that falls into the next cell then sits next to that cell's lanes, in the same wave (wave_sort_dest)
                kk = (kk & ~63) | ((st.base[0] & 3) << 4) | ((st.base[2] & 3) << 2) | (st.base[1] & 3);
#endif
            }
        }
    }
    // The sentinel key (unused / out-of-grid slots -> tail) has its own histogram slot and does not take part in the window's
    // minimum: freshly injected particles are scattered over the tail, and a tail workgroup whose window starts at one of
    // their cell keys used to send its other 255 lanes to cnt[ncell] with one same-address global atomic each (215 us per
    // sort in IceCreamDynamic-v0).
    const bool is_tail = kk == S.ncell;
    if (tid == 0) kmin = 0x7fffffff;
    for (int l = tid; l <= SORT_HB; l += 256) hist[l] = 0;
    __syncthreads();
    if (valid && !is_tail) atomicMin(&kmin, kk);
    __syncthreads();
    const int rel = is_tail ? SORT_HB : kk - kmin;
    const bool local = valid && rel <= SORT_HB && (is_tail || rel < SORT_HB);
    int r = 0;
    if (local) r = atomicAdd(&hist[rel], 1);
    else if (valid) r = atomicAdd(&cnt[kk], 1);
    {   // keys outside the window: their block counts, one atomic per distinct block and wave (a window's worth of particles that
        // moved on together all name the same block: 250 same-address atomics per workgroup took the kernel from 13 to 40 us)
        unsigned long long todo = __ballot(valid && !local);
        const int myb = kk >> 6;
        while (todo) {
            const int b = __builtin_amdgcn_readlane(myb, __builtin_ctzll(todo));
            const unsigned long long same = __ballot(valid && !local && myb == b);
            if ((tid & 63) == __builtin_ctzll(todo)) atomicAdd(&bcnt[b], __popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    // per-block counts of the window (it starts anywhere, so it overlaps up to 65 blocks; block nblk = the tail sentinel): a wave takes
    // a block at a time, a lane one cell -- consecutive LDS words -- and the sum comes out of a DPP scan (65 threads walking 64 cells each
    // were a 64-deep chain of LDS reads; 16 cells per thread were 32-way bank conflicts)
    if (kmin != 0x7fffffff) {
        const int k0 = kmin, base = k0 & ~63, lane = tid & 63;
        for (int j = tid >> 6; j <= SORT_HB / 64; j += 4) {
            const int l = base + j * 64 + lane - k0;
            const int n = wave_iscan((l >= 0 && l < SORT_HB) ? hist[l] : 0);
            if (lane == 63 && n > 0) atomicAdd(&bcnt[(base >> 6) + j], n);
        }
    }
    if (tid == 255 && hist[SORT_HB] > 0) atomicAdd(&bcnt[S.ncell >> 6], hist[SORT_HB]);
    __syncthreads();
    {   // one returning global atomic per distinct cell of the window -- all of a thread's sixteen asked for before the first answer is
        // waited for (in a loop every iteration that had a count in any lane was a round trip of its own: the keys of a workgroup
        // cluster in a few blocks, i.e. in a few of the sixteen iterations, and those came back one after the other)
        const int k0 = kmin;
        int c[SORT_HB / 256], b[SORT_HB / 256];
#pragma unroll
        for (int i = 0; i < SORT_HB / 256; i++) c[i] = hist[tid + 256 * i];
        const int ct = tid == 0 ? hist[SORT_HB] : 0;
        int bt = 0;
#pragma unroll
        for (int i = 0; i < SORT_HB / 256; i++) { b[i] = 0; if (c[i] > 0) b[i] = atomicAdd(&cnt[k0 + tid + 256 * i], c[i]); }
        if (ct > 0) bt = atomicAdd(&cnt[S.ncell], ct);
#pragma unroll
        for (int i = 0; i < SORT_HB / 256; i++) if (c[i] > 0) hist[tid + 256 * i] = b[i];
        if (ct > 0) hist[SORT_HB] = bt;
    }
    __syncthreads();
    if (local) r += hist[rel];
    if (valid) { key[s] = kk; rank[s] = r; }
}

// Work items of a block with n particles: k = ceil(n / ITEM_MAX) items, all full but the last.  The items of a block with several
// items pair up among themselves (pair list: items 2j and 2j + 1 share one LDS tile and one slab); the last one of an odd count is
// the block's LEFTOVER.  A block with one item is a (true) SINGLE.  Leftovers and singles are listed in four classes -- big / small
// (at most quad_max particles: one wave is enough) -- because the unit lists combine them in two ways (build_unit_list).
struct BlkWork { int k, full, single, left, last; };
__device__ __forceinline__ BlkWork block_work(int n, int ITEM_MAX) {
    BlkWork w;
    w.k = (n + ITEM_MAX - 1) / ITEM_MAX;
    w.full = w.k > 1 ? w.k >> 1 : 0;
    w.single = w.k == 1 ? 1 : 0;
    w.left = (w.k > 1 && (w.k & 1)) ? 1 : 0;
    w.last = n - (w.k - 1) * ITEM_MAX;                          // particles of the last item
    return w;
}

// The scan over the blocks, two launches of ceil((nblk + 1) / 1024) workgroups (33 at 128^3): a workgroup owns 1024 consecutive
// blocks, four per thread (one 16-byte load, coalesced; bcnt is padded with zeros); k_sort_blk_partial leaves the workgroup's
// sums, k_sort_blk_final adds up the partials before it (and all of them: the loose particles start behind ALL dense ones), scans
// its own threads and hands out slot ranges, items, pairs, singles and the list of occupied blocks.  nblk + 1 counts instead of the
// n^3 + 1 cell counts the two-launch scan of round 2 went over; the cells are dealt with block by block in k_sort_fill.
// (A single workgroup walking thread-contiguous stretches was tried first: 70 us at 128^3 and 420 us at 256^3 -- every load and
// store instruction of a wave touched 64 different cache lines.)
#define SORT_BLK_WG 1024
#define NSUM 11
#define PART_STRIDE 16
// dense particles, loose particles, items, full pairs, big singles, occupied blocks, small singles of more than 21 particles, big leftovers, small leftovers,
// small singles of 8 .. 21 particles, small singles of at most 7
struct BlkSums { int v[NSUM]; };
// (straight-line selects: with early returns the sums lived in memory -- scratch, then LDS when a seventh one was added -- and the scan
//  stage took 29 us instead of 18)
__device__ __forceinline__ void blk_accumulate(BlkSums& a, int n, int ITEM_MAX, int loose_max, int quad_max) {
    const bool occ = n > 0, loose = occ && n <= loose_max, dense = occ && !loose;
    const BlkWork w = block_work(n > 0 ? n : 1, ITEM_MAX);
    const bool small = w.last <= quad_max;
    a.v[5] += occ ? 1 : 0;
    a.v[1] += loose ? n : 0;
    a.v[0] += dense ? n : 0;
    a.v[2] += dense ? w.k : 0;
    a.v[3] += dense ? w.full : 0;
    a.v[4] += (dense && w.single && !small) ? 1 : 0;
    // the small true singles by size: the quad units take the LAST entries of the order's `singles`, and a unit lasts as long as its
    // slowest wave -- four items of one size class to a unit (lane_split: at most 7, at most 21 particles, the rest), not whatever
    // four blocks follow each other in space
    a.v[6] += (dense && w.single && small && w.last > FE_SPLIT3_MAX) ? 1 : 0;
    a.v[9] += (dense && w.single && small && w.last <= FE_SPLIT3_MAX && w.last > FE_SPLIT9_MAX) ? 1 : 0;
    a.v[10] += (dense && w.single && small && w.last <= FE_SPLIT9_MAX) ? 1 : 0;
    a.v[7] += (dense && w.left && !small) ? 1 : 0;
    a.v[8] += (dense && w.left && small) ? 1 : 0;
}
// the thread's four block counts (one 16-byte load: blk_ask4) and their sums.  The counts of blocks past the grid read as 0 through a
// SELECT, not a branch: the first use of the loaded registers is then unconditional and the one wait for the load sits right there --
// under a branch every later use got its own `s_waitcnt vmcnt(0)`, which in k_sort_blk_final also drained the stores of the block
// before (four store round trips one after the other).
__device__ __forceinline__ int4 blk_ask4(const int* __restrict__ bcnt) { return *(const int4*)(bcnt + blockIdx.x * SORT_BLK_WG + threadIdx.x * 4); }
__device__ __forceinline__ BlkSums blk_sums4(const int4 n4, int nblk, int ITEM_MAX, int loose_max, int quad_max, int n[4]) {
    const int b0 = blockIdx.x * SORT_BLK_WG + threadIdx.x * 4;
    n[0] = b0 < nblk ? n4.x : 0; n[1] = b0 + 1 < nblk ? n4.y : 0; n[2] = b0 + 2 < nblk ? n4.z : 0; n[3] = b0 + 3 < nblk ? n4.w : 0;
    BlkSums m = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
#pragma unroll
    for (int u = 0; u < 4; u++) blk_accumulate(m, n[u], ITEM_MAX, loose_max, quad_max);
    return m;
}
// NSUM sums over the 256 threads of the workgroup: exclusive prefix of this thread in ex[], workgroup totals in tot[]
__device__ __forceinline__ void wg_scan6(const BlkSums& m, int (*sh)[NSUM], int ex[NSUM], int tot[NSUM]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < NSUM; k++) {
        const int incl = wave_iscan(m.v[k]);
        ex[k] = incl - m.v[k];
        if (lane == 63) sh[wave][k] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NSUM; k++) {
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const int t = sh[w][k]; all += t; if (w < wave) before += t; }
        ex[k] += before; tot[k] = all;
    }
    __syncthreads();
}
// Three jobs in one launch, all of which need the block counts and nothing else:
//   workgroups [0, blk_wgs)            the partial sums of the block scan;
//   the next scan_wgs                  where a block's 64 cells start INSIDE the block (exclusive scan of its cell counts; the counts
//                                      are zeroed for the next sort): a wave takes 16 blocks, four at a time -- a lane owns four
//                                      consecutive cells (one 16-byte load) and the 16 lanes of a DPP row one block, so a row-wide
//                                      scan is a block's scan.  k_sort_apply adds the block's first slot (blk_base, k_sort_blk_final).
//                                      (Round 3 did this behind the scan, one wave per OCCUPIED block and one dependent round trip
//                                      per block: 8.1 us of its own launch, k_sort_fill.)
//   the rest                           one thread per block of the grid -- is any of its 27 neighbours occupied?  Then it joins the
//                                      order's active list: every block some tile or some loose particle's stencil can reach (asking
//                                      costs 27 cached loads per block and one atomic per wave, and the list comes out nearly in block
//                                      order; round 2 pushed instead -- 200,000 returning atomics on 25,000 words, 36 us).
// The side jobs (cell starts, active list): true when the workgroup was one of theirs.  ONE (k_sort_blk_scan): the last workgroup of the active-list job to finish
// leaves the list's length in meta[2] (its counter `act_done` is monotonic: it reads act_target - 1 when all the others of this sort have passed).
template <bool ONE>
__device__ __forceinline__ bool sort_side_jobs(int nblk, int nb, const int* __restrict__ bcnt, int blk_wgs, int scan_wgs, int* cnt, int* start, int* nact, int* active, int* blk_slot,
                                               int* act_done, int act_target, int* meta) {
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= blk_wgs + scan_wgs) {
        const int b = (blockIdx.x - blk_wgs - scan_wgs) * 256 + threadIdx.x;
        bool on = false;
        {   // all 27 counts asked for at once, from clamped (always valid) indices, and masked afterwards: written as `on = on || bcnt[..] > 0 || ...`
            // every load sat behind a branch on the one before it -- 27 dependent round trips per wave, the 8 us this launch (and round 3's
            // k_sort_fill) took
            const bool inb = b < nblk;
            const int bb = inb ? b : 0;
            const int bi = bb / (nb * nb), bj = (bb / nb) % nb, bk = bb % nb;
            int v[27];
            unsigned okm = 0;                                   // (the masks applied behind ALL the loads: a use right behind its load is a wait right there)
#pragma unroll
            for (int t = 0; t < 27; t++) {
                const int i2 = bi + t / 9 - 1, j2 = bj + (t / 3) % 3 - 1, k2 = bk + t % 3 - 1;
                const bool ok = (unsigned)i2 < (unsigned)nb && (unsigned)j2 < (unsigned)nb && (unsigned)k2 < (unsigned)nb;
                okm |= (ok ? 1u : 0u) << t;
                v[t] = bcnt[ok ? (i2 * nb + j2) * nb + k2 : bb];
            }
            int any = 0;
#pragma unroll
            for (int t = 0; t < 27; t++) any |= ((okm >> t) & 1u) ? v[t] : 0;
            on = inb && any > 0;
        }
        const unsigned long long wm = __ballot(on);
        if (wm) {                                                  // the wave's entries with ONE returning atomic on the list's length
            int base = 0;
            if (lane == 0) base = atomicAdd(nact, __popcll(wm));
            base = __shfl(base, 0, 64);
            if (on) { const int e = base + __popcll(wm & ((1ull << lane) - 1ull)); active[e] = b; blk_slot[b] = e; }
        }
        // (every block of the grid passes here: the ones that are not on the new list lose whatever slot the table's previous life gave them -- k_sort_count's
        //  trailing workgroups used to walk the old list for that; with the keys counted inside k_g2p there is no launch to carry them)
        if (!on && b < nblk) blk_slot[b] = -1;
        if (ONE) {
            // (no fences: a release fence at agent scope writes the XCD's whole L2 back -- 13 us per sort when this was written with __threadfence().  A wave's atomic on
            //  the length has returned its value before the wave gets here; the counter and the length are read and written at agent scope, past the L2)
            __syncthreads();
            if (threadIdx.x == 0 && __hip_atomic_fetch_add(act_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == act_target - 1)
                meta[2] = __hip_atomic_load(nact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    }
    if ((int)blockIdx.x >= blk_wgs) {
        const int g = (blockIdx.x - blk_wgs) * 4 + (threadIdx.x >> 6);         // this wave's 16 blocks: [16 g, 16 g + 16)
        const int bl = g * 16 + (lane & 15);
        const unsigned occ = (unsigned)__ballot(lane < 16 && bl < nblk && bcnt[bl] > 0);
        if (!occ) return true;
        int4 c[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {                                          // the loads of all four groups before the first scan
            const int b = g * 16 + i * 4 + (lane >> 4);
            c[i] = make_int4(0, 0, 0, 0);
            if (((occ >> (4 * i)) & 15u) && b < nblk) c[i] = *(const int4*)(cnt + (size_t)b * 64 + (lane & 15) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!((occ >> (4 * i)) & 15u)) continue;
            const int b = g * 16 + i * 4 + (lane >> 4);
            const int s0 = c[i].x, s1 = s0 + c[i].y, s2 = s1 + c[i].z, s3 = s2 + c[i].w;
            const int ex = row_iscan(s3) - s3;
            if (b < nblk && ((occ >> (4 * i + (lane >> 4))) & 1u)) {
                *(int4*)(start + (size_t)b * 64 + (lane & 15) * 4) = make_int4(ex, ex + s0, ex + s1, ex + s2);
                if (s3) *(int4*)(cnt + (size_t)b * 64 + (lane & 15) * 4) = make_int4(0, 0, 0, 0);      // ready for the next sort
            }
        }
        return true;
    }
    return false;
}
__global__ __launch_bounds__(256) void k_sort_blk_partial(int nblk, int nb, int ITEM_MAX, int loose_max, int quad_max, const int* __restrict__ bcnt, int* partial,
                                                          int blk_wgs, int scan_wgs, int* cnt, int* start, int* nact, int* active, int* blk_slot) {
    __shared__ int sh[4][NSUM];
    if (sort_side_jobs<false>(nblk, nb, bcnt, blk_wgs, scan_wgs, cnt, start, nact, active, blk_slot, nullptr, 0, nullptr)) return;
    int n[4], ex[NSUM], tot[NSUM];
    const BlkSums m = blk_sums4(blk_ask4(bcnt), nblk, ITEM_MAX, loose_max, quad_max, n);
    wg_scan6(m, sh, ex, tot);
#pragma unroll
    for (int k = 0; k < NSUM; k++) if ((int)threadIdx.x == k) partial[blockIdx.x * PART_STRIDE + k] = tot[k];      // (static indices: tot[threadIdx.x] sends the sums through memory)
}
// (one launch in which every workgroup goes over the whole array again instead of reading partial sums was tried: the six sums need
// a division per block, 35 us for the launch against 18 for these two)
// the scan proper, from this thread's four counts and its share of the workgroups' partial sums (pb: of the workgroups before this one, pa: of all).  nact: where the
// active list's length is read for meta[2], or null (k_sort_blk_scan: the active-list job's last workgroup writes it).
__device__ __forceinline__ void sort_blk_final_body(int (*sh)[NSUM], const int4 n4, const BlkSums& pb, const BlkSums& pa, int nblk, int nb, int ncell, int ITEM_MAX, int loose_max, int quad_max,
                                                    int* cnt, int* start, int4* items, int2* pairs, int* singles, int* singles_c, int2* blk_first, int* blk_base, const int* __restrict__ nact, int* meta) {
    const int tid = threadIdx.x;
    int n[4], ex[NSUM], tot[NSUM];
    const BlkSums m = blk_sums4(n4, nblk, ITEM_MAX, loose_max, quad_max, n);
    int exb[NSUM], before[NSUM], exa[NSUM], total[NSUM];
    wg_scan6(pb, sh, exb, before);
    wg_scan6(pa, sh, exa, total);
    wg_scan6(m, sh, ex, tot);
    // meta: [0] items, [1] first slot behind the dense blocks, [2] active blocks (k_sort_fill), [3] full pairs, [4] big singles, [5] / [9] slots
    // of the two unit lists, [6] occupied blocks, [7] first slot of the tail, [8] small singles, [10] quad units, [11] big / [12] small leftovers, [13] scatter list packed, [14] / [15] work units of the two lists
    if (blockIdx.x == 0 && tid == 0) { meta[0] = total[2]; meta[1] = total[0]; if (nact) meta[2] = *nact; meta[3] = total[3]; meta[4] = total[4]; meta[6] = total[5]; meta[7] = total[0] + total[1]; meta[8] = total[6] + total[9] + total[10];
                                       meta[11] = total[7]; meta[12] = total[8]; }
    const int b0 = blockIdx.x * SORT_BLK_WG + tid * 4;
    if (b0 > nblk) return;
    int D = before[0] + ex[0], L = total[0] + before[1] + ex[1], bi = before[2] + ex[2], bm = before[3] + ex[3];
    // `singles` = [big singles][big leftovers][small leftovers][small singles]: the big ones, the small ones and the leftovers are each
    // one stretch of it, the true singles two (build_unit_list)
    int p_tb = before[4] + ex[4], p_lb = total[4] + before[7] + ex[7], p_ls = total[4] + total[7] + before[8] + ex[8];
    // `singles` keeps the small true singles in block order (the gather kernels' pairs-only list: two spatial neighbours to a workgroup share most
    // of the nodes their tiles load); `singles_c`, the same list for the scatter kernels, has them by size class: [more than 21 particles][8 .. 21][at most 7]
    const int ts0 = total[4] + total[7] + total[8];
    int p_ts = ts0 + before[6] + ex[6], p_t3 = ts0 + total[6] + before[9] + ex[9], p_t9 = ts0 + total[6] + total[9] + before[10] + ex[10];
    int p_bo = ts0 + before[6] + ex[6] + before[9] + ex[9] + before[10] + ex[10];
    int2 bf[4];
    int base[4] = {0, 0, 0, 0};                                 // first slot of the block's particles (its cells follow in order: k_sort_blk_partial's launch)
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int b = b0 + u, nn = n[u];
        bf[u] = make_int2(0, 0);
        if (b > nblk) continue;
        if (b == nblk) { base[u] = total[0] + total[1]; start[ncell] = 0; cnt[ncell] = 0; continue; }     // the tail: behind everything
        if (nn <= 0) continue;
        if (nn <= loose_max) { base[u] = L; L += nn; continue; }
        const BlkWork w = block_work(nn, ITEM_MAX);
        bf[u] = make_int2(bi, w.k);
        base[u] = D;
        const bool small = w.last <= quad_max;
        if (w.single) {
            if (!small) { singles[p_tb] = bi; singles_c[p_tb++] = bi; }
            else {
                singles[p_bo++] = bi;
                if (w.last > FE_SPLIT3_MAX) singles_c[p_ts++] = bi;
                else if (w.last > FE_SPLIT9_MAX) singles_c[p_t3++] = bi;
                else singles_c[p_t9++] = bi;
            }
        }
        if (w.left) { if (small) { singles[p_ls] = bi + w.k - 1; singles_c[p_ls++] = bi + w.k - 1; } else { singles[p_lb] = bi + w.k - 1; singles_c[p_lb++] = bi + w.k - 1; } }
        for (int j = 0; j < w.full; j++) pairs[bm++] = make_int2(bi + 2 * j, bi + 2 * j + 1);
        const int pk = BLK_PACK(b / (nb * nb), (b / nb) % nb, b % nb);       // (the block's coordinates: tile_origin, neighbour_entry)
        for (int o = 0; o < nn; o += ITEM_MAX) items[bi++] = make_int4(pk, D + o, min(ITEM_MAX, nn - o), 0);
        D += nn;
    }
    *(int4*)(blk_base + b0) = make_int4(base[0], base[1], base[2], base[3]);        // (padded like bcnt)
    if (b0 + 3 < nblk) { *(int4*)(blk_first + b0) = make_int4(bf[0].x, bf[0].y, bf[1].x, bf[1].y); *(int4*)(blk_first + b0 + 2) = make_int4(bf[2].x, bf[2].y, bf[3].x, bf[3].y); }
    else for (int u = 0; u < 4; u++) if (b0 + u < nblk) blk_first[b0 + u] = bf[u];
}
__global__ __launch_bounds__(256) void k_sort_blk_final(int nblk, int nb, int ncell, int ITEM_MAX, int loose_max, int quad_max, const int* __restrict__ bcnt, int* cnt, int* start,
                                                        const int* __restrict__ partial, int4* items, int2* pairs, int* singles, int* singles_c, int2* blk_first, int* blk_base, const int* __restrict__ nact, int* meta) {
    __shared__ int sh[4][NSUM];
    const int tid = threadIdx.x;
    const int4 n4 = blk_ask4(bcnt);                  // (on its way while the partial sums are read)
    // the partials of the workgroups before this one, and of all of them
    BlkSums pb = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, pa = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
    for (int w = tid; w < (int)gridDim.x; w += 256) {
#pragma unroll
        for (int k = 0; k < NSUM; k++) { const int t = partial[w * PART_STRIDE + k]; pa.v[k] += t; if (w < (int)blockIdx.x) pb.v[k] += t; }
    }
    sort_blk_final_body(sh, n4, pb, pa, nblk, nb, ncell, ITEM_MAX, loose_max, quad_max, cnt, start, items, pairs, singles, singles_c, blk_first, blk_base, nact, meta);
}
// Round 6: the two launches as ONE (option sort_one_scan, default OFF -- built because VERDICT r5 asked for it, measured: 33.2 us per sort against 32.6 ... 33.3 for the two
// launches at 128^3 / 200k, 114 against 106 at 256^3 / 1M, profiles/r06_ab_sort_one_scan.txt: what the kernel boundary cost, the wait for the counter costs -- one hop to
// publish, one to see it; the first version with __threadfence() cost 13 us per sort, the fence writes the XCD's L2 back --, while the scan's workgroups -- the launch's first ones -- are few enough to be resident together: at most two per CU; 257 at 256^3).
// A scan workgroup publishes its eleven sums (agent-scope stores, waited for; one agent-scope atomic on a counter), waits until the counter says that all
// blk_wgs of this sort have (the counter is monotonic: `ready_target` = what it reads then), reads everybody's sums past the L2 (agent-scope loads: the other XCDs' stores)
// and goes on with the scan from the counts it still holds.  One kernel boundary and the second pass over the counts less per sort; the active list's length reaches
// meta[2] through the last workgroup of that job (sort_side_jobs<true>).
struct SortScanSync { int* ready; int ready_target; int* act_done; int act_target; };
__global__ __launch_bounds__(256) void k_sort_blk_scan(int nblk, int nb, int ncell, int ITEM_MAX, int loose_max, int quad_max, const int* __restrict__ bcnt, int* partial, int blk_wgs, int scan_wgs,
                                                       int* cnt, int* start, int* nact, int* active, int* blk_slot, int4* items, int2* pairs, int* singles, int* singles_c, int2* blk_first,
                                                       int* blk_base, int* meta, SortScanSync Y) {
    __shared__ int sh[4][NSUM];
    if (sort_side_jobs<true>(nblk, nb, bcnt, blk_wgs, scan_wgs, cnt, start, nact, active, blk_slot, Y.act_done, Y.act_target, meta)) return;
    const int tid = threadIdx.x;
    const int4 n4 = blk_ask4(bcnt);
    {
        int n[4], ex[NSUM], tot[NSUM];
        const BlkSums m = blk_sums4(n4, nblk, ITEM_MAX, loose_max, quad_max, n);
        wg_scan6(m, sh, ex, tot);
#pragma unroll
        for (int k = 0; k < NSUM; k++) if (tid == k) __hip_atomic_store(partial + blockIdx.x * PART_STRIDE + k, tot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the sums went through to memory: agent-scope stores; no fence -- see sort_side_jobs)
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(Y.ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int)(__hip_atomic_load(Y.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - Y.ready_target) < 0) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    BlkSums pb = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}, pa = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
    for (int w = tid; w < blk_wgs; w += 256) {
#pragma unroll
        for (int k = 0; k < NSUM; k++) { const int t = __hip_atomic_load(partial + w * PART_STRIDE + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pa.v[k] += t; if (w < (int)blockIdx.x) pb.v[k] += t; }
    }
    sort_blk_final_body(sh, n4, pb, pa, nblk, nb, ncell, ITEM_MAX, loose_max, quad_max, cnt, start, items, pairs, singles, singles_c, blk_first, blk_base, nullptr, meta);
}

// What the substep kernels would otherwise look up through chains of dependent loads, laid out once per sort: the unit list of
// the particle kernels (struct Unit) and, per active-list entry, the item ranges of its block's 27 neighbours (gather_slabs).
// (device function: runs in the extra workgroups of k_sort_apply's launch, and alone for the identity order)
// One unit list out of the sort's lists: nF full pairs (two items of one block, shared tile) and `singles` = [tB big singles][lB big
// leftovers][lS small leftovers][tS small singles] (item indices), in one of two layouts:
//   pack = false   [full pairs] [every leftover alone in its workgroup, the other half idle] [true singles two to a workgroup]
//                  [the last x small singles four to a QUAD unit] [tail units]                  -- round 3's list
//   pack = true    [full pairs] [big ones -- singles and leftovers alike -- then the small ones not in quads, two to a workgroup]
//                  [the last x small ones (leftovers included) four to a QUAD unit] [tail units]
// Packing takes the idle halves out: a leftover has a tile and a slab of its own either way (gather_slabs: the slabs of a block are
// its items first, first + 2, ... -- the leftover of an odd count is one of them).
struct UnitLists { int nF, tB, lB, lS, tS; };
__device__ __forceinline__ int unit_count(const UnitLists& u, bool pack, int x, int n_tail) {
    const int nsing = u.tB + u.lB + u.lS + u.tS;
    return u.nF + (pack ? (nsing - x + 1) >> 1 : u.lB + u.lS + ((u.tB + u.tS - x + 1) >> 1)) + ((x + 3) >> 2) + n_tail;
}
__device__ __forceinline__ void build_unit_list(int gtid, int nth, int N, int xcd_on, const int4* __restrict__ items, const int2* __restrict__ pairs,
                                                const int* __restrict__ singles, int tail_start, const UnitLists L, bool pack, int x, int* n_slots_out, int* n_work_out, UnitRec* units, int units_cap) {
    const int nsing = L.tB + L.lB + L.lS + L.tS, n_left = L.lB + L.lS;
    const int n_alone = pack ? 0 : n_left;                                   // leftovers with a workgroup of their own
    const int n_two = pack ? nsing - x : L.tB + L.tS - x;                    // list entries taken two to a workgroup
    const int n_pairs = L.nF + n_alone + ((n_two + 1) >> 1), n_quads = (x + 3) >> 2, n_tail = (N - tail_start + WG - 1) / WG;
    const int n_work = n_pairs + n_quads + n_tail;
    const int per_xcd = xcd_on >= 2 ? (((n_work + xcd_on - 1) / xcd_on + 7) >> 3) * xcd_on : (n_work + 7) >> 3;      // slots per XCD
    const int n_slots = per_xcd * 8 < units_cap ? per_xcd * 8 : units_cap;      // (units_cap covers the worst case)
    if (gtid == 0) { *n_slots_out = n_slots; *n_work_out = n_work; }
    // entry j of the stretch taken two at a time: all of `singles` when packing, else the true singles (big, then small)
    auto two = [&](int j) { return singles[pack ? j : (j < L.tB ? j : j + n_left)]; };
    for (int w = gtid; w < n_slots; w += nth) {
        const int u = xcd_item(w, per_xcd, xcd_on);
        UnitRec un;
        un.a = make_int4(0, 0, -2, 0); un.b = un.c = un.d = make_int4(0, 0, 0, -1);
        if (u < n_pairs) {
            int ia, ib, same;
            if (u < L.nF) { const int2 pr = pairs[u]; ia = pr.x; ib = pr.y; same = 1; }
            else if (u < L.nF + n_alone) { ia = singles[L.tB + (u - L.nF)]; ib = -1; same = 1; }
            else { const int q = 2 * (u - L.nF - n_alone); ia = two(q); ib = q + 1 < n_two ? two(q + 1) : -1; same = 0; }
            un.a = items[ia]; un.a.w = ia | (same << 30);
            if (ib >= 0) { un.b = items[ib]; un.b.w = ib; }
        } else if (u < n_pairs + n_quads) {
            const int q = nsing - x + 4 * (u - n_pairs), nq = min(4, nsing - q);      // (the last x entries of `singles`: small ones)
            { const int ii = singles[q]; un.a = items[ii]; un.a.w = ii | QUAD_BIT; }
            if (nq > 1) { const int ii = singles[q + 1]; un.b = items[ii]; un.b.w = ii; }
            if (nq > 2) { const int ii = singles[q + 2]; un.c = items[ii]; un.c.w = ii; }
            if (nq > 3) { const int ii = singles[q + 3]; un.d = items[ii]; un.d.w = ii; }
        } else if (u < n_work) un.a = make_int4(0, tail_start + (u - n_pairs - n_quads) * WG, -1, 0);
        units[w] = un;
    }
}
// Two lists.  `units_p` is round 3's pairs-only list, for the GATHER kernels (k_g2p, k_p2g_grad): quads never helped them (k_p2g_grad
// has no room for four 4-plane tiles beside its stash) and a second tile to load per workgroup costs them more than an idle half.
// `units` is what the SCATTER kernels (k_p2g, k_g2p_grad2) walk.  Their launches are ONE round of the chip's resident workgroups
// (`quad_fit` = 4 per CU) while the block falls -- a latency chain, in which a quad's longer chain (one wave zeroes and hands over a
// whole tile) costs +0.3 ... 0.5 us: pairs only.  Where the water has come apart they are several rounds -- throughput, and halving
// the workgroups pays (splash: p2g 43 -> 35 us, g2p_grad 45 -> 37): every small single in a quad (`quad_min_units`).  In between --
// the block hitting the floor, 1,030 ... 2,000 pair units -- round 3 left the list as pairs: the few hundred workgroups beyond the
// first 1,024 started when the first ones retired and the scatter kernels took two chains instead of one (31.8 / 32.3 us against
// 20.9 / 19.0 on the falling block).  There the list is now PACKED (no idle halves) and takes as many quads as it needs to fit one
// round again, if that is enough.
__device__ __forceinline__ void build_units_dev(int gtid, int nth, int nb, int N, int xcd_on, int quad_min_units, int quad_fit, int pack_units, int pgg_quad_min_units, const int4* __restrict__ items, const int2* __restrict__ pairs,
                                                const int* __restrict__ singles, const int* __restrict__ singles_c, const int2* __restrict__ blk_first, const int* __restrict__ active,
                                                int* meta, UnitRec* units, UnitRec* units_p, int units_cap, int2* nbr, int* expected) {
    const int tail_start = meta[1], n_active = meta[2];
    const UnitLists L = {meta[3], meta[4], meta[11], meta[12], meta[8]};
    const int n_tail = (N - tail_start + WG - 1) / WG;
    const int U0 = unit_count(L, false, 0, n_tail);
    bool pack = false; int x = 0;
    if (U0 > quad_fit && quad_fit > 0) {
        const int U1 = unit_count(L, true, 0, n_tail), small = L.lS + L.tS;
        const int need = U1 > quad_fit ? 4 * (U1 - quad_fit) + 4 : 0;       // four small ones as a quad instead of two pairs: one workgroup less
        if (pack_units > 0 && need <= small) { pack = true; x = need; }      // back in one round
        else if (U0 > quad_min_units) { pack = pack_units > 1; x = pack ? small : L.tS; }
        else pack = pack_units > 1;
    } else if (U0 > quad_min_units) x = L.tS;
    if (gtid == 0) { meta[10] = (x + 3) >> 2; meta[13] = pack ? 1 : 0; }       // (fe_get_work_stats)
    // k_p2g_grad walks the scatter list only where quad units pay for IT (option "pgg_quad_min_units"): between ~1,400 and ~1,800 pair units the
    // scatter kernels gain from quads what it loses (a quad's four tiles to load per workgroup) -- impact 21.5 vs 20.3 us, profiles/r05_ab_lane_split.txt
    if (gtid == 0) meta[16] = (x > 0 && U0 > pgg_quad_min_units) ? 1 : 0;
    build_unit_list(gtid, nth, N, xcd_on, items, pairs, singles_c, tail_start, L, pack, x, meta + 5, meta + 14, units, units_cap);
    build_unit_list(gtid, nth, N, xcd_on, items, pairs, singles, tail_start, L, false, 0, meta + 9, meta + 15, units_p, units_cap);
    // neighbour records, four entries of a thread at a time: their block numbers in one round trip, their item ranges in the next (one at
    // a time this loop was two dependent round trips per entry, 5 ... 20 entries per thread -- the longest chain of the sort's last launch)
    for (int t0 = gtid; t0 < n_active * 27; t0 += 4 * nth) {
        int b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int t = t0 + u * nth; b[u] = t < n_active * 27 ? active[t / 27] : 0; }
        int2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u * nth, n = t % 27;
            const int i2 = b[u] / (nb * nb) + n / 9 - 1, j2 = (b[u] / nb) % nb + (n / 3) % 3 - 1, k2 = b[u] % nb + n % 3 - 1;
            v[u] = make_int2(0, 0);
            if (t < n_active * 27 && (unsigned)i2 < (unsigned)nb && (unsigned)j2 < (unsigned)nb && (unsigned)k2 < (unsigned)nb) v[u] = blk_first[(i2 * nb + j2) * nb + k2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int t = t0 + u * nth; if (t < n_active * 27) nbr[t] = v[u]; }
    }
    // the fused grid pass (FG kernels): how many tiles are handed over around each entry -- the slabs of its block's 27 neighbours (a block's items pair up from
    // its first one: gather_slabs) -- is what its arrival word counts up to
    // (only for tables built while the option fuse_grid is on: 27 loads per entry, ~5 us of k_sort_apply at 128^3 / 200k)
    if (expected) for (int e = gtid; e < n_active; e += nth) {
        const int b = active[e], bi = b / (nb * nb), bj = (b / nb) % nb, bk = b % nb;
        int cnt[27];
#pragma unroll
        for (int n = 0; n < 27; n++) {                      // (all 27 asked for together, masked behind the loads: k_sort_blk_partial's active list)
            const int i2 = bi + n / 9 - 1, j2 = bj + (n / 3) % 3 - 1, k2 = bk + n % 3 - 1;
            const bool ok = (unsigned)i2 < (unsigned)nb && (unsigned)j2 < (unsigned)nb && (unsigned)k2 < (unsigned)nb;
            cnt[n] = ok ? blk_first[(i2 * nb + j2) * nb + k2].y : 0;
        }
        int sum = 0;
#pragma unroll
        for (int n = 0; n < 27; n++) sum += (cnt[n] + 1) >> 1;
        expected[e] = sum;
    }
}
__global__ __launch_bounds__(256) void k_build_units(int nb, int N, int xcd_on, const int4* __restrict__ items, const int2* __restrict__ pairs,
                                                     const int* __restrict__ singles, const int2* __restrict__ blk_first, const int* __restrict__ active,
                                                     int* meta, UnitRec* units, UnitRec* units_p, int units_cap, int2* nbr, int* expected) {
    build_units_dev(blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, nb, N, xcd_on, 0x7fffffff, 0, 0, 0x7fffffff, items, pairs, singles, singles, blk_first, active, meta, units, units_p, units_cap, nbr, expected);
}

// The permutation itself, one pass: slot s of the old order goes to d = start[key] + rank -- its particle id, material record and
// all 25 planes.  Reads are coalesced, writes land near s (the old order was sorted too: particles move less than a cell between
// sorts), so the write combining of the L2 sees them almost in order.  (Round 1: index kernel, copy of the id table, gather kernel.)
#ifndef SORT_UNIT_WGS
#define SORT_UNIT_WGS 512      // (128 until late in round 4: where the water has come apart the neighbour records were the launch's longest chain)
#endif
struct UnitsArgs { int* bcnt; int nb, xcd_on, quad_min_units, quad_fit, pack_units, pgg_quad_min_units; const int4* items; const int2* pairs; const int* singles; const int* singles_c; const int2* blk_first; const int* active; int* meta; UnitRec* units; UnitRec* units_p; int units_cap; int2* nbr; int* expected; };
__global__ __launch_bounds__(256) void k_sort_apply(int N, size_t Np, int n_pwg, const int* __restrict__ key, const int* __restrict__ rank,
                                                    const int* __restrict__ start, const int* __restrict__ blk_base, const int* __restrict__ pid_old, int* pid_new, int* slot_of_pid,
                                                    const float4* __restrict__ pinfo, float4* info_new, float* dst_, float* src_, UnitsArgs U, int uni, int iso) {      // iso: the frame's F is compact (FrameV::iso): two planes less to move
    // the first SORT_UNIT_WGS workgroups (dispatched first: theirs are the longer chains): unit lists and neighbour records
    // (independent of the permutation: both only need what the block scan's two launches left)
    if ((int)blockIdx.x < SORT_UNIT_WGS) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i <= U.nb * U.nb * U.nb; i += SORT_UNIT_WGS * 256) U.bcnt[i] = 0;      // block counts: ready for the next sort
        build_units_dev(blockIdx.x * 256 + threadIdx.x, SORT_UNIT_WGS * 256, U.nb, N, U.xcd_on, U.quad_min_units, U.quad_fit, U.pack_units, U.pgg_quad_min_units, U.items, U.pairs, U.singles, U.singles_c, U.blk_first, U.active, U.meta,
                        U.units, U.units_p, U.units_cap, U.nbr, U.expected);
        return;
    }
    const int s = (blockIdx.x - SORT_UNIT_WGS) * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int kk = key[s];
    const int d = blk_base[kk >> 6] + start[kk] + rank[s];     // the block's first slot + the cell's start inside the block + the rank inside the cell
    const int pid = pid_old[s];
    FrameV q = frame_view(src_, Np), o = frame_view(dst_, Np);
    const float4 a0 = q.A0[s], a1 = q.A1[s], a2 = q.A2[s];
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (!iso) { b0 = q.B0[s]; b1 = q.B1[s]; }
    const float a3 = q.a3[s], a4 = q.a4[s], a5 = q.a5[s], b2 = q.b2[s];
    const int u = q.used[s];
    pid_new[d] = pid;
    if (!uni) info_new[d] = pinfo[pid];                        // the order's slot-indexed material record (TableP::info); nobody reads it in a single-material scene
    slot_of_pid[pid] = d;
    o.A0[d] = a0; o.A1[d] = a1; o.A2[d] = a2; o.a3[d] = a3; o.a4[d] = a4; o.a5[d] = a5;
    if (!iso) { o.B0[d] = b0; o.B1[d] = b1; }
    o.b2[d] = b2; o.used[d] = u;
}

// dst[s] = src[idx[s]] over all 24 planes (+ used when WITH_USED): coalesced writes, gathered 16-byte reads
template <bool WITH_USED>
__global__ __launch_bounds__(256) void k_perm_gather(int N, size_t Np, float* dst_, float* src_, const int* __restrict__ idx) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int o = idx[s];
    FrameV d = frame_view(dst_, Np), q = frame_view(src_, Np);
    d.A0[s] = q.A0[o]; d.A1[s] = q.A1[o]; d.A2[s] = q.A2[o];
    d.a3[s] = q.a3[o]; d.a4[s] = q.a4[o]; d.a5[s] = q.a5[o];
    d.B0[s] = q.B0[o]; d.B1[s] = q.B1[o]; d.b2[s] = q.b2[o];
    if (WITH_USED) d.used[s] = q.used[o];
}
// dst[s] = src[inv_from[pid_to[s]]]: adjoint frame from order `from` to order `to` in one pass
__global__ __launch_bounds__(256) void k_perm_reorder(int N, size_t Np, float* dst_, float* src_, const int* __restrict__ pid_to, const int* __restrict__ inv_from) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int o = inv_from[pid_to[s]];
    FrameV d = frame_view(dst_, Np), q = frame_view(src_, Np);
    // (all nine planes asked for before the first store: written plane by plane -- dst[s] = src[o]; ... -- every load was waited for and stored
    //  before the next one went out, 21.6 us for 40 MB: DESIGN section 11)
    const float4 a0 = q.A0[o], a1 = q.A1[o], a2 = q.A2[o], b0 = q.B0[o], b1 = q.B1[o];
    const float a3 = q.a3[o], a4 = q.a4[o], a5 = q.a5[o], b2 = q.b2[o];
    d.A0[s] = a0; d.A1[s] = a1; d.A2[s] = a2; d.a3[s] = a3; d.a4[s] = a4; d.a5[s] = a5;
    d.B0[s] = b0; d.B1[s] = b1; d.b2[s] = b2;
}
// dst[idx[s]] = src[s]
__global__ __launch_bounds__(256) void k_perm_scatter(int N, size_t Np, float* dst_, float* src_, const int* __restrict__ idx) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int o = idx[s];
    FrameV d = frame_view(dst_, Np), q = frame_view(src_, Np);
    d.A0[o] = q.A0[s]; d.A1[o] = q.A1[s]; d.A2[o] = q.A2[s];
    d.a3[o] = q.a3[s]; d.a4[o] = q.a4[s]; d.a5[o] = q.a5[s];
    d.B0[o] = q.B0[s]; d.B1[o] = q.B1[s]; d.b2[o] = q.b2[s];
}

// =========================================================================================
// small kernels: effectors, loss, state I/O
// =========================================================================================

struct Act6 { float a[8]; };       // an action vector (up to 8 entries: AirCon)

// set_action_kernel + set_velocity (effector.py:218-221, 252-260)
__global__ void k_eff_set_action(EffP e, int s, int s_global, int n_substeps, Act6 act) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int ad = e.action_dim;
    for (int j = 0; j < ad; j++) e.abuf[(size_t)s_global * ad + j] = act.a[j];
    const float nf = (float)n_substeps;
    for (int j = s * n_substeps; j < (s + 1) * n_substeps; j++) {
        for (int k = 0; k < 3; k++) e.v[j * 3 + k] = e.abuf[(size_t)s_global * ad + k] * e.scale_v[k] / nf;
        if (ad > 3) for (int k = 0; k < 3; k++) e.w[j * 3 + k] = e.abuf[(size_t)s_global * ad + k + 3] * e.scale_v[k + 3] / nf;
        if (ad > 6) {                                                                               // aircon.py:211-213
            e.sa[j] = e.abuf[(size_t)s_global * ad + 6] * e.scale_v[6];
            e.ra[j] = e.abuf[(size_t)s_global * ad + 7] * e.scale_v[7];
        }
    }
}
// set_velocity.grad (effector.py:270-274)
__global__ void k_eff_set_action_grad(EffP e, int s, int s_global, int n_substeps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int ad = e.action_dim;
    const float nf = (float)n_substeps;
    for (int j = s * n_substeps; j < (s + 1) * n_substeps; j++) {
        for (int k = 0; k < 3; k++) e.gabuf[(size_t)s_global * ad + k] += e.gv[j * 3 + k] * e.scale_v[k] / nf;
        if (ad > 3) for (int k = 0; k < 3; k++) e.gabuf[(size_t)s_global * ad + k + 3] += e.gw[j * 3 + k] * e.scale_v[k + 3] / nf;
        if (ad > 6) {
            e.gabuf[(size_t)s_global * ad + 6] += e.gsa[j] * e.scale_v[6];
            e.gabuf[(size_t)s_global * ad + 7] += e.gra[j] * e.scale_v[7];
        }
    }
}
// set_action_p_kernel + apply_action_p_kernel (effector.py:223-231, 236-239)
__global__ void k_eff_apply_p(EffP e, Act6 act) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int j = 0; j < e.action_dim; j++) e.abuf_p[j] = act.a[j];
    float xin[3] = {e.abuf_p[0] * e.scale_p[0], e.abuf_p[1] * e.scale_p[1], e.abuf_p[2] * e.scale_p[2]};
    float xn[3], J[3][3];
    boundary_x(e.bnd, xin, xn, J);
    e.pos[0] = xn[0]; e.pos[1] = xn[1]; e.pos[2] = xn[2];
}
// apply_action_p_kernel.grad (effector.py:233-234)
__global__ void k_eff_apply_p_grad(EffP e) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float xin[3] = {e.abuf_p[0] * e.scale_p[0], e.abuf_p[1] * e.scale_p[1], e.abuf_p[2] * e.scale_p[2]};
    float xn[3], J[3][3];
    boundary_x(e.bnd, xin, xn, J);
    for (int d = 0; d < 3; d++) {
        float g = J[0][d] * e.gpos[0] + J[1][d] * e.gpos[1] + J[2][d] * e.gpos[2];
        e.gabuf_p[d] += g * e.scale_p[d];
    }
}
// Effector/Injector.copy_frame, copy_grad (effector.py:164-176, injector.py:174-186)
__global__ void k_eff_copy(EffP e, int src, int dst, int grad) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float *pos = grad ? e.gpos : e.pos, *quat = grad ? e.gquat : e.quat, *v = grad ? e.gv : e.v, *w = grad ? e.gw : e.w;
    for (int j = 0; j < 3; j++) { pos[dst * 3 + j] = pos[src * 3 + j]; v[dst * 3 + j] = v[src * 3 + j]; w[dst * 3 + j] = w[src * 3 + j]; }
    for (int j = 0; j < 4; j++) quat[dst * 4 + j] = quat[src * 4 + j];
    if (grad) { e.gsa[dst] = e.gsa[src]; e.gra[dst] = e.gra[src]; } else { e.sa[dst] = e.sa[src]; e.ra[dst] = e.ra[src]; }     // aircon.py:148-163
}


// =========================================================================================
// MAT_RIGID shape matching (advect / advect_grad, mpm:428-505).  Bodies are few and their particles a small part of
// the scene: one workgroup per body walks the body's particle list, per-body moments are reduced in fp64 through LDS, one
// thread does the 3x3 SVD.  The reference accumulates in fp32 with atomics (mpm:456-478).
// =========================================================================================
struct RigidBody {
    float c0[3], c1[3], R[9], U[9], sig[3], V[9], gH[9];
    float inv_n;         // 1 / bodies_i.n_particles (all particles of the body, used or not: mpm:201)
    int rigid;           // bodies_i.mat_cls == MAT_RIGID
};

// Sum val[0..K) over the workgroup (256 threads) into out[0..K) (LDS); every thread may read out[] after the call.
template <int K>
__device__ __forceinline__ void block_sum(const double (&val)[K], double* out, double* part /* [4][K] */) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < K; k++) {
        double v = val[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wave * K + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) out[threadIdx.x] = part[threadIdx.x] + part[K + threadIdx.x] + part[2 * K + threadIdx.x] + part[3 * K + threadIdx.x];
    __syncthreads();
}

struct RigidLane { int s; float x[3], y[3]; };
// x = x[f], y = x[f] + dt v[f+1] of the i-th particle of a body's list if it is in use (s = -1 otherwise)
__device__ __forceinline__ RigidLane rigid_lane(const SimP& S, const FrameV& cur, const FrameV& nxt, const int* __restrict__ slot_of_pid,
                                                const int* __restrict__ pids, int i, int hi) {
    RigidLane l; l.s = -1;
    if (i < hi) { const int s = slot_of_pid[pids[i]]; if (cur.used[s]) l.s = s; }
    if (l.s >= 0) {
        const float4 a0 = cur.A0[l.s], n0 = nxt.A0[l.s], n1 = nxt.A1[l.s];
        l.x[0] = a0.x; l.x[1] = a0.y; l.x[2] = a0.z;
        l.y[0] = a0.x + S.dt * n0.w; l.y[1] = a0.y + S.dt * n1.x; l.y[2] = a0.z + S.dt * n1.y;
    }
    return l;
}

__device__ __forceinline__ void rigid_H_adjoint(const RigidBody& b, const RigidLane& l, float ga[3], float gb[3]) {
    float a[3], bb[3];
    for (int d = 0; d < 3; d++) { a[d] = l.x[d] - b.c0[d]; bb[d] = l.y[d] - b.c1[d]; }
    for (int d = 0; d < 3; d++) {
        ga[d] = 0; gb[d] = 0;
        for (int e = 0; e < 3; e++) { ga[d] += b.gH[d * 3 + e] * bb[e]; gb[d] += b.gH[e * 3 + d] * a[e]; }
    }
}

// One workgroup per body does the whole chain for its particles (a per-body list of particle ids, fixed at init; slots through
// the order's slot_of_pid): reset_bodies + compute_COM + compute_H (mpm:449-478) reduced in fp64 through LDS, compute_H_svd +
// compute_R on one thread (mpm:480-495), then the rigid branch of advect_kernel (mpm:500-502).  It used to be five N-wide
// launches per substep (and four more in the backward pass) that touched every particle to find the few hundred rigid ones.
// BACKWARD: the forward values are rebuilt, then advect_kernel.grad -> compute_R.grad / compute_H_svd_grad (mpm:485-489) ->
// compute_H.grad -> compute_COM.grad, rewriting the incoming adjoint so that k_g2p_grad's generic advect adjoint finishes it:
//   X = R^T g + H.grad b + H.grad^T a + (COM_t0.grad + COM_t1.grad)/n   (what x.grad[f] receives)
//   W = H.grad^T a + COM_t1.grad/n                                       (what v.grad[f+1] receives, times dt)
//   x.grad[f+1] := X, v.grad[f+1] += dt (W - X)   =>   x.grad[f] += x.grad[f+1]; v.grad[f+1] += dt x.grad[f+1] yields X and dt W.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void k_rigid_body(SimP S, float* fr_f, float* fr_n, float* G1_, const int* __restrict__ slot_of_pid,
                                                    const int* __restrict__ body_start, const int* __restrict__ body_pids, RigidBody* B) {
    __shared__ RigidBody sb;
    __shared__ double s_out[15], s_part[4 * 15];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (!B[b].rigid) return;                                   // (uniform)
    const int lo = body_start[b], hi = body_start[b + 1];
    const int n_iter = (hi - lo + 255) / 256;
    FrameV cur = frame_view(fr_f, S.Np), nxt = frame_view(fr_n, S.Np);
    const double inv_n = B[b].inv_n;
    {   // compute_COM
        double val[6] = {0, 0, 0, 0, 0, 0};
        for (int it = 0; it < n_iter; it++) {
            const RigidLane l = rigid_lane(S, cur, nxt, slot_of_pid, body_pids, lo + it * 256 + tid, hi);
            if (l.s >= 0) for (int d = 0; d < 3; d++) { val[d] += l.x[d] * inv_n; val[3 + d] += l.y[d] * inv_n; }
        }
        block_sum<6>(val, s_out, s_part);
    }
    const double com[6] = {s_out[0], s_out[1], s_out[2], s_out[3], s_out[4], s_out[5]};
    __syncthreads();
    {   // compute_H
        double val[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < n_iter; it++) {
            const RigidLane l = rigid_lane(S, cur, nxt, slot_of_pid, body_pids, lo + it * 256 + tid, hi);
            if (l.s >= 0) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) val[i * 3 + j] += ((double)l.x[i] - com[i]) * ((double)l.y[j] - com[3 + j]);
        }
        block_sum<9>(val, s_out, s_part);
    }
    if (tid == 0) {                                            // compute_H_svd + compute_R
        m3 H, U, V; float sig[3];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H.a[r][c] = (float)s_out[r * 3 + c];
        svd3(H, U, sig, V);
        const m3 R = m3_mul_nt(V, U);
        for (int d = 0; d < 3; d++) { sb.c0[d] = (float)com[d]; sb.c1[d] = (float)com[3 + d]; sb.sig[d] = sig[d]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sb.R[r * 3 + c] = R.a[r][c]; sb.U[r * 3 + c] = U.a[r][c]; sb.V[r * 3 + c] = V.a[r][c]; }
        sb.inv_n = (float)inv_n;
    }
    __syncthreads();
    if (!BACKWARD) {                                           // advect_kernel, rigid branch: x[f+1] = R (x[f] - COM_t0) + COM_t1
        for (int it = 0; it < n_iter; it++) {
            const RigidLane l = rigid_lane(S, cur, nxt, slot_of_pid, body_pids, lo + it * 256 + tid, hi);
            if (l.s < 0) continue;
            float o[3];
            for (int i = 0; i < 3; i++) { o[i] = sb.c1[i]; for (int j = 0; j < 3; j++) o[i] += sb.R[i * 3 + j] * (l.x[j] - sb.c0[j]); }
            float4 n0 = nxt.A0[l.s];
            n0.x = o[0]; n0.y = o[1]; n0.z = o[2];
            nxt.A0[l.s] = n0;
        }
        return;
    }
    FrameV G1 = frame_view(G1_, S.Np);
    {   // advect_kernel.grad, rigid branch: R.grad += g (x - COM_t0)^T, COM_t0.grad -= R^T g, COM_t1.grad += g
        double val[15];
        for (int k = 0; k < 15; k++) val[k] = 0.0;
        for (int it = 0; it < n_iter; it++) {
            const RigidLane l = rigid_lane(S, cur, nxt, slot_of_pid, body_pids, lo + it * 256 + tid, hi);
            if (l.s < 0) continue;
            const float4 g0 = G1.A0[l.s];
            const float g[3] = {g0.x, g0.y, g0.z};
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) val[i * 3 + j] += (double)g[i] * (double)(l.x[j] - sb.c0[j]);
            for (int j = 0; j < 3; j++) {
                float rtg = 0; for (int i = 0; i < 3; i++) rtg += sb.R[i * 3 + j] * g[i];
                val[9 + j] -= (double)rtg; val[12 + j] += g[j];
            }
        }
        block_sum<15>(val, s_out, s_part);
    }
    const double gcom_a[6] = {s_out[9], s_out[10], s_out[11], s_out[12], s_out[13], s_out[14]};
    if (tid == 0) {                                            // compute_R.grad (R = V U^T) and compute_H_svd_grad
        m3 gR, U, V;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { gR.a[r][c] = (float)s_out[r * 3 + c]; U.a[r][c] = sb.U[r * 3 + c]; V.a[r][c] = sb.V[r * 3 + c]; }
        const m3 gV = m3_mul(gR, U), gU = m3_mul_tn(gR, V);
        const float gS[3] = {0.f, 0.f, 0.f};
        const m3 gH = backward_svd(gU, gS, gV, U, sb.sig, V);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) sb.gH[r * 3 + c] = gH.a[r][c];
    }
    __syncthreads();
    {   // compute_H.grad, the part that flows into the COM adjoints
        double val[6] = {0, 0, 0, 0, 0, 0};
        for (int it = 0; it < n_iter; it++) {
            const RigidLane l = rigid_lane(S, cur, nxt, slot_of_pid, body_pids, lo + it * 256 + tid, hi);
            if (l.s < 0) continue;
            float ga[3], gb[3];
            rigid_H_adjoint(sb, l, ga, gb);
            for (int d = 0; d < 3; d++) { val[d] -= (double)ga[d]; val[3 + d] -= (double)gb[d]; }
        }
        block_sum<6>(val, s_out, s_part);
    }
    float gc0[3], gc1[3];
    for (int j = 0; j < 3; j++) { gc0[j] = (float)(gcom_a[j] + s_out[j]); gc1[j] = (float)(gcom_a[3 + j] + s_out[3 + j]); }
    for (int it = 0; it < n_iter; it++) {                      // particle side
        const RigidLane l = rigid_lane(S, cur, nxt, slot_of_pid, body_pids, lo + it * 256 + tid, hi);
        if (l.s < 0) continue;
        float4 g0 = G1.A0[l.s], g1 = G1.A1[l.s];
        const float g[3] = {g0.x, g0.y, g0.z};
        float ga[3], gb[3], X[3], W[3];
        rigid_H_adjoint(sb, l, ga, gb);
        for (int j = 0; j < 3; j++) {
            float rtg = 0; for (int i = 0; i < 3; i++) rtg += sb.R[i * 3 + j] * g[i];
            X[j] = rtg + ga[j] + gb[j] + (gc0[j] + gc1[j]) * sb.inv_n;
            W[j] = gb[j] + gc1[j] * sb.inv_n;
        }
        g0.x = X[0]; g0.y = X[1]; g0.z = X[2];
        g0.w += S.dt * (W[0] - X[0]); g1.x += S.dt * (W[1] - X[1]); g1.y += S.dt * (W[2] - X[2]);
        G1.A0[l.s] = g0; G1.A1[l.s] = g1;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// compute_chamfer_loss_kernel (shapematching_loss.py:80-84)
__global__ __launch_bounds__(256) void k_loss_fwd(SimP S, float* fr, const int* __restrict__ pid_of_slot,
                                                  const float4* __restrict__ pinfo, const float* __restrict__ tgt,
                                                  int matching_mat, float* chamfer_s) {
    __shared__ float part[4];
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if (s < S.N) {
        FrameV cur = frame_view(fr, S.Np);
        if (cur.used[s]) {
            const int pid = pid_of_slot[s];
            if (matching_mat < 0 || load_info(pinfo, pid).mat == matching_mat) {
                float4 a0 = cur.A0[s];
                float d0 = a0.x - tgt[pid * 3], d1 = a0.y - tgt[pid * 3 + 1], d2 = a0.z - tgt[pid * 3 + 2];
                acc = d0 * d0 + d1 * d1 + d2 * d2;
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float t = part[0] + part[1] + part[2] + part[3]; if (t != 0.f) atomicAdd(chamfer_s, t); }
}
// sum_up_loss_kernel (shapematching_loss.py:86-88)
__global__ void k_loss_sum(float* chamfer_s, float* step_loss_s, float weight) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step_loss_s += *chamfer_s * weight;
}
// adjoint of the two kernels above: x.grad[f,p] += 2 (x - tgt) * weight * step_loss.grad[s]
// The adjoint slot may be stored in another particle order than the frame (the reverse sweep leaves the adjoint of a call's first frame in
// the order the NEXT call's first substep works in, substep_bwd): slot s of the adjoint belongs to particle pid_of_slot[s], whose state
// sits in slot frame_slot_of_pid[pid] of the frame (nullptr: the same order).
__global__ __launch_bounds__(256) void k_loss_bwd(SimP S, float* fr, float* G_, const int* __restrict__ pid_of_slot, const int* __restrict__ frame_slot_of_pid,
                                                  const float4* __restrict__ pinfo, const float* __restrict__ tgt,
                                                  int matching_mat, float g) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S.N) return;
    FrameV cur = frame_view(fr, S.Np);
    const int pid = pid_of_slot[s];
    const int sf = frame_slot_of_pid ? frame_slot_of_pid[pid] : s;
    if (!cur.used[sf]) return;
    if (matching_mat >= 0 && load_info(pinfo, pid).mat != matching_mat) return;
    FrameV G = frame_view(G_, S.Np);
    float4 a0 = cur.A0[sf];
    float4 g0 = G.A0[s];
    g0.x += 2.f * (a0.x - tgt[pid * 3]) * g;
    g0.y += 2.f * (a0.y - tgt[pid * 3 + 1]) * g;
    g0.z += 2.f * (a0.z - tgt[pid * 3 + 2]) * g;
    G.A0[s] = g0;
}

// a compact F (FrameV::iso) written out into the full planes: c I of a state frame (scale 1), a third of the trace on the diagonal of an adjoint
// frame (scale 1/3: the adjoint of an isotropic F is isotropic).  Ahead of API calls that hand F out or add to it.
// `used` (adjoint frames): the flags of the frame the adjoint belongs to -- only the used slots are compact, an unused particle's adjoint is kept in full
// (slot_p2g_grad); they are indexed by the FRAME's slots, which the adjoint's slot reaches through its own id table when the two orders differ.
__global__ __launch_bounds__(256) void k_expand_F(int N, size_t Np, float* planes, float scale, const int* __restrict__ used, const int* __restrict__ pid_of_slot,
                                                  const int* __restrict__ frame_slot_of_pid) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    if (used && used[frame_slot_of_pid ? frame_slot_of_pid[pid_of_slot[s]] : s] == 0) return;
    FrameV fr = frame_view(planes, Np);
    const float c = fr.b2[s] * scale;
    fr.B0[s] = make_float4(c, 0.f, 0.f, 0.f); fr.B1[s] = make_float4(c, 0.f, 0.f, 0.f); fr.b2[s] = c;
}
// staging (caller's particle-id order, AoS) <-> planes (slot order).  mask bits: 1 x, 2 v, 4 C, 8 F, 16 used
__global__ __launch_bounds__(256) void k_pack(int N, size_t Np, float* planes, const int* __restrict__ pid_of_slot,
                                              const float* sx, const float* sv, const float* sC, const float* sF,
                                              const int* sused, int mask, int add) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int pid = pid_of_slot[s];
    FrameV fr = frame_view(planes, Np);
    PState p; load_xvC(fr, s, p); load_F(fr, s, p.F);
    if (mask & 1) for (int d = 0; d < 3; d++) p.x[d] = (add ? p.x[d] : 0.f) + sx[pid * 3 + d];
    if (mask & 2) for (int d = 0; d < 3; d++) p.v[d] = (add ? p.v[d] : 0.f) + sv[pid * 3 + d];
    if (mask & 4) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p.C.a[i][j] = (add ? p.C.a[i][j] : 0.f) + sC[pid * 9 + i * 3 + j];
    if (mask & 8) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p.F.a[i][j] = (add ? p.F.a[i][j] : 0.f) + sF[pid * 9 + i * 3 + j];
    store_xvC(fr, s, p.x, p.v, p.C); store_F(fr, s, p.F);
    if (mask & 16) fr.used[s] = sused[pid];
}
__global__ __launch_bounds__(256) void k_unpack(int N, size_t Np, float* planes, const int* __restrict__ pid_of_slot,
                                                float* sx, float* sv, float* sC, float* sF, int* sused, int mask) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int pid = pid_of_slot[s];
    FrameV fr = frame_view(planes, Np);
    PState p; load_xvC(fr, s, p); load_F(fr, s, p.F);
    if (mask & 1) for (int d = 0; d < 3; d++) sx[pid * 3 + d] = p.x[d];
    if (mask & 2) for (int d = 0; d < 3; d++) sv[pid * 3 + d] = p.v[d];
    if (mask & 4) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sC[pid * 9 + i * 3 + j] = p.C.a[i][j];
    if (mask & 8) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sF[pid * 9 + i * 3 + j] = p.F.a[i][j];
    if (mask & 16) sused[pid] = fr.used[s];
}

// stats: mark touched nodes of frame f, then count
__global__ __launch_bounds__(256) void k_stats_mark(SimP S, float* fr, unsigned char* node_mark, unsigned long long* counters) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool used = false;
    if (s < S.N) {
        FrameV cur = frame_view(fr, S.Np);
        used = cur.used[s] != 0;
        if (used) {
            float4 a0 = cur.A0[s];
            float x[3] = {a0.x, a0.y, a0.z};
            Stencil st; stencil_make(x, S.inv_dx, st);
            if (stencil_inside(st, S.n))
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
                    node_mark[cell_addr(st.base[0] + i, st.base[1] + j, st.base[2] + k, S.nb)] = 1;
        }
    }
    unsigned long long b = __ballot(used);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counters[0], (unsigned long long)__popcll(b));
}
__global__ __launch_bounds__(256) void k_stats_count(int ncells, unsigned char* node_mark, unsigned long long* counters) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;    // one wave per 4^3 block
    if (b * 64 >= ncells) return;
    bool t = node_mark[b * 64 + lane] != 0;
    node_mark[b * 64 + lane] = 0;
    unsigned long long m = __ballot(t);
    if (lane == 0 && m) { atomicAdd(&counters[1], (unsigned long long)__popcll(m)); atomicAdd(&counters[2], 1ull); }
}

// =========================================================================================
// host side
// =========================================================================================
enum { KID_P2G = 0, KID_GRID, KID_G2P, KID_P2G_RE, KID_GRID_KEEP, KID_G2P_GRAD, KID_GRID_GRAD, KID_P2G_GRAD, KID_SORT, KID_REORDER_GRAD, KID_SORT_COUNT, KID_SORT_SCAN, KID_SORT_ACTIVE, KID_SORT_PERM, KID_G2P_P2G, KID_PGG_G2PG, KID_COUNT };
static const char* KNAMES[KID_COUNT] = {"p2g", "grid_op", "g2p", "p2g_recompute", "grid_op_keep", "g2p_grad", "grid_op_grad", "p2g_grad", "sort", "reorder_grad", "sort_count", "sort_scan", "sort_active", "sort_perm", "g2p_p2g", "pgg_g2pg"};

struct EffHost {
    EffP p;
    std::vector<int> act_id;        // [L+1]; deterministic, kept on the host (injector.py:29,105)
    std::vector<int> act_range;
};

struct FeEngine {
    FeConfig cfg;
    int N = 0, Np = 0, L = 0, n = 0, nb = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    SimP S;
    float* frames = nullptr; size_t frame_stride = 0;      // (L+2) buffers of FR_WORDS*Np floats: L+1 frames + 1 spare
    std::vector<float*> frame_ptr;                          // frame f -> buffer (a sort swaps a frame with the spare)
    float* grads = nullptr;                                 // 3 x (GR_WORDS*Np + Np) floats: ring of two + 1 spare
    float* grad_ptr[3] = {nullptr, nullptr, nullptr};
    // particle orders ("tables"): id 0 = identity; id 1+f = order produced by the sort at frame f
    struct Table { int* pid = nullptr; float4* info = nullptr; int2* pairs = nullptr; int* singles = nullptr; int4* items = nullptr; int* meta = nullptr; int2* blk_first = nullptr; int* active = nullptr; int* blk_slot = nullptr; int* slot_of_pid = nullptr; UnitRec* units = nullptr; UnitRec* units_p = nullptr; int2* nbr = nullptr; unsigned long long* arrive = nullptr; /* [nblk] arrival words, then int expected[nblk] (fused grid pass) */ bool has_expected = false; /* built while fuse_grid was on */ };
    int loose_max = 0;                                      // blocks with <= this many particles get no work item (option "loose_max")
    std::vector<int> gs_host; bool gs_host_valid = false;   // host copy of gs_flag, refreshed once per backward sweep
    float4* gstore = nullptr; int* gs_flag = nullptr; int gs_cap = 0;     // forward grid store (see GridStore)
    unsigned char *ent_touched = nullptr, *ent_dirty = nullptr; unsigned stamp = 0;    // [nblk] marks of the scatter kernels per active-list entry (see GridStore)
    unsigned char *gs_live = nullptr, *cur_live = nullptr;                // k_grid's record of the entries it worked on: [(L+1) * cap] beside the store, [nblk] current
    float4* slab = nullptr;                                 // one 6^3-node float4 slab per work item (scatter hand-over)
    std::vector<Table> tables;
    std::vector<int> tbl_of_frame;                          // [L+1]
    int gtbl[2] = {-1, -1};                                 // order of each adjoint ring slot; -1 = all zero
    std::vector<char> fiso;                                 // [L+1] frame f's F is stored compactly (FrameV::iso = 1): written by the SVD-free k_p2g
    bool gcompact[2] = {false, false};                      // ... and the adjoint of F in a ring slot (iso = 2): written by k_p2g_grad inside a ranged call
    bool gpartial[2] = {false, false};                      // the slot's x, v, C planes were passed on in registers by a fused launch (k_pgg_g2pg) and never stored: only F's adjoint is in memory.
                                                            // After fe_step_grad(f0, n > 1) that is the state of frame f0 + 1's slot; the API refuses to hand it out
    int n_cus = 256;                                        // compute units of the device (fe_create)
    bool compact_F = true;                                  // option "compact_F"
    int fuse_bwd = 1;                                       // option "fuse_bwd": inside a fe_step_grad call a substep's p2g_grad takes the next substep's g2p_grad along (k_pgg_g2pg)
    bool fuse_g2p = true;                                   // option "fuse_g2p": inside a fe_step call the g2p of a substep runs at the head of the next substep's p2g launch (k_g2p_p2g)
    int fuse_grid = 0;                                      // option "fuse_grid": grid_op rides on the forward scatter launches (FG kernels, fused grid pass); 0 = off (default: measured slower),
                                                            // 1 = where nothing slow is expected (fuse_grid_ok), 2 = wherever it is possible; + 4 = its waves never wait (every entry goes the skipped road: tests)
    FgDev fg_host = {nullptr, nullptr, nullptr, nullptr, nullptr}; FgDev* fg_dev = nullptr;     // the pass' device block
    unsigned fg_started = 0;                                // workgroups of all FG launches so far: what the pass' monotonic counters read before the next one (SimP::fg_base)
    bool tail_used = true;                                  // used particles may sit behind the work items of the current order: injected or edited by the host since the last sort (-> late deposits, fg_final)
    int tbl_bank = 0, last_sorted_f = -1;                   // two banks of table ids, one per sweep over the window (sort_frame)
    int sort_keys_in_g2p = 1;                               // option "sort_keys_in_g2p": the k_g2p launch in front of a sort counts the sort's keys (k_g2p_sortkey), the sort skips k_sort_count
    int keys_frame = -1;                                    // the frame whose keys / ranks / counts the sort's scratch arrays hold (written by k_g2p_sortkey), or -1
    bool sort_scratch_dirty = false;                        // ... they hold counts no sort has consumed yet
    int p2g_grad_waves = 4;                                 // occupancy target of the SVD-free p2g_grad build (tuning)
    int pack_units = 2;                                     // option "pack_units": 0 never, 1 pack the scatter list (no idle halves) when that brings it back into one round, 2 whenever it is more than one round
    int quad_fit = 1024;                                    // option "quad_fit": the workgroups of one resident round (set from the device in fe_create: 4 per CU); 0 = round 3's rule
    int quad_min_units = 1400;                              // option "quad_min_units": quad units only when pairs alone would be more workgroups than this (build_units_dev)
    int pgg_quad_min_units = 1800;                          // option "pgg_quad_min_units": ... and k_p2g_grad takes them only beyond this many (else the pairs-only list)
    int quad = QUAD_MAX;                                    // option "quad_max": single-item blocks of at most this many particles go four to a workgroup (0: never)
    int g2p_grad_v = 3;                                     // build of the G2P adjoint: 3 = split passes, x offset rolled (k_g2p_grad2<4>, default); 2 = split, unrolled completely (<3>) (1, round 2's one-loop kernel, was removed in round 6)
    int item_max = ITEM_MAX_CAP;                            // particles per work item (<= ITEM_MAX_CAP = one half workgroup)
    int sort_interval = 10;                                 // K: re-sort every K substeps (0 = never: global path only)
    size_t items_cap = 0, units_cap = 0;
    int *sort_key = nullptr, *sort_rank = nullptr, *sort_cnt = nullptr, *sort_start = nullptr, *sort_pid = nullptr, *sort_bcnt = nullptr, *sort_partial = nullptr, *sort_base = nullptr, *sort_nact = nullptr;
    int sort_one_scan = 0, scan_ready = 0, scan_act = 0;    // option sort_one_scan (default off: it measured the same at 128^3 and 8 % slower per sort at 256^3): the sort's two scan launches as one (k_sort_blk_scan); what its two monotonic counters read after the last sort
    int* slow_dev = nullptr;
    int* frame_slow_dev = nullptr;                          // set by a slow-path scatter of the current forward substep
    float4* pinfo = nullptr; int* pool_idx = nullptr;
    std::vector<int> mat_host;
    float *g_in = nullptr, *gg_out = nullptr;              // SoA accumulator planes (4 and 3 x ncell floats)
    float4 *g_out = nullptr, *gg_in = nullptr;
    bool all_simple_liquid = false;                         // every particle is an inviscid MAT_LIQUID: SVD-free kernels
    std::vector<SdfP> statics_host; std::vector<float*> statics_vox; SdfP* statics_dev = nullptr;   // static SDF colliders
    struct SmokeState* smoke = nullptr;                     // SmokeField (fe_smoke.h), optional
    unsigned char* hit_dev = nullptr;                                 // contact flags per (frame, slot)
    int* hit_list = nullptr; int* hit_count = nullptr;                // the flagged slots of the frame being differentiated
    NodeWork* node_work = nullptr; int* node_work_count = nullptr;    // grid nodes inside an agent collider (collide_type grid / both)
    int ggrid_cap = 1024;                                  // workgroups of the grid kernels: one round of the chip's resident ones (option "ggrid_cap")
    bool fold_reorder = true;                               // option "fold_reorder": k_p2g_grad writes across a sort boundary in the next substep's order (substep_bwd)
    int wgrid_cap_pgg = 2048;                              // ... and for k_p2g_grad (option "wgrid_cap_pgg")
    int wgrid_cap_g2p = 2048;                              // the same for k_g2p (option "wgrid_cap_g2p")
    int wgrid_cap = 2048;                                  // workgroups of the work-list kernels (option "wgrid_cap")
    int collide_type = 1;                                  // Agent.collide_type (agent.py:17-26): 1 particle, 2 grid, 3 both
    BoundaryP* collector_dev = nullptr; bool has_collector = false; int collector_mat = -1;     // collector_act_kernel (agent_pouring.py:30-41)
    int inject_till = -1; float collide_min_y = -1e30f;    // AgentIceCreamDynamic (agent_icecreamdynamic.py:11,23-43)
    bool prof_fine = false;
    bool has_mesh_effector = false; std::vector<float*> mesh_vox;   // Rigid effectors with an SDF mesh (dynamic.py)
    bool has_rigid = false; int n_bodies = 0;               // MAT_RIGID shape-matching bodies (mpm:176-201)
    RigidBody* bodies_dev = nullptr;                        // [n_bodies]
    int* body_start = nullptr; int* body_pids = nullptr;    // [n_bodies + 1], [n rigid]: the MAT_RIGID particle ids of each body
    int *blk_flag = nullptr, *blk_list = nullptr, *blk_count = nullptr, *err_dev = nullptr;
    float* stage_r = nullptr; int* stage_i = nullptr;       // 24 N floats, N ints
    unsigned char* node_mark = nullptr; unsigned long long* counters = nullptr;
    std::vector<EffHost> effs;
    EffP* effs_dev = nullptr;                               // [FE_MAX_EFF] parameter blocks read by the kernels
    int loss_steps = 0; float *tgt = nullptr, *chamfer = nullptr, *step_loss = nullptr;
    size_t bytes = 0;
    std::string err;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_batch = nullptr;
    hipStream_t own_stream = nullptr;                       // the engine's stream; `stream` is the leader's while a batch call runs
    // per-kernel profiling
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev; std::vector<int> prof_kid; size_t prof_used = 0;
    double prof_ms[KID_COUNT] = {0}; long long prof_n[KID_COUNT] = {0};
#ifdef FE_TIMELINE
    unsigned long long* tl_dev = nullptr;                   // [KID_COUNT][TL_WGS][8] phase stamps of the last launch of each kernel
#endif

    float* frame(int f) { return frame_ptr[f]; }
    float*& spare_frame() { return frame_ptr[L + 1]; }
    float* grad(int f) { return grad_ptr[f & 1]; }
    size_t grad_words() const { return (size_t)GR_WORDS * Np + Np; }
    TableP tableP(int id) const { TableP t; t.pid_of_slot = tables[id].pid; t.info = tables[id].info; t.meta = tables[id].meta; t.active = tables[id].active; t.blk_slot = tables[id].blk_slot; t.units = tables[id].units; t.units_p = tables[id].units_p; t.units_cap = (int)units_cap; t.nbr = tables[id].nbr; t.arrive = tables[id].arrive; return t; }
    int* expected_of(int id) const { return (int*)(tables[id].arrive + (size_t)nb * nb * nb); }
    const int* pid_of(int f) const { return tables[tbl_of_frame[f]].pid; }
};

static std::string g_create_err;

// The engine's stream is created hipStreamNonBlocking, so a plain hipMemcpy (null stream) is not ordered with the
// hipMemsetAsync zero-fills dev_alloc queues on it: with a 90 GB frame buffer still being cleared, the memset of a small
// table used to land AFTER the table had been uploaded.  All uploads go through the engine's stream.
static hipError_t hipMemcpyOnStream(FeEngine* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);

#define FAIL(h, msg) do { (h)->err = (msg); return 1; } while (0)
// every ABI entry runs on its engine's device, whatever the calling thread's current device is (two engines on two GPUs in
// one process; a caller that switched devices after fe_create)
#define FE_ENTRY(h) do { if ((h) != nullptr) (void)hipSetDevice((h)->device); } while (0)
#define HIPCK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return 1; } } while (0)
#define CHECK_FRAME(h, f) do { if ((f) < 0 || (f) > (h)->L) FAIL(h, "frame index out of range"); } while (0)
#define CHECK_EFF(h, e) do { if ((e) < 0 || (e) >= (int)(h)->effs.size()) FAIL(h, "effector index out of range"); } while (0)

static hipError_t hipMemcpyOnStream(FeEngine* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(h->stream);
}

namespace {

template <typename T>
int dev_alloc(FeEngine* h, T** p, size_t count, bool zero = true) {
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIPCK(h, hipMalloc((void**)p, bytes));
    if (zero) HIPCK(h, hipMemsetAsync(*p, 0, bytes, h->stream));
    h->bytes += bytes;
    return 0;
}

BoundaryP to_boundary(const FeBoundary& b) {
    BoundaryP r;
    r.type = b.type;
    for (int i = 0; i < 3; i++) { r.lower[i] = b.lower[i]; r.upper[i] = b.upper[i]; }
    r.cx = b.xz_center[0]; r.cz = b.xz_center[1]; r.radius = b.xz_radius; r.restitution = b.restitution; r.lock_dims = b.lock_dims;
    return r;
}

AgentP agent_params(FeEngine* h) {
    AgentP a; a.n = (int)h->effs.size(); a.inj = 0; a.e = h->effs_dev; a.collide_min_y = h->collide_min_y;
    a.collector = h->has_collector ? h->collector_dev : nullptr; a.collector_mat = h->collector_mat;
    a.hit = h->hit_dev;
    for (size_t i = 0; i < h->effs.size(); i++) if (h->effs[i].p.type == FE_EFF_INJECTOR) a.inj = (int)i;
    return a;
}

inline dim3 pgrid(FeEngine* h) { return dim3((h->N + 255) / 256); }
// work-list kernels loop over (items + tail chunks); the count lives on the device, so launch a bounded grid
// ("wgrid_cap": upper bound; the lower bound keeps sparse scenes -- few particles per item -- from serialising their items)
inline dim3 wgrid(FeEngine* h) { int g = (h->N + 63) / 64 + 8; if (g < 512) g = 512; return dim3(g < h->wgrid_cap ? g : h->wgrid_cap); }
// k_g2p keeps six workgroups per CU resident (80 registers, 12 KB of LDS) where the other particle kernels keep four: its own bound ("wgrid_cap_g2p")
inline dim3 wgrid_g2p(FeEngine* h) { int g = (h->N + 63) / 64 + 8; if (g < 512) g = 512; return dim3(g < h->wgrid_cap_g2p ? g : h->wgrid_cap_g2p); }
inline dim3 wgrid_pgg(FeEngine* h) { int g = (h->N + 63) / 64 + 8; if (g < 512) g = 512; return dim3(g < h->wgrid_cap_pgg ? g : h->wgrid_cap_pgg); }      // k_p2g_grad ("wgrid_cap_pgg")
inline dim3 ggrid(FeEngine* h) { int blocks = h->nb * h->nb * h->nb; int g = (blocks + 3) / 4; return dim3(g < h->ggrid_cap ? g : h->ggrid_cap); }      // (option "ggrid_cap": tests shrink it so that small scenes take the long-list road of the grid kernels)

void prof_drain(FeEngine* h);
void prof_begin(FeEngine* h, int kid) {
#ifdef FE_TIMELINE
    if (!h->tl_dev) { (void)hipMalloc((void**)&h->tl_dev, sizeof(unsigned long long) * KID_COUNT * TL_WGS * 9); }
    h->S.tl = h->tl_dev + (size_t)kid * TL_WGS * 9;
    (void)hipMemsetAsync(h->S.tl, 0, sizeof(unsigned long long) * TL_WGS * 9, h->stream);
#endif
    if (!h->prof_on) return;
    if (h->prof_used >= 8192) prof_drain(h);
    if (h->prof_used + 2 > h->prof_ev.size()) {
        for (int i = 0; i < 256; i++) { hipEvent_t e; (void)hipEventCreate(&e); h->prof_ev.push_back(e); }
    }
    h->prof_kid.push_back(kid);
    (void)hipEventRecord(h->prof_ev[h->prof_used++], h->stream);
}
void prof_end(FeEngine* h) {
    if (!h->prof_on) return;
    (void)hipEventRecord(h->prof_ev[h->prof_used++], h->stream);
}
void prof_drain(FeEngine* h) {
    if (h->prof_used == 0) return;
    (void)hipStreamSynchronize(h->stream);
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]);
        int kid = h->prof_kid[i / 2];
        h->prof_ms[kid] += ms; h->prof_n[kid]++;
    }
    h->prof_used = 0; h->prof_kid.clear();
}

// the injector (at most one, as AgentInjector asserts: agent_injector.py:17-18)
int find_injector(FeEngine* h) {
    for (size_t i = 0; i < h->effs.size(); i++) if (h->effs[i].p.type == FE_EFF_INJECTOR) return (int)i;
    return -1;
}

int make_inject(FeEngine* h, int f, int f_global, int act, bool forward, InjectP& inj) {
    inj.on = 0; inj.act_id = 0; inj.row = 0; inj.flux = 0;
    int ie = find_injector(h);
    if (ie < 0) return 0;
    EffHost& E = h->effs[ie];
    if (!act || (h->inject_till >= 0 && f_global >= h->inject_till)) {       // no injection this substep: act_id carries over
        if (forward) E.act_id[f + 1] = E.act_id[f];
        return 0;
    }
    if (E.act_range.empty()) FAIL(h, "injector has no act_range");
    int row = E.p.locally_random ? f : f_global;
    if (row < 0 || row >= E.p.random_length) FAIL(h, "injector random_vector row out of range");
    if (forward) {
        if (E.act_id[f] + E.p.flux > (int)E.act_range.size()) FAIL(h, "too many particles added");   // agent_injector.py:38-39
        E.act_id[f + 1] = E.act_id[f] + E.p.flux;                                                     // injector.py:105
    }
    inj.on = 1; inj.act_id = E.act_id[f]; inj.row = row; inj.flux = E.p.flux;
    return 0;
}

int check_async(FeEngine* h);

void build_units(FeEngine* h, FeEngine::Table& t) {
    hipLaunchKernelGGL(k_build_units, dim3(256), dim3(256), 0, h->stream, h->nb, h->N, h->S.xcd, t.items, t.pairs, t.singles, t.blk_first, t.active, t.meta,
                       t.units, t.units_p, (int)h->units_cap, t.nbr, (h->fuse_grid & 3) ? (int*)(t.arrive + (size_t)h->nb * h->nb * h->nb) : (int*)nullptr);
    t.has_expected = (h->fuse_grid & 3) != 0;
}
int ensure_table(FeEngine* h, int id) {
    if ((int)h->tables.size() <= id) h->tables.resize(id + 1);
    FeEngine::Table& t = h->tables[id];
    if (t.pid) return 0;
    const size_t nblk = (size_t)h->nb * h->nb * h->nb;
    if (id == 0) t.info = h->pinfo;                            // identity order: slot == particle id
    else if (dev_alloc(h, &t.info, h->Np)) return 1;
    // (pairs: a block with k > 1 items makes ceil(k / 2) pairs -- two from three -- so the bound is the item count, not half of it)
    if (dev_alloc(h, &t.pairs, h->items_cap) || dev_alloc(h, &t.singles, 2 * h->items_cap)) return 1;      // (singles: block order, then the same list by size class)
    if (dev_alloc(h, &t.pid, h->Np) || dev_alloc(h, &t.items, h->items_cap) || dev_alloc(h, &t.meta, 24) ||
        dev_alloc(h, &t.blk_first, nblk) || dev_alloc(h, &t.active, nblk) || dev_alloc(h, &t.blk_slot, nblk) || dev_alloc(h, &t.slot_of_pid, h->Np) ||
        dev_alloc(h, &t.units, h->units_cap, false) || dev_alloc(h, &t.units_p, h->units_cap, false) || dev_alloc(h, &t.nbr, nblk * 27, false) ||
        dev_alloc(h, &t.arrive, nblk + (nblk + 1) / 2)) return 1;       // (arrival words: zero between launches -- the owner of an entry clears it; then `expected`, ints)
    HIPCK(h, hipMemsetAsync(t.blk_slot, 0xff, sizeof(int) * nblk, h->stream));
    if (id == 0) build_units(h, t);                            // the identity order: no items, every slot on the tail
    return 0;
}

// bring adjoint ring slot `slot` into particle order `to` (via the identity order; rare: once per K substeps)
int reorder_grad(FeEngine* h, int slot, int to) {
    const int from = h->gtbl[slot];
    if (from == to || from < 0) { if (from >= 0) h->gtbl[slot] = to; return 0; }
    prof_begin(h, KID_REORDER_GRAD);
    hipLaunchKernelGGL(k_perm_reorder, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, h->grad_ptr[2], h->grad_ptr[slot],
                       h->tables[to].pid, h->tables[from].slot_of_pid);
    std::swap(h->grad_ptr[2], h->grad_ptr[slot]);
    prof_end(h);
    h->gtbl[slot] = to;
    return 0;
}

// The particle order the adjoint slot of frame f is stored in, for whatever accumulates into it (losses, fe_add_grad): a cleared slot
// takes the frame's order; otherwise the slot keeps its own -- round 4 forced it back to the frame's here, one k_perm_reorder pass per
// env step in fluidlab's step_grad flow (the loss of a step lands on the frame a sort follows), which the accumulating kernels can
// do without: they address the slot through its own id table.
int grad_table_for_frame(FeEngine* h, int f) {
    if (h->gtbl[f & 1] < 0) h->gtbl[f & 1] = h->tbl_of_frame[f];
    return h->gtbl[f & 1];
}

// (round 2 switched the blocks flagged "static" here, two launches; an order's active list is now recognised by its own blk_slot)
int use_static_table(FeEngine*, int) { return 0; }

inline int quad_max(FeEngine* h) { return h->quad; }
// counting sort of frame f by 4^3 block; the new order becomes table 1 + f of the current bank.  The bank changes whenever the frame
// number does not grow (a new sweep over the window): the adjoint slots of the sweep before still name that sweep's tables -- until the
// caller resets them, which fluidlab's solver does AFTER the next forward pass (solver.py:36) -- and with one bank the first sort of every
// window found both slots "stored in the order about to be overwritten" and moved them to the identity order first: two k_perm_reorder
// launches (21.6 us each) per window for adjoints nobody reads again.
int sort_frame(FeEngine* h, int f) {
    if (f <= h->last_sorted_f) h->tbl_bank ^= 1;
    h->last_sorted_f = f;
    const int id_new = 1 + f + h->tbl_bank * (h->L + 1), id_old = h->tbl_of_frame[f];
    if (ensure_table(h, id_new)) return 1;
    for (int slot = 0; slot < 2; slot++)            // an adjoint slot still stored in the order about to be overwritten
        if (h->gtbl[slot] == id_new && reorder_grad(h, slot, 0)) return 1;
    FeEngine::Table& tn = h->tables[id_new];
    const int ncell = h->S.ncell;
    const bool fine = h->prof_on && h->prof_fine;          // option "prof_fine": time the sort's stages instead of the whole
    if (!fine) prof_begin(h, KID_SORT);
    const int n_pwg = (int)pgrid(h).x;
    const int nblk = h->nb * h->nb * h->nb;
    if (fine) prof_begin(h, KID_SORT_COUNT);
    // (the keys, ranks and counts of this very frame may be there already: the k_g2p launch that wrote it counted them -- k_g2p_sortkey)
    const bool have_keys = h->keys_frame == f && h->sort_scratch_dirty;
    if (!have_keys && h->sort_scratch_dirty) {                // counts of a frame that was never sorted (the caller went elsewhere): back to zero first
        HIPCK(h, hipMemsetAsync(h->sort_cnt, 0, sizeof(int) * ((size_t)ncell + 1), h->stream));
        HIPCK(h, hipMemsetAsync(h->sort_bcnt, 0, sizeof(int) * (size_t)(((nblk + 1 + SORT_BLK_WG - 1) / SORT_BLK_WG) * SORT_BLK_WG), h->stream));
    }
    h->keys_frame = -1; h->sort_scratch_dirty = false;
    if (!have_keys)
    hipLaunchKernelGGL(k_sort_count, dim3(n_pwg + SORT_CLR_WGS), dim3(256), 0, h->stream, h->S, h->frame(f), n_pwg, h->sort_key, h->sort_rank, h->sort_cnt, h->sort_bcnt,
                       tn.active, tn.meta, tn.blk_slot, h->sort_nact);
    if (fine) { prof_end(h); prof_begin(h, KID_SORT_SCAN); }
    const int blk_wgs = (nblk + 1 + SORT_BLK_WG - 1) / SORT_BLK_WG, scan_wgs = (nblk + 63) / 64, act_wgs = (nblk + 255) / 256;
    if (h->sort_one_scan && blk_wgs <= 2 * h->n_cus) {    // (the scan's workgroups wait for each other: all of them resident -- the launch's first ones, three fit a CU at the kernel's 142 registers)
        h->scan_ready += blk_wgs; h->scan_act += act_wgs;
        const SortScanSync Y = {h->sort_nact + 1, h->scan_ready, h->sort_nact + 2, h->scan_act};
        hipLaunchKernelGGL(k_sort_blk_scan, dim3(blk_wgs + scan_wgs + act_wgs), dim3(256), 0, h->stream, nblk, h->nb, ncell, h->item_max, h->loose_max, quad_max(h), h->sort_bcnt, h->sort_partial,
                           blk_wgs, scan_wgs, h->sort_cnt, h->sort_start, h->sort_nact, tn.active, tn.blk_slot, tn.items, tn.pairs, tn.singles, tn.singles + h->items_cap, tn.blk_first,
                           h->sort_base, tn.meta, Y);
    } else {
        hipLaunchKernelGGL(k_sort_blk_partial, dim3(blk_wgs + scan_wgs + act_wgs), dim3(256), 0, h->stream, nblk, h->nb, h->item_max, h->loose_max, quad_max(h), h->sort_bcnt, h->sort_partial,
                           blk_wgs, scan_wgs, h->sort_cnt, h->sort_start, h->sort_nact, tn.active, tn.blk_slot);
        hipLaunchKernelGGL(k_sort_blk_final, dim3(blk_wgs), dim3(256), 0, h->stream, nblk, h->nb, ncell, h->item_max, h->loose_max, quad_max(h), h->sort_bcnt, h->sort_cnt, h->sort_start, h->sort_partial,
                           tn.items, tn.pairs, tn.singles, tn.singles + h->items_cap, tn.blk_first, h->sort_base, h->sort_nact, tn.meta);
    }
    if (fine) { prof_end(h); prof_begin(h, KID_SORT_PERM); }
    // re-sorting a frame that is already in this table's order reads the id table it rewrites: stage it
    int* pid_dst = id_old == id_new ? h->sort_pid : tn.pid;
    const UnitsArgs U = {h->sort_bcnt, h->nb, h->S.xcd, h->quad_min_units, h->quad_fit, h->pack_units, h->pgg_quad_min_units, tn.items, tn.pairs, tn.singles, tn.singles + h->items_cap, tn.blk_first, tn.active, tn.meta, tn.units, tn.units_p, (int)h->units_cap, tn.nbr, (h->fuse_grid & 3) ? (int*)(tn.arrive + (size_t)nblk) : (int*)nullptr};
    tn.has_expected = (h->fuse_grid & 3) != 0;
    hipLaunchKernelGGL(k_sort_apply, dim3(n_pwg + SORT_UNIT_WGS), dim3(256), 0, h->stream, h->N, (size_t)h->Np, n_pwg, h->sort_key, h->sort_rank, h->sort_start, h->sort_base,
                       h->tables[id_old].pid, pid_dst, tn.slot_of_pid, h->pinfo, tn.info, h->spare_frame(), h->frame(f), U, h->S.uni, h->fiso[f] ? 1 : 0);
    if (id_old == id_new) HIPCK(h, hipMemcpyAsync(tn.pid, h->sort_pid, sizeof(int) * h->Np, hipMemcpyDeviceToDevice, h->stream));
    prof_end(h);
    std::swap(h->frame_ptr[f], h->spare_frame());
    h->tbl_of_frame[f] = id_new;
    return 0;
}

StaticsP statics_p(FeEngine* h) { StaticsP p; p.n = (int)h->statics_host.size(); p.s = h->statics_dev; return p; }

GridStore grid_store(FeEngine* h) {
    GridStore g; g.data = h->gstore; g.flag = h->gs_flag; g.cap = h->gs_cap;
    g.touched = h->ent_touched; g.dirty = h->ent_dirty; g.live = h->gs_live; g.cur = h->cur_live; g.stamp = (unsigned char)(1 + h->stamp % 255);
    return g;
}

// The length of substep f's dynamic block list: two counters, by the parity of f -- k_g2p_p2g adds to substep f's list and, in the same launch, clears
// the one grid_op of substep f - 1 has just read (g2p_body's reset).  Every window of launches that uses a counter ends with the kernel that clears it.
inline int* bcount(FeEngine* h, int f) { return h->blk_count + (f & 1); }
// v_out of substep f / the adjoint grid (d v_in, d m) of substep f: two buffers each, by the parity of f (fused grid pass, FG kernels)
inline float4* gout(FeEngine* h, int f) { return h->g_out + (size_t)(f & 1) * h->S.ncell; }
inline float4* ggin(FeEngine* h, int f) { return h->gg_in + (size_t)(f & 1) * h->S.ncell; }
GridW grid_w(FeEngine* h, int f) {
    GridW g; g.g_in = h->g_in; g.slab = h->slab; g.ncell = h->S.ncell; g.frame_slow = h->frame_slow_dev; g.blk_flag = h->blk_flag; g.blk_list = h->blk_list; g.blk_count = bcount(h, f); g.err = h->err_dev; g.slow = h->slow_dev;
    g.fg = h->fg_dev; g.g_out = gout(h, f);
    return g;
}

// Agent.collide_type (agent.py:17-26): bit 0 = at the particles (g2p), bit 1 = at the grid nodes (grid_op)
inline bool grid_collide(FeEngine* h) { return h->has_mesh_effector && (h->collide_type & 2); }
inline bool particle_collide(FeEngine* h) { return h->has_mesh_effector && (h->collide_type & 1); }

template <bool KEEP>
void launch_grid(FeEngine* h, const TableP& T, int f, const AgentP& ag) {
#define LAUNCH_GRID(ST_, DY_) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid<KEEP, ST_, DY_>), ggrid(h), dim3(256), 0, h->stream, h->S, T, h->slab, h->g_in, gout(h, f), \
                           h->blk_list, bcount(h, f), h->blk_flag, grid_store(h), f, h->frame_slow_dev, statics_p(h), ag)
    if (grid_collide(h)) { if (h->statics_host.empty()) LAUNCH_GRID(false, true); else LAUNCH_GRID(true, true); }
    else { if (h->statics_host.empty()) LAUNCH_GRID(false, false); else LAUNCH_GRID(true, false); }
#undef LAUNCH_GRID
}

// compact-F flags of a forward substep: bit 0 = frame f is stored compactly, bit 1 = frame f + 1 is going to be (the SVD-free kernel, option on)
inline int fwd_iso(FeEngine* h, int f) { return (h->fiso[f] ? 1 : 0) | ((h->all_simple_liquid && h->compact_F) ? 2 : 0); }

// Can the g2p of substep f - 1 run at the head of substep f's p2g launch (k_g2p_p2g)?  Nothing may come between the two: no sort of frame f, no
// effector mesh acting on the gathered velocities, no rigid-body pass over frame f, no collector reading the positions (option "fuse_g2p").
inline bool fusable_fwd(FeEngine* h, int f) {
    return h->fuse_g2p && f > 0 && !(h->sort_interval > 0 && f % h->sort_interval == 0) && !particle_collide(h) && !h->has_rigid && !h->has_collector;
}
// Does grid_op of substep f ride on its scatter launch (the fused grid pass, FG kernels; option "fuse_grid")?  What the pass cannot do: colliders at the
// nodes (the lean grid_op only), blocks without work items (nobody arrives for them), a store that does not hold every entry (late deposits are added to what the
// owner stored), the identity order (no entries at all).  What it can do but slowly (the launch's final wave, alone): used particles behind the work items --
// the injector's, or the host's edits since the last sort -- so with fuse_grid = 1 such launches keep the separate k_grid.
inline bool fuse_grid_possible(FeEngine* h, int t) {
    return (h->fuse_grid & 3) != 0 && h->sort_interval > 0 && t > 0 && h->loose_max == 0 && h->statics_host.empty() && !h->has_mesh_effector &&
           h->gs_cap == h->nb * h->nb * h->nb && h->tables[t].has_expected;      // (a table sorted before the option was switched on has no arrival counts: separate grid kernels until the next sort)
}
inline bool fuse_grid_ok(FeEngine* h, int f) { return fuse_grid_possible(h, h->tbl_of_frame[f]) && ((h->fuse_grid & 3) >= 2 || !h->tail_used); }
// an FG launch: the pass' `started` / `finished` counters are monotonic (nothing to reset between launches); they read fg_started before it and fg_started + workgroups behind it
inline void fg_launch_begin(FeEngine* h) { h->S.fg_base = h->fg_started; h->S.fg_nowait = (h->fuse_grid & 4) ? 1 : 0; }
inline void fg_launch_end(FeEngine* h, dim3 grid) { h->fg_started += grid.x; }
// `g2p_pending`: the caller's last substep left its g2p to this one (fusable_fwd(h, f) held);  `defer_g2p`: leave this substep's to the next one
int substep_fwd(FeEngine* h, int f, int f_global, int act, bool g2p_pending = false, bool defer_g2p = false) {
    h->gs_host_valid = false;
    h->stamp++;                                           // p2g marks, grid_op reads (GridStore)
    InjectP inj;
    if (make_inject(h, f, f_global, act, true, inj)) return 1;
    if (h->sort_interval > 0 && f % h->sort_interval == 0) { if (sort_frame(h, f)) return 1; h->tail_used = false; }
    h->tbl_of_frame[f + 1] = h->tbl_of_frame[f];        // a substep keeps the slot order
    use_static_table(h, h->tbl_of_frame[f]);
    const TableP T = h->tableP(h->tbl_of_frame[f]);
    AgentP ag = agent_params(h);
    const int fiso = fwd_iso(h, f);
    const bool fg = fuse_grid_ok(h, f);                   // (before this substep's injection counts: its particles are used from frame f + 1 on)
    if (inj.on) h->tail_used = true;
    if (fg) fg_launch_begin(h);
    if (g2p_pending) {
        const FuseP FU = {h->frame(f - 1), gout(h, f - 1), h->slow_dev, bcount(h, f - 1)};
        prof_begin(h, KID_G2P_P2G);
        if (fg) {
            const P2GArgs A = {h->S, h->frame(f), h->frame(f + 1), T, h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), h->all_simple_liquid ? fiso : 0, FU};
            if (h->all_simple_liquid) hipLaunchKernelGGL((k_g2p_p2g_fg<false>), wgrid(h), dim3(WG), 0, h->stream, A);
            else hipLaunchKernelGGL((k_g2p_p2g_fg<true>), wgrid(h), dim3(WG), 0, h->stream, A);
        } else if (h->all_simple_liquid)
            hipLaunchKernelGGL((k_g2p_p2g<false>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T,
                               h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), fiso, FU);
        else
            hipLaunchKernelGGL((k_g2p_p2g<true>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T,
                               h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), 0, FU);
        prof_end(h);
    } else {
    prof_begin(h, KID_P2G);
    if (fg) {
        const P2GArgs A = {h->S, h->frame(f), h->frame(f + 1), T, h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), h->all_simple_liquid ? fiso : 0, FuseP{nullptr, nullptr, nullptr, nullptr}};
        if (h->all_simple_liquid) hipLaunchKernelGGL((k_p2g_fg<false>), wgrid(h), dim3(WG), 0, h->stream, A);
        else hipLaunchKernelGGL((k_p2g_fg<true>), wgrid(h), dim3(WG), 0, h->stream, A);
    } else if (h->all_simple_liquid)
        hipLaunchKernelGGL((k_p2g<true, false>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T,
                           h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), fiso);
    else
        hipLaunchKernelGGL((k_p2g<true, true>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T,
                           h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), 0);
    prof_end(h);
    }
    if (fg) fg_launch_end(h, wgrid(h));
    h->fiso[f + 1] = (fiso & 2) != 0;
    if (!fg) {
    prof_begin(h, KID_GRID);
    launch_grid<false>(h, T, f, ag);
    prof_end(h);
    }
    if (defer_g2p) return 0;                              // (fusable_fwd(h, f + 1): no rigid-body pass either)
    prof_begin(h, KID_G2P);
    // frame f + 1 is a frame the order is rebuilt on: this launch counts the sort's keys as it writes the positions (the sort then skips k_sort_count).
    // Not with rigid bodies (their particles are moved again behind this launch).
    const bool sortkey = h->sort_keys_in_g2p && h->sort_interval > 0 && (f + 1) % h->sort_interval == 0 && !h->has_rigid;
    if (sortkey) {
        if (h->sort_scratch_dirty) {                          // (counts of a frame that was never sorted)
            HIPCK(h, hipMemsetAsync(h->sort_cnt, 0, sizeof(int) * ((size_t)h->S.ncell + 1), h->stream));
            HIPCK(h, hipMemsetAsync(h->sort_bcnt, 0, sizeof(int) * (size_t)((((size_t)h->nb * h->nb * h->nb + 1 + SORT_BLK_WG - 1) / SORT_BLK_WG) * SORT_BLK_WG), h->stream));
        }
        const SortKeyP K = {h->sort_key, h->sort_rank, h->sort_cnt, h->sort_bcnt, h->sort_nact};
        if (particle_collide(h))
            hipLaunchKernelGGL(k_g2p_sortkey<true>, wgrid_g2p(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T, gout(h, f), bcount(h, f), h->slow_dev, ag, f, K);
        else
            hipLaunchKernelGGL(k_g2p_sortkey<false>, wgrid_g2p(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T, gout(h, f), bcount(h, f), h->slow_dev, ag, f, K);
        h->keys_frame = f + 1; h->sort_scratch_dirty = true;
    } else if (particle_collide(h))
        hipLaunchKernelGGL(k_g2p<true>, wgrid_g2p(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T, gout(h, f), bcount(h, f), h->slow_dev, ag, f);
    else
        hipLaunchKernelGGL(k_g2p<false>, wgrid_g2p(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T, gout(h, f), bcount(h, f), h->slow_dev, ag, f);
    prof_end(h);
    if (h->has_rigid) {
        hipLaunchKernelGGL(k_rigid_body<false>, dim3(h->n_bodies), dim3(256), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), (float*)nullptr,
                           h->tables[h->tbl_of_frame[f]].slot_of_pid, h->body_start, h->body_pids, h->bodies_dev);
    }
    return 0;
}

// `next_f` >= 0: the caller goes on with substep next_f of the same sweep (fe_step_grad) -- when that one works in another particle order,
// k_p2g_grad leaves the adjoint of frame f in that order right away (GradDst) instead of a reorder pass at the head of the next substep.
// `compact_out`: the adjoint of frame f may be left with a compact F (nobody but the next substep of the same call reads it)
// one small D2H per backward sweep: which frames have a stored grid
int fetch_gs_flags(FeEngine* h) {
    if (h->gs_cap > 0 && !h->gs_host_valid) {
        h->gs_host.resize(h->L + 1);
        HIPCK(h, hipMemcpyAsync(h->gs_host.data(), h->gs_flag, sizeof(int) * (h->L + 1), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        h->gs_host_valid = true;
    }
    return 0;
}
// Can substep f's p2g_grad take the g2p_grad of substep f - 1 along (k_pgg_g2pg)?  Nothing may lie between the two: the same particle order in frames
// f - 1 and f (no reorder), no collide / rigid-body adjoint pass, no collector; the SVD-free build with its default kernels; and grid[f - 1] in the
// per-frame store (a recompute would have to run first).  Option "fuse_bwd".
inline bool fusable_bwd(FeEngine* h, int f) {
    // (the SVD build's fused kernel keeps three workgroups per CU where k_g2p_grad2 keeps four: it pays while a launch is a round or two of workgroups -- +0.6 % at 200k particles --
    //  and costs where the kernels are bound by what they issue: 155.8 us against 61.3 + 77.6 at 1M.  fuse_bwd = 1 fuses it up to two rounds' worth of particles, 2 always)
    if (!h->all_simple_liquid && h->fuse_bwd < 2 && h->Np / 256 > 2 * 3 * (size_t)h->n_cus) return false;      // (two rounds of three workgroups per CU; the tuning option quad_fit has no say in this: ADVICE r5)
    return h->fuse_bwd && f > 0 && h->g2p_grad_v == 3 && (h->p2g_grad_waves >= 4 || !h->all_simple_liquid) && !particle_collide(h) && !h->has_rigid && !h->has_collector &&
           h->tbl_of_frame[f - 1] == h->tbl_of_frame[f] && h->tbl_of_frame[f] >= 0 && h->gs_cap > 0 && h->gs_host_valid && h->gs_host[f - 1] != 0;
}
// `g2p_done`: the g2p_grad of this substep ran at the tail of the last launch of substep f + 1 (k_pgg_g2pg).  `fuse_next`: this substep's p2g_grad
// takes the g2p_grad of substep f - 1 along (fusable_bwd(h, f) held and the caller goes on with f - 1).
int substep_bwd(FeEngine* h, int f, int f_global, int act, int next_f = -1, bool compact_out = false, bool g2p_done = false, bool fuse_next = false) {
    InjectP inj;
    if (make_inject(h, f, f_global, act, false, inj)) return 1;
    // grad[f+1] arrives in the order frame f+1 is stored in; substep f works in frame f's order
    const int t = h->tbl_of_frame[f];
    if (!g2p_done && h->gpartial[(f + 1) & 1]) FAIL(h, "the adjoint of frame f + 1 is incomplete: a fused fe_step_grad passed it on in registers (only the adjoint of a call's first frame is defined afterwards; option fuse_bwd = 0 keeps every frame's)");
    if (reorder_grad(h, (f + 1) & 1, t)) return 1;
    use_static_table(h, t);
    const TableP T = h->tableP(t);
    AgentP ag = agent_params(h);
    InjectP noinj = {0, 0, 0, 0};
    if (fetch_gs_flags(h)) return 1;
    const bool stored = h->gs_cap > 0 && h->gs_host[f] != 0;
    if (!g2p_done) {
    if (!stored) {
    h->stamp++;                                           // marks of the recompute (p2g -> grid_op)
    prof_begin(h, KID_P2G_RE);
    if (h->all_simple_liquid)
        hipLaunchKernelGGL((k_p2g<false, false>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T,
                           h->pool_idx, grid_w(h, f), ag, noinj, 0, f, grid_store(h), h->fiso[f] ? 1 : 0);
    else
        hipLaunchKernelGGL((k_p2g<false, true>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), T,
                           h->pool_idx, grid_w(h, f), ag, noinj, 0, f, grid_store(h), 0);
    prof_end(h);
    prof_begin(h, KID_GRID_KEEP);
    launch_grid<true>(h, T, f, ag);
    prof_end(h);
    }
    h->stamp++;                                           // marks of the adjoint scatter (g2p.grad -> grid_op.grad)
    if (h->has_rigid) {                                   // advect_grad (mpm:436-447) for the rigid bodies, see k_rigid_body
        hipLaunchKernelGGL(k_rigid_body<true>, dim3(h->n_bodies), dim3(256), 0, h->stream, h->S, h->frame(f), h->frame(f + 1), h->grad(f + 1),
                           h->tables[t].slot_of_pid, h->body_start, h->body_pids, h->bodies_dev);
    }
    prof_begin(h, KID_G2P_GRAD);
    if (particle_collide(h)) {
        HIPCK(h, hipMemsetAsync(h->hit_count, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_collide_list, pgrid(h), dim3(256), 0, h->stream, h->S, h->frame(f), f, ag, h->hit_list, h->hit_count);
        hipLaunchKernelGGL(k_collide_grad, dim3(std::min((h->N + 15) / 16, 2048)), dim3(256), 0, h->stream, h->S, h->frame(f), h->grad(f + 1), gout(h, f), T, grid_store(h), f, ag,
                           h->hit_list, h->hit_count);
    }
    if (h->g2p_grad_v == 2) hipLaunchKernelGGL(k_g2p_grad2<3>, wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, gout(h, f), h->gg_out, h->slab, h->slow_dev, grid_store(h), f, ag);
    else hipLaunchKernelGGL(k_g2p_grad2<4>, wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, gout(h, f), h->gg_out, h->slab, h->slow_dev, grid_store(h), f, ag);
    prof_end(h);
    }                                                     // (!g2p_done; the marks of its scatter carry the stamp of the launch it ran in: no stamp++ since)
    prof_begin(h, KID_GRID_GRAD);
#define LAUNCH_GRID_GRAD(ST_, DY_) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_grad<ST_, DY_>), ggrid(h), dim3(256), 0, h->stream, h->S, T, h->slab, h->g_in, h->gg_out, \
                           ggin(h, f), h->blk_list, bcount(h, f), h->blk_flag, grid_store(h), f, statics_p(h), ag, h->node_work, h->node_work_count)
    if (grid_collide(h)) {
        if (!h->node_work && (dev_alloc(h, &h->node_work, (size_t)h->S.ncell, false) || dev_alloc(h, &h->node_work_count, 1))) return 1;
        HIPCK(h, hipMemsetAsync(h->node_work_count, 0, sizeof(int), h->stream));
        if (h->statics_host.empty()) LAUNCH_GRID_GRAD(false, true); else LAUNCH_GRID_GRAD(true, true);
        const dim3 wg((unsigned)std::min((h->S.ncell + 15) / 16, 1024));
        if (h->statics_host.empty()) hipLaunchKernelGGL(k_grid_collide_grad<false>, wg, dim3(256), 0, h->stream, h->S, ggin(h, f), f, statics_p(h), ag, h->node_work, h->node_work_count);
        else hipLaunchKernelGGL(k_grid_collide_grad<true>, wg, dim3(256), 0, h->stream, h->S, ggin(h, f), f, statics_p(h), ag, h->node_work, h->node_work_count);
    } else { if (h->statics_host.empty()) LAUNCH_GRID_GRAD(false, false); else LAUNCH_GRID_GRAD(true, false); }
    prof_end(h);
    const int t_next = (h->fold_reorder && next_f >= 0) ? h->tbl_of_frame[next_f] : t;
    const bool fold = t_next != t && t_next >= 0;
    float* g_dst = fold ? h->grad_ptr[2] : h->grad(f);
    const int* to_slot = fold ? h->tables[t_next].slot_of_pid : nullptr;
    // (not with a collector: collector_takes clears `used` of a frame after the fact, and the slot of a particle not in use holds all nine words of F's adjoint -- ADVICE r5)
    const bool gc_out = compact_out && h->compact_F && h->all_simple_liquid && !h->has_collector;
    const int giso = (h->fiso[f] ? 1 : 0) | (h->gcompact[(f + 1) & 1] ? 2 : 0) | (gc_out ? 4 : 0);
#define LAUNCH_P2G_GRAD(G, W) hipLaunchKernelGGL((k_p2g_grad<G, W>), wgrid_pgg(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->grad(f + 1), \
                           h->grad(f), T, h->pool_idx, ggin(h, f), bcount(h, f), h->slow_dev, ag, inj, act, f, g_dst, to_slot, giso)
    if (!fuse_next) prof_begin(h, KID_P2G_GRAD);
    if (fuse_next) {                                      // (same order in frames f - 1 and f: no fold; the SVD-free build)
        h->stamp++;                                       // marks of substep f - 1's adjoint scatter (the launch's g2p_grad part -> its grid_op.grad)
        const BwdFuseP B = {h->frame(f - 1), h->grad(f - 1), gout(h, f - 1), h->gg_out, h->slab};
        prof_begin(h, KID_PGG_G2PG);
        if (h->all_simple_liquid)
            hipLaunchKernelGGL(k_pgg_g2pg<4>, wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, h->pool_idx, ggin(h, f), bcount(h, f), h->slow_dev,
                               ag, inj, act, f, giso, B, grid_store(h));
        else                                              // (the SVD build: three workgroups per CU, as k_p2g_grad<true>)
            hipLaunchKernelGGL((k_pgg_g2pg<3, true>), wgrid(h), dim3(WG), 0, h->stream, h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, h->pool_idx, ggin(h, f), bcount(h, f), h->slow_dev,
                               ag, inj, act, f, 0, B, grid_store(h));
    } else if (h->all_simple_liquid) {
        if (h->p2g_grad_waves >= 4) LAUNCH_P2G_GRAD(false, 4);
        else if (h->p2g_grad_waves == 3) LAUNCH_P2G_GRAD(false, 3);
        else LAUNCH_P2G_GRAD(false, 2);
    } else LAUNCH_P2G_GRAD(true, 1);
    prof_end(h);
    if (fold) std::swap(h->grad_ptr[2], h->grad_ptr[f & 1]);
    h->gtbl[f & 1] = fold ? t_next : t;
    h->gcompact[f & 1] = gc_out;
    h->gpartial[f & 1] = fuse_next;                       // (k_pgg_g2pg stores x v C of frame f's adjoint only for the particles somebody else reads them of)
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// batched environments (fe_step_batch / fe_step_grad_batch): see "Batched environments" above the kernels
// ---------------------------------------------------------------------------------------------------------------
// The launches of a batch need one variant of every kernel and one grid geometry: liquid-only or not alike, no SDF colliders, no
// mesh effectors, no MAT_RIGID bodies (those scenes step one engine at a time: fe_step_batch falls back to a loop).
bool batchable(FeEngine** hs, int B) {
    if (B < 2 || B > FE_MAX_BATCH) return false;
    for (int i = 0; i < B; i++) if (!hs[i]) return false;
    for (int i = 0; i < B; i++) {
        FeEngine* h = hs[i];
        if (h->device != hs[0]->device || h->n != hs[0]->n || h->L != hs[0]->L || h->all_simple_liquid != hs[0]->all_simple_liquid ||
            h->sort_interval != hs[0]->sort_interval || h->p2g_grad_waves != hs[0]->p2g_grad_waves || h->g2p_grad_v != hs[0]->g2p_grad_v || h->S.wt != hs[0]->S.wt || !h->statics_host.empty() || h->has_mesh_effector || h->has_rigid || h->prof_fine ||
            (h->gs_cap > 0) != (hs[0]->gs_cap > 0)) return false;
        for (int j = 0; j < i; j++) if (hs[j] == h) return false;
    }
    return true;
}
inline dim3 wgrid_b(FeEngine** hs, int B) { unsigned g = 0; for (int i = 0; i < B; i++) g = std::max(g, wgrid(hs[i]).x); return dim3(g, B); }

// Inside a batch call every engine enqueues on the leader's stream.  The per-engine stretches of a substep -- the sort's nine
// small launches, the adjoint reorder -- are independent of each other: they fork onto the engines' own streams, where the
// launches of different engines overlap, and join the leader's stream again before the next shared launch.
void batch_fork(FeEngine** hs, int B) {
    (void)hipEventRecord(hs[0]->ev_batch, hs[0]->own_stream);
    for (int i = 1; i < B; i++) { (void)hipStreamWaitEvent(hs[i]->own_stream, hs[0]->ev_batch, 0); hs[i]->stream = hs[i]->own_stream; }
}
void batch_join(FeEngine** hs, int B) {
    for (int i = 1; i < B; i++) {
        (void)hipEventRecord(hs[i]->ev_batch, hs[i]->own_stream);
        (void)hipStreamWaitEvent(hs[0]->own_stream, hs[i]->ev_batch, 0);
        hs[i]->stream = hs[0]->own_stream;
    }
}

int substep_fwd_batch(FeEngine** hs, int B, int f, int f_global, int act, bool g2p_pending = false, bool defer_g2p = false) {      // (g2p_pending / defer_g2p: as in substep_fwd)
    FeEngine* h0 = hs[0];
    Batch<P2GArgs> bp; Batch<GridArgs> bg; Batch<G2PArgs> bq;
    const bool sorting = h0->sort_interval > 0 && f % h0->sort_interval == 0;     // (batchable: the same interval in every engine)
    if (sorting) batch_fork(hs, B);
    struct Join { FeEngine** hs; int B; bool on; ~Join() { if (on) batch_join(hs, B); } };
    {
    Join join{hs, B, sorting};
    for (int i = 0; i < B; i++) {
        FeEngine* h = hs[i];
        h->gs_host_valid = false;
        h->stamp++;
        InjectP inj;
        if (make_inject(h, f, f_global, act, true, inj)) return 1;
        if (h->sort_interval > 0 && f % h->sort_interval == 0 && sort_frame(h, f)) return 1;
        h->tbl_of_frame[f + 1] = h->tbl_of_frame[f];
        use_static_table(h, h->tbl_of_frame[f]);
        const TableP T = h->tableP(h->tbl_of_frame[f]);
        const AgentP ag = agent_params(h);
        const int fiso = fwd_iso(h, f);
        bp.a[i] = P2GArgs{h->S, h->frame(f), h->frame(f + 1), T, h->pool_idx, grid_w(h, f), ag, inj, act, f, grid_store(h), h->all_simple_liquid ? fiso : 0,
                           g2p_pending ? FuseP{h->frame(f - 1), gout(h, f - 1), h->slow_dev, bcount(h, f - 1)} : FuseP{nullptr, nullptr, nullptr, nullptr}};
        h->fiso[f + 1] = h->all_simple_liquid && (fiso & 2) != 0;
        bg.a[i] = GridArgs{h->S, T, h->slab, h->g_in, gout(h, f), h->blk_list, bcount(h, f), h->blk_flag, grid_store(h), f, h->frame_slow_dev, statics_p(h), ag};
        bq.a[i] = G2PArgs{h->S, h->frame(f), h->frame(f + 1), T, gout(h, f), bcount(h, f), h->slow_dev, ag, f};
    }
    }
    if (g2p_pending) {
        prof_begin(h0, KID_G2P_P2G);
        if (h0->all_simple_liquid) hipLaunchKernelGGL((k_g2p_p2g_b<false>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bp);
        else hipLaunchKernelGGL((k_g2p_p2g_b<true>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bp);
        prof_end(h0);
    } else {
    prof_begin(h0, KID_P2G);
    if (h0->all_simple_liquid) hipLaunchKernelGGL((k_p2g_b<true, false>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bp);
    else hipLaunchKernelGGL((k_p2g_b<true, true>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bp);
    prof_end(h0);
    }
    prof_begin(h0, KID_GRID);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_b<false, false, false>), dim3(ggrid(h0).x, B), dim3(256), 0, h0->stream, bg);
    prof_end(h0);
    if (defer_g2p) return 0;
    prof_begin(h0, KID_G2P);
    hipLaunchKernelGGL(k_g2p_b<false>, wgrid_b(hs, B), dim3(WG), 0, h0->stream, bq);
    prof_end(h0);
    return 0;
}

int substep_bwd_batch(FeEngine** hs, int B, int f, int f_global, int act, bool g2p_done = false, bool fuse_next = false) {      // (g2p_done / fuse_next: as in substep_bwd)
    FeEngine* h0 = hs[0];
    Batch<P2GArgs> bp; Batch<GridArgs> bg; Batch<G2PGradArgs> bq; Batch<GridGradArgs> bgg; Batch<P2GGradArgs> bpg; Batch<PggArgs> bf;
    const InjectP noinj = {0, 0, 0, 0};
    bool all_stored = true;
    bool reordering = false;                               // some engine's adjoint crosses a sort here
    for (int i = 0; i < B; i++) { const int from = hs[i]->gtbl[(f + 1) & 1]; reordering = reordering || (from >= 0 && from != hs[i]->tbl_of_frame[f]); }
    if (reordering) batch_fork(hs, B);
    struct Join { FeEngine** hs; int B; bool on; ~Join() { if (on) batch_join(hs, B); } };
    {
    Join join{hs, B, reordering};
    for (int i = 0; i < B; i++) {
        FeEngine* h = hs[i];
        InjectP inj;
        if (make_inject(h, f, f_global, act, false, inj)) return 1;
        const int t = h->tbl_of_frame[f];
        if (reorder_grad(h, (f + 1) & 1, t)) return 1;
        use_static_table(h, t);
        const TableP T = h->tableP(t);
        const AgentP ag = agent_params(h);
        if (fetch_gs_flags(h)) return 1;
        all_stored = all_stored && h->gs_cap > 0 && h->gs_host[f] != 0;
        if (!g2p_done) h->stamp++;                        // recompute marks, then the adjoint scatter's (as in substep_bwd; g2p_done: the scatter's marks carry the stamp of the launch it ran in)
        bp.a[i] = P2GArgs{h->S, h->frame(f), h->frame(f + 1), T, h->pool_idx, grid_w(h, f), ag, noinj, 0, f, grid_store(h), (h->all_simple_liquid && h->fiso[f]) ? 1 : 0, FuseP{nullptr, nullptr, nullptr, nullptr}};
        bg.a[i] = GridArgs{h->S, T, h->slab, h->g_in, gout(h, f), h->blk_list, bcount(h, f), h->blk_flag, grid_store(h), f, h->frame_slow_dev, statics_p(h), ag};
        if (!g2p_done) h->stamp++;
        bq.a[i] = G2PGradArgs{h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, gout(h, f), h->gg_out, h->slab, h->slow_dev, grid_store(h), f, ag};
        bgg.a[i] = GridGradArgs{h->S, T, h->slab, h->g_in, h->gg_out, ggin(h, f), h->blk_list, bcount(h, f), h->blk_flag, grid_store(h), f, statics_p(h), ag, h->node_work, h->node_work_count};
        bpg.a[i] = P2GGradArgs{h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, h->pool_idx, ggin(h, f), bcount(h, f), h->slow_dev, ag, inj, act, f, h->grad(f), nullptr,
                               (h->fiso[f] ? 1 : 0) | (h->gcompact[(f + 1) & 1] ? 2 : 0)};      // (the batch writes full adjoints)
        if (fuse_next) {
            h->stamp++;                                   // marks of substep f - 1's adjoint scatter (the fused launch's g2p_grad part -> its grid_op.grad)
            bf.a[i] = PggArgs{h->S, h->frame(f), h->grad(f + 1), h->grad(f), T, h->pool_idx, ggin(h, f), bcount(h, f), h->slow_dev, ag, inj, act, f,
                              (h->fiso[f] ? 1 : 0) | (h->gcompact[(f + 1) & 1] ? 2 : 0), BwdFuseP{h->frame(f - 1), h->grad(f - 1), gout(h, f - 1), h->gg_out, h->slab}, grid_store(h)};
        }
        h->gtbl[f & 1] = t;
        h->gcompact[f & 1] = false;
        h->gpartial[f & 1] = fuse_next;
    }
    }
    if (!g2p_done) {
    if (!all_stored) {                                    // (the kernels of an env whose frame is stored return at once)
        prof_begin(h0, KID_P2G_RE);
        if (h0->all_simple_liquid) hipLaunchKernelGGL((k_p2g_b<false, false>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bp);
        else hipLaunchKernelGGL((k_p2g_b<false, true>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bp);
        prof_end(h0);
        prof_begin(h0, KID_GRID_KEEP);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_b<true, false, false>), dim3(ggrid(h0).x, B), dim3(256), 0, h0->stream, bg);
        prof_end(h0);
    }
    prof_begin(h0, KID_G2P_GRAD);
    if (h0->g2p_grad_v == 2) hipLaunchKernelGGL(k_g2p_grad2_b<3>, wgrid_b(hs, B), dim3(WG), 0, h0->stream, bq);
    else hipLaunchKernelGGL(k_g2p_grad2_b<4>, wgrid_b(hs, B), dim3(WG), 0, h0->stream, bq);
    prof_end(h0);
    }                                                     // (!g2p_done)
    prof_begin(h0, KID_GRID_GRAD);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_grad_b<false, false>), dim3(ggrid(h0).x, B), dim3(256), 0, h0->stream, bgg);
    prof_end(h0);
    if (fuse_next) {
        prof_begin(h0, KID_PGG_G2PG);
        hipLaunchKernelGGL(k_pgg_g2pg_b<4>, wgrid_b(hs, B), dim3(WG), 0, h0->stream, bf);
        prof_end(h0);
        return 0;
    }
    prof_begin(h0, KID_P2G_GRAD);
    if (h0->all_simple_liquid) {                          // (batchable: the same p2g_grad_waves in every engine)
        if (h0->p2g_grad_waves >= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_p2g_grad_b<false, 4>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bpg);
        else if (h0->p2g_grad_waves == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_p2g_grad_b<false, 3>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bpg);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_p2g_grad_b<false, 2>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bpg);
    } else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_p2g_grad_b<true, 1>), wgrid_b(hs, B), dim3(WG), 0, h0->stream, bpg);
    prof_end(h0);
    return 0;
}

// All engines of a batch work on the leader's stream for the duration of the call: it first waits for whatever the others have
// queued on their own streams, and they wait for it afterwards.
struct BatchStreams {
    FeEngine** hs; int B;
    BatchStreams(FeEngine** hs_, int B_) : hs(hs_), B(B_) {
        for (int i = 1; i < B; i++) {
            (void)hipEventRecord(hs[i]->ev_batch, hs[i]->own_stream);
            (void)hipStreamWaitEvent(hs[0]->own_stream, hs[i]->ev_batch, 0);
            hs[i]->stream = hs[0]->own_stream;
        }
    }
    ~BatchStreams() {
        (void)hipEventRecord(hs[0]->ev_batch, hs[0]->own_stream);
        for (int i = 1; i < B; i++) { hs[i]->stream = hs[i]->own_stream; (void)hipStreamWaitEvent(hs[i]->own_stream, hs[0]->ev_batch, 0); }
    }
};

int check_async(FeEngine* h) {
    HIPCK(h, hipGetLastError());
    return 0;
}

// H2D staging + pack into frame / grad planes
int upload_planes(FeEngine* h, float* planes, const int* pid, const float* x, const float* v, const float* C, const float* F, const int* used, int add) {
    const size_t N = h->N;
    int mask = 0;
    float *sx = h->stage_r, *sv = h->stage_r + 3 * N, *sC = h->stage_r + 6 * N, *sF = h->stage_r + 15 * N;
    if (x) { mask |= 1; HIPCK(h, hipMemcpyAsync(sx, x, sizeof(float) * 3 * N, hipMemcpyHostToDevice, h->stream)); }
    if (v) { mask |= 2; HIPCK(h, hipMemcpyAsync(sv, v, sizeof(float) * 3 * N, hipMemcpyHostToDevice, h->stream)); }
    if (C) { mask |= 4; HIPCK(h, hipMemcpyAsync(sC, C, sizeof(float) * 9 * N, hipMemcpyHostToDevice, h->stream)); }
    if (F) { mask |= 8; HIPCK(h, hipMemcpyAsync(sF, F, sizeof(float) * 9 * N, hipMemcpyHostToDevice, h->stream)); }
    if (used) { mask |= 16; HIPCK(h, hipMemcpyAsync(h->stage_i, used, sizeof(int) * N, hipMemcpyHostToDevice, h->stream)); }
    if (!mask || N == 0) return 0;
    hipLaunchKernelGGL(k_pack, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, planes, pid, sx, sv, sC, sF, h->stage_i, mask, add);
    HIPCK(h, hipStreamSynchronize(h->stream));      // host buffers are borrowed only for the call
    return check_async(h);
}
int download_planes(FeEngine* h, float* planes, const int* pid, float* x, float* v, float* C, float* F, int* used) {
    const size_t N = h->N;
    int mask = (x ? 1 : 0) | (v ? 2 : 0) | (C ? 4 : 0) | (F ? 8 : 0) | (used ? 16 : 0);
    if (!mask || N == 0) return 0;
    float *sx = h->stage_r, *sv = h->stage_r + 3 * N, *sC = h->stage_r + 6 * N, *sF = h->stage_r + 15 * N;
    hipLaunchKernelGGL(k_unpack, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, planes, pid, sx, sv, sC, sF, h->stage_i, mask);
    if (x) HIPCK(h, hipMemcpyAsync(x, sx, sizeof(float) * 3 * N, hipMemcpyDeviceToHost, h->stream));
    if (v) HIPCK(h, hipMemcpyAsync(v, sv, sizeof(float) * 3 * N, hipMemcpyDeviceToHost, h->stream));
    if (C) HIPCK(h, hipMemcpyAsync(C, sC, sizeof(float) * 9 * N, hipMemcpyDeviceToHost, h->stream));
    if (F) HIPCK(h, hipMemcpyAsync(F, sF, sizeof(float) * 9 * N, hipMemcpyDeviceToHost, h->stream));
    if (used) HIPCK(h, hipMemcpyAsync(used, h->stage_i, sizeof(int) * N, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return check_async(h);
}

// API calls that hand F (or its adjoint) out, or add to it, see full planes: a compact F is written out first (FrameV::iso, k_expand_F)
void full_F_of_frame(FeEngine* h, int f) {
    if (!h->fiso[f]) return;
    hipLaunchKernelGGL(k_expand_F, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, h->frame(f), 1.f, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr);
    h->fiso[f] = 0;
}
void full_F_of_grad(FeEngine* h, int f) {
    if (!h->gcompact[f & 1]) return;
    const int gt = h->gtbl[f & 1] < 0 ? h->tbl_of_frame[f] : h->gtbl[f & 1], ft = h->tbl_of_frame[f];
    hipLaunchKernelGGL(k_expand_F, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, h->grad(f), 1.f / 3.f, (const int*)frame_view(h->frame(f), h->Np).used.ptr(),
                       (const int*)h->tables[gt].pid, gt == ft ? (const int*)nullptr : (const int*)h->tables[ft].slot_of_pid);
    h->gcompact[f & 1] = false;
}

int check_device_errors(FeEngine* h) {
    int e = 0;
    HIPCK(h, hipMemcpyAsync(&e, h->err_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (e) {
        (void)hipMemsetAsync(h->err_dev, 0, sizeof(int), h->stream);
        FAIL(h, "particle stencil left the grid (p2g)");
    }
    if (h->fg_started) {                                      // a wave of the fused grid pass gave up on a wait (fg_give_up): results of that launch are incomplete
        int ge = 0;
        HIPCK(h, hipMemcpyAsync(&ge, h->fg_host.ctr + FGC_ERR * FG_LINE, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        if (ge) {
            (void)hipMemsetAsync(h->fg_host.ctr + FGC_ERR * FG_LINE, 0, sizeof(int), h->stream);
            FAIL(h, "fused grid pass: a wait did not end (option fuse_grid = 0 runs the grid kernels as launches of their own)");
        }
    }
    return 0;
}

} // namespace

// =========================================================================================
// C ABI
// =========================================================================================
#include "fe_smoke.h"
#include "fe_mesh.h"

extern "C" {

const char* fe_backend(void) { return "hip-gfx950"; }
int fe_real_size(void) { return 4; }

FeEngine* fe_create(const FeConfig* cfg) {
    if (!cfg || cfg->struct_size != (int)sizeof(FeConfig)) { g_create_err = "FeConfig size mismatch"; return nullptr; }
    if (cfg->n_particles > 40000000) { g_create_err = "n_particles > 40M: a frame no longer fits 32-bit plane offsets"; return nullptr; }
    if (cfg->n_grid > 4092) { g_create_err = "n_grid > 4092: a work item names its block by three 10-bit coordinates"; return nullptr; }
    if (cfg->n_grid < 4 || cfg->n_grid % 4 != 0 || cfg->n_particles < 0 || cfg->max_substeps_local < 1 || cfg->n_substeps < 1) {
        g_create_err = "invalid FeConfig (n_grid must be a multiple of 4)"; return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device visible: the MI355X engine has no CPU fallback"; return nullptr; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "HIP device ordinal out of range"; return nullptr; }
    FeEngine* h = new FeEngine();
    if (const char* e = std::getenv("FE_SORT_INTERVAL")) h->sort_interval = std::atoi(e);     // tuning experiments (the option of the same name wins)
    if (const char* e = std::getenv("FE_QUAD_MIN_UNITS")) h->quad_min_units = std::atoi(e);   // (task-level A/B of the quad units: scripts/run_envs.py)
    if (const char* e = std::getenv("FE_G2P_GRAD_V")) h->g2p_grad_v = std::atoi(e) == 2 ? 2 : 3;           // (the parity suite is run once per build of the G2P adjoint)
    if (const char* e = std::getenv("FE_FUSE_BWD")) h->fuse_bwd = std::atoi(e);
    if (const char* e = std::getenv("FE_FUSE_G2P")) h->fuse_g2p = std::atoi(e) != 0;           // (the parity suite with and without the fused forward launch)
    if (const char* e = std::getenv("FE_FUSE_GRID")) h->fuse_grid = std::atoi(e);              // (... with and without the fused grid pass; 2 = wherever possible, late deposits included)
    const char* env_lsplit = std::getenv("FE_LANE_SPLIT");                                    // (the parity suite with and without lane_split)
    h->cfg = *cfg; h->N = cfg->n_particles; h->L = cfg->max_substeps_local; h->n = cfg->n_grid; h->nb = cfg->n_grid / 4;
    h->Np = ((h->N + 63) / 64) * 64; if (h->Np == 0) h->Np = 64;
    h->device = cfg->device;
    auto fail = [&](const std::string& m) { g_create_err = m.empty() ? h->err : m; fe_destroy(h); return (FeEngine*)nullptr; };
    if (hipSetDevice(h->device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    {   // One round of resident workgroups: 4 per CU for the kernels at 128 registers, 6 for k_g2p.  The scatter kernels' and k_g2p's work-list
        // launches are no larger than that (round 4: a launch of 2,048 started a second round of workgroups where the first ones could have
        // looped on: early splash -2.4 %, the layer -2 %, the whole run -1.9 % in time; `profiles/r04_ab_wgrid_caps.txt`).  k_p2g_grad keeps the
        // larger launch: its extra workgroups fill whatever slot frees first, which is worth 3.7 % on LatteArt at 128^3 (1,995 units of
        // full 128-particle items: 8,216 -> 8,520 pairs/s) and costs the water block 0.2 ... 0.4 % (`r04_ab_wgrid_caps_latteart.txt`).
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) { h->n_cus = cus; h->quad_fit = 4 * cus; h->wgrid_cap = 4 * cus; h->wgrid_cap_pgg = 8 * cus; h->wgrid_cap_g2p = 6 * cus; }
    }
    h->own_stream = h->stream;
    SimP& S = h->S;
    S.xcd = 16;                                            // blocked-cyclic unit mapping (A/B in DESIGN.md section 6)
    S.uni = 0; S.fg_base = 0; S.fg_nowait = 0;
    S.wsort = 1;                                           // lanes regrouped by stencil base before the scan (A/B in DESIGN.md section 6)
    S.lsplit = env_lsplit ? std::atoi(env_lsplit) : 3;            // small waves give every particle three or nine lanes (lane_split) in the two scatter kernels; in k_p2g_grad (bit 2)
                                                                  // the fifteen sums' cross-lane reads cost what the shorter loop saves (profiles/r05_ab_lane_split.txt)
    S.wt = 5;                                              // p2g and g2p_grad: their bulk stores come early (A/B in DESIGN.md section 6)
    S.N = h->N; S.Np = h->Np; S.n = h->n; S.nb = h->nb; S.ncell = h->nb * h->nb * h->nb * 64;
    S.dx = 1.0f / (float)h->n; S.inv_dx = (float)h->n; S.dt = cfg->dt;
    S.stress_scale = -cfg->dt * cfg->p_vol * 4.f * S.inv_dx * S.inv_dx;
    for (int i = 0; i < 3; i++) S.g[i] = cfg->gravity[i];
    S.bnd = to_boundary(cfg->boundary);
    h->frame_stride = (size_t)FR_WORDS * h->Np;
    const size_t ncell = (size_t)h->nb * h->nb * h->nb * 64;
    if (dev_alloc(h, &h->frames, h->frame_stride * (h->L + 2))) return fail("");
    h->frame_ptr.resize(h->L + 2);
    for (int i = 0; i < h->L + 2; i++) h->frame_ptr[i] = h->frames + (size_t)i * h->frame_stride;
    if (dev_alloc(h, &h->grads, 3 * h->grad_words())) return fail("");
    for (int i = 0; i < 3; i++) h->grad_ptr[i] = h->grads + (size_t)i * h->grad_words();
    h->tbl_of_frame.assign(h->L + 1, 0);
    h->fiso.assign(h->L + 2, 0);
    {
        const size_t nblk = (size_t)h->nb * h->nb * h->nb;
        h->items_cap = (nblk < (size_t)h->Np ? nblk : (size_t)h->Np) + (size_t)h->Np / 64 + 2;      // item_max >= 64
        h->units_cap = h->items_cap + (size_t)h->Np / WG + 16 + 1024;                                      // work units: items (at worst one each) + tail workgroups, rounded up to 8
        // (the slab gathers address a node of slab i as a signed 32-bit byte offset i * SLAB_N * 16 + ...: ADVICE r4)
        if (h->items_cap * (size_t)SLAB_N * 16 >= ((size_t)1 << 31)) return fail("grid / particle count too large: the slab buffer no longer fits 32-bit byte offsets");
        if (dev_alloc(h, &h->sort_key, h->Np) || dev_alloc(h, &h->sort_rank, h->Np) || dev_alloc(h, &h->sort_cnt, ncell + 1) ||
            dev_alloc(h, &h->sort_start, ncell + 1) || dev_alloc(h, &h->sort_bcnt, ((nblk + 1 + SORT_BLK_WG - 1) / SORT_BLK_WG) * SORT_BLK_WG) || dev_alloc(h, &h->sort_partial, ((nblk + 1 + SORT_BLK_WG - 1) / SORT_BLK_WG) * PART_STRIDE) || dev_alloc(h, &h->sort_base, ((nblk + 1 + SORT_BLK_WG - 1) / SORT_BLK_WG) * SORT_BLK_WG) || dev_alloc(h, &h->sort_nact, 4)      /* [0] the active list's length, [1] / [2] k_sort_blk_scan's counters */ || dev_alloc(h, &h->sort_pid, h->Np) ||
            dev_alloc(h, &h->slow_dev, 1) || dev_alloc(h, &h->frame_slow_dev, 1) || dev_alloc(h, &h->slab, h->items_cap * SLAB_N, false)) return fail("");
    }
    if (dev_alloc(h, &h->effs_dev, FE_MAX_EFF)) return fail("");
    {   // forward grid store: cap blocks per frame, 1,792 bytes each (GS_BLK); bounded to 64 GiB
        const size_t nblk = (size_t)h->nb * h->nb * h->nb;
        // (round 1 clamped this to 4096 blocks: a block of water that has spread into a thin layer over the floor of a 128^3 box
        // has more active blocks than that, every frame lost its store and the backward pass recomputed P2G + grid_op throughout)
        size_t cap = nblk;
        // (FE_GRID_STORE_GIB: another budget for the store -- two LatteArt replicas at 128^3 fit one GPU with 32 GiB each, bench.py --envs-per-gpu 2;
        //  a frame with more active blocks than the store has slots falls back to the recompute, as ever)
        size_t budget_gib = 64;
        if (const char* e = std::getenv("FE_GRID_STORE_GIB")) { const long v = std::atol(e); if (v >= 0 && v <= 256) budget_gib = (size_t)v; }
        while (cap > 0 && (size_t)(h->L + 1) * cap * (GS_BLK * 16) > (budget_gib << 30)) cap /= 2;
        while (cap * (size_t)(GS_BLK * 16) >= ((size_t)1 << 31)) cap /= 2;       // (store_vout: slot * 1,792 as a signed 32-bit byte offset into a frame's store)
        h->gs_cap = (int)cap;
        if (cap > 0 && (dev_alloc(h, &h->gstore, (size_t)(h->L + 1) * cap * GS_BLK, false) || dev_alloc(h, &h->gs_flag, h->L + 1) ||
                        dev_alloc(h, &h->gs_live, (size_t)(h->L + 1) * cap))) return fail("");
        if (dev_alloc(h, &h->ent_touched, nblk) || dev_alloc(h, &h->ent_dirty, nblk) || dev_alloc(h, &h->cur_live, nblk)) return fail("");
    }
    if (dev_alloc(h, &h->pinfo, h->Np) || dev_alloc(h, &h->pool_idx, h->Np)) return fail("");
    if (ensure_table(h, 0)) return fail("");                 // identity order: no items, everything is "tail"; its `info` is pinfo itself
    // (v_out and the adjoint grid twice, by the parity of the substep: a launch with the grid pass on board reads the substep before's while its owners write its own)
    if (dev_alloc(h, &h->g_in, 4 * ncell) || dev_alloc(h, &h->g_out, 2 * ncell) || dev_alloc(h, &h->gg_out, 3 * ncell) || dev_alloc(h, &h->gg_in, 2 * ncell)) return fail("");
    if (dev_alloc(h, &h->blk_flag, ncell / 64) || dev_alloc(h, &h->blk_list, ncell / 64) || dev_alloc(h, &h->blk_count, 2) || dev_alloc(h, &h->err_dev, 1)) return fail("");
    {   // the fused grid pass (FG kernels)
        const size_t nblk = ncell / 64;
        if (dev_alloc(h, &h->fg_host.late, 4 * ncell) || dev_alloc(h, &h->fg_host.late_flag, nblk) || dev_alloc(h, &h->fg_host.late_list, nblk) ||
            dev_alloc(h, &h->fg_host.skipm, nblk) || dev_alloc(h, &h->fg_host.ctr, FGC_N * FG_LINE) || dev_alloc(h, &h->fg_dev, 1)) return fail("");
        if (hipMemcpyOnStream(h, h->fg_dev, &h->fg_host, sizeof(FgDev), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
    }
    if (dev_alloc(h, &h->stage_r, (size_t)24 * h->Np) || dev_alloc(h, &h->stage_i, h->Np)) return fail("");
    if (dev_alloc(h, &h->node_mark, ncell) || dev_alloc(h, &h->counters, 4)) return fail("");
    {   // identity particle order
        std::vector<int> id(h->Np);
        for (int i = 0; i < h->Np; i++) id[i] = i < h->N ? i : 0;
        if (hipMemcpyOnStream(h, h->tables[0].pid, id.data(), sizeof(int) * h->Np, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
        if (hipMemcpyOnStream(h, h->tables[0].slot_of_pid, id.data(), sizeof(int) * h->Np, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
    }
    if (hipEventCreate(&h->ev_t0) != hipSuccess || hipEventCreate(&h->ev_t1) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_batch, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail("device initialisation failed");
    return h;
}

void fe_destroy(FeEngine* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    smoke_destroy(h);
    for (auto& t : h->tables) { if (t.info && t.info != h->pinfo) (void)hipFree(t.info);
        for (void* q : {(void*)t.pairs, (void*)t.singles, (void*)t.pid, (void*)t.items, (void*)t.meta, (void*)t.blk_first, (void*)t.active, (void*)t.blk_slot, (void*)t.slot_of_pid, (void*)t.units, (void*)t.units_p, (void*)t.nbr, (void*)t.arrive}) if (q) (void)hipFree(q); }
    void* ptrs[] = {h->frames, h->grads, h->sort_key, h->sort_rank, h->sort_cnt, h->sort_start, h->sort_bcnt, h->sort_partial, h->sort_base, h->sort_nact, h->sort_pid, h->slow_dev, h->frame_slow_dev, h->gstore, h->gs_flag, h->gs_live, h->ent_touched, h->ent_dirty, h->cur_live, h->slab, h->effs_dev, h->pinfo, h->pool_idx, h->g_in, h->g_out, h->gg_out, h->gg_in,
                    h->blk_flag, h->blk_list, h->blk_count, h->err_dev, h->stage_r, h->stage_i, h->node_mark, h->counters,
                    h->fg_host.late, h->fg_host.late_flag, h->fg_host.late_list, h->fg_host.skipm, h->fg_host.ctr, h->fg_dev,
                    h->tgt, h->chamfer, h->step_loss, h->body_start, h->body_pids, h->bodies_dev, h->statics_dev, h->collector_dev, h->hit_dev, h->hit_list, h->hit_count, h->node_work, h->node_work_count};
    for (float* v : h->statics_vox) if (v) (void)hipFree(v);
    for (float* v : h->mesh_vox) if (v) (void)hipFree(v);
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& E : h->effs) {
        void* ep[] = {E.p.pos, E.p.quat, E.p.v, E.p.w, E.p.gpos, E.p.gquat, E.p.gv, E.p.gw, E.p.abuf, E.p.gabuf, E.p.abuf_p, E.p.gabuf_p, E.p.random_vector,
                      E.p.sa, E.p.ra, E.p.gsa, E.p.gra};
        for (void* p : ep) if (p) (void)hipFree(p);
    }
    for (auto e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->ev_t0) (void)hipEventDestroy(h->ev_t0);
    if (h->ev_t1) (void)hipEventDestroy(h->ev_t1);
    if (h->ev_batch) (void)hipEventDestroy(h->ev_batch);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

const char* fe_last_error(FeEngine* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int fe_sync(FeEngine* h) {
    FE_ENTRY(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (check_async(h)) return 1;
    return check_device_errors(h);
}

int fe_set_option(FeEngine* h, const char* name, double value) {
    FE_ENTRY(h);
    if (!std::strcmp(name, "sort_interval")) {
        if (value < 0) FAIL(h, "sort_interval must be >= 0");
        h->sort_interval = (int)value;
        return 0;
    }
    if (!std::strcmp(name, "item_max")) {
        if (value < 64 || value > ITEM_MAX_CAP) FAIL(h, "item_max must be in [64, 128]");
        h->item_max = (int)value;
        return 0;
    }
    if (!std::strcmp(name, "grid_store")) {                  // 0 disables the forward grid store (backward always recomputes)
        if (value == 0) h->gs_cap = 0;
        h->gs_host_valid = false;
        return 0;
    }
    if (!std::strcmp(name, "p2g_grad_waves")) { h->p2g_grad_waves = (int)value; return 0; }
    if (!std::strcmp(name, "g2p_grad_v")) { h->g2p_grad_v = (int)value == 2 ? 2 : 3; return 0; }      // (1, round 2's one-loop kernel, is gone: the default)
    if (!std::strcmp(name, "loose_max")) { if (value < 0 || value > ITEM_MAX_CAP) FAIL(h, "loose_max must be in [0, 128]"); h->loose_max = (int)value; return 0; }
    if (!std::strcmp(name, "inject_till")) { h->inject_till = (int)value; return 0; }
    if (!std::strcmp(name, "collide_min_y")) { h->collide_min_y = (float)value; return 0; }
    if (!std::strcmp(name, "collide_type")) {
        const int t = (int)value;
        if (t < 1 || t > 3) { h->err = "collide_type must be 1 (particle), 2 (grid) or 3 (both)"; return 1; }
        h->collide_type = t; return 0;
    }
    if (!std::strcmp(name, "prof_fine")) { h->prof_fine = value != 0; return 0; }
    // (run length k rounds the unit list up to a multiple of 8 k slots; units_cap has ~1,040 slots of slack: k <= 64 always fits)
    if (!std::strcmp(name, "xcd_map")) { if (value < 0 || value > 64) FAIL(h, "xcd_map must be 0 (none), 1 (contiguous eighths) or a run length 2..64"); h->S.xcd = (int)value; return 0; }
    if (!std::strcmp(name, "write_through")) { h->S.wt = (int)value; return 0; }
    if (!std::strcmp(name, "wave_sort")) { h->S.wsort = value != 0; return 0; }
    if (!std::strcmp(name, "lane_split")) { if (value < 0 || value > 7) FAIL(h, "lane_split is a bit set: 1 k_p2g, 2 k_g2p_grad2, 4 k_p2g_grad"); h->S.lsplit = (int)value; return 0; }
    if (!std::strcmp(name, "fold_reorder")) { h->fold_reorder = value != 0; return 0; }
    if (!std::strcmp(name, "compact_F")) { h->compact_F = value != 0; return 0; }
    if (!std::strcmp(name, "fuse_g2p")) { h->fuse_g2p = value != 0; return 0; }
    if (!std::strcmp(name, "sort_keys_in_g2p")) { h->sort_keys_in_g2p = value != 0; return 0; }
    if (!std::strcmp(name, "sort_one_scan")) { h->sort_one_scan = value != 0; return 0; }
    if (!std::strcmp(name, "fuse_bwd")) { h->fuse_bwd = (int)value; return 0; }
    if (!std::strcmp(name, "fuse_grid")) { if (value < 0 || value > 6 || ((int)value & 3) == 3) FAIL(h, "fuse_grid must be 0 (separate grid kernels), 1 (fused where nothing slow is expected) or 2 (wherever possible), + 4: never wait"); h->fuse_grid = (int)value; return 0; }
    if (!std::strcmp(name, "quad_min_units")) { h->quad_min_units = (int)value; return 0; }
    if (!std::strcmp(name, "pgg_quad_min_units")) { h->pgg_quad_min_units = (int)value; return 0; }
    if (!std::strcmp(name, "pack_units")) { if (value < 0 || value > 2) FAIL(h, "pack_units must be 0, 1 or 2"); h->pack_units = (int)value; return 0; }
    if (!std::strcmp(name, "quad_fit")) { if (value < 0) FAIL(h, "quad_fit must be >= 0"); h->quad_fit = (int)value; return 0; }
    if (!std::strcmp(name, "quad_max")) { if (value < 0 || value > QUAD_MAX) FAIL(h, "quad_max must be in [0, 64]"); h->quad = (int)value; return 0; }
    if (!std::strcmp(name, "ggrid_cap")) { if (value < 1) FAIL(h, "ggrid_cap must be >= 1"); h->ggrid_cap = (int)value; return 0; }
    if (!std::strcmp(name, "wgrid_cap_pgg")) { if (value < 64) FAIL(h, "wgrid_cap_pgg must be >= 64"); h->wgrid_cap_pgg = (int)value; return 0; }
    if (!std::strcmp(name, "wgrid_cap_g2p")) { if (value < 64) FAIL(h, "wgrid_cap_g2p must be >= 64"); h->wgrid_cap_g2p = (int)value; return 0; }
    if (!std::strcmp(name, "wgrid_cap")) { if (value < 64) { h->err = "wgrid_cap must be >= 64"; return 1; } h->wgrid_cap = (int)value; return 0; }
    if (!std::strcmp(name, "threads")) return 0;             // oracle-only tunable
    FAIL(h, std::string("unknown option: ") + name);
}

int fe_get_option(FeEngine* h, const char* name, double* value) {
    FE_ENTRY(h);
    if (!value) FAIL(h, "fe_get_option: null output");
    const struct { const char* n; double v; } tab[] = {
        {"sort_interval", (double)h->sort_interval}, {"item_max", (double)h->item_max}, {"grid_store", h->gs_cap > 0 ? 1.0 : 0.0},
        {"p2g_grad_waves", (double)h->p2g_grad_waves}, {"g2p_grad_v", (double)h->g2p_grad_v}, {"loose_max", (double)h->loose_max},
        {"inject_till", (double)h->inject_till}, {"collide_min_y", (double)h->collide_min_y}, {"collide_type", (double)h->collide_type},
        {"prof_fine", h->prof_fine ? 1.0 : 0.0}, {"xcd_map", (double)h->S.xcd}, {"write_through", (double)h->S.wt}, {"wave_sort", (double)h->S.wsort}, {"lane_split", (double)h->S.lsplit}, {"fold_reorder", h->fold_reorder ? 1.0 : 0.0}, {"compact_F", h->compact_F ? 1.0 : 0.0}, {"fuse_g2p", h->fuse_g2p ? 1.0 : 0.0}, {"fuse_bwd", (double)h->fuse_bwd}, {"fuse_grid", (double)h->fuse_grid}, {"sort_keys_in_g2p", (double)h->sort_keys_in_g2p}, {"sort_one_scan", (double)h->sort_one_scan},
        {"quad_min_units", (double)h->quad_min_units}, {"pgg_quad_min_units", (double)h->pgg_quad_min_units}, {"quad_max", (double)h->quad}, {"quad_fit", (double)h->quad_fit}, {"pack_units", (double)h->pack_units},
        {"wgrid_cap", (double)h->wgrid_cap}, {"wgrid_cap_g2p", (double)h->wgrid_cap_g2p}, {"wgrid_cap_pgg", (double)h->wgrid_cap_pgg}, {"ggrid_cap", (double)h->ggrid_cap}, {"threads", 0.0}};
    for (const auto& t : tab) if (!std::strcmp(name, t.n)) { *value = t.v; return 0; }
    FAIL(h, std::string("unknown option: ") + name);
}

int fe_init_particles(FeEngine* h, const fe_real* x, const int* used, const int* mat, const int* mat_cls,
                      const fe_real* mu, const fe_real* lam, const fe_real* rho, const int* body_id) {
    FE_ENTRY(h);
    const int N = h->N;
    std::vector<float4> info(h->Np, make_float4(0, 0, 0, 0));
    std::vector<float> C0((size_t)9 * N, 0.f), F0((size_t)9 * N, 0.f), v0((size_t)3 * N, 0.f);
    h->mat_host.assign(mat, mat + N);
    bool simple = true;
    for (int i = 0; i < N; i++) {
        if (mat_cls[i] < 0 || mat_cls[i] > 0xffff || mat[i] < 0 || mat[i] > 0xffff) FAIL(h, "material id out of range");
        int bits = (mat_cls[i] & 0xffff) | ((mat[i] & 0xffff) << 16);
        float w; std::memcpy(&w, &bits, 4);
        info[i] = make_float4(mu[i], lam[i], h->cfg.p_vol * rho[i], w);      // mass = p_vol * rho, mpm:174
        simple = simple && mu[i] == 0.f && mat_cls[i] == FE_MAT_LIQUID;
        F0[(size_t)i * 9] = F0[(size_t)i * 9 + 4] = F0[(size_t)i * 9 + 8] = 1.f;
    }
    h->all_simple_liquid = simple;
    {   // one material record for every particle?  (collector scenes pick particles by material through the per-slot record: they keep it)
        bool uni = N > 0;
        for (int i = 1; i < N && uni; i++) uni = std::memcmp(&info[i], &info[0], sizeof(float4)) == 0;
        h->S.uni = uni ? 1 : 0;
        if (uni) { h->S.uinfo[0] = info[0].x; h->S.uinfo[1] = info[0].y; h->S.uinfo[2] = info[0].z; h->S.uinfo[3] = info[0].w; }
    }
    // init_bodies, mpm:176-201
    h->has_rigid = false; h->n_bodies = 0;
    for (int i = 0; i < N; i++) {
        const int b = body_id ? body_id[i] : 0;
        if (b < 0) FAIL(h, "negative body_id");
        if (b + 1 > h->n_bodies) h->n_bodies = b + 1;
        if (mat_cls[i] == FE_MAT_RIGID) h->has_rigid = true;
    }
    for (void* q : {(void*)h->bodies_dev, (void*)h->body_start, (void*)h->body_pids}) if (q) (void)hipFree(q);
    h->bodies_dev = nullptr; h->body_start = nullptr; h->body_pids = nullptr;
    if (h->has_rigid) {
        std::vector<RigidBody> bodies(h->n_bodies);
        std::vector<int> cnt(h->n_bodies, 0);
        std::vector<std::vector<int>> members(h->n_bodies);
        std::memset(bodies.data(), 0, sizeof(RigidBody) * bodies.size());
        for (int i = 0; i < N; i++) {
            const int b = body_id ? body_id[i] : 0;
            if (cnt[b]++ == 0) bodies[b].rigid = mat_cls[i] == FE_MAT_RIGID;       // mat_cls[body_id == b][0], mpm:201
            if (mat_cls[i] == FE_MAT_RIGID) members[b].push_back(i);
        }
        std::vector<int> start(h->n_bodies + 1, 0), pids;
        for (int b = 0; b < h->n_bodies; b++) {
            bodies[b].inv_n = cnt[b] ? 1.f / (float)cnt[b] : 0.f;
            pids.insert(pids.end(), members[b].begin(), members[b].end());
            start[b + 1] = (int)pids.size();
        }
        if (dev_alloc(h, &h->bodies_dev, h->n_bodies) || dev_alloc(h, &h->body_start, h->n_bodies + 1) || dev_alloc(h, &h->body_pids, pids.size())) return 1;
        HIPCK(h, hipMemcpyAsync(h->bodies_dev, bodies.data(), sizeof(RigidBody) * h->n_bodies, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(h->body_start, start.data(), sizeof(int) * start.size(), hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(h->body_pids, pids.data(), sizeof(int) * pids.size(), hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    HIPCK(h, hipMemcpyAsync(h->pinfo, info.data(), sizeof(float4) * h->Np, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->tbl_of_frame[0] = 0;
    h->tail_used = true;
    h->keys_frame = -1;
    std::fill(h->fiso.begin(), h->fiso.end(), 0);
    h->gcompact[0] = h->gcompact[1] = false;
    h->gpartial[0] = h->gpartial[1] = false;
    return upload_planes(h, h->frame(0), h->pid_of(0), x, v0.data(), C0.data(), F0.data(), used, 0);
}

int fe_substep(FeEngine* h, int f, int f_global, int act) {
    FE_ENTRY(h);
    if (f < 0 || f >= h->L) FAIL(h, "substep frame out of range");
    if (substep_fwd(h, f, f_global, act)) return 1;
    return check_async(h);
}
int fe_substep_grad(FeEngine* h, int f, int f_global, int act) {
    FE_ENTRY(h);
    if (f < 0 || f >= h->L) FAIL(h, "substep frame out of range");
    if (substep_bwd(h, f, f_global, act)) return 1;
    return check_async(h);
}
int fe_step(FeEngine* h, int f0, int f_global0, int n, int act) {
    FE_ENTRY(h);
    if (f0 < 0 || f0 + n > h->L) FAIL(h, "step frames out of range");
    bool pending = false;
    for (int i = 0; i < n; i++) {
        const bool defer = i + 1 < n && fusable_fwd(h, f0 + i + 1);
        if (substep_fwd(h, f0 + i, f_global0 + i, act, pending, defer)) return 1;
        pending = defer;
    }
    return check_async(h);
}
int fe_step_grad(FeEngine* h, int f0, int f_global0, int n, int act) {
    FE_ENTRY(h);
    if (f0 < 0 || f0 + n > h->L) FAIL(h, "step frames out of range");
    // (the call's last substep leaves the adjoint of frame f0 in the order of frame f0 - 1: where the next call of the sweep starts --
    //  fluidlab's step_grad is one call per env step, and with K = n_substeps a sort lies on every call boundary)
    // (i > 0: the adjoint of that frame is read by the next substep of this call and by nobody else -- its F may stay compact)
    bool done = false;
    if (n > 1 && fetch_gs_flags(h)) return 1;            // (fusable_bwd asks which frames have their grid stored)
    for (int i = n - 1; i >= 0; i--) {
        const bool fuse = i > 0 && fusable_bwd(h, f0 + i);
        if (substep_bwd(h, f0 + i, f_global0 + i, act, f0 + i > 0 ? f0 + i - 1 : -1, i > 0, done, fuse)) return 1;
        done = fuse;
    }
    return check_async(h);
}
int fe_step_batch(FeEngine** hs, int n_env, int f0, int f_global0, int n, int act) {
    if (!hs || n_env < 1) return 1;
    for (int e = 0; e < n_env; e++) if (!hs[e]) return 1;
    FeEngine* h = hs[0];
    FE_ENTRY(h);
    if (!batchable(hs, n_env)) {                              // scenes that cannot share launches: one engine after the other
        for (int e = 0; e < n_env; e++) if (fe_step(hs[e], f0, f_global0, n, act)) { h->err = hs[e]->err; return 1; }
        return 0;
    }
    if (f0 < 0 || f0 + n > h->L) FAIL(h, "step frames out of range");
    BatchStreams bs(hs, n_env);
    bool pending = false;
    for (int i = 0; i < n; i++) {
        bool defer = i + 1 < n;
        for (int e = 0; e < n_env && defer; e++) defer = fusable_fwd(hs[e], f0 + i + 1);
        if (substep_fwd_batch(hs, n_env, f0 + i, f_global0 + i, act, pending, defer)) { for (int e = 1; e < n_env; e++) if (!hs[e]->err.empty()) h->err = hs[e]->err; return 1; }
        pending = defer;
    }
    return check_async(h);
}
int fe_step_grad_batch(FeEngine** hs, int n_env, int f0, int f_global0, int n, int act) {
    if (!hs || n_env < 1) return 1;
    for (int e = 0; e < n_env; e++) if (!hs[e]) return 1;
    FeEngine* h = hs[0];
    FE_ENTRY(h);
    if (!batchable(hs, n_env)) {
        for (int e = 0; e < n_env; e++) if (fe_step_grad(hs[e], f0, f_global0, n, act)) { h->err = hs[e]->err; return 1; }
        return 0;
    }
    if (f0 < 0 || f0 + n > h->L) FAIL(h, "step frames out of range");
    BatchStreams bs(hs, n_env);
    bool done = false;
    if (n > 1) for (int e = 0; e < n_env; e++) if (fetch_gs_flags(hs[e])) { h->err = hs[e]->err; return 1; }
    for (int i = n - 1; i >= 0; i--) {
        bool fuse = i > 0;
        for (int e = 0; e < n_env && fuse; e++) fuse = hs[e]->all_simple_liquid && fusable_bwd(hs[e], f0 + i);      // (the batch has the SVD-free fused kernel only)
        if (substep_bwd_batch(hs, n_env, f0 + i, f_global0 + i, act, done, fuse)) { for (int e = 1; e < n_env; e++) if (!hs[e]->err.empty()) h->err = hs[e]->err; return 1; }
        done = fuse;
    }
    return check_async(h);
}

int fe_get_frame(FeEngine* h, int f, fe_real* x, fe_real* v, fe_real* C, fe_real* F, int* used) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (F) full_F_of_frame(h, f);
    if (download_planes(h, h->frame(f), h->pid_of(f), x, v, C, F, used)) return 1;
    return check_device_errors(h);
}
int fe_set_frame(FeEngine* h, int f, const fe_real* x, const fe_real* v, const fe_real* C, const fe_real* F, const int* used) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (f == h->keys_frame) h->keys_frame = -1;                  // (the sort's pre-counted keys describe what the frame held)
    h->tail_used = true;                                      // (the host may have put particles in use behind the order's work items: fuse_grid_ok)
    h->gs_host_valid = false;
    if (h->gs_cap > 0) HIPCK(h, hipMemsetAsync(h->gs_flag + f, 0, sizeof(int), h->stream));     // the stored grid of this frame is stale now
    if (F) h->fiso[f] = 0;                                    // (k_pack overwrites all nine words of F)
    return upload_planes(h, h->frame(f), h->pid_of(f), x, v, C, F, used, 0);
}
// device-pointer variants: k_unpack / k_pack work straight on the caller's device arrays, nothing crosses PCIe
int fe_get_frame_dev(FeEngine* h, int f, fe_real* x, fe_real* v, fe_real* C, fe_real* F, int* used) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    const int mask = (x ? 1 : 0) | (v ? 2 : 0) | (C ? 4 : 0) | (F ? 8 : 0) | (used ? 16 : 0);
    if (!mask || h->N == 0) return 0;
    if (F) full_F_of_frame(h, f);
    hipLaunchKernelGGL(k_unpack, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, h->frame(f), h->pid_of(f), x, v, C, F, used, mask);
    HIPCK(h, hipStreamSynchronize(h->stream));
    return check_device_errors(h);
}
int fe_set_frame_dev(FeEngine* h, int f, const fe_real* x, const fe_real* v, const fe_real* C, const fe_real* F, const int* used) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (f == h->keys_frame) h->keys_frame = -1;                  // (the sort's pre-counted keys describe what the frame held)
    h->tail_used = true;                                      // (the host may have put particles in use behind the order's work items: fuse_grid_ok)
    h->gs_host_valid = false;
    if (h->gs_cap > 0) HIPCK(h, hipMemsetAsync(h->gs_flag + f, 0, sizeof(int), h->stream));
    const int mask = (x ? 1 : 0) | (v ? 2 : 0) | (C ? 4 : 0) | (F ? 8 : 0) | (used ? 16 : 0);
    if (!mask || h->N == 0) return 0;
    if (F) h->fiso[f] = 0;
    hipLaunchKernelGGL(k_pack, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, h->frame(f), h->pid_of(f), x, v, C, F, used, mask, 0);
    HIPCK(h, hipStreamSynchronize(h->stream));
    return check_async(h);
}
int fe_copy_frame(FeEngine* h, int src, int dst) {
    FE_ENTRY(h);
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    if (src == dst) return 0;
    HIPCK(h, hipMemcpyAsync(h->frame(dst), h->frame(src), sizeof(float) * h->frame_stride, hipMemcpyDeviceToDevice, h->stream));
    h->tbl_of_frame[dst] = h->tbl_of_frame[src];
    h->fiso[dst] = h->fiso[src];
    h->tail_used = true;
    if (dst == h->keys_frame) h->keys_frame = -1;
    if (src == h->keys_frame) h->keys_frame = dst;             // (the copy holds the same particles in the same slots: the pre-counted keys are its as well -- fluidlab's window wraps frame L to frame 0 and sorts it)
    h->gs_host_valid = false;
    if (h->gs_cap > 0) HIPCK(h, hipMemsetAsync(h->gs_flag + dst, 0, sizeof(int), h->stream));
    return 0;
}
int fe_copy_grad(FeEngine* h, int src, int dst) {
    FE_ENTRY(h);
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    // adjoint frames are a ring of two (slot = f & 1); the `used` copy of mpm:604 is a frame copy
    if (h->gpartial[src & 1]) FAIL(h, "this frame's adjoint was passed on in registers inside a fused fe_step_grad call and is not in memory (after fe_step_grad(f0, n) only the adjoint of frame f0 is defined; option fuse_bwd = 0 keeps every frame's)");
    if ((src & 1) != (dst & 1)) {
        HIPCK(h, hipMemcpyAsync(h->grad(dst), h->grad(src), sizeof(float) * GR_WORDS * h->Np, hipMemcpyDeviceToDevice, h->stream));
        h->gtbl[dst & 1] = h->gtbl[src & 1];
        h->gcompact[dst & 1] = h->gcompact[src & 1];
        h->gpartial[dst & 1] = false;
    }
    if (src != dst) {
        FrameV s = frame_view(h->frame(src), h->Np), d = frame_view(h->frame(dst), h->Np);
        HIPCK(h, hipMemcpyAsync(d.used.ptr(), s.used.ptr(), sizeof(int) * h->Np, hipMemcpyDeviceToDevice, h->stream));
    }
    return 0;
}
int fe_reset_grad(FeEngine* h) {
    FE_ENTRY(h);
    if (smoke_reset_grad_impl(h)) return 1;
    HIPCK(h, hipMemsetAsync(h->grad_ptr[0], 0, sizeof(float) * h->grad_words(), h->stream));
    HIPCK(h, hipMemsetAsync(h->grad_ptr[1], 0, sizeof(float) * h->grad_words(), h->stream));
    h->gtbl[0] = h->gtbl[1] = -1;
    h->gcompact[0] = h->gcompact[1] = false;
    h->gpartial[0] = h->gpartial[1] = false;
    for (auto& E : h->effs) {
        const int Fm = h->L + 1, ad = E.p.action_dim > 0 ? E.p.action_dim : 1;
        HIPCK(h, hipMemsetAsync(E.p.gpos, 0, sizeof(float) * 3 * Fm, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gquat, 0, sizeof(float) * 4 * Fm, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gv, 0, sizeof(float) * 3 * Fm, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gw, 0, sizeof(float) * 3 * Fm, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gsa, 0, sizeof(float) * Fm, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gra, 0, sizeof(float) * Fm, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gabuf, 0, sizeof(float) * (size_t)h->cfg.max_action_steps * ad, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gabuf_p, 0, sizeof(float) * ad, h->stream));
    }
    return 0;
}
int fe_reset_grad_till_frame(FeEngine* h, int f) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    // particle adjoints: every substep_grad overwrites its ring slot, nothing to clear.
    return 0;
}
int fe_agent_set_collector(FeEngine* h, const FeBoundary* b, int mat) {
    FE_ENTRY(h);
    (void)hipSetDevice(h->device);
    h->has_collector = b != nullptr;
    h->collector_mat = mat;
    if (!b) return 0;
    const BoundaryP p = to_boundary(*b);
    if (!h->collector_dev && dev_alloc(h, &h->collector_dev, 1)) return 1;
    if (hipMemcpyOnStream(h, h->collector_dev, &p, sizeof(BoundaryP), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy failed"; return 1; }
    return 0;
}
int fe_agent_reset_grad_till_frame(FeEngine* h, int f) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    for (auto& E : h->effs) {
        if (f == 0) break;
        HIPCK(h, hipMemsetAsync(E.p.gpos, 0, sizeof(float) * 3 * f, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gquat, 0, sizeof(float) * 4 * f, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gv, 0, sizeof(float) * 3 * f, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gw, 0, sizeof(float) * 3 * f, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gsa, 0, sizeof(float) * f, h->stream));
        HIPCK(h, hipMemsetAsync(E.p.gra, 0, sizeof(float) * f, h->stream));
    }
    return 0;
}
int fe_get_grad(FeEngine* h, int f, fe_real* gx, fe_real* gv, fe_real* gC, fe_real* gF) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (h->gpartial[f & 1]) FAIL(h, "this frame's adjoint was passed on in registers inside a fused fe_step_grad call and is not in memory (after fe_step_grad(f0, n) only the adjoint of frame f0 is defined; option fuse_bwd = 0 keeps every frame's)");
    const int t = h->gtbl[f & 1] < 0 ? 0 : h->gtbl[f & 1];
    if (gF) full_F_of_grad(h, f);
    return download_planes(h, h->grad(f), h->tables[t].pid, gx, gv, gC, gF, nullptr);
}
int fe_add_grad(FeEngine* h, int f, const fe_real* gx, const fe_real* gv, const fe_real* gC, const fe_real* gF) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (h->gpartial[f & 1]) FAIL(h, "this frame's adjoint was passed on in registers inside a fused fe_step_grad call and is not in memory (after fe_step_grad(f0, n) only the adjoint of frame f0 is defined; option fuse_bwd = 0 keeps every frame's)");
    full_F_of_grad(h, f);                                     // (k_pack reads and rewrites all the planes)
    return upload_planes(h, h->grad(f), h->tables[grad_table_for_frame(h, f)].pid, gx, gv, gC, gF, nullptr, 1);
}
// device-pointer variant (a loss evaluated on the GPU hands its adjoint over without crossing PCIe)
int fe_add_grad_dev(FeEngine* h, int f, const fe_real* gx, const fe_real* gv, const fe_real* gC, const fe_real* gF) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (h->gpartial[f & 1]) FAIL(h, "this frame's adjoint was passed on in registers inside a fused fe_step_grad call and is not in memory (after fe_step_grad(f0, n) only the adjoint of frame f0 is defined; option fuse_bwd = 0 keeps every frame's)");
    const int gt = grad_table_for_frame(h, f);
    const int mask = (gx ? 1 : 0) | (gv ? 2 : 0) | (gC ? 4 : 0) | (gF ? 8 : 0);
    if (!mask || h->N == 0) return 0;
    full_F_of_grad(h, f);
    hipLaunchKernelGGL(k_pack, pgrid(h), dim3(256), 0, h->stream, h->N, (size_t)h->Np, h->grad(f), h->tables[gt].pid, gx, gv, gC, gF, (const int*)nullptr, mask, 1);
    HIPCK(h, hipStreamSynchronize(h->stream));
    return check_async(h);
}
int fe_get_mat(FeEngine* h, int* mat) {
    FE_ENTRY(h);
    if ((int)h->mat_host.size() != h->N) FAIL(h, "particles not initialised");
    std::memcpy(mat, h->mat_host.data(), sizeof(int) * h->N);
    return 0;
}

// ---- effectors
int fe_add_effector(FeEngine* h, const FeEffectorDesc* d, const fe_real* random_vector) {
    FE_ENTRY(h);
    auto bad = [&](const char* m) { h->err = m; return -1; };
    if (!d || d->struct_size != (int)sizeof(FeEffectorDesc)) return bad("FeEffectorDesc size mismatch");
    if (!(d->action_dim == 0 || d->action_dim == 3 || d->action_dim == 6 || (d->type == FE_EFF_AIRCON && d->action_dim == 8)))
        return bad("action_dim must be 0, 3 or 6 (8 for an AirCon)");
    if ((int)h->effs.size() >= FE_MAX_EFF) return bad("too many effectors");
    if (d->type == FE_EFF_INJECTOR && find_injector(h) >= 0) return bad("only one injector per agent (agent_injector.py:17)");
    EffHost E; std::memset(&E.p, 0, sizeof(E.p));
    EffP& p = E.p;
    p.type = d->type; p.action_dim = d->action_dim;
    for (int i = 0; i < 8; i++) { p.scale_v[i] = d->action_scale_v[i]; p.scale_p[i] = d->action_scale_p[i]; }
    p.bnd = to_boundary(d->boundary);
    p.flux = d->flux; p.radius = d->radius;
    for (int i = 0; i < 3; i++) { p.inject_v[i] = d->inject_v[i]; p.inject_p[i] = d->inject_p[i]; }
    p.locally_random = d->locally_random; p.randomize_inject_v = d->randomize_inject_v; p.random_length = d->random_length;
    const int Fm = h->L + 1, ad = d->action_dim > 0 ? d->action_dim : 1;
    if (dev_alloc(h, &p.pos, 3 * Fm) || dev_alloc(h, &p.quat, 4 * Fm) || dev_alloc(h, &p.v, 3 * Fm) || dev_alloc(h, &p.w, 3 * Fm) ||
        dev_alloc(h, &p.gpos, 3 * Fm) || dev_alloc(h, &p.gquat, 4 * Fm) || dev_alloc(h, &p.gv, 3 * Fm) || dev_alloc(h, &p.gw, 3 * Fm) ||
        dev_alloc(h, &p.sa, Fm) || dev_alloc(h, &p.ra, Fm) || dev_alloc(h, &p.gsa, Fm) || dev_alloc(h, &p.gra, Fm) ||
        dev_alloc(h, &p.abuf, (size_t)h->cfg.max_action_steps * ad) || dev_alloc(h, &p.gabuf, (size_t)h->cfg.max_action_steps * ad) ||
        dev_alloc(h, &p.abuf_p, ad) || dev_alloc(h, &p.gabuf_p, ad)) return -1;
    if (d->type == FE_EFF_INJECTOR) {
        if (!random_vector || d->random_length <= 0 || d->flux <= 0) return bad("injector needs random_vector, random_length, flux");
        size_t cnt = (size_t)d->random_length * d->flux * 3;
        if (dev_alloc(h, &p.random_vector, cnt, false)) return -1;
        if (hipMemcpyOnStream(h, p.random_vector, random_vector, sizeof(float) * cnt, hipMemcpyHostToDevice) != hipSuccess) return bad("hipMemcpy failed");
    }
    E.act_id.assign(Fm, 0);
    h->effs.push_back(E);
    if (hipMemcpyOnStream(h, h->effs_dev + (h->effs.size() - 1), &h->effs.back().p, sizeof(EffP), hipMemcpyHostToDevice) != hipSuccess) return bad("hipMemcpy failed");
    return (int)h->effs.size() - 1;
}
int fe_eff_set_act_range(FeEngine* h, int e, const int* act_range, int n) {
    FE_ENTRY(h);
    CHECK_EFF(h, e);
    EffHost& E = h->effs[e];
    E.act_range.assign(act_range, act_range + n);
    if (n > 0) E.act_id[0] = act_range[0];         // injector.py:68 (sic: the first pool id)
    std::vector<int> pool(h->Np, -1);
    for (int i = 0; i < n; i++) {
        if (act_range[i] < 0 || act_range[i] >= h->N) FAIL(h, "act_range entry out of range");
        pool[act_range[i]] = i;
    }
    HIPCK(h, hipMemcpyAsync(h->pool_idx, pool.data(), sizeof(int) * h->Np, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_eff_get_state(FeEngine* h, int e, int f, fe_real* s) {
    FE_ENTRY(h);
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    EffHost& E = h->effs[e];
    HIPCK(h, hipMemcpyAsync(s, E.p.pos + f * 3, sizeof(float) * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(s + 3, E.p.quat + f * 4, sizeof(float) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    s[7] = (float)E.act_id[f];
    return 0;
}
int fe_eff_set_state(FeEngine* h, int e, int f, const fe_real* s) {
    FE_ENTRY(h);
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    EffHost& E = h->effs[e];
    HIPCK(h, hipMemcpyAsync(E.p.pos + f * 3, s, sizeof(float) * 3, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(E.p.quat + f * 4, s + 3, sizeof(float) * 4, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    E.act_id[f] = (int)s[7];
    return 0;
}
int fe_eff_get_vw(FeEngine* h, int e, int f, fe_real* v3, fe_real* w3) {
    FE_ENTRY(h);
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    HIPCK(h, hipMemcpyAsync(v3, h->effs[e].p.v + f * 3, sizeof(float) * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(w3, h->effs[e].p.w + f * 3, sizeof(float) * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_eff_get_sr(FeEngine* h, int e, int f, fe_real* s, fe_real* r) {
    FE_ENTRY(h);
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    HIPCK(h, hipMemcpyAsync(s, h->effs[e].p.sa + f, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(r, h->effs[e].p.ra + f, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_eff_set_sr(FeEngine* h, int e, int f, fe_real s, fe_real r) {
    FE_ENTRY(h);
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    HIPCK(h, hipMemcpyAsync(h->effs[e].p.sa + f, &s, sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->effs[e].p.ra + f, &r, sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_eff_set_vw(FeEngine* h, int e, int f, const fe_real* v3, const fe_real* w3) {
    FE_ENTRY(h);
    CHECK_EFF(h, e); CHECK_FRAME(h, f);
    HIPCK(h, hipMemcpyAsync(h->effs[e].p.v + f * 3, v3, sizeof(float) * 3, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->effs[e].p.w + f * 3, w3, sizeof(float) * 3, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_eff_set_action(FeEngine* h, int e, int s, int s_global, int n_substeps, const fe_real* action) {
    FE_ENTRY(h);
    CHECK_EFF(h, e);
    EffP& p = h->effs[e].p;
    if (p.action_dim == 0) return 0;
    if (s_global < 0 || s_global >= h->cfg.max_action_steps) FAIL(h, "s_global out of range");     // effector.py:263
    if (s < 0 || (s + 1) * n_substeps > h->L + 1) FAIL(h, "s out of range");                          // effector.py:264
    Act6 a; std::memset(&a, 0, sizeof(a));
    for (int j = 0; j < p.action_dim; j++) a.a[j] = action[j];
    hipLaunchKernelGGL(k_eff_set_action, dim3(1), dim3(64), 0, h->stream, p, s, s_global, n_substeps, a);
    return check_async(h);
}
int fe_eff_set_action_grad(FeEngine* h, int e, int s, int s_global, int n_substeps) {
    FE_ENTRY(h);
    CHECK_EFF(h, e);
    EffP& p = h->effs[e].p;
    if (p.action_dim == 0) return 0;
    if (s_global < 0 || s_global >= h->cfg.max_action_steps) FAIL(h, "s_global out of range");
    if (s < 0 || (s + 1) * n_substeps > h->L + 1) FAIL(h, "s out of range");
    hipLaunchKernelGGL(k_eff_set_action_grad, dim3(1), dim3(64), 0, h->stream, p, s, s_global, n_substeps);
    return check_async(h);
}
int fe_eff_apply_action_p(FeEngine* h, int e, const fe_real* action_p) {
    FE_ENTRY(h);
    CHECK_EFF(h, e);
    EffP& p = h->effs[e].p;
    if (p.action_dim == 0) return 0;
    Act6 a; std::memset(&a, 0, sizeof(a));
    for (int j = 0; j < p.action_dim; j++) a.a[j] = action_p[j];
    hipLaunchKernelGGL(k_eff_apply_p, dim3(1), dim3(64), 0, h->stream, p, a);
    return check_async(h);
}
int fe_eff_apply_action_p_grad(FeEngine* h, int e) {
    FE_ENTRY(h);
    CHECK_EFF(h, e);
    EffP& p = h->effs[e].p;
    if (p.action_dim == 0) return 0;
    hipLaunchKernelGGL(k_eff_apply_p_grad, dim3(1), dim3(64), 0, h->stream, p);
    return check_async(h);
}
int fe_eff_get_action_grad(FeEngine* h, int e, int s, int n, fe_real* grad) {
    FE_ENTRY(h);
    CHECK_EFF(h, e);
    EffP& p = h->effs[e].p;
    const int ad = p.action_dim;
    if (ad == 0) return 0;
    if (s < 0 || s + n > h->cfg.max_action_steps) FAIL(h, "action grad range out of bounds");
    if (n > 0) HIPCK(h, hipMemcpyAsync(grad, p.gabuf + (size_t)s * ad, sizeof(float) * (size_t)n * ad, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(grad + (size_t)n * ad, p.gabuf_p, sizeof(float) * ad, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_agent_copy_frame(FeEngine* h, int src, int dst) {
    FE_ENTRY(h);
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    for (auto& E : h->effs) {
        hipLaunchKernelGGL(k_eff_copy, dim3(1), dim3(64), 0, h->stream, E.p, src, dst, 0);
        if (E.p.type == FE_EFF_INJECTOR) E.act_id[dst] = E.act_id[src];
    }
    return check_async(h);
}
int fe_agent_copy_grad(FeEngine* h, int src, int dst) {
    FE_ENTRY(h);
    CHECK_FRAME(h, src); CHECK_FRAME(h, dst);
    for (auto& E : h->effs) hipLaunchKernelGGL(k_eff_copy, dim3(1), dim3(64), 0, h->stream, E.p, src, dst, 1);
    return check_async(h);
}

// ---- SDF colliders: statics (statics.py, static.py:25-104) and Rigid effector meshes (rigid.py:19-24, dynamic.py)
static int build_sdf(FeEngine* h, const FeSdfDesc* d, const fe_real* voxels, SdfP& s, float** vox_out) {
    if (hipSetDevice(h->device) != hipSuccess) FAIL(h, "hipSetDevice failed");
    if (!d || d->struct_size != (int)sizeof(FeSdfDesc) || d->res < 2 || !voxels) FAIL(h, "bad FeSdfDesc");
    s.res = d->res; s.friction = d->friction; s.softness = d->softness;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) s.T[r * 4 + c] = d->T_mesh_to_voxels[r * 4 + c];
    double A[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = d->T_mesh_to_voxels[r * 4 + c];
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (det == 0.0) FAIL(h, "singular T_mesh_to_voxels");
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        s.Rinv[j * 3 + i] = (float)((A[i1][j1] * A[i2][j2] - A[i1][j2] * A[i2][j1]) / det);
    }
    const size_t nv = (size_t)d->res * d->res * d->res;
    float* vox = nullptr;
    if (dev_alloc(h, &vox, nv, false)) return 1;
    HIPCK(h, hipMemcpyOnStream(h, vox, voxels, sizeof(float) * nv, hipMemcpyHostToDevice));
    s.vox = vox;
    *vox_out = vox;
    return 0;
}
int fe_add_static(FeEngine* h, const FeSdfDesc* d, const fe_real* voxels) {
    FE_ENTRY(h);
    if ((int)h->statics_host.size() >= FE_MAX_STATICS) { h->err = "add_static: too many static colliders"; return -1; }
    SdfP s; float* vox = nullptr;
    if (build_sdf(h, d, voxels, s, &vox)) return -1;
    h->statics_vox.push_back(vox);
    h->statics_host.push_back(s);
    if (!h->statics_dev && dev_alloc(h, &h->statics_dev, FE_MAX_STATICS)) return -1;
    if (hipMemcpyOnStream(h, h->statics_dev, h->statics_host.data(), sizeof(SdfP) * h->statics_host.size(), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy failed"; return -1; }
    h->gs_host_valid = false;
    return (int)h->statics_host.size() - 1;
}
int fe_eff_set_mesh(FeEngine* h, int e, const FeSdfDesc* d, const fe_real* voxels) {
    FE_ENTRY(h);
    if (e < 0 || e >= (int)h->effs.size()) FAIL(h, "effector index out of range");
    SdfP s; float* vox = nullptr;
    if (build_sdf(h, d, voxels, s, &vox)) return 1;
    h->mesh_vox.push_back(vox);
    h->effs[e].p.has_mesh = 1; h->effs[e].p.mesh = s;
    HIPCK(h, hipMemcpyOnStream(h, h->effs_dev + e, &h->effs[e].p, sizeof(EffP), hipMemcpyHostToDevice));
    h->has_mesh_effector = true;
    if (!h->hit_dev) {
        if (dev_alloc(h, &h->hit_dev, (size_t)(h->L + 1) * h->Np) ||
            dev_alloc(h, &h->hit_list, (size_t)h->Np) || dev_alloc(h, &h->hit_count, 1)) return 1;
    }
    return 0;
}

// ---- loss
int fe_loss_alloc(FeEngine* h, int max_loss_steps) {
    FE_ENTRY(h);
    if (max_loss_steps <= 0) FAIL(h, "max_loss_steps must be positive");
    if (h->tgt) { (void)hipFree(h->tgt); (void)hipFree(h->chamfer); (void)hipFree(h->step_loss); h->tgt = h->chamfer = h->step_loss = nullptr; }
    h->loss_steps = max_loss_steps;
    // all targets stay resident in HBM (the reference re-uploads one per step: shapematching_loss.py:73,77)
    if (dev_alloc(h, &h->tgt, (size_t)max_loss_steps * h->N * 3) || dev_alloc(h, &h->chamfer, max_loss_steps) || dev_alloc(h, &h->step_loss, max_loss_steps)) return 1;
    return 0;
}
int fe_loss_set_target(FeEngine* h, int s, const fe_real* x) {
    FE_ENTRY(h);
    if (s < 0 || s >= h->loss_steps) FAIL(h, "loss step out of range");
    HIPCK(h, hipMemcpyAsync(h->tgt + (size_t)s * h->N * 3, x, sizeof(float) * 3 * h->N, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int fe_loss_clear(FeEngine* h) {
    FE_ENTRY(h);
    if (!h->loss_steps) return 0;
    HIPCK(h, hipMemsetAsync(h->chamfer, 0, sizeof(float) * h->loss_steps, h->stream));
    HIPCK(h, hipMemsetAsync(h->step_loss, 0, sizeof(float) * h->loss_steps, h->stream));
    return 0;
}
int fe_loss_step(FeEngine* h, int s, int f, int matching_mat, fe_real weight) {
    FE_ENTRY(h);
    if (s < 0 || s >= h->loss_steps) FAIL(h, "loss step out of range");
    CHECK_FRAME(h, f);
    hipLaunchKernelGGL(k_loss_fwd, pgrid(h), dim3(256), 0, h->stream, h->S, h->frame(f), h->pid_of(f), h->pinfo,
                       h->tgt + (size_t)s * h->N * 3, matching_mat, h->chamfer + s);
    hipLaunchKernelGGL(k_loss_sum, dim3(1), dim3(64), 0, h->stream, h->chamfer + s, h->step_loss + s, weight);
    return check_async(h);
}
int fe_loss_step_grad(FeEngine* h, int s, int f, int matching_mat, fe_real weight, fe_real step_loss_grad) {
    FE_ENTRY(h);
    if (s < 0 || s >= h->loss_steps) FAIL(h, "loss step out of range");
    CHECK_FRAME(h, f);
    if (h->gpartial[f & 1]) FAIL(h, "this frame's adjoint was passed on in registers inside a fused fe_step_grad call and is not in memory (after fe_step_grad(f0, n) only the adjoint of frame f0 is defined; option fuse_bwd = 0 keeps every frame's)");
    const int gt = grad_table_for_frame(h, f), ft = h->tbl_of_frame[f];
    hipLaunchKernelGGL(k_loss_bwd, pgrid(h), dim3(256), 0, h->stream, h->S, h->frame(f), h->grad(f), h->tables[gt].pid, gt == ft ? (const int*)nullptr : h->tables[ft].slot_of_pid, h->pinfo,
                       h->tgt + (size_t)s * h->N * 3, matching_mat, weight * step_loss_grad);
    return check_async(h);
}
int fe_loss_get(FeEngine* h, fe_real* step_loss, int n) {
    FE_ENTRY(h);
    if (n > h->loss_steps) FAIL(h, "loss_get: n too large");
    HIPCK(h, hipMemcpyAsync(step_loss, h->step_loss, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- measurement
// utils/mesh.py:63-96 (mesh_to_sdf): exact signed distance of arbitrary points to a triangle mesh, see fe_mesh.h
int fe_mesh_sdf(int device, const float* verts, int nv, const int* faces, int nf, const float* points, long long n_points, float* sdf) {
    if (nv <= 0 || nf <= 0 || n_points < 0 || !verts || !faces || (n_points > 0 && (!points || !sdf))) { g_create_err = "fe_mesh_sdf: invalid arguments"; return 1; }
    for (long long i = 0; i < (long long)nf * 3; i++) if (faces[i] < 0 || faces[i] >= nv) { g_create_err = "fe_mesh_sdf: face index out of range"; return 1; }
    if (n_points == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device visible: the MI355X engine has no CPU fallback"; return 1; }
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) { g_create_err = "HIP device ordinal out of range"; return 1; }
    float *dv = nullptr, *dp = nullptr, *ds = nullptr; int* df = nullptr;
    auto done = [&](int rc, const char* msg) { if (msg) g_create_err = msg; (void)hipFree(dv); (void)hipFree(df); (void)hipFree(dp); (void)hipFree(ds); return rc; };
    if (hipMalloc(&dv, sizeof(float) * 3 * nv) != hipSuccess || hipMalloc(&df, sizeof(int) * 3 * nf) != hipSuccess ||
        hipMalloc(&dp, sizeof(float) * 3 * n_points) != hipSuccess || hipMalloc(&ds, sizeof(float) * n_points) != hipSuccess)
        return done(1, "fe_mesh_sdf: hipMalloc failed");
    if (hipMemcpy(dv, verts, sizeof(float) * 3 * nv, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(df, faces, sizeof(int) * 3 * nf, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dp, points, sizeof(float) * 3 * n_points, hipMemcpyHostToDevice) != hipSuccess)
        return done(1, "fe_mesh_sdf: upload failed");
    hipLaunchKernelGGL(k_mesh_sdf, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, 0, dv, df, nf, dp, n_points, ds);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return done(1, "fe_mesh_sdf: kernel failed");
    if (hipMemcpy(sdf, ds, sizeof(float) * n_points, hipMemcpyDeviceToHost) != hipSuccess) return done(1, "fe_mesh_sdf: download failed");
    return done(0, nullptr);
}

int fe_get_stats(FeEngine* h, int f, FeStats* out) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    const int ncell = h->nb * h->nb * h->nb * 64;
    HIPCK(h, hipMemsetAsync(h->counters, 0, sizeof(unsigned long long) * 4, h->stream));
    hipLaunchKernelGGL(k_stats_mark, pgrid(h), dim3(256), 0, h->stream, h->S, h->frame(f), h->node_mark, h->counters);
    hipLaunchKernelGGL(k_stats_count, dim3((ncell + 255) / 256), dim3(256), 0, h->stream, ncell, h->node_mark, h->counters);
    unsigned long long c[4];
    int slow = 0;
    HIPCK(h, hipMemcpyAsync(c, h->counters, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(&slow, h->slow_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemsetAsync(h->slow_dev, 0, sizeof(int), h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    out->n_used = (long long)c[0]; out->n_cells_touched = (long long)c[1]; out->n_blocks_active = (long long)c[2];
    out->n_slow_path = slow; out->bytes_state = (long long)h->bytes;
    return check_async(h);
}
int fe_get_work_stats(FeEngine* h, int f, long long out[FE_WORK_STATS]) { return fe_get_work_stats_n(h, f, out, FE_WORK_STATS); }
int fe_get_work_stats_n(FeEngine* h, int f, long long* out_, int n_out) {
    FE_ENTRY(h);
    CHECK_FRAME(h, f);
    if (!out_ || n_out < 0) FAIL(h, "fe_get_work_stats_n: bad output buffer");
    long long out[FE_WORK_STATS];
    struct Copy { long long* src; long long* dst; int n; ~Copy() { for (int i = 0; i < n && i < FE_WORK_STATS; i++) dst[i] = src[i]; } } copy_out{out, out_, n_out};
    for (int i = 0; i < 24; i++) out[i] = 0;
    const int t = h->tbl_of_frame[f];
    if (t < 0 || t >= (int)h->tables.size() || !h->tables[t].meta) return 0;
    int meta[16];
    HIPCK(h, hipMemcpyAsync(meta, h->tables[t].meta, sizeof(meta), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 5; i++) out[i] = meta[i];
    out[3] += meta[11] + meta[12];                            // full pairs + the leftovers of odd item counts: the workgroups multi-item blocks get in the pairs-only list
    out[4] += meta[8]; out[14] = meta[8]; out[15] = meta[10];                     // single-item blocks: the small ones (four to a quad unit) are counted apart
    out[13] = meta[7] - meta[1];                              // particles of loose blocks: slots [tail_start, tail_start + this)
    out[16] = meta[14]; out[17] = meta[15]; out[18] = meta[13]; out[19] = meta[11]; out[20] = meta[12];
    std::vector<int4> items((size_t)std::max(meta[0], 0));
    if (!items.empty()) {
        HIPCK(h, hipMemcpyAsync(items.data(), h->tables[t].items, sizeof(int4) * items.size(), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    int last_block = -1;
    for (const int4& it : items) {
        const int c = it.z;
        const int b = c <= 1 ? 0 : c <= 4 ? 1 : c <= 8 ? 2 : c <= 16 ? 3 : c <= 32 ? 4 : c <= 64 ? 5 : 6;
        out[5 + b]++;
        if (it.x != last_block) { out[12]++; last_block = it.x; }
        if (h->S.lsplit) {                                    // the item's waves (64 particles each) that split their stencils over idle lanes (lane_split)
            for (int w0 = 0; w0 < c; w0 += 64) { const int cnt = std::min(64, c - w0); if (cnt <= FE_SPLIT9_MAX) out[21]++; else if (cnt <= FE_SPLIT3_MAX) out[22]++; }
        }
    }
    return check_async(h);
}
#ifdef FE_TIMELINE
// profiling builds only (not part of include/fluidengine.h): the stamps of the last launch of kernel `kid` (order of KNAMES)
int fe_timeline_read(FeEngine* h, int kid, unsigned long long* out) {
    FE_ENTRY(h);
    if (kid < 0 || kid >= KID_COUNT || !h->tl_dev) FAIL(h, "no timeline");
    HIPCK(h, hipMemcpyAsync(out, h->tl_dev + (size_t)kid * TL_WGS * 9, sizeof(unsigned long long) * TL_WGS * 9, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return 0;
}
#endif
int fe_timer_start(FeEngine* h) { FE_ENTRY(h); HIPCK(h, hipEventRecord(h->ev_t0, h->stream)); return 0; }
double fe_timer_stop_ms(FeEngine* h) {
    FE_ENTRY(h);
    if (hipEventRecord(h->ev_t1, h->stream) != hipSuccess || hipEventSynchronize(h->ev_t1) != hipSuccess) { h->err = "timer stop failed"; return -1.0; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1) != hipSuccess) { h->err = "hipEventElapsedTime failed"; return -1.0; }
    return (double)ms;
}
int fe_profile_enable(FeEngine* h, int on) {
    FE_ENTRY(h);
    prof_drain(h);
    h->prof_on = on != 0;
    if (on) for (int i = 0; i < KID_COUNT; i++) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
    return 0;
}
int fe_profile_read(FeEngine* h, char* buf, int buf_len, double* ms_total, long long* launches, int cap) {
    FE_ENTRY(h);
    prof_drain(h);
    std::string names;
    for (int i = 0; i < KID_COUNT; i++) { if (i) names += "\n"; names += KNAMES[i]; }
    std::snprintf(buf, buf_len, "%s", names.c_str());
    for (int i = 0; i < KID_COUNT && i < cap; i++) { ms_total[i] = h->prof_ms[i]; launches[i] = h->prof_n[i]; }
    return KID_COUNT;
}

} // extern "C"
