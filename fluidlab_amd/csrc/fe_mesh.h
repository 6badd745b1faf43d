// fe_mesh.h -- triangle mesh -> signed distance at arbitrary query points, brute force on the GPU.
//
// The reference builds its collision SDFs (utils/mesh.py:63-87, compute_sdf_data) and its mesh-filled particle bodies
// (bodies.py:187-210) with the third-party packages mesh_to_sdf and trimesh: a point cloud from ~200 virtual scans, the
// distance to the nearest scan point, the sign from scan normals -- minutes per mesh on the CPU.  Here every query
// point visits every triangle: exact point-triangle distance, sign from the generalized winding number (robust for
// meshes that are not watertight).  128^3 points x 20k triangles is 4e10 pairs x ~150 VALU ops: tens of milliseconds on
// 256 CUs, so no spatial hierarchy is worth its code.
//
// Layout: one workgroup = 256 query points, one point per lane; the triangle list streams through LDS in tiles of 128
// triangles (9 floats each, SoA so that a tile read is a broadcast: every lane reads the same triangle).  VALU-bound by
// construction; HBM traffic is nf * 36 B per workgroup from L2 and 16 B per point.
#pragma once
#include <hip/hip_runtime.h>

#define MESH_TILE 128

struct MeshAcc { float d2; float w; };            // running min of squared distance, running sum of solid angles

// squared distance from p to triangle (a, b, c) -- closest-point regions of the triangle (Ericson, Real-Time Collision
// Detection 5.1.5), written on the vectors relative to p
__device__ __forceinline__ float tri_dist2(const float a[3], const float b[3], const float c[3]) {
    // a, b, c are already relative to the query point: the query is the origin
    const float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float d1 = -(ab[0] * a[0] + ab[1] * a[1] + ab[2] * a[2]);
    const float d2 = -(ac[0] * a[0] + ac[1] * a[1] + ac[2] * a[2]);
    const float d3 = -(ab[0] * b[0] + ab[1] * b[1] + ab[2] * b[2]);
    const float d4 = -(ac[0] * b[0] + ac[1] * b[1] + ac[2] * b[2]);
    const float d5 = -(ab[0] * c[0] + ab[1] * c[1] + ab[2] * c[2]);
    const float d6 = -(ac[0] * c[0] + ac[1] * c[1] + ac[2] * c[2]);
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    float q[3];
    if (d1 <= 0.f && d2 <= 0.f) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; }                                   // vertex a
    else if (d3 >= 0.f && d4 <= d3) { q[0] = b[0]; q[1] = b[1]; q[2] = b[2]; }                               // vertex b
    else if (d6 >= 0.f && d5 <= d6) { q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; }                               // vertex c
    else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {                                                           // edge ab
        const float t = d1 / (d1 - d3);
        q[0] = a[0] + t * ab[0]; q[1] = a[1] + t * ab[1]; q[2] = a[2] + t * ab[2];
    } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {                                                         // edge ac
        const float t = d2 / (d2 - d6);
        q[0] = a[0] + t * ac[0]; q[1] = a[1] + t * ac[1]; q[2] = a[2] + t * ac[2];
    } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {                                           // edge bc
        const float t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        q[0] = b[0] + t * (c[0] - b[0]); q[1] = b[1] + t * (c[1] - b[1]); q[2] = b[2] + t * (c[2] - b[2]);
    } else {                                                                                                  // face
        const float den = 1.f / (va + vb + vc), v = vb * den, w = vc * den;
        q[0] = a[0] + ab[0] * v + ac[0] * w; q[1] = a[1] + ab[1] * v + ac[1] * w; q[2] = a[2] + ab[2] * v + ac[2] * w;
    }
    return q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
}

// solid angle of triangle (a, b, c) seen from the origin (van Oosterom & Strackee 1983); the sum over a closed mesh is
// 4 pi times the winding number
__device__ __forceinline__ float tri_solid_angle(const float a[3], const float b[3], const float c[3]) {
    const float la = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const float lb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    const float lc = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    const float ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    const float bc = b[0] * c[0] + b[1] * c[1] + b[2] * c[2];
    const float ca = c[0] * a[0] + c[1] * a[1] + c[2] * a[2];
    return 2.f * atan2f(det, la * lb * lc + ab * lc + bc * la + ca * lb);
}

__global__ __launch_bounds__(256) void k_mesh_sdf(const float* __restrict__ verts, const int* __restrict__ faces, int nf,
                                                  const float* __restrict__ points, long long n_points, float* __restrict__ sdf) {
    __shared__ float s_tri[9][MESH_TILE];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_points;
    float p[3] = {0.f, 0.f, 0.f};
    if (live) { p[0] = points[i * 3]; p[1] = points[i * 3 + 1]; p[2] = points[i * 3 + 2]; }
    float best = 3.0e38f, omega = 0.f;
    for (int t0 = 0; t0 < nf; t0 += MESH_TILE) {
        __syncthreads();
        for (int l = threadIdx.x; l < MESH_TILE * 3; l += 256) {              // one (triangle, corner) per lane
            const int t = l / 3, k = l - 3 * t;
            const bool has = t0 + t < nf;
            const int vi = has ? faces[(size_t)(t0 + t) * 3 + k] : 0;
            s_tri[k * 3 + 0][t] = has ? verts[(size_t)vi * 3 + 0] : 0.f;
            s_tri[k * 3 + 1][t] = has ? verts[(size_t)vi * 3 + 1] : 0.f;
            s_tri[k * 3 + 2][t] = has ? verts[(size_t)vi * 3 + 2] : 0.f;
        }
        __syncthreads();
        const int n = min(MESH_TILE, nf - t0);
#pragma unroll 2
        for (int t = 0; t < n; t++) {
            const float a[3] = {s_tri[0][t] - p[0], s_tri[1][t] - p[1], s_tri[2][t] - p[2]};
            const float b[3] = {s_tri[3][t] - p[0], s_tri[4][t] - p[1], s_tri[5][t] - p[2]};
            const float c[3] = {s_tri[6][t] - p[0], s_tri[7][t] - p[1], s_tri[8][t] - p[2]};
            best = fminf(best, tri_dist2(a, b, c));
            omega += tri_solid_angle(a, b, c);
        }
    }
    if (live) {
        const float d = sqrtf(best);
        sdf[i] = fabsf(omega) > 6.2831853f ? -d : d;                          // |winding number| > 1/2 (either face orientation)  <=>  |sum of solid angles| > 2 pi
    }
}
