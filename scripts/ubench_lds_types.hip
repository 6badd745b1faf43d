// LDS atomic-add throughput by operand type on MI355X (same P2G-like pattern as ubench_atomics.hip), plus
// plain LDS read-modify-write and the DPP segmented pre-reduction alternative.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T> struct Tile { static __device__ T* get(); };
__shared__ unsigned long long s_raw[4 * 512];

template <typename T>
__global__ __launch_bounds__(256) void k_lds(int mode, int rounds, const int* __restrict__ rnd, float* out) {
    T* tile = (T*)s_raw;
    const int tid = threadIdx.x;
    for (int l = tid; l < 4 * 512; l += 256) tile[l] = (T)0;
    __syncthreads();
    int r = rnd[(blockIdx.x * 256 + tid) & 0xffff];
    for (int it = 0; it < rounds; it++) {
#pragma unroll
        for (int n = 0; n < 27; n++) {
            int node = mode == 0 ? ((tid + n * 7) & 511) : (((r >> 3) + n * 19 + it) % 216 + (n & 1));
            // mode 2: groups of 8 adjacent lanes share one node (cell-sorted particles, no aggregation)
            // mode 3: same nodes, but only the last lane of each group of 8 issues (after a wavefront segmented scan)
            if (mode >= 2) node = (((tid >> 3) * 37 + n * 19 + it) % 216) + (n & 1);
            if (mode == 3 && (tid & 7) != 7) continue;
#pragma unroll
            for (int c = 0; c < 4; c++) atomicAdd(&tile[c * 512 + node], (T)1);
        }
        r = r * 1664525 + 1013904223;
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (float)tile[5];
}

// non-atomic LDS read+write of the same volume (upper bound of what a conflict-free owner-computes scheme could reach)
__global__ __launch_bounds__(256) void k_lds_rw(int rounds, const int* __restrict__ rnd, float* out) {
    float* tile = (float*)s_raw;
    const int tid = threadIdx.x;
    for (int l = tid; l < 4 * 512; l += 256) tile[l] = 0.f;
    __syncthreads();
    float acc = 0.f;
    int r = rnd[(blockIdx.x * 256 + tid) & 0xffff];
    for (int it = 0; it < rounds; it++) {
#pragma unroll
        for (int n = 0; n < 27; n++) {
            int node = ((r >> 3) + n * 19 + it) % 216 + (n & 1);
#pragma unroll
            for (int c = 0; c < 4; c++) acc += tile[c * 512 + node];
        }
        r = r * 1664525 + 1013904223;
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = acc;
}

int main() {
    int* rnd; float* out;
    std::vector<int> h(1 << 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int)(i * 2654435761u) >> 1;
    CK(hipMalloc(&rnd, h.size() * 4)); CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 4 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = 1024, rounds = 4;
    const double ops = (double)wgs * 256 * rounds * 108;
#define RUN(name, launch) do { launch; CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-44s %8.1f us  %9.2f G lane-ops/s\n", name, ms * 1e3, ops / ms / 1e6); } while (0)
    const char* mnames[] = {"distinct words per wave", "random of 216 nodes (P2G-like)", "8 adjacent lanes share a node", "1 of 8 lanes active (aggregated); rate counts all 64 lanes"};
    for (int mode = 0; mode < 4; mode++) {
        printf("-- %s\n", mnames[mode]);
        RUN("ds_add_f32 (float)", hipLaunchKernelGGL(k_lds<float>, dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
        RUN("ds_add_u32 (int)", hipLaunchKernelGGL(k_lds<int>, dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
        RUN("ds_add_u64 (unsigned long long)", hipLaunchKernelGGL(k_lds<unsigned long long>, dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
        RUN("ds_add_f64 (double)", hipLaunchKernelGGL(k_lds<double>, dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
    }
    RUN("plain ds_read_b32 of the same pattern", hipLaunchKernelGGL(k_lds_rw, dim3(wgs), dim3(256), 0, 0, rounds, rnd, out));
    return 0;
}
