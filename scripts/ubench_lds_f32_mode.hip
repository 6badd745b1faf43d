// Does the FP32 denormal mode decide the rate of ds_add_f32?  (ubench_lds_types.hip measured 0.2 T lane-ops/s against 4-6 T for
// ds_add_u32 with ordinary operands.)  Variants: default mode, MODE.fp_denorm(f32) = flush set by s_setreg in the kernel, and the
// whole file built with -fgpu-flush-denormals-to-zero (run both binaries).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__shared__ float s_tile[4 * 512];

template <int SETREG, typename T>
__global__ __launch_bounds__(256) void k_lds(int mode, int rounds, const int* __restrict__ rnd, float* out) {
    T* tile = (T*)s_tile;
    const int tid = threadIdx.x;
    if (SETREG == 1) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);      // MODE[5:4] = 0: flush f32 denormals (in and out)
    if (SETREG == 2) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 3);      // MODE[5:4] = 3: keep them
    for (int l = tid; l < 4 * 512; l += 256) tile[l] = (T)0;
    __syncthreads();
    int r = rnd[(blockIdx.x * 256 + tid) & 0xffff];
    for (int it = 0; it < rounds; it++) {
#pragma unroll
        for (int n = 0; n < 27; n++) {
            int node = mode == 0 ? ((tid + n * 7) & 511) : (((r >> 3) + n * 19 + it) % 216 + (n & 1));
#pragma unroll
            for (int c = 0; c < 4; c++) atomicAdd(&tile[c * 512 + node], (T)1);
        }
        r = r * 1664525 + 1013904223;
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (float)tile[5];
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
    const char* mnames[] = {"distinct words per wave", "random of 216 nodes (P2G-like)"};
    for (int mode = 0; mode < 2; mode++) {
        printf("-- %s\n", mnames[mode]);
        RUN("ds_add_f32, mode as compiled", hipLaunchKernelGGL((k_lds<0, float>), dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
        RUN("ds_add_f32, s_setreg flush", hipLaunchKernelGGL((k_lds<1, float>), dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
        RUN("ds_add_f32, s_setreg keep denormals", hipLaunchKernelGGL((k_lds<2, float>), dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
        RUN("ds_add_u32", hipLaunchKernelGGL((k_lds<0, int>), dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out));
    }
    return 0;
}
