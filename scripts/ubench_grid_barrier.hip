// What does a grid-wide barrier cost on this part, against a kernel boundary?  (Next round's question: the six substep kernels as one or
// two persistent launches.)  G workgroups of 256 threads, all resident (hipLaunchCooperativeKernel refuses otherwise), R rounds of:
// every thread writes `bytes_per_wg / 256` bytes of a buffer another workgroup (on another XCD: id + 1) reads in the next round,
// then a barrier = release fence, one atomic add on a counter, spin (agent-scope acquire loads, s_sleep) until all have arrived.
// Compared with R launches of the same body without the barrier.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_grid_barrier scripts/ubench_grid_barrier.hip && ./ubench_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned goal) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < goal) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}
__device__ __forceinline__ float body(float* buf, int words_per_wg, int round, int G) {
    // read what workgroup id + 1 wrote last round, write this round's
    const int src = ((blockIdx.x + 1) % G) * words_per_wg, dst = blockIdx.x * words_per_wg;
    float acc = 0.f;
    for (int i = threadIdx.x; i < words_per_wg; i += 256) acc += buf[src + i];
    for (int i = threadIdx.x; i < words_per_wg; i += 256) buf[dst + i] = acc * 1e-6f + (float)round;
    return acc;
}
__global__ __launch_bounds__(256) void k_persistent(float* buf, int words_per_wg, int R, unsigned* counter, float* out) {
    const int G = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < R; r++) {
        acc += body(buf, words_per_wg, r, G);
        grid_barrier(counter, (unsigned)(r + 1) * G);
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one(float* buf, int words_per_wg, int r, float* out) {
    const float acc = body(buf, words_per_wg, r, gridDim.x);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main() {
    const int R = 200;
    float* buf; float* out; unsigned* counter;
    CK(hipMalloc(&buf, 1024 * 65536 * sizeof(float))); CK(hipMalloc(&out, 4096 * sizeof(float))); CK(hipMalloc(&counter, 4));
    CK(hipMemset(buf, 0, 1024 * 65536 * sizeof(float)));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {256, 1024, 2048}) for (int kb : {0, 4, 32}) {
        int words = kb * 256; if (words == 0) words = 256;                 // (at least one word per thread)
        float ms_p = -1.f, ms_l = -1.f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemsetAsync(counter, 0, 4, st));
            int R_ = R; void* args[] = {&buf, &words, &R_, &counter, &out};
            CK(hipEventRecord(e0, st));
            hipError_t e = hipLaunchCooperativeKernel((void*)k_persistent, dim3(G), dim3(256), args, 0, st);
            if (e != hipSuccess) { printf("G=%d: cooperative launch refused (%s)\n", G, hipGetErrorString(e)); (void)hipGetLastError(); break; }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms_p, e0, e1));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < R; r++) hipLaunchKernelGGL(k_one, dim3(G), dim3(256), 0, st, buf, words, r, out);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms_l, e0, e1));
        }
        printf("G=%5d workgroups, %3d KB written + read per workgroup and round: persistent + grid barrier %7.2f us per round, one launch per round %7.2f us\n",
               G, kb ? kb : 1, 1e3f * ms_p / R, 1e3f * ms_l / R);
    }
    return 0;
}
