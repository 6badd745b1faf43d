// Microbenchmark: fp32 atomic-add throughput on MI355X, LDS (ds_add_f32) and global (global_atomic_add_f32),
// in the access patterns the P2G scatter / tile flush produce.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__shared__ float s_tile[4 * 512];

// mode 0: every lane a distinct word (conflict free)   1: random node of 216 (P2G-like, plane per instruction)
// mode 2: all lanes one word                            3: random node, but 8 distinct nodes per wave (cell-sorted particles)
__global__ __launch_bounds__(256) void k_lds(int mode, int rounds, const int* __restrict__ rnd, float* out) {
    const int tid = threadIdx.x;
    for (int l = tid; l < 4 * 512; l += 256) s_tile[l] = 0.f;
    __syncthreads();
    int r = rnd[(blockIdx.x * 256 + tid) & 0xffff];
    for (int it = 0; it < rounds; it++) {
#pragma unroll
        for (int n = 0; n < 27; n++) {
            int node;
            if (mode == 0) node = (tid + n * 7) & 511;
            else if (mode == 1) node = ((r >> 3) + n * 19 + it) % 216 + (n & 1);
            else if (mode == 2) node = 5;
            else node = (((r >> 3) & 7) * 27 + n + it) & 511;
#pragma unroll
            for (int c = 0; c < 4; c++) atomicAdd(&s_tile[c * 512 + node], 1.0f);
        }
        r = r * 1664525 + 1013904223;
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = s_tile[5];
}

// global: mode 0 random word of `span`; 1 coalesced contiguous (lane -> consecutive floats);
// 2 flush-like: lane -> node (16 B stride), 4 instructions for the 4 components, tiles overlap between neighbouring WGs
// 3 like 2 but plain stores (no atomics) for reference
__global__ __launch_bounds__(256) void k_glb(int mode, int rounds, int span, const int* __restrict__ rnd, float* buf) {
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    int r = rnd[gid & 0xffff];
    for (int it = 0; it < rounds; it++) {
        if (mode == 0) {
#pragma unroll
            for (int n = 0; n < 8; n++) { unsafeAtomicAdd(&buf[(unsigned)(r + n * 7919) % (unsigned)span], 1.0f); }
        } else if (mode == 1) {
#pragma unroll
            for (int n = 0; n < 8; n++) unsafeAtomicAdd(&buf[((size_t)(blockIdx.x * 8 + n) * 256 + tid + it * 64) % (size_t)span], 1.0f);
        } else {
            // WG b owns a 512-node tile starting at node b*64 (so 8 WGs overlap on every node, like halo overlap)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                size_t node = ((size_t)blockIdx.x * 64 + h * 256 + tid + it * 8) % ((size_t)span / 4);
                float* dst = buf + node * 4;
                if (mode == 2) { unsafeAtomicAdd(dst, 1.f); unsafeAtomicAdd(dst + 1, 1.f); unsafeAtomicAdd(dst + 2, 1.f); unsafeAtomicAdd(dst + 3, 1.f); }
                else { *(float4*)dst = make_float4(1.f, 1.f, 1.f, 1.f); }
            }
        }
        r = r * 1664525 + 1013904223;
    }
}

int main() {
    int* rnd; float *out, *buf;
    const int span = 1 << 22;                  // 16 MiB of floats
    std::vector<int> h(1 << 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int)(i * 2654435761u) >> 1;
    CK(hipMalloc(&rnd, h.size() * 4)); CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&buf, (size_t)span * 4)); CK(hipMemset(buf, 0, (size_t)span * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* lds_names[] = {"distinct words", "random of 216 nodes (P2G-like)", "one word", "8 distinct nodes per wave"};
    for (int wgs : {512, 2048}) for (int mode = 0; mode < 4; mode++) {
        const int rounds = 4;
        hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), 0, 0, mode, rounds, rnd, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double ops = (double)wgs * 256 * rounds * 108;
        printf("LDS  ds_add_f32  %4d WGs  %-34s %8.1f us  %8.2f G lane-ops/s\n", wgs, lds_names[mode], ms * 1e3, ops / ms / 1e6);
    }
    const char* g_names[] = {"random word in 16 MiB", "coalesced contiguous", "tile flush (4 comps, overlap)", "tile flush as plain float4 stores"};
    for (int wgs : {600, 2400}) for (int mode = 0; mode < 4; mode++) {
        const int rounds = 4;
        hipLaunchKernelGGL(k_glb, dim3(wgs), dim3(256), 0, 0, mode, rounds, span, rnd, buf);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_glb, dim3(wgs), dim3(256), 0, 0, mode, rounds, span, rnd, buf);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double ops = (double)wgs * 256 * rounds * 8;
        printf("GLOBAL atomic f32 %4d WGs  %-34s %8.1f us  %8.2f G lane-ops/s\n", wgs, g_names[mode], ms * 1e3, ops / ms / 1e6);
    }
    return 0;
}
