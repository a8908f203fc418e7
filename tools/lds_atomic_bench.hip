// Micro-benchmark: LDS atomic / store rates on gfx950 for the access patterns of scatter_binned.h.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/lds_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, uint32_t slots_mask) {
    __shared__ float acc[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) acc[i] = 0.f;
    __syncthreads();
    uint32_t h = mix(threadIdx.x + blockIdx.x * 977u + 1u);
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            h = h * 1664525u + 1013904223u;
            uint32_t a = (h >> 8) & slots_mask;
            if (MODE == 3) a = ((threadIdx.x & 63u) + 64u * (uint32_t)u) & slots_mask;      // conflict-free
            if (MODE == 0) atomicAdd(&acc[a], 1.0f);                                          // ds_add_f32
            else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(&acc[a]), 1u);          // ds_add_u32
            else if (MODE == 2) sink += atomicAdd(reinterpret_cast<uint32_t*>(&acc[a]), 1u);  // ds_add_rtn_u32
            else if (MODE == 3) atomicAdd(&acc[a], 1.0f);
            else if (MODE == 4) acc[a] = (float)h;                                            // ds_write_b32
            else if (MODE == 5) { float v = acc[a]; acc[a] = v + 1.0f; }                      // racy read-modify-write
            else if (MODE == 6) __hip_atomic_fetch_add(&acc[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 7) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[(a & ~1u)]), 1ull);   // ds_add_u64
            else if (MODE == 8) atomicAdd(reinterpret_cast<double*>(&acc[(a & ~1u)]), 1.0);                 // ds_add_f64
            else if (MODE == 9) atomicMax(reinterpret_cast<uint32_t*>(&acc[a]), h);                          // ds_max_u32
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + (float)sink;
}

template <int MODE> void run(const char* name, uint32_t slots) {
    float* out; hipMalloc(&out, 4096 * 4);
    const int blocks = 256 * 4, iters = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 8, slots - 1);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters, slots - 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 16;
    printf("%-28s slots %6u : %8.3f ms  %8.2f Gops/s  = %.2f lane-ops/clk/CU (2.4 GHz, 256 CU)\n", name, slots, ms, ops / ms * 1e-6,
           ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(out);
}

int main() {
    for (uint32_t slots : {4096u}) {
        run<0>("ds_add_f32 random", slots);
        run<1>("ds_add_u32 random", slots);
        run<2>("ds_add_rtn_u32 random", slots);
        run<6>("ds_add_f32 wg-scope builtin", slots);
        run<4>("ds_write_b32 random", slots);
        run<5>("ds read+write random", slots);
        run<7>("ds_add_u64 random", slots);
        run<8>("ds_add_f64 random", slots);
        run<9>("ds_max_u32 random", slots);
    }
    run<3>("ds_add_f32 conflict-free", 16384);
    return 0;
}
