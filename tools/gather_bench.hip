// Micro-benchmark: rate of independent random 4/8-byte gathers from a table of a given size (gfx950).
// Tells how close the hash-encoder forward is to what the L1/L2/Infinity-Cache path can deliver.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <typename T, int G>
__global__ void __launch_bounds__(256) k(const T* __restrict__ table, uint32_t mask, float* out, int iters) {
    uint32_t h = mix(threadIdx.x + blockIdx.x * 256u + 12345u);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        T v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { h = h * 1664525u + 1013904223u; v[g] = table[(h >> 4) & mask]; }
#pragma unroll
        for (int g = 0; g < G; ++g) acc += (float)v[g].x;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename T> void run(const char* name, size_t bytes) {
    T* table; hipMalloc(&table, bytes); hipMemset(table, 0, bytes);
    float* out; hipMalloc(&out, 4);
    const uint32_t mask = (uint32_t)(bytes / sizeof(T)) - 1;
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, 8><<<blocks, 256>>>(table, mask, out, 4);
    hipEventRecord(a);
    k<T, 8><<<blocks, 256>>>(table, mask, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 8;
    printf("%-10s table %7.1f MB : %7.3f ms  %7.1f Ggather/s = %.2f gathers/clk/CU (2.4 GHz)  useful %.2f TB/s\n", name, bytes / 1048576.0, ms,
           ops / ms * 1e-6, ops / (ms * 1e-3) / 256 / 2.4e9, ops * sizeof(T) / (ms * 1e-3) / 1e12);
    hipFree(table); hipFree(out);
}

int main() {
    for (size_t mb : {1, 2, 4, 8, 32, 128, 512}) {
        run<uint1>("4B", mb << 20);
        run<uint2>("8B", mb << 20);
        run<uint4>("16B", mb << 20);
    }
    return 0;
}
