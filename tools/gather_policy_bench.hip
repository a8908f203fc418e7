// Micro-benchmark: do the cache-policy bits of a gfx950 load (sc0 / sc1 / nt) change the rate of independent random
// 4-byte gathers?  (The encoder's fine levels are bound by L2 -> L1 line fills; a policy that moved less than a 128-byte
// line per miss would lift that ceiling.)  Raw buffer loads, so that the compiler tracks the loads (waitcnt) itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

// aux bits on gfx940+: 1 = sc0, 2 = nt, 16 = sc1
template <int AUX, int G>
__global__ void __launch_bounds__(256) k(const uint32_t* __restrict__ table, uint32_t bytes, float* out, int iters) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)bytes, 0x00027000);
    const uint32_t mask = bytes / 4u - 1u;
    uint32_t h = mix(threadIdx.x + blockIdx.x * 256u + 12345u);
    uint32_t acc = 0u;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { h = h * 1664525u + 1013904223u; v[g] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(((h >> 4) & mask) * 4u), 0, AUX); }
#pragma unroll
        for (int g = 0; g < G; ++g) acc += v[g];
    }
    if (acc == 0x12345678u) out[0] = (float)acc;
}

template <int AUX> void run(const char* name, size_t bytes) {
    uint32_t* table; hipMalloc(&table, bytes); hipMemset(table, 0, bytes);
    float* out; hipMalloc(&out, 4);
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<AUX, 8><<<blocks, 256>>>(table, (uint32_t)bytes, out, 4);
    hipEventRecord(a);
    k<AUX, 8><<<blocks, 256>>>(table, (uint32_t)bytes, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 8;
    printf("%-10s table %7.1f MB : %7.3f ms  %7.1f Ggather/s\n", name, bytes / 1048576.0, ms, ops / ms * 1e-6);
    hipFree(table); hipFree(out);
}

int main() {
    for (size_t mb : {2, 16, 512}) {
        run<0>("plain", mb << 20);
        run<1>("sc0", mb << 20);
        run<16>("sc1", mb << 20);
        run<17>("sc0 sc1", mb << 20);
        run<2>("nt", mb << 20);
        run<3>("sc0 nt", mb << 20);
        run<18>("sc1 nt", mb << 20);
        run<19>("sc0 sc1 nt", mb << 20);
    }
    return 0;
}
