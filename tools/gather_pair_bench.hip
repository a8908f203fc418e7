// Micro-benchmark for the hash encoder's x-neighbour corner pairs (gfx950): the two corners (x, x+1) of a cell sit in rows r and
// r ^ m with m = x ^ (x + 1) (hashed levels: the prime of dimension 0 is 1, hashencoder.cu:36-52), i.e. 1, 3, 7, ... with
// probability 1/2, 1/4, 1/8, ...  Counter evidence (profiles/round3_encode_cache_counters.md) says the encoder is bound by the
// number of L1 (TCP) accesses, one per lane and instruction, not by bytes.  Variants, per pair of 4-byte rows:
//   two      : two dword gathers (what encode_kernel did through round 2)
//   window   : ONE 4-byte-aligned dwordx4 gather at min(r, r^m) brings both rows when they are < 4 rows apart (83 %), a second,
//              exec-masked dword gather fetches the far row otherwise
//   single   : one dword gather per pair (lower bound: the first corner only)
// Prints pairs/s for a table that fits one XCD's L2 (2 MB) and one that does not (32 MB), and checks that the window variant
// returns the same words as the two-gather variant (an unaligned dwordx4 must behave like four dword loads).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint32_t *__restrict__ table, uint32_t mask, uint32_t *out, int iters) {
    uint32_t h = mix(threadIdx.x + blockIdx.x * 256u + 12345u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t ra[8], rb[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            h = h * 1664525u + 1013904223u;
            const uint32_t x = h >> 7, hz = mix(h);
            ra[g] = (x ^ hz) & mask;
            rb[g] = ((x + 1u) ^ hz) & mask;
        }
        if (MODE == 0) {
            uint32_t va[8], vb[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) { va[g] = table[ra[g]]; vb[g] = table[rb[g]]; }
#pragma unroll
            for (int g = 0; g < 8; ++g) acc += va[g] * 3u + vb[g];
        } else if (MODE == 1) {
            u32x4_a4 w[8];
            uint32_t far[8], d[8];
            bool swap[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const uint32_t lo = min(ra[g], rb[g]), hi = max(ra[g], rb[g]);
                const uint32_t base = min(lo, mask - 3u);
                swap[g] = ra[g] > rb[g];
                d[g] = hi - base;
                w[g] = *reinterpret_cast<const u32x4_a4 *>(table + base);
                far[g] = 0;
                if (d[g] >= 4u || base != lo) far[g] = table[hi];
                if (base != lo) w[g].x = table[lo];
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const uint32_t vlo = w[g].x;
                const uint32_t vhi = d[g] == 1u ? w[g].y : d[g] == 2u ? w[g].z : d[g] == 3u ? w[g].w : far[g];
                acc += (swap[g] ? vhi : vlo) * 3u + (swap[g] ? vlo : vhi);
            }
        } else {
            uint32_t va[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) va[g] = table[ra[g]];
#pragma unroll
            for (int g = 0; g < 8; ++g) acc += va[g] * 3u;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE> float run(const uint32_t *table, uint32_t mask, uint32_t *out, int blocks, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(table, mask, out, 2);
    (void)hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(table, mask, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 64;
    uint32_t *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    for (size_t mb : {2, 32}) {
        const size_t rows = (mb << 20) / 4;
        uint32_t *table;
        (void)hipMalloc(&table, rows * 4);
        std::vector<uint32_t> host(rows);
        for (size_t i = 0; i < rows; ++i) host[i] = (uint32_t)(i * 2654435761u);
        (void)hipMemcpy(table, host.data(), rows * 4, hipMemcpyHostToDevice);
        std::vector<uint32_t> r0((size_t)blocks * 256), r1((size_t)blocks * 256);
        const float t0 = run<0>(table, (uint32_t)rows - 1, out, blocks, iters);
        (void)hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost);
        const float t1 = run<1>(table, (uint32_t)rows - 1, out, blocks, iters);
        (void)hipMemcpy(r1.data(), out, r1.size() * 4, hipMemcpyDeviceToHost);
        const float t2 = run<2>(table, (uint32_t)rows - 1, out, blocks, iters);
        size_t bad = 0;
        for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
        const double pairs = (double)blocks * 256 * iters * 8;
        printf("table %3zu MB: two dword gathers %.3f ms (%.1f G pairs/s) | dwordx4 window %.3f ms (%.1f G pairs/s) | one dword %.3f ms (%.1f G/s) | window == two: %s\n",
               mb, t0, pairs / t0 * 1e-6, t1, pairs / t1 * 1e-6, t2, pairs / t2 * 1e-6, bad ? "MISMATCH" : "yes");
        (void)hipFree(table);
    }
    return 0;
}
