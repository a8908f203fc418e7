// Calibration of rocprofv3's FETCH_SIZE for scattered 4-byte gathers (gfx950): every gather of the measured kernel goes to a
// DIFFERENT 64-byte sector of a table far larger than the caches, each sector exactly once (bijective index), after the caches
// were flushed by streaming over another buffer.  Known quantities: gathers = sectors = bytes / 64.  If a miss moves one 64-byte
// sector, HBM traffic is `bytes`; if it moves a whole 128-byte line (whose sibling sector is requested much later, by another
// wave, long after the line left the 4 MB L2), it is 2 x `bytes`.  Run under `rocprofv3 --pmc FETCH_SIZE` (and, in a second
// pass, the TCC request counters) and compare the counter with `bytes`; the printed time gives the implied bandwidth for
// either reading.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) distinct_sector_gather(const uint32_t *__restrict__ table, uint32_t sector_mask, float *out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t v[8];
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        const uint32_t sector = ((gid * 8u + g) * 0x9E3779B1u) & sector_mask;      // odd multiplier: a bijection of [0, 2^k)
        v[g] = table[(size_t)sector * 16u + (sector & 15u)];                       // some dword inside the sector
    }
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) acc ^= v[g];
    if (acc == 0x12345678u) out[0] = 1.0f;
}

// Same number of gathers, but lanes 2i and 2i+1 take the TWO sectors of one 128-byte line (lines in bijective random order): if
// a miss moved a whole line, this variant would touch half the lines of the first and run about twice as fast.
__global__ void __launch_bounds__(256) both_sectors_gather(const uint32_t *__restrict__ table, uint32_t line_mask, float *out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t v[8];
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        const uint32_t line = (((gid >> 1) * 8u + g) * 0x9E3779B1u) & line_mask;
        v[g] = table[(size_t)line * 32u + (gid & 1u) * 16u + (line & 15u)];
    }
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) acc ^= v[g];
    if (acc == 0x12345678u) out[0] = 1.0f;
}

__global__ void __launch_bounds__(256) stream_fill(uint4 *buf, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) buf[i] = make_uint4(1, 2, 3, 4);
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 2048;                       // table size in MiB (a power of two)
    const size_t bytes = mb << 20, sectors = bytes / 64;
    uint32_t *table;
    uint4 *flush;
    float *out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&flush, (size_t)1 << 30) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(table, 0, bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        stream_fill<<<2048, 256>>>(flush, ((size_t)1 << 30) / 16);                 // 1 GiB of stores: nothing of the table stays cached
        (void)hipEventRecord(a);
        distinct_sector_gather<<<(uint32_t)(sectors / 8 / 256), 256>>>(table, (uint32_t)(sectors - 1), out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("table %zu MiB: %zu gathers, one per 64-B sector: %.3f ms = %.1f G gathers/s; %.2f TB/s if a miss moves 64 B, %.2f TB/s if 128 B\n",
               mb, sectors, ms, sectors / ms * 1e-6, bytes / (ms * 1e-3) / 1e12, 2.0 * bytes / (ms * 1e-3) / 1e12);
    }
    for (int rep = 0; rep < 3; ++rep) {
        stream_fill<<<2048, 256>>>(flush, ((size_t)1 << 30) / 16);
        (void)hipEventRecord(a);
        both_sectors_gather<<<(uint32_t)(sectors / 8 / 256), 256>>>(table, (uint32_t)(sectors / 2 - 1), out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("table %zu MiB: %zu gathers, both sectors of each 128-B line from adjacent lanes: %.3f ms = %.1f G gathers/s (%.2f TB/s of lines)\n",
               mb, sectors, ms, sectors / ms * 1e-6, bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
