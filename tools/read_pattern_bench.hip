// HBM read bandwidth for the access pattern of the binned scatter's pass 2 (gfx950): workgroup (bucket b, level l) reads
// the first `bytes` of region [l][tile][b] for all tiles (region stride 768 B, tile-major), 8 regions in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/read_pattern_bench.hip -o /tmp/rp && /tmp/rp
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(1024) reader(const uint2 *buf, unsigned n_tiles, unsigned stride8, unsigned lanes,
                                               int tile_major, unsigned long long *sink) {
    const unsigned bucket = blockIdx.x, level = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned per_wave = (n_tiles + 15) / 16, t0 = wave * per_wave, t1 = min(n_tiles, t0 + per_wave);
    unsigned long long acc = 0;
    for (unsigned t = t0; t < t1; t += 8) {
        uint2 v[8];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) {
            const unsigned tt = min(t + u, t1 - 1);
            const size_t reg = tile_major ? ((size_t)level * n_tiles + tt) * 64 + bucket : ((size_t)level * 64 + bucket) * n_tiles + tt;
            v[u] = buf[reg * stride8 + (lane < lanes ? lane : 0)];
        }
#pragma unroll
        for (unsigned u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y;
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}

int main() {
    const unsigned n_tiles = 24576, levels = 16;
    const size_t bytes = (size_t)n_tiles * levels * 64 * 768;
    uint2 *buf; unsigned long long *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; unsigned lanes; int tm; } cases[] = {
        {"512 of 768 B, tile-major (as pass 2 reads)", 64, 1}, {"512 of 768 B, bucket-major", 64, 0},
        {"384 of 768 B, tile-major", 48, 1}, {"256 of 768 B, tile-major", 32, 1}};
    for (auto &c : cases)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(reader, dim3(64, levels), dim3(1024), 0, 0, buf, n_tiles, 768 / 8, c.lanes, c.tm, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-46s %7.3f ms  %6.2f TB/s read\n", c.name, ms, (double)n_tiles * levels * 64 * c.lanes * 8 / ms / 1e9);
        }
    return 0;
}
