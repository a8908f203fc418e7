// HBM write bandwidth for the access pattern of the binned scatter's pass 1 (gfx950):
// a workgroup (512 threads) writes 64 chunks of `chunk` bytes, one per bucket, at [bucket][tile] x `stride` bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/write_pattern_bench.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(512) writer(uint4 *buf, unsigned n_tiles, unsigned stride16, unsigned lanes, int tile_major) {
    const unsigned tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (unsigned b = wave; b < 64; b += 8) {
        const size_t reg = tile_major ? (size_t)tile * 64 + b : (size_t)b * n_tiles + tile;
        if (lane < lanes) buf[reg * stride16 + lane] = make_uint4(tile, b, lane, 7);
    }
}

int main() {
    const unsigned n_tiles = 24576 * 8;                      // 8 "levels" worth of tiles
    const size_t bytes = (size_t)n_tiles * 64 * 1024;          // the largest stride below
    uint4 *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; unsigned stride, lanes; int tm; } cases[] = {
        {"512 of 832 B, bucket-major", 832, 32, 0}, {"832 of 832 B, bucket-major", 832, 52, 0},
        {"512 of 832 B, tile-major", 832, 32, 1},   {"832 of 832 B, tile-major", 832, 52, 1},
        {"512 of 512 B, bucket-major", 512, 32, 0}, {"512 of 512 B, tile-major (contiguous)", 512, 32, 1},
        {"1024 of 1024 B, bucket-major", 1024, 64, 0}, {"768 of 832 B, bucket-major", 832, 48, 0},
        {"512 of 1024 B, bucket-major", 1024, 32, 0},
        {"512 of 768 B, bucket-major", 768, 32, 0}, {"640 of 768 B, bucket-major", 768, 40, 0},
        {"576 of 768 B, bucket-major (64-B tail)", 768, 36, 0}, {"512 of 768 B, tile-major", 768, 32, 1},
    };
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(writer, dim3(n_tiles), dim3(512), 0, 0, buf, n_tiles, c.stride / 16, c.lanes, c.tm);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) printf("%-40s %7.3f ms  %6.2f TB/s written\n", c.name, ms, (double)n_tiles * 64 * c.lanes * 16 / ms / 1e9);
        }
    }
    return 0;
}
