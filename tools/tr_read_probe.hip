// Probe of ds_read_b64_tr_b16 (gfx950): fills LDS with element indices of a [rows][32 x 16-bit] image and prints what
// each lane receives when lane 4q+p of a 16-lane group supplies the address of row 4*group+q, columns 4p..4p+3.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/tr_read_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __bf16 v4bf16 __attribute__((ext_vector_type(4)));
__global__ void k(short *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 2048; i += 64) reinterpret_cast<short *>(smem)[i] = (short)i;
    __syncthreads();
    const unsigned lane = threadIdx.x;
    const unsigned i = lane & 15, q = i >> 2, p = i & 3, grp = lane >> 4;
    // group grp reads block rows 4*grp.., columns 0..15 of a [rows][32 shorts] image
    unsigned char *addr = smem + (4 * grp + q) * 64 + 8 * p;
    v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16 *)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short *d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}
