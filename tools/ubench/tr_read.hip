// tools/ubench/tr_read.hip -- what ds_read_b64_tr_b16 returns (gfx950): every lane reads 8 bytes at its own LDS address
// and receives four 16-bit elements; prints, for a chosen address pattern, which LDS element each (lane, element) got.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read.hip -o tools/ubench/tr_read && tools/ubench/tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode, int S) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int byteoff;
    if (mode == 0) byteoff = l * 8;                                            // contiguous 8-byte chunks
    else byteoff = (l >> 4) * 4 * S + ((l & 15) >> 2) * S + (l & 3) * 8;        // group g: 4 rows at pitch S, lane i -> row i/4, 8-byte piece i%4
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)((__attribute__((address_space(3))) char*)lds + byteoff));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, 512);
    for (int mode = 0; mode < 2; ++mode) {
        const int S = 160;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode, S);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d%s\n", mode, mode ? " (rows at pitch 160 B = 80 elements)" : " (contiguous)");
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]);
            printf("%s", (l & 3) == 3 ? "\n" : "   |");
        }
    }
    return 0;
}
