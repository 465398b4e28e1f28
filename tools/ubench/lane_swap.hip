// tools/ubench/lane_swap.hip -- lane semantics of v_permlane32_swap_b32 on gfx950 (the register transposition of the conv epilogue,
// step_amd/csrc/common.h: lane32_swap).  hipcc --offload-arch=gfx950 -O2 lane_swap.hip -o lane_swap && ./lane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    p[threadIdx.x] = r[0];
    p[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("x':"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]);
    printf("\ny':"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]);
    printf("\n");
    // expected: x' = 0..31, 100..131 ; y' = 32..63, 132..163
    bool ok = true;
    for (int i = 0; i < 32; ++i) ok = ok && h[i] == (unsigned)i && h[32 + i] == 100u + i && h[64 + i] == 32u + i && h[96 + i] == 132u + i;
    printf("%s\n", ok ? "OK: x.upper <-> y.lower" : "DIFFERENT SEMANTICS");
    return ok ? 0 : 1;
}
