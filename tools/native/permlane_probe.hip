// What v_permlane32_swap_b32 does on this machine: prints, per lane, the two results for the
// inputs (lane, 100 + lane).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    out[2 * l] = r[0]; out[2 * l + 1] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int l = 0; l < 64; l += 8) printf("lane %2d: r0 %3u r1 %3u | lane %2d: r0 %3u r1 %3u\n", l, h[2 * l], h[2 * l + 1], l + 1, h[2 * l + 2], h[2 * l + 3]);
    return 0;
}
