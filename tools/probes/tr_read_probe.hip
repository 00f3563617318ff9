// What does ds_read_b64_tr_b16 deliver?  (gfx950; `hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe && ./tr_read_probe`)
// LDS holds lds[i] = i (16-bit); every lane reads 8 bytes "transposed" at an address pattern; the 4 values a lane receives are printed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;

__global__ void k(short* out, int pattern, int rowstride) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int a;
    if (pattern == 0) a = lane * 4;                                                       // 8 contiguous bytes per lane
    else { const int i = lane & 15, g = lane >> 4; a = (i >> 2) * rowstride + (i & 3) * 4 + g * 16; }   // [4 rows][16 cols] per 16 lanes
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + a));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    short h[256];
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int rs : {64, 32}) {
            if (pattern == 0 && rs != 64) continue;
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pattern, rs);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("pattern %d rowstride %d (values = LDS element index; row = idx / rowstride, col = idx %% rowstride)\n", pattern, rs);
            for (int l = 0; l < 64; ++l) {
                printf("  lane %2d:", l);
                for (int j = 0; j < 4; ++j) printf(" %5d (r%d c%2d)", h[l * 4 + j], h[l * 4 + j] / rs, h[l * 4 + j] % rs);
                printf("\n");
            }
        }
    return 0;
}
