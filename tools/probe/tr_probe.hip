// Semantics probe for ds_read_b64_tr_b16 (gfx950): LDS holds u16 element indices; every lane passes its own byte address.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out, int row_stride_elems) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // 16-lane group g: block of 4 rows x 16 columns; lane i of the group points at row i/4, columns 4(i%4)..+3
    const int g = l >> 4, i = l & 15;
    const int row = 4 * (g >> 1) + (i >> 2), col = 16 * (g & 1) + 4 * (i & 3);
    const unsigned addr = (unsigned)(size_t)lds + (row * row_stride_elems + col) * 2;   // LDS byte address
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
    out[l * 4 + 0] = r[0] & 0xffff; out[l * 4 + 1] = r[0] >> 16; out[l * 4 + 2] = r[1] & 0xffff; out[l * 4 + 3] = r[1] >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {64, 160}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("row stride %d elements: lane -> 4 elements as (row,col)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%d,%2d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
