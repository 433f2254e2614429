// Semantics check for ds_read_b64_tr_b16 as conv_igemm3.hip uses it: within each 16-lane group,
// lane 4q+p supplies the address of row q / columns 4p..4p+3 of a 4 x 16 block and lane i gets
// column i of the four rows (row q in element q).
//   hipcc --offload-arch=gfx950 -O2 tr16_test.hip -o tr16_test && ./tr16_test
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short s[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) s[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int q = (lane & 15) >> 2, p = lane & 3, g = lane >> 4;
    unsigned short* addr = s + (4 * (g >> 1) + q) * 64 + 16 * (g & 1) + 4 * p;   // image [16 k][64 px]
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short h[16 * 64], o[256], *din, *dout;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = (unsigned short)(r * 100 + c);
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        for (int e = 0; e < 4; ++e) {
            const int want = (4 * (g >> 1) + e) * 100 + 16 * (g & 1) + i;   // row e of the block, column i
            if (o[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", lane, e, o[lane*4+e], want); ++bad; }
        }
    }
    printf(bad ? "tr16: MISMATCH (%d)\n" : "tr16: semantics as documented\n", bad);
    return bad != 0;
}
