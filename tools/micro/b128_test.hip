// Checks raw_buffer_load_b128 semantics on gfx950: 4-byte-aligned (unaligned to 16) addresses,
// and all-zero return for an out-of-range offset.  Build: hipcc --offload-arch=gfx950 -O3 -o b128_test b128_test.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int v4u __attribute__((vector_size(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, float* c, int n, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, n * 4, 0x00020000);
    unsigned off = (threadIdx.x * 4 + shift) * 4;
    if (threadIdx.x & 1) off = 0x80000000u;
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    *(f32x4*)(c + threadIdx.x * 4) = __builtin_bit_cast(f32x4, v);
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; i++) h[i] = (float)i;
    float *a, *c;
    hipMalloc(&a, n * 4); hipMalloc(&c, 256 * 4 * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift = 0; shift < 4; shift++) {
        k<<<1, 256>>>(a, c, n, shift);
        std::vector<float> o(1024);
        hipMemcpy(o.data(), c, 1024 * 4, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; t++)
            for (int e = 0; e < 4; e++) {
                float want = (t & 1) ? 0.f : (float)(t * 4 + shift + e);
                if (o[t * 4 + e] != want) { if (bad < 5) printf("shift %d t %d e %d got %f want %f\n", shift, t, e, o[t*4+e], want); bad++; }
            }
    }
    printf("b128_test: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
