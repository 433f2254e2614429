// How fast can one block per CU drain a 256 x 256 fp32 accumulator tile to HBM?  Variants of the
// conv epilogue's store pattern on the l3c3 forward shape (1024 rows x 131072 pixels = 537 MB).
//   hipcc --offload-arch=gfx950 -O3 store_bench.hip -o store_bench && ./store_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int P = 131072, M = 1024;

// V=0: dword stores, lane = column (2 rows x 128 B per instruction)       [igemm3]
// V=1: dwordx4 stores, lane = 4 columns (2 rows x 512 B per instruction)  [igemm2]
// V=2: dwordx4 stores, one row x 1 KB per instruction
template <int V>
__global__ void __launch_bounds__(256) k(float* out, int lds_dummy) {
    extern __shared__ float sm[];
    if (lds_dummy == 12345) sm[threadIdx.x] = 1.f;
    const int tiles_m = M / 256;
    const int group = 8 * tiles_m;
    const int g = blockIdx.x / group, local = blockIdx.x - g * group;
    const int nt = g * 8 + (local & 7), mt = local >> 3;
    const int p0 = nt * 256, m0 = mt * 256;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lhi = lane >> 5;
    const float val = (float)threadIdx.x;
    if (V == 0) {
        float* o = out + (long long)(m0 + wm * 128 + 4 * lhi) * P + p0 + wn * 128 + l31;
#pragma unroll 4
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[(long long)row * P + 32 * j] = val;
            }
    } else if (V == 1) {
        float* o = out + (long long)(m0 + wm * 128 + 4 * lhi) * P + p0 + wn * 128 + 4 * l31;
#pragma unroll 4
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                *reinterpret_cast<f32x4*>(o + (long long)row * P) = f32x4{val, val, val, val};
            }
    } else {
        // wave w writes rows w*64 .. w*64+63, one full 1 KB row per instruction
        float* o = out + (long long)(m0 + wid * 64) * P + p0 + 4 * lane;
#pragma unroll 8
        for (int r = 0; r < 64; ++r)
            *reinterpret_cast<f32x4*>(o + (long long)r * P) = f32x4{val, val, val, val};
    }
}

template <int V>
void run(float* d, size_t lds, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = (M / 256) * (P / 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, d, 0);
    hipEventRecord(a);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, d, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-34s lds %6zu B: %.3f ms  %.2f TB/s\n", name, lds, ms, (double)M * P * 4 / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, (size_t)M * P * 4);
    for (size_t lds : {(size_t)129024, (size_t)65536, (size_t)16384}) {
        run<0>(d, lds, "dword, 2 rows x 128 B");
        run<1>(d, lds, "dwordx4, 2 rows x 512 B");
        run<2>(d, lds, "dwordx4, 1 row x 1 KB");
    }
    hipMemset(d, 0, (size_t)M * P * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); for (int i = 0; i < 10; ++i) hipMemsetAsync(d, 0, (size_t)M * P * 4, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("hipMemset: %.3f ms  %.2f TB/s\n", ms, (double)M * P * 4 / ms / 1e9);
    return 0;
}
