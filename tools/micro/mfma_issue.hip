// Does ONE wave per SIMD keep the fp32 matrix pipe full?  Independent v_mfma_f32_32x32x2_f32 streams,
// W waves per SIMD (W = 1: 16 accumulator tiles per wave, W = 2: 8 tiles per wave), no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o mfma_issue && ./mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TILES, int NT>
__global__ void __launch_bounds__(NT) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / TILES; ++rep)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int TILES, int NT>
void run(const char* name, float* out, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TILES, NT>), dim3(blocks), dim3(NT), 0, 0, out, iters, 0.f, 0.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (NT / 64) * iters * 16.0 * 4096.0;
        printf("%-34s blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    }
}

int main() {
    float* out; hipMalloc(&out, 4096);
    run<16, 256>("1 wave/SIMD, 16 tiles", out, 256);
    run<8, 512>("2 waves/SIMD, 8 tiles (1 block/CU)", out, 256);
    run<8, 256>("2 waves/SIMD, 8 tiles (2 blocks/CU)", out, 512);
    run<4, 1024>("4 waves/SIMD, 4 tiles", out, 256);
    return 0;
}
