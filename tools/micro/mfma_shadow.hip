// What does a producer cost in the shadow of the fp32 matrix pipe?  One wave per SIMD, 16 accumulator tiles (the
// fused Winograd kernels' budget), a "K-step" of 64 v_mfma_f32_32x32x2_f32 with, per MFMA, NV independent v_add_f32 and,
// per PAIR of MFMAs, NR ds_read_b128 and NW ds_write_b64 (conflict-free, 1 KB / 512 B per wave instruction), pinned
// between the MFMAs with sched_barrier.  No global memory, no barrier: issue and LDS-pipe cost only.
//   hipcc --offload-arch=gfx950 -O3 mfma_shadow.hip -o mfma_shadow_test && ./mfma_shadow_test
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NV v_add_f32 (PK: v_pk_add_f32 on register pairs) per MFMA on average, issued in bursts every G-th MFMA;
// per slot of 4 MFMAs NR ds_read_b128 whose results feed the NEXT slot's MFMAs (as in the kernels) and NW ds_write_b64.
template <int NV, int G, bool PK, int NR, int NW, int MODE = 0>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, float a0, float b0, const float* gsrc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x16 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    float a = a0 + tid, b = b0 - tid;
    f32x2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a0 + i, b0 - i};
    const f32x2 inc = {b, a};
    f32x4 fr[2][2] = {{{a, a, a, a}, {b, b, b, b}}, {{a, b, a, b}, {b, a, b, a}}};
    const float* rd = smem + (tid >> 6) * 4096 + lane * 4;
    float* wr = smem + 16384 + (tid >> 6) * 2048 + lane * 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = m >> 2, cur = slot & 1;
            if ((m & 3) == 0) {          // top of a slot: fragment reads for the next slot
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    fr[cur ^ 1][i & 1] = *reinterpret_cast<const f32x4*>(rd + (slot & 3) * 256 + i * 1024);
            }
            acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][0][m & 3], fr[cur][1][m & 3], acc[m & 15], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((m % G) == 0) {
#pragma unroll
                for (int i = 0; i < NV * G; ++i) {
                    if constexpr (PK) { v[i & 7] += inc; asm volatile("" : "+v"(v[i & 7])); }
                    else { v[i & 7][0] += b; asm volatile("" : "+v"(v[i & 7][0])); }
                    if constexpr (MODE == 1) {      // an LDS write behind every second add, inside the burst
                        if (i & 1) *reinterpret_cast<f32x2*>(wr + (slot & 3) * 128 + (i >> 1) * 512) = v[i & 7];
                    }
                }
            }
            if constexpr (MODE != 1) {
                if ((m & 3) == 1) {
#pragma unroll
                    for (int i = 0; i < NW; ++i) *reinterpret_cast<f32x2*>(wr + (slot & 3) * 128 + i * 512) = v[i & 7];
                }
            }
            if constexpr (MODE == 2) {      // one 16-byte global load per slot (L2-resident source), consumed 8 slots later
                if ((m & 3) == 2) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(gsrc + ((it * 16 + slot) & 1023) * 1024 + tid * 4);
                    v[slot & 7][0] += t[0];
                }
            }
            if constexpr (MODE == 3) {      // scalar work: 4 SALU ops per MFMA
                int sidx = __builtin_amdgcn_readfirstlane(it + m);
                asm volatile("s_add_u32 %0, %0, 1\n\ts_lshl_b32 %0, %0, 1\n\ts_and_b32 %0, %0, 0xffff\n\ts_add_u32 %0, %0, 3" : "+s"(sidx));
                if (sidx == 0x12345678) v[0][0] += 1.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.678f) out[tid] = s;
}

template <int NV, int G, bool PK, int NR, int NW, int MODE = 0>
void run(float* out, const float* gsrc) {
    const int iters = 4000, blocks = 256;
    const size_t lds = 96 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, G, PK, NR, NW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, G, PK, NR, NW, MODE>), dim3(blocks), dim3(256), lds, 0, out, iters, 0.f, 0.f, gsrc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    const double cyc = best * 1e-3 / (iters * 64.0) * 2.4e9;
    printf("mode %d per 64 MFMAs: %3d %s in bursts of %2d  %2d ds_read_b128  %2d ds_write_b64   %8.3f ms  %6.1f TFLOP/s  %5.1f cycles/MFMA @2.4GHz\n",
           MODE, NV * 64, PK ? "v_pk_add" : "v_add   ", NV * G, NR * 16, NW * 16, best, flops / best / 1e9, cyc);
}

// Two waves per SIMD (512 threads, 8 accumulator tiles each): does one wave's ALU burst overlap the other wave's MFMAs?
// Per wave and "K-step": 32 MFMAs + NV v_add_f32 per MFMA in bursts of 4 x NV (one burst per slot of 4 MFMAs).
// PH: the odd wave of a SIMD starts half a period (BURST / 2 MFMAs) out of phase; BURST = MFMAs between two ALU bursts
template <int NV, int BURST = 4, bool PH = false>
__global__ void __launch_bounds__(512, 1) k2(float* out, int iters, float a0, float b0) {
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int tid = threadIdx.x;
    float a = a0 + tid, b = b0 - tid;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0 + i;
    if (PH && ((tid >> 6) & 4)) {       // waves 4..7 share the SIMDs of waves 0..3
#pragma unroll
        for (int m = 0; m < BURST / 2; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 7], 0, 0, 0);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if ((m % BURST) == 0) {
#pragma unroll
                for (int i = 0; i < BURST * NV; ++i) { v[i & 7] += b; asm volatile("" : "+v"(v[i & 7])); }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[tid] = s;
}
template <int NV, int BURST = 4, bool PH = false>
void run2(float* out) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k2<NV, BURST, PH>), dim3(blocks), dim3(512), 0, 0, out, iters, 0.f, 0.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 8 * iters * 32.0 * 4096.0;
    printf("2 waves/SIMD%s, per wave and 32 MFMAs: %3d v_add in bursts of %2d   %8.3f ms  %6.1f TFLOP/s  %5.1f cycles/MFMA @2.4GHz\n",
           PH ? " (out of phase)" : "", NV * 32, NV * BURST, best, flops / best / 1e9, best * 1e-3 / (iters * 64.0) * 2.4e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    float* g; (void)hipMalloc(&g, 1024 * 1024 * 4); (void)hipMemset(g, 0, 1024 * 1024 * 4);
    run<0, 1, false, 0, 0>(out, g);
    run<1, 1, false, 0, 0>(out, g); run<1, 4, false, 0, 0>(out, g); run<1, 16, false, 0, 0>(out, g); run<1, 64, false, 0, 0>(out, g);
    run<2, 4, false, 0, 0>(out, g); run<2, 16, false, 0, 0>(out, g); run<2, 32, false, 0, 0>(out, g);
    run<1, 4, false, 2, 2>(out, g);
    run<1, 4, false, 2, 0, 1>(out, g);      // the same adds with 2 LDS writes inside each burst of 4
    run<2, 8, false, 2, 0, 1>(out, g);
    run<0, 1, false, 2, 2, 2>(out, g);      // + 16 global loads per K-step
    run<1, 4, false, 2, 2, 2>(out, g);
    run<0, 1, false, 0, 0, 3>(out, g);      // 256 SALU ops per K-step
    run2<0>(out); run2<1>(out); run2<2>(out); run2<3>(out); run2<4>(out);
    run2<2, 8>(out); run2<2, 16>(out); run2<2, 32>(out); run2<3, 8>(out); run2<3, 16>(out);
    run2<2, 4, true>(out); run2<2, 8, true>(out); run2<2, 16, true>(out); run2<3, 8, true>(out); run2<3, 16, true>(out);
    return 0;
}
