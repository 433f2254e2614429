// VARIANT: 8 waves (two per SIMD), v_mfma_f32_16x16x4_f32, wave tile = 16 components x (32 channels x 16 tiles).
// Feasibility probe for a Winograd GEMM that owns all 16 components of a (64 output channels x 64 tiles) block
// and applies the output transform in registers.  Operand traffic into LDS is 4x that of
// the 256 x 256 tile (16 flop/B): does the LDS-DMA path carry it beside the MFMAs?
//   hipcc --offload-arch=gfx950 -O3 wino_fused_probe.hip -o wino_fused_probe_test && ./wino_fused_probe_test
// dbg bits: 1 no A copies, 2 no B copies, 4 no epilogue stores, 8 no MFMA
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ u32x4 make_desc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return d;
}

constexpr int BKC = 8;                    // channels per K-step
constexpr int STAGE = 16 * BKC * 64;      // floats of one operand of one stage (32 KB)

// Ug[cb][mb][xi][wm][lane][e]: e = 2*mt + kq -> U[xi][c = cb*8 + 4*kq + (lane>>4)][m = mb*64 + wm*32 + mt*16 + (lane&15)]
// V[xi][c][T];  out[o][k][T]
template <int dbg>
__global__ void __launch_bounds__(512, 2) probe(const float* __restrict__ Ug, const float* __restrict__ V,
                                                float* __restrict__ out, int C, int K, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 2, wn = wid & 3;
    const int l15 = lane & 15, lk = lane >> 4;
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const int mblocks = K / 64;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int mb = j % mblocks, tb = (j / mblocks) * 8 + xcd;
    const int t0 = tb * 64;
    if (t0 >= T) return;
    const int nk = C / BKC;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const u32x4 a_desc = make_desc(Ug, 0x7ffffffcu);
    const unsigned lane16 = lane * 16u;
    const unsigned b_lane = (unsigned)((lane >> 4) * T + t0 + (lane & 15) * 4) * 4u;
    const long long planeV = (long long)C * T;

    auto issue = [&](int kt, int buf) {
        const unsigned abase = (unsigned)((kt * mblocks + mb) * (2 * STAGE / 2)) * 4u;   // 8192 floats per (cb, mb)
        static_for<0, 4>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            const int piece = wid_s * 4 + q;
            if constexpr (!(dbg & 1)) {
                const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * 2 * STAGE + piece * 256) * 4u);
                const unsigned a_s = __builtin_amdgcn_readfirstlane(abase + (unsigned)piece * 1024u);
                const unsigned av = lane16;
                const u32x4 ad = a_desc;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(la), "v"(av), "s"(ad), "s"(a_s) : "memory", "m0");
            }
            if constexpr (!(dbg & 2)) {
                const int xi = piece >> 1, half = piece & 1;
                const u32x4 bd = make_desc(V + (long long)xi * planeV, (unsigned)(planeV * 4));
                const unsigned lb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * 2 * STAGE + STAGE + piece * 256) * 4u);
                const unsigned b_s = __builtin_amdgcn_readfirstlane((unsigned)((kt * BKC + half * 4) * T) * 4u);
                const unsigned bv = b_lane;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(lb), "v"(bv), "s"(bd), "s"(b_s) : "memory", "m0");
            }
        });
    };
    auto retire = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x4a acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }

    issue(0, 0);
    retire();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const float* As = smem + cur * 2 * STAGE;
        const float* Bs = As + STAGE;
        static_for<0, 16>([&](auto xi_) {
            constexpr int xi = decltype(xi_)::value;
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(As + (xi * 2 + wm) * 256 + lane * 4);
            float b[2];
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) b[kq] = Bs[(xi * BKC + 4 * kq + lk) * 64 + wn * 16 + l15];
            if constexpr (!(dbg & 8)) {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b[0], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b[0], acc[xi][1], 0, 0, 0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b[1], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b[1], acc[xi][1], 0, 0, 0);
            } else {
                acc[xi][0][0] += a4[0] * b[0] + a4[1] * b[1] + a4[2] * b[0] + a4[3] * b[1];
            }
        });
        retire();
        cur ^= 1;
    }
    if constexpr (dbg & 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][0][r] + acc[i][1][r];
        if (s == 123.456f) out[0] = s;
        return;
    }
    // output transform per (row, tile): A^T m A
    const long long plane = (long long)K * T;
#pragma unroll
    for (int rr8 = 0; rr8 < 8; ++rr8) {
        const int mt = rr8 >> 2, r = rr8 & 3;
        const int row = mb * 64 + wm * 32 + mt * 16 + 4 * lk + r;
        float u[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u[0][s] = (acc[0 + s][mt][r] + acc[4 + s][mt][r]) + acc[8 + s][mt][r];
            u[1][s] = (acc[4 + s][mt][r] - acc[8 + s][mt][r]) - acc[12 + s][mt][r];
        }
        float* dst = out + (long long)row * T + t0 + wn * 16 + l15;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            dst[(2 * rr + 0) * plane] = (u[rr][0] + u[rr][1]) + u[rr][2];
            dst[(2 * rr + 1) * plane] = (u[rr][1] - u[rr][2]) - u[rr][3];
        }
    }
}

int main(int argc, char** argv) {
    const int K = 256, T = 32768;
    for (int C : {256, 1024}) {
        const size_t nU = (size_t)16 * C * K, nV = (size_t)16 * C * T, nO = (size_t)4 * K * T;
        std::vector<float> hU(nU), hV(nV);
        unsigned long long sd = 88172645463325252ull; auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)(sd & 0xffffff) / 16777216.0f - 0.5f; };
        for (auto& v : hU) v = rnd();
        for (auto& v : hV) v = rnd();
        // device layout of U
        std::vector<float> hUg(nU);
        for (int cb = 0; cb < C / 8; ++cb)
            for (int mb = 0; mb < K / 64; ++mb)
                for (int xi = 0; xi < 16; ++xi)
                    for (int wm = 0; wm < 2; ++wm)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int mt = e >> 1, kq = e & 1;
                                const int c = cb * 8 + 4 * kq + (lane >> 4), m = mb * 64 + wm * 32 + mt * 16 + (lane & 15);
                                hUg[((((size_t)(cb * (K / 64) + mb) * 16 + xi) * 2 + wm) * 64 + lane) * 4 + e] =
                                    hU[((size_t)xi * C + c) * K + m];
                            }
        float *dU, *dV, *dO;
        hipMalloc(&dU, nU * 4); hipMalloc(&dV, nV * 4); hipMalloc(&dO, nO * 4);
        hipMemcpy(dU, hUg.data(), nU * 4, hipMemcpyHostToDevice);
        hipMemcpy(dV, hV.data(), nV * 4, hipMemcpyHostToDevice);
        const size_t lds = 4 * STAGE * sizeof(float);

        const int blocks = (K / 64) * (T / 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](auto kern, int dbg) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, dU, dV, dO, C, K, T);
            hipDeviceSynchronize();
            const int reps = 10;
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, dU, dV, dO, C, K, T);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("C %4d dbg %d: %8.3f ms  %7.1f TF (%s)\n", C, dbg, ms, 16.0 * 2.0 * K * C * T / ms / 1e9,
                   hipGetErrorString(hipGetLastError()));
        };
        run(probe<0>, 0); run(probe<1>, 1); run(probe<2>, 2); run(probe<3>, 3); run(probe<4>, 4); run(probe<7>, 7); run(probe<8>, 8);
        // correctness (dbg 0): a few sampled outputs against the host
        hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), lds, 0, dU, dV, dO, C, K, T);
        std::vector<float> hO(nO);
        hipMemcpy(hO.data(), dO, nO * 4, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int s = 0; s < 64; ++s) {
            const int k = rand() % K, t = rand() % T;
            double m[16];
            for (int xi = 0; xi < 16; ++xi) {
                double a = 0;
                for (int c = 0; c < C; ++c) a += (double)hU[((size_t)xi * C + c) * K + k] * hV[((size_t)xi * C + c) * T + t];
                m[xi] = a;
            }
            double u[2][4];
            for (int q = 0; q < 4; ++q) { u[0][q] = m[q] + m[4 + q] + m[8 + q]; u[1][q] = m[4 + q] - m[8 + q] - m[12 + q]; }
            for (int rr = 0; rr < 2; ++rr) {
                const double o0 = u[rr][0] + u[rr][1] + u[rr][2], o1 = u[rr][1] - u[rr][2] - u[rr][3];
                const double g0 = hO[((size_t)(2 * rr) * K + k) * T + t], g1 = hO[((size_t)(2 * rr + 1) * K + k) * T + t];
                maxerr = fmax(maxerr, fmax(fabs(g0 - o0), fabs(g1 - o1)));
            }
        }
        printf("C %d max abs err vs host fp64 on 64 samples: %.3e\n", C, maxerr);
        hipFree(dU); hipFree(dV); hipFree(dO);
    }
    return 0;
}
