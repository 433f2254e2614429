// wino_patch.h - what the fused Winograd kernels (conv_winograd2.hip: forward / dgrad; conv_winograd3.hip: weight
// gradient) share: the patch loads of a tile PAIR and the B^T d B transform in the operation order of
// conv_winograd.hip's wino_input_kernel.
#pragma once
#include "igemm2_common.h"

namespace {

typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef unsigned u32x2 __attribute__((vector_size(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ u32x4 make_desc(const void* base, unsigned bytes) {   // raw buffer: stride 0, 32-bit data
    const unsigned long long a = (unsigned long long)base;
    u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return d;
}

// ---- patch loads.  A thread owns the tile pair (tp, tp + 1) of one channel; R[r][col] are the columns its two patches
// touch in patch row r.  Column of patch A / B for patch column s:
//   DM 1 (d = 1, pitched):  R[r][0..5] = 6 consecutive floats, A: s, B: s + 2          loads: x4 + x2 per row
//   DM 2 (d = 2, pitched):  R[r][0..7] = 8 consecutive floats, A: 2s, B: 2s + 1        loads: x4 + x4 per row
//   DM 4 (d >= 4 even, W even): R[r][2s..2s+1] = 2 floats at column w0 + s d            loads: 4 x x2 per row
// (anything else - a dense x with dilation 1 / 2, odd dilations - stays on the three-pass path of conv_winograd.hip)
template <int DM> struct PatchCfg;
template <> struct PatchCfg<1> { static constexpr int NV = 8, NCOL = 6; };
template <> struct PatchCfg<2> { static constexpr int NV = 8, NCOL = 8; };
template <> struct PatchCfg<4> { static constexpr int NV = 16, NCOL = 8; };
template <int DM> __device__ __forceinline__ constexpr int colA(int s) { return DM == 1 ? s : 2 * s; }
template <int DM> __device__ __forceinline__ constexpr int colB(int s) { return DM == 1 ? s + 2 : 2 * s + 1; }

template <int DM>
__device__ __forceinline__ void load_patches(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[PatchCfg<DM>::NV],
                                             unsigned soff, float (&R)[4][8]) {
    static_for<0, 4>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        if constexpr (DM == 1) {
            const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[2 * r], soff, 0));
            const f32x2 b = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[2 * r + 1], soff, 0));
            R[r][0] = a[0]; R[r][1] = a[1]; R[r][2] = a[2]; R[r][3] = a[3]; R[r][4] = b[0]; R[r][5] = b[1];
        } else if constexpr (DM == 2) {
            const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[2 * r], soff, 0));
            const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[2 * r + 1], soff, 0));
            R[r][0] = a[0]; R[r][1] = a[1]; R[r][2] = a[2]; R[r][3] = a[3];
            R[r][4] = b[0]; R[r][5] = b[1]; R[r][6] = b[2]; R[r][7] = b[3];
        } else {
            static_for<0, 4>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                const f32x2 a = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[4 * r + s], soff, 0));
                R[r][2 * s] = a[0]; R[r][2 * s + 1] = a[1];
            });
        }
    });
}

// load I of the NV loads of a step (the K loop spreads them over its MFMA groups)
template <int DM, int I>
__device__ __forceinline__ void load_one(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[PatchCfg<DM>::NV],
                                         unsigned soff, float (&R)[4][8]) {
    if constexpr (DM == 1) {
        constexpr int r = I >> 1;
        if constexpr ((I & 1) == 0) {
            const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[I], soff, 0));
            R[r][0] = a[0]; R[r][1] = a[1]; R[r][2] = a[2]; R[r][3] = a[3];
        } else {
            const f32x2 b = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[I], soff, 0));
            R[r][4] = b[0]; R[r][5] = b[1];
        }
    } else if constexpr (DM == 2) {
        constexpr int r = I >> 1, c0 = 4 * (I & 1);
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[I], soff, 0));
        R[r][c0] = a[0]; R[r][c0 + 1] = a[1]; R[r][c0 + 2] = a[2]; R[r][c0 + 3] = a[3];
    } else {
        constexpr int r = I >> 2, s = I & 3;
        const f32x2 a = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[I], soff, 0));
        R[r][2 * s] = a[0]; R[r][2 * s + 1] = a[1];
    }
}

// B^T p over the patch rows (the same expressions as wino_input_kernel)
template <int DM>
__device__ __forceinline__ void row_transform(const float (&R)[4][8], float (&q)[4][8]) {
#pragma unroll
    for (int c = 0; c < PatchCfg<DM>::NCOL; ++c) {
        q[0][c] = R[0][c] - R[2][c];
        q[1][c] = R[1][c] + R[2][c];
        q[2][c] = R[2][c] - R[1][c];
        q[3][c] = R[1][c] - R[3][c];
    }
}
template <int DM, int ROW>
__device__ __forceinline__ void row_transform_one(const float (&R)[4][8], float (&q)[4][8]) {
#pragma unroll
    for (int c = 0; c < PatchCfg<DM>::NCOL; ++c) {
        if constexpr (ROW == 0) q[0][c] = R[0][c] - R[2][c];
        if constexpr (ROW == 1) q[1][c] = R[1][c] + R[2][c];
        if constexpr (ROW == 2) q[2][c] = R[2][c] - R[1][c];
        if constexpr (ROW == 3) q[3][c] = R[1][c] - R[3][c];
    }
}
// component xi = 4 r + s of patch A and patch B
template <int DM, int XI>
__device__ __forceinline__ f32x2 col_transform(const float (&q)[4][8]) {
    constexpr int r = XI >> 2, s = XI & 3;
    f32x2 o;
    if constexpr (s == 0) { o[0] = q[r][colA<DM>(0)] - q[r][colA<DM>(2)]; o[1] = q[r][colB<DM>(0)] - q[r][colB<DM>(2)]; }
    if constexpr (s == 1) { o[0] = q[r][colA<DM>(1)] + q[r][colA<DM>(2)]; o[1] = q[r][colB<DM>(1)] + q[r][colB<DM>(2)]; }
    if constexpr (s == 2) { o[0] = q[r][colA<DM>(2)] - q[r][colA<DM>(1)]; o[1] = q[r][colB<DM>(2)] - q[r][colB<DM>(1)]; }
    if constexpr (s == 3) { o[0] = q[r][colA<DM>(1)] - q[r][colA<DM>(3)]; o[1] = q[r][colB<DM>(1)] - q[r][colB<DM>(3)]; }
    return o;
}

}  // namespace
