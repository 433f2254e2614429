// syncbn_p2p.hip — the SyncBatchNorm statistics exchange (engine.py:65, nn.SyncBatchNorm.convert_sync_batchnorm)
// as ONE kernel launch per BatchNorm layer and direction, instead of one RCCL collective each (115 all-gathers
// forward + 115 all-reduces backward per DeepLabv3-R101 step, every one latency-bound: 2C+1 <= 4097 floats).
//
// Every rank owns a MAILBOX in its own HBM (fine-grained device memory, mapped into the peers through
// hipIpcOpenMemHandle): [slot][source rank][128-byte header | payload].  Exchange number `seq` (1, 2, 3, ... the
// same on all ranks: they walk the same layers in the same order) uses slot seq % 4:
//   push   the one workgroup writes its row into entry (slot, my rank) of EVERY rank's mailbox over xGMI
//          (system-scope stores), fences, and then stores `seq` into the header of each of those entries;
//   pull   lane r polls the header of entry (slot, r) of its OWN mailbox (local memory) until it reads `seq`;
//   reduce the rows are combined in RANK ORDER in fp64 - the same expression on every rank, hence bit-identical
//          results everywhere, and the same numbers the all-gather + dcfp_syncbn_combine_f32 path produces.
// Slot reuse: a rank can finish exchange s only after every peer has posted s, and a peer posts s+1 only after it
// has finished s, so no rank is ever more than one exchange ahead of another's reads; 4 slots are plenty.
// Exit condition: the poll gives up after `spin_limit` rounds (a dead or diverged peer), poisons the outputs with
// NaN and records `seq` in *status - the kernel always drains.
#include "common.h"
#include <string.h>

namespace {

constexpr int kMaxWorld = 8;    // one xGMI node
constexpr int kSlots = 4;
constexpr int kHdr = 32;   // floats: one 128-byte line of its own in front of each payload; word 0 = seq
constexpr int kP2pThreads = 1024;
constexpr int kPer = 5;      // floats per lane and push pass: 5 x 1024 covers the 4097 floats of a 2048-channel layer

struct Peers {
    float* box[kMaxWorld];
};

__device__ __forceinline__ float* entry_of(float* box, int world, int cap, int slot, int src) {
    return box + ((size_t)slot * world + src) * (size_t)(kHdr + cap);
}
__device__ __forceinline__ float ld_sys(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void st_sys(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// MODE 0: out[world][n] = the gathered rows.  MODE 1: out[n] = sum of the rows in rank order (SyncBN backward:
// [sum g, sum g*(x-mean)]).  MODE 2: rows are (mean[C], var[C], count), n = 2C+1; out = (pooled mean[C], pooled
// biased variance[C], total count) and the running statistics of `run` (SyncBN forward).
template <int MODE>
__global__ void __launch_bounds__(kP2pThreads)
syncbn_p2p_kernel(Peers peers, int world, int rank, unsigned seq, int cap, const float* __restrict__ local, int n,
                  float* __restrict__ out, DcfpBnRunning run, unsigned spin_limit, int* __restrict__ status) {
#pragma clang fp contract(off)   // the rank-order reductions round every product and sum on their own, as the host reference does
    __shared__ int ok;
    const int tid = threadIdx.x;
    const int slot = (int)(seq % kSlots);
    if (tid == 0) ok = 1;
    // ---- push my row to every rank (my own mailbox included: one code path, and the reduce reads one place).
    // The row is read into registers once (kPer floats per lane and pass) and then stored to all peers back to back:
    // no store waits for a load, the xGMI writes to the different peers are all in flight together.
    for (int base = 0; base < n; base += kP2pThreads * kPer) {
        float v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int i = base + k * kP2pThreads + tid;
            v[k] = i < n ? local[i] : 0.f;
        }
        for (int p = 0; p < world; ++p) {
            float* e = entry_of(peers.box[p], world, cap, slot, rank) + kHdr;
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i = base + k * kP2pThreads + tid;
                if (i < n) st_sys(e + i, v[k]);
            }
        }
    }
    // Every mailbox access of this kernel is a system-scope (sc0 sc1) access: write-through stores, cache-bypassing
    // loads.  So "my row is visible" needs no L2 write-back (a release fence's buffer_wbl2 would flush whatever the
    // previous conv left dirty in this XCD's L2), only that the stores have been acknowledged: vmcnt(0) in every
    // lane, then the barrier, then the flags - which travel the same path to the same peer behind the rows.
    // (inline asm with a memory clobber, in EVERY storing wave: the guide's publish recipe - a builtin wait may be moved or
    //  merged by the compiler, and the flag must not overtake the rows)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < world) {
        unsigned* f = reinterpret_cast<unsigned*>(entry_of(peers.box[tid], world, cap, slot, rank));
        __hip_atomic_store(f, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- pull: wait for every rank's row in my mailbox
    float* mine = peers.box[rank];
    if (tid < world) {
        const unsigned* f = reinterpret_cast<const unsigned*>(entry_of(mine, world, cap, slot, tid));
        unsigned it = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++it > spin_limit) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();
    const int nout = MODE == 0 ? world * n : n;
    if (!ok) {
        for (int i = tid; i < nout; i += kP2pThreads) out[i] = __int_as_float(0x7fc00000);
        if (tid == 0) atomicExch(status, (int)seq);
        return;
    }
    // (the row loads below are system-scope loads issued after the barrier the polling lanes joined: no cached copy
    //  can stand in for them, so no L2 invalidate is needed either)
    const float* rows = entry_of(mine, world, cap, slot, 0) + kHdr;
    const size_t rs = (size_t)(kHdr + cap);   // row r of this exchange starts at rows + r * rs
    // Loads first, arithmetic second: a system-scope load goes to memory (~1-2 us), so the `world` loads of one output
    // element are issued together (fixed-trip loops over kMaxWorld with r < world predicates) before any is consumed.
    if (MODE == 0) {
        for (int i = tid; i < n; i += kP2pThreads) {
            float v[kMaxWorld];
#pragma unroll
            for (int r = 0; r < kMaxWorld; ++r) v[r] = r < world ? ld_sys(rows + r * rs + i) : 0.f;
#pragma unroll
            for (int r = 0; r < kMaxWorld; ++r)
                if (r < world) out[(size_t)r * n + i] = v[r];
        }
    } else if (MODE == 1) {
        for (int i = tid; i < n; i += kP2pThreads) {
            float v[kMaxWorld];
#pragma unroll
            for (int r = 0; r < kMaxWorld; ++r) v[r] = r < world ? ld_sys(rows + r * rs + i) : 0.f;
            float s = v[0];
#pragma unroll
            for (int r = 1; r < kMaxWorld; ++r)
                if (r < world) s += v[r];
            out[i] = s;
        }
    } else {
        // the expression of syncbn_combine_kernel (bn.hip) / ops.syncbn_combine_reference: fp64, rank order, every
        // product and sum rounded on its own
        const int C = (n - 1) / 2;
        float cnt[kMaxWorld];
#pragma unroll
        for (int r = 0; r < kMaxWorld; ++r) cnt[r] = r < world ? ld_sys(rows + r * rs + 2 * C) : 0.f;
        double tot = 0.0;
#pragma unroll
        for (int r = 0; r < kMaxWorld; ++r)
            if (r < world) tot = __dadd_rn(tot, (double)cnt[r]);
        for (int c = tid; c < C; c += kP2pThreads) {
            float mr[kMaxWorld], vr[kMaxWorld];
#pragma unroll
            for (int r = 0; r < kMaxWorld; ++r) {
                mr[r] = r < world ? ld_sys(rows + r * rs + c) : 0.f;
                vr[r] = r < world ? ld_sys(rows + r * rs + C + c) : 0.f;
            }
            double m = 0.0, v = 0.0;
#pragma unroll
            for (int r = 0; r < kMaxWorld; ++r)
                if (r < world) m = __dadd_rn(m, __dmul_rn((double)mr[r], (double)cnt[r]));
            m /= tot;
#pragma unroll
            for (int r = 0; r < kMaxWorld; ++r)
                if (r < world) {
                    const double d = __dadd_rn((double)mr[r], -m);
                    v = __dadd_rn(v, __dmul_rn(__dadd_rn((double)vr[r], __dmul_rn(d, d)), (double)cnt[r]));
                }
            out[c] = (float)m;
            out[C + c] = (float)(v / tot);
            running_update(run, c, (float)m, (float)(v / tot), (float)tot);
        }
        if (tid == 0) out[2 * C] = (float)tot;
    }
}

}  // namespace

extern "C" size_t dcfp_syncbn_p2p_mailbox_bytes(int world, int cap_floats) {
    if (world <= 0 || world > kMaxWorld || cap_floats <= 0) return 0;
    return (size_t)kSlots * world * (size_t)(kHdr + cap_floats) * sizeof(float);
}

extern "C" int dcfp_p2p_alloc(size_t bytes, int kind, void** ptr) {
    if (!ptr || bytes == 0 || kind < 0 || kind > 2) return DCFP_E_BADDESC;
    *ptr = nullptr;
    hipError_t e = kind == 2 ? hipMalloc(ptr, bytes)
                             : hipExtMallocWithFlags(ptr, bytes, kind == 1 ? hipDeviceMallocUncached
                                                                           : hipDeviceMallocFinegrained);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*ptr, 0, bytes);                  // sequence numbers start above 0
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
        return (int)e;
    }
    return DCFP_OK;
}

extern "C" int dcfp_p2p_free(void* ptr) {
    if (!ptr) return DCFP_E_BADDESC;
    const hipError_t e = hipFree(ptr);
    return e == hipSuccess ? DCFP_OK : (int)e;
}

extern "C" int dcfp_p2p_export(void* ptr, void* handle64) {
    if (!ptr || !handle64) return DCFP_E_BADDESC;
    static_assert(sizeof(hipIpcMemHandle_t) == DCFP_P2P_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return (int)e;
    memcpy(handle64, &h, sizeof(h));
    return DCFP_OK;
}

extern "C" int dcfp_p2p_import(const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return DCFP_E_BADDESC;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    *ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? DCFP_OK : (int)e;
}

extern "C" int dcfp_p2p_unmap(void* ptr) {
    if (!ptr) return DCFP_E_BADDESC;
    const hipError_t e = hipIpcCloseMemHandle(ptr);
    return e == hipSuccess ? DCFP_OK : (int)e;
}

extern "C" int dcfp_syncbn_p2p_exchange_f32(void* const* mailboxes, int world, int rank, uint32_t seq,
                                            int cap_floats, const float* local, int n, int mode, float* out,
                                            const DcfpBnRunning* run, uint32_t spin_limit, int32_t* status,
                                            dcfp_stream_t stream) {
    if (!mailboxes || world <= 0 || world > kMaxWorld || rank < 0 || rank >= world || seq == 0 || !local || !out ||
        !status || n <= 0 || n > cap_floats || mode < 0 || mode > 2 || spin_limit == 0)
        return DCFP_E_BADDESC;
    if (mode == 2 && (n < 3 || (n & 1) == 0)) return DCFP_E_BADDESC;
    Peers peers;
    for (int r = 0; r < kMaxWorld; ++r) {
        peers.box[r] = r < world ? static_cast<float*>(mailboxes[r]) : nullptr;
        if (r < world && (!mailboxes[r] || (reinterpret_cast<uintptr_t>(mailboxes[r]) & 127u))) return DCFP_E_BADDESC;
    }
    DcfpBnRunning rn = {nullptr, nullptr, nullptr, 0.f, 0};
    if (run) {
        if (mode != 2 || (run->running_mean && !run->running_var)) return DCFP_E_BADDESC;
        rn = *run;
    }
    int* st = reinterpret_cast<int*>(status);
#define LAUNCH_P2P(M)                                                                                        \
    hipLaunchKernelGGL(syncbn_p2p_kernel<M>, dim3(1), dim3(kP2pThreads), 0, dcfp_s(stream), peers, world, rank, \
                       seq, cap_floats, local, n, out, rn, spin_limit, st)
    if (mode == 0) LAUNCH_P2P(0);
    else if (mode == 1) LAUNCH_P2P(1);
    else LAUNCH_P2P(2);
#undef LAUNCH_P2P
    DCFP_RETURN_LAUNCH();
}
