// Device-side helpers shared by the kernel translation units of libvds (gfx950 / CDNA4 only): wave64 lane helpers, DPP row
// reductions, LDS ordering points, arrival posting (ring / far inbox) and the generic far-list / ring drain of one bucket.
// Everything here is `static` / inline: each .hip file of csrc/ is its own translation unit (no relocatable device code).
#pragma once
#include "vds_device.h"
#include <type_traits>
#include <cstring>

namespace vds {

#define WAVE 64
#define IMAX 0x7FFFFFFF
#define WORK_FULL 0x80000000u

// Instrumented build only (make prof): timing-only ablation switches (results are INVALID when non-zero) and per-wave,
// per-section cycle sums.  Each translation unit owns its copy; vds_api.hip sets / sums them all (VDS_PROF_ACCESSORS).
#define PROF_WAVES (1 << 16)
#define PROF_SLOTS 32
#ifdef VDS_PROF
static __device__ int g_ablate = 0;
static __device__ unsigned long long g_prof[PROF_WAVES * PROF_SLOTS];   // bit7 of g_ablate: per-wave, per-section cycles
#define PROF_STAMP(i) do { if (prof) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) g_prof[(size_t)pwave * PROF_SLOTS + (i)] += t_ - tprev; tprev = __builtin_amdgcn_s_memtime(); } } while (0)
#define PROF_STAMP_NW(i) do { if (prof) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) g_prof[(size_t)pwave * PROF_SLOTS + (i)] += t_ - tprev; tprev = t_; } } while (0)
// out[0..14]: slots 0..14 summed over the wavefronts, out[15]: wavefronts that recorded slot 6, out[16..31]: slots 16..31
// (accumulated INTO out: the caller zeroes it and adds up the translation units)
#define VDS_PROF_ACCESSORS(tag) \
void read_prof_##tag(unsigned long long *out, hipStream_t st) { \
    (void)hipStreamSynchronize(st); \
    static unsigned long long host[PROF_WAVES * PROF_SLOTS]; \
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(host)); \
    for (size_t w = 0; w < PROF_WAVES; ++w) { \
        for (int i = 0; i < 15; ++i) out[i] += host[w * PROF_SLOTS + i]; \
        for (int i = 16; i < 32; ++i) out[i] += host[w * PROF_SLOTS + i]; \
        if (host[w * PROF_SLOTS + 6]) out[15]++; \
    } \
    memset(host, 0, sizeof(host)); \
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), host, sizeof(host)); \
} \
void set_ablate_##tag(int f, hipStream_t st) { (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ablate), &f, sizeof(int), 0, hipMemcpyHostToDevice, st); }
#else
#define PROF_STAMP(i) do { } while (0)
#define PROF_STAMP_NW(i) do { } while (0)
#define VDS_PROF_ACCESSORS(tag) \
void read_prof_##tag(unsigned long long *, hipStream_t) { } \
void set_ablate_##tag(int, hipStream_t) { }
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int v) {
    // full row mask: every lane has a valid source for the controls used here (quad_perm, row mirrors, row_ror), so
    // bound_ctrl:0 with an undefined `old` is equivalent - and lets the compiler fold the move into the consuming
    // VALU op (v_min_i32_dpp / v_add_u32_dpp) instead of copy + v_mov_b32_dpp + op
    if (ROW_MASK == 0xF) return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ int dpp_zero(int v) {   // out-of-row sources read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Minimum over all 64 lanes (all lanes must be active); wave-uniform result.
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_mov<0xB1, 0xF>(v, v));   // quad_perm [1,0,3,2]
    v = min(v, dpp_mov<0x4E, 0xF>(v, v));   // quad_perm [2,3,0,1]
    v = min(v, dpp_mov<0x141, 0xF>(v, v));  // row_half_mirror
    v = min(v, dpp_mov<0x140, 0xF>(v, v));  // row_mirror: every lane holds its row's min
    v = min(v, dpp_mov<0x142, 0xA>(v, v));  // row_bcast:15 into rows 1,3
    v = min(v, dpp_mov<0x143, 0xC>(v, v));  // row_bcast:31 into rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}
// Per-16-lane-row reductions; every lane of a row receives the row's result.
__device__ __forceinline__ int row_min_i32(int v) {
    v = min(v, dpp_mov<0xB1, 0xF>(v, v));
    v = min(v, dpp_mov<0x4E, 0xF>(v, v));
    v = min(v, dpp_mov<0x141, 0xF>(v, v));
    v = min(v, dpp_mov<0x140, 0xF>(v, v));
    return v;
}
__device__ __forceinline__ int row_sum_i32(int v) {
    v += dpp_mov<0xB1, 0xF>(v, v);
    v += dpp_mov<0x4E, 0xF>(v, v);
    v += dpp_mov<0x141, 0xF>(v, v);
    v += dpp_mov<0x140, 0xF>(v, v);
    return v;
}
__device__ __forceinline__ int row_incl_scan_i32(int v) {   // row_shr:1,2,4,8 (Hillis-Steele)
    v += dpp_zero<0x111>(v);
    v += dpp_zero<0x112>(v);
    v += dpp_zero<0x114>(v);
    v += dpp_zero<0x118>(v);
    return v;
}

__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(unsigned long long x) { return __popcll(x); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// Ordering points for LDS traffic.  Both are release + acquire fences RESTRICTED TO THE LDS ADDRESS SPACE ("local"): the
// language-level ordering the protocols below need, and - unlike an unrestricted fence, which is lowered to s_waitcnt vmcnt(0),
// a full HBM round trip per use - nothing waits for the global loads / stores in flight (the ISA shows at most
// s_waitcnt lgkmcnt(0)).  wave_order(): between accesses of different lanes of ONE wavefront to its own scratch;
// wg_order(): between the wavefronts of a workgroup (k_dfs_walk's flags, records and stamps).
__device__ __forceinline__ void wave_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local"); }
__device__ __forceinline__ void wg_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local"); }

// ---------------------------------------------------------------------------------------
// Arrival posting.  An entry goes to the ring slot of the tick at which UpdateFunction will see
// `arrive <= RealExpTime` (:1016) - or, when that is >= H ticks away, to the destination's far inbox.
// LT: the caller may run on the dense layout (dispatch kernels, shared by both layouts); the tick kernels of the wide layout
// leave the branch out (k_tick_rows sits exactly at its 72-VGPR budget)
template <bool LT = false>
__device__ __forceinline__ void ring_post(const Static &S, const State &D, int dc, int r, int tick, int4 e) {
    const size_t i = ((size_t)(tick & (S.H - 1)) * S.C + dc) * S.R + r;
    const int old = atomicAdd(&D.ring_cnt[i], meta_is_dispatch(e.w) ? 1 : 0x10001);   // high half: carries an order (:889)
    const int pos = old & 0xFFFF;
    if (LT && D.sup != nullptr && !meta_is_dispatch(e.w)) atomicAdd(&D.sup[sup_index(S.C, S.R, tick, dc, r)], 1);      // (SupplyExpect in place)
    if (pos >= S.ring_cap) atomicOr(&D.err[0], ERR_RING_CAP);
    else if (LT && S.dense) {
        // dense layout: {veh << 8 | dest_local, key}; the arrival minute of a dispatched vehicle goes to ring_min (read side only)
        ring2(D)[i * S.ring_cap + pos] = make_uint2(dense_pack((unsigned)e.x, (unsigned)meta_dest(e.w)), dense_key((int)((unsigned)e.w >> 16), meta_is_dispatch(e.w), e.y));
        if (meta_is_dispatch(e.w)) D.ring_min[i * S.ring_cap + pos] = e.z;
    } else D.ring[i * S.ring_cap + pos] = e;
}

// ceil(rel / tick_minutes) for rel > 0.  SMALL: the caller guarantees rel < 2^25 (costs < 2^23), so the multiply-high form
// is exact whenever the host could build the magic number (tick_minutes >= 2) - a wave-uniform choice, no per-lane fallback
template <bool SMALL>
__device__ __forceinline__ int ticks_until(const Static &S, int rel) {
    const int n = rel + S.tick_minutes - 1;
    if (SMALL && S.tick_div_limit > (1 << 26)) return (int)__umulhi((unsigned)n, S.tick_magic);
    return n / S.tick_minutes;
}

template <bool SMALL = false, bool LT = false>
__device__ __forceinline__ void post_arrival(const Static &S, const State &D, int dc, int r, int t, int now,
                                             int veh, int id, int arrive, int is_dispatch, int dest_local) {
    const int rel = arrive - now;
    const int d = rel <= 0 ? 1 : ticks_until<SMALL>(S, rel);   // next Update is at t+1 at the earliest
    const int4 e = make_int4(veh, id, arrive, meta_pack(t, is_dispatch, dest_local));
    if (d < S.H) {
        ring_post<LT>(S, D, dc, r, t + d, e);
    } else {
        const size_t db = (size_t)dc * S.R + r;
        const int np = (t + 1) & 1;
        const int slot = atomicAdd(&D.hdr[db * HDR_WORDS + HDR_INBOX0 + np], 1);
        if (slot < S.in_cap) D.inbox[((size_t)np * S.C * S.R + db) * S.in_cap + slot] = e;
        else atomicOr(&D.err[0], ERR_INBOX_CAP);
    }
}

// ---------------------------------------------------------------------------------------
// Generic (one wavefront per bucket) update phase.
__device__ __forceinline__ int4 pending_load(const int4 *fl, const int4 *inb, int f, int P, int e) {
    int4 z = make_int4(0, 0, IMAX, 0);
    if (e >= P) return z;
    return e < f ? fl[e] : inb[e - f];
}

// far entries: those that are now < H ticks away move into the ring, the rest is compacted into fl
static __device__ void update_far(const Static &S, const State &D, int c, int r, int t, int now, int f, int qin,
                           int4 *fl, const int4 *inb, int &newf) {
    const int lane = lane_id();
    const int P = f + qin;
    newf = 0;
    const int nch = (P + WAVE - 1) / WAVE;
    for (int I = 0; I < nch; ++I) {
        const int idx = I * WAVE + lane;
        int4 e = pending_load(fl, inb, f, P, idx);
        const int rel = e.z - now;
        const int d = rel <= 0 ? 0 : ticks_until<false>(S, rel);
        const bool valid = idx < P;
        const bool keep = valid && d >= S.H;
        if (valid && !keep) ring_post(S, D, c, r, t + d, e);
        unsigned long long kb = ballot(keep);
        if (keep) {
            int pos = newf + popc64(kb & lanemask_lt());
            if (pos < S.fl_cap) fl[pos] = e;
        }
        newf += popc64(kb);
    }
    if (newf > S.fl_cap) {
        if (lane == 0) atomicOr(&D.err[0], ERR_FL_CAP);
        newf = S.fl_cap;
    }
}

// drain the ring slot due at tick t: every entry has arrived; append to the idle list in dict
// insertion order (ascending key).  Returns the number of arrivals, m is advanced.
static __device__ int drain_ring(const Static &S, const State &D, size_t b, int t, int &m, uint2 *idle) {
    const int lane = lane_id();
    const size_t si = (size_t)(t & (S.H - 1)) * S.C * S.R + b;
    int A = D.ring_cnt[si] & 0xFFFF;
    if (A == 0) return 0;
    if (A > S.ring_cap) A = S.ring_cap;
    const int4 *ring = D.ring + si * S.ring_cap;
    const int nch = (A + WAVE - 1) / WAVE;
    for (int I = 0; I < nch; ++I) {
        const int idx = I * WAVE + lane;
        const bool valid = idx < A;
        int4 e = valid ? ring[idx] : make_int4(0, 0, 0, 0);
        const unsigned long long key = entry_key(e.y, e.w);
        int rank = 0;
        for (int Jc = 0; Jc < nch; ++Jc) {
            const int jdx = Jc * WAVE + lane;
            int4 ej = (Jc == I) ? e : (jdx < A ? ring[jdx] : make_int4(0, 0, 0, 0));
            const int khi = (int)((unsigned)ej.w >> 15), klo = ej.y;
            const int nj = min(WAVE, A - Jc * WAVE);
            for (int j = 0; j < nj; ++j) {
                unsigned long long kj = ((unsigned long long)(unsigned)rdlane(khi, j) << 32) | (unsigned)rdlane(klo, j);
                rank += (kj < key) ? 1 : 0;
            }
        }
        if (valid) {
            int pos = m + rank;
            if (pos < S.idle_cap) idle[pos] = make_uint2((unsigned)e.x, (unsigned)meta_dest(e.w));
        }
    }
    if (lane == 0) D.ring_cnt[si] = 0;
    if (m + A > S.idle_cap) {
        if (lane == 0) atomicOr(&D.err[0], ERR_IDLE_CAP);
        A = S.idle_cap - m;
    }
    m += A;
    return A;
}

}  // namespace vds
