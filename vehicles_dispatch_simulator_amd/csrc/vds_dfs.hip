// Neighbour-search kernels of libvds (gfx950 only): the tick with FindServerVehicleFunction (:978-996).
//   k_dfs_walk<U8, JB>   second half of the HYBRID neighbour-search tick (the default with NeighborCanServer; first half:
//                  k_tick_rows in stamp mode, vds_tick.hip)
//   k_tick_replica2 the fallback when the hybrid tick's preconditions fail: exact lower-bound rounds, one workgroup per replica
//   k_match_dfs    the serial reference form (one wavefront walks a replica's orders in id order): last fallback and a second,
//                  independent GPU statement for the tests
#include "vds_kernels_common.h"

namespace vds {

VDS_PROF_ACCESSORS(dfs)

// ---------------------------------------------------------------------------------------
// k_match_dfs: neighbour-search mode.  One wavefront per replica walks the tick's orders in
// Order.ID order (the reference's cursor, :912-973); an order whose own cluster has idle
// vehicles takes the nearest one (:924-933), otherwise (quirk Q2, `elif` :936) the clusters of
// the precomputed FindServerVehicleFunction visit sequence are scanned in visit order
// (:978-996) with the same first-strict-minimum rule.  Removal is order preserving.
__device__ __forceinline__ void list_remove(uint2 *idle, int m, int pos) {
    const int lane = lane_id();
    for (int base = pos; base < m - 1; base += WAVE) {
        int i = base + lane;
        uint2 e = make_uint2(0u, 0u);
        if (i < m - 1) e = idle[i + 1];
        wave_fence();
        if (i < m - 1) idle[i] = e;
        wave_fence();
    }
}

__global__ __launch_bounds__(64) void k_match_dfs(Static S, State D, int t) {
    const int r = blockIdx.x;
    const int lane = lane_id();
    const DayView dv = day_view(S, r);
    if (t >= dv.T) return;
    const int now = dv.now0 + t * S.tick_minutes;
    const int o0 = dv.tick_off[t], o1 = dv.tick_off[t + 1];
    int2 *out_r = out_row(S, D, r);
    for (int oi = o0; oi < o1; ++oi) {
        const int q = S.ord_q[oi];
        const int4 rec = S.so_rec[q];
        const int pc = (int)((unsigned)rec.z >> 16), pl = rec.y & 0xFFFF;
        const size_t b = (size_t)pc * S.R + r;
        int *hdr = D.hdr + b * HDR_WORDS;
        int m = hdr[HDR_IDLE];
        long long evals = 0;
        int best_c = IMAX, best_pos = -1, best_cl = -1;   // winner: cost, position, cluster
        if (m > 0) {
            const int nc = S.cl_off[pc + 1] - S.cl_off[pc];
            const int *row = S.blk + S.blk_off[pc] + (size_t)pl * nc;
            const uint2 *idle = D.idle + b * S.idle_cap;
            int lc = IMAX, lp = -1;
            for (int base = 0; base < m; base += WAVE) {
                int i = base + lane;
                if (i < m) {
                    int cst = row[idle[i].y];
                    if (lp < 0 || cst < lc) { lc = cst; lp = i; }
                }
            }
            evals += m;
            const int minc = wave_min_i32(lp >= 0 ? lc : IMAX);
            const int minp = wave_min_i32((lp >= 0 && lc == minc) ? lp : IMAX);
            best_c = minc; best_pos = minp; best_cl = pc;
        } else {
            // DFS visit sequence (own cluster, already known empty, is position 0 and skipped)
            const int pnode = S.cl_nodes[S.cl_off[pc] + pl];
            const int *crow = S.cost + (size_t)pnode * S.N;
            for (int si = S.dfs_off[pc]; si < S.dfs_off[pc + 1]; ++si) {
                const int c2 = S.dfs_seq[si];
                const size_t b2 = (size_t)c2 * S.R + r;
                const int m2 = D.hdr[b2 * HDR_WORDS + HDR_IDLE];
                if (m2 == 0) continue;
                const uint2 *idle2 = D.idle + b2 * S.idle_cap;
                const int off2 = S.cl_off[c2];
                int lc = IMAX, lp = -1;
                for (int base = 0; base < m2; base += WAVE) {
                    int i = base + lane;
                    if (i < m2) {
                        int cst = crow[off2 + idle2[i].y];
                        if (lp < 0 || cst < lc) { lc = cst; lp = i; }
                    }
                }
                evals += m2;
                const int minc = wave_min_i32(lp >= 0 ? lc : IMAX);
                if (best_cl < 0 || minc < best_c) {              // strict < across clusters in visit order
                    const int minp = wave_min_i32((lp >= 0 && lc == minc) ? lp : IMAX);
                    best_c = minc; best_pos = minp; best_cl = c2;
                }
            }
        }
        long long *cnt = D.cnt + b * CNT_WORDS;
        int res_veh = -1, res_wait = -1;
        bool matched = best_cl >= 0 && (long long)best_c <= S.reject_threshold;
        if (matched) {
            const size_t bw = (size_t)best_cl * S.R + r;
            uint2 *widle = D.idle + bw * S.idle_cap;
            const int mw = D.hdr[bw * HDR_WORDS + HDR_IDLE];
            res_veh = (int)widle[best_pos].x;
            res_wait = best_c;
            wave_fence();
            list_remove(widle, mw, best_pos);
            if (lane == 0) {
                D.hdr[bw * HDR_WORDS + HDR_IDLE] = mw - 1;
                post_arrival(S, D, rec.z & 0xFFFF, r, t, now, res_veh, rec.x, now + res_wait + rec.w, 0, (int)((unsigned)rec.y >> 16));
            }
        }
        if (lane == 0) {
            out_r[q] = make_int2(res_veh, res_wait);
            cnt[CNT_ORDERS] += 1;
            hdr[HDR_ORDERS] += 1;
            if (!matched) cnt[CNT_REJECTS] += 1;
            else { cnt[CNT_WAIT] += res_wait; cnt[CNT_VALUE] += rec.w; }
            cnt[CNT_EVALS] += evals;
        }
        wave_fence();
    }
}
#ifndef REPL_THREADS
#define REPL_THREADS 256
#endif
#define REPL_WAVES (REPL_THREADS / WAVE)

// ---------------------------------------------------------------------------------------
// k_tick_replica2: neighbour-search mode, second generation (same lower-bound rounds as k_tick_replica, same
// results).  What changed is how a round is executed:
//   * idle lists are NOT edited during the tick.  A replica's lists stay in HBM at their positions after Update
//     ("original positions"); LDS holds a u16 mirror of them: the cost-matrix column of every entry's node, or
//     DEAD once the vehicle has been taken.  List order == original order, so "first strict minimum" is still
//     the lowest position.
//   * own-cluster matching runs on 8-LANE GROUPS, one bucket per group, fed by per-wavefront worklists of the
//     buckets that have orders older than LB: lists of up to 96 entries sit in the group's registers (12 per
//     lane), all candidates of an order are gathered in one round trip, three DPP steps give the group minimum
//     of (cost << 16 | position).  A workgroup advances up to 32 buckets concurrently.
//   * the neighbour search reads its cluster list once per wavefront (lane j = j-th candidate cluster), prefetched
//     together with the dry order's pickup node BEFORE the own-cluster pass; it walks (cluster, 64-entry chunk)
//     slots eight at a time - all LDS reads, then all cost gathers, then the minima of 32-bit keys
//     cost << 16 | slot (lane = last tie-break) - and hands (cost, visit position | list position, cluster) to
//     the winner step.
//   * Update is row-mapped: four buckets per wavefront, arrivals ranked by dict-insertion key with DPP rotations.
//   * results are written in a preliminary form {victim cluster << 16 | original position, wait}; after the
//     last round one thread per order resolves the vehicle id from the untouched HBM list, posts the arrival
//     (:954-960) and accumulates the counters, and one wavefront per bucket compacts the HBM list once (:963).
// Preconditions (Static / vds_api dfs2_ok): order ids < 2^20, clusters <= 2047 nodes, 0 <= cost < 2^15,
// V <= 20480, N <= 65534, C <= 3072, idle_cap <= 32767, < 32768 orders per tick; otherwise k_tick_replica runs.
#define ID_BITS 20
#define ID_MASK ((1 << ID_BITS) - 1)
#define GRP 8                               // lanes per own-cluster group
#define GRPS_WAVE (WAVE / GRP)
#ifndef OB
#define OB 1                                // orders of one bucket whose cost gathers are in flight together
#endif
#define RCNT 4                              // counters accumulated by the resolve pass: orders, rejects, wait, value

__device__ __forceinline__ int grp_min_i32(int v) {     // minimum over each aligned group of 8 lanes
    v = min(v, dpp_mov<0xB1, 0xF>(v, v));   // quad_perm [1,0,3,2]
    v = min(v, dpp_mov<0x4E, 0xF>(v, v));   // quad_perm [2,3,0,1]
    v = min(v, dpp_mov<0x141, 0xF>(v, v));  // row_half_mirror
    return v;
}

__host__ __device__ inline size_t replica2_lds_ints(int C, int V, int max_tick_orders) {
    const int ids = max_tick_orders > RCNT * C ? max_tick_orders : RCNT * C;
    return (size_t)9 * C + 1 + ids + ((size_t)V + 2) / 2;
}

// rank of a 64-bit key (hi, lo) among the 16 keys of its row: number of strictly smaller ones (row_ror:1..15)
template <int N>
struct RowRank {
    static __device__ __forceinline__ void run(unsigned hi, unsigned lo, int &rank) {
        const unsigned h2 = (unsigned)dpp_mov<0x120 + N, 0xF>((int)hi, (int)hi), l2 = (unsigned)dpp_mov<0x120 + N, 0xF>((int)lo, (int)lo);
        rank += (h2 < hi || (h2 == hi && l2 < lo)) ? 1 : 0;
        RowRank<N - 1>::run(hi, lo, rank);
    }
};
template <>
struct RowRank<0> {
    static __device__ __forceinline__ void run(unsigned, unsigned, int &) {}
};

#ifndef REPL2_MIN_WAVES
#define REPL2_MIN_WAVES 1
#endif
#define DEAD 0xFFFF
#define CAPABLE (1 << 30)
#ifndef PRUNE_DELTA
#ifndef PRUNE_DELTA
#define PRUNE_DELTA 2                       // first scan pass: candidate clusters whose cost bound is within this of the smallest
#endif
#endif
#ifndef SLOTS
#define SLOTS 12                            // candidates per lane of an 8-lane group held in registers
#endif
// U8 (Static.u8_ok): cost entries are read from the byte copies of the cluster blocks and of the cost matrix - a quarter of
// the cache footprint of every gather
template <bool U8>
__global__ __launch_bounds__(REPL_THREADS, REPL2_MIN_WAVES) void k_tick_replica2(Static S, State D, int t) {
    extern __shared__ int lds_dyn[];
    const char *blk_b = U8 ? reinterpret_cast<const char *>(S.blk8) : reinterpret_cast<const char *>(S.blk);
    auto cost_at = [](const char *base, unsigned elem) -> int {
        return U8 ? (int)*reinterpret_cast<const unsigned char *>(base + elem) : *reinterpret_cast<const int *>(base + (elem << 2));
    };
    const int C = S.C;
    int *m_l = lds_dyn;                 // [C] idle vehicles still alive
    int *qcur_l = lds_dyn + C;          // [C] sorted position of the bucket's next pending order
    int *qend_l = lds_dyn + 2 * C;      // [C]
    int *dry_l = lds_dyn + 3 * C;       // [C] id of the first order that can find the cluster dry
    int *moff_l = lds_dyn + 4 * C;      // [C+1] start of the cluster's segment in mirror
    int *ev_l = lds_dyn + 5 * C + 1;    // [C] match evaluations of this tick
    int *arr_l = ev_l + C;              // [C] arrivals of this tick
    int *cdA_l = arr_l + C;             // [C] n_c | first cost column << 11 | (cluster has neighbours: can search) << 30
    int *cdB_l = cdA_l + C;             // [C] start of the cluster's cost block
    int *ids_l = cdB_l + C;             // [max(max_tick_orders, RCNT*C)] id | pickup_local << ID_BITS by sorted position;
                                        //     after the last round: [C][RCNT] counters of the resolve pass
    const int ids_n = S.max_tick_orders > RCNT * C ? S.max_tick_orders : RCNT * C;
    unsigned short *mirror = reinterpret_cast<unsigned short *>(ids_l + ids_n);       // [V] cost column of the entry's node,
                                        //     DEAD (0xFFFF) once the vehicle has been taken
    __shared__ int s_cand[REPL_WAVES][2];
    __shared__ int s_wl[REPL_WAVES][WAVE];      // per-wavefront worklist of buckets with pending orders older than LB
    const int r = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();   // wave-uniform: keep it scalar
    const int p = t & 1;
    const DayView dv = day_view(S, r);
    if (t >= dv.T) return;                       // whole workgroup: this replica's day is over
    const int now = dv.now0 + t * S.tick_minutes;
    const int *bkt_off = dv.bkt_off;
    const int tq0 = bkt_off[(size_t)t * C], tq1 = bkt_off[(size_t)(t + 1) * C];
#ifdef VDS_PROF
    const bool prof = (g_ablate & 128) != 0;
    unsigned long long tprev = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    const int pwave = (int)((blockIdx.x * REPL_WAVES + wave) & (PROF_WAVES - 1));
#endif
    for (int i = threadIdx.x; i < tq1 - tq0; i += REPL_THREADS) {
        const int4 rec = S.so_rec[tq0 + i];
        ids_l[i] = rec.x | ((rec.y & 0xFFFF) << ID_BITS);
    }
    for (int c = threadIdx.x; c < C; c += REPL_THREADS) {
        const int4 cd = S.cdesc[c];
        ev_l[c] = 0; arr_l[c] = 0;
        cdA_l[c] = cd.x | (S.cl_off[c] << 11) | (S.dfs_off[c + 1] > S.dfs_off[c] ? CAPABLE : 0);
        cdB_l[c] = U8 ? cd.z : cd.y;
    }
    __syncthreads();
#define ORDER_ID2(q, qend) ((q) < (qend) ? (ids_l[(q) - tq0] & ID_MASK) : IMAX)
    // ---- UpdateFunction (:1006-1024).  Four buckets per wavefront, one per 16-lane row: a bucket with no far
    //      entries and at most 16 arrivals this tick ranks them by dict-insertion key with 15 row rotations and
    //      appends them to its idle list; the others are handled afterwards by the whole wavefront.
    auto update_wave = [&](int c) {
        const size_t b = (size_t)c * S.R + r;
        int *hdr = D.hdr + b * HDR_WORDS;
        int hv = lane < HDR_WORDS ? hdr[lane] : 0;
        int m = rdlane(hv, HDR_IDLE);
        const int f = rdlane(hv, HDR_FL), qin = rdlane(hv, HDR_INBOX0 + p);
        int newf = f;
        if (f + qin > 0) {
            update_far(S, D, c, r, t, now, f, qin, D.fl + b * S.fl_cap, D.inbox + ((size_t)p * S.C * S.R + b) * S.in_cap, newf);
            wave_fence();
        }
        const int A = drain_ring(S, D, b, t, m, D.idle + b * S.idle_cap);
        const int q0 = bkt_off[(size_t)t * C + c], q1 = bkt_off[(size_t)t * C + c + 1];
        if (lane == 0) {
            hdr[HDR_FL] = newf; hdr[HDR_INBOX0 + p] = 0; hdr[HDR_IDLE_PRE] = m; hdr[HDR_ORDERS] = q1 - q0;
            m_l[c] = m; qcur_l[c] = q0; qend_l[c] = q1;
            if (A > 0) arr_l[c] = A;
        }
    };
    {
        const int g16 = lane >> 4, l16 = lane & 15;
        for (int c0 = 0; c0 < C; c0 += 4 * REPL_WAVES) {
            const int c = c0 + wave * 4 + g16;
            const bool valid = c < C;
            const size_t b = (size_t)(valid ? c : 0) * S.R + r;
            int *hdr = D.hdr + b * HDR_WORDS;
            const size_t si = (size_t)(t & (S.H - 1)) * S.C * S.R + b;
            int m = 0, far = 0, A = 0, q0 = 0, q1 = 0;
            if (valid) {
                m = hdr[HDR_IDLE];
                far = hdr[HDR_FL] | hdr[HDR_INBOX0 + p];
                A = D.ring_cnt[si] & 0xFFFF;
                q0 = bkt_off[(size_t)t * C + c]; q1 = bkt_off[(size_t)t * C + c + 1];
            }
            const bool slow = valid && (far != 0 || A > 16 || A > S.ring_cap);
            const bool fast = valid && !slow;
            if (ballot(fast && A > 0) != 0) {
                int4 e = make_int4(0, 0, 0, 0);
                unsigned khi = 0xFFFFFFFFu, klo = 0xFFFFFFFFu;
                const bool mine = fast && l16 < A;
                if (mine) {
                    e = D.ring[si * S.ring_cap + l16];
                    const unsigned long long key = entry_key(e.y, e.w);
                    khi = (unsigned)(key >> 32); klo = (unsigned)key;
                }
                int rank = 0;
                RowRank<15>::run(khi, klo, rank);
                if (mine) {
                    const int pos = m + rank;
                    if (pos < S.idle_cap) D.idle[b * S.idle_cap + pos] = make_uint2((unsigned)e.x, (unsigned)meta_dest(e.w));
                    else atomicOr(&D.err[0], ERR_IDLE_CAP);
                }
            }
            if (fast && l16 == 0) {
                if (A > 0) D.ring_cnt[si] = 0;
                const int mn = min(m + A, S.idle_cap);
                hdr[HDR_IDLE_PRE] = mn; hdr[HDR_ORDERS] = q1 - q0;
                m_l[c] = mn; qcur_l[c] = q0; qend_l[c] = q1;
                if (mn > m) arr_l[c] = mn - m;
            }
            for (unsigned long long rest = ballot(slow && l16 == 0); rest; rest &= rest - 1)
                update_wave(c0 + wave * 4 + ((__ffsll((long long)rest) - 1) >> 4));
        }
    }
    __syncthreads();
    PROF_STAMP(0);
    for (int c = threadIdx.x; c < C; c += REPL_THREADS)
        dry_l[c] = (cdA_l[c] & CAPABLE) ? ORDER_ID2(qcur_l[c] + m_l[c], qend_l[c]) : IMAX;
    if (wave == 0) {            // exclusive prefix of the list lengths
        int run = 0;
        for (int base = 0; base < C; base += WAVE) {
            const int c = base + lane;
            const int v = c < C ? m_l[c] : 0;
            int inc = v;
            for (int o = 1; o < WAVE; o <<= 1) { const int u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
            if (c < C) moff_l[c] = run + inc - v;
            run += rdlane(inc, WAVE - 1);
        }
        if (lane == 0) moff_l[C] = run;
    }
    __syncthreads();
    for (int c = wave; c < C; c += 4 * REPL_WAVES) {       // four buckets' list loads in flight per wavefront
        int m4[4], mo4[4], clo4[4];
        unsigned loc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cu = c + u * REPL_WAVES;
            m4[u] = cu < C ? m_l[cu] : 0;
            mo4[u] = cu < C ? moff_l[cu] : 0;
            clo4[u] = cu < C ? (cdA_l[cu] >> 11) & 0xFFFF : 0;
            loc4[u] = 0;
            if (lane < m4[u]) loc4[u] = D.idle[((size_t)cu * S.R + r) * S.idle_cap + lane].y;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (lane < m4[u]) mirror[mo4[u] + lane] = (unsigned short)(clo4[u] + (int)loc4[u]);
            if (m4[u] > WAVE) {
                const uint2 *idle = D.idle + ((size_t)(c + u * REPL_WAVES) * S.R + r) * S.idle_cap;
                for (int i = WAVE + lane; i < m4[u]; i += WAVE) mirror[mo4[u] + i] = (unsigned short)(clo4[u] + (int)idle[i].y);
            }
        }
    }
    __syncthreads();
    PROF_STAMP(1);
    int2 *out_r = D.out + (size_t)r * S.Oq - dv.q_base;

    // own-cluster match of bucket c by ONE thread (rare path: order LB found its cluster not dry after all)
    auto own_match_thread = [&](int c, int limit) {
        int qc = qcur_l[c];
        const int qe = qend_l[c];
        if (qc >= qe) return;
        int idw = ids_l[qc - tq0];
        if ((idw & ID_MASK) >= limit) return;
        const int nc = cdA_l[c] & 2047;
        const int boff = cdB_l[c] - ((cdA_l[c] >> 11) & 0xFFFF);
        const int mo = moff_l[c], m0 = moff_l[c + 1] - mo;
        int m = m_l[c];
        int evals = 0;
        do {
            evals += m;
            int best = IMAX, bpos = -1;
            if (m > 0) {
                const int rowoff = boff + (idw >> ID_BITS) * nc;
                for (int i = 0; i < m0; ++i) {
                    const int col = mirror[mo + i];
                    if (col == DEAD) continue;
                    const int cst = cost_at(blk_b, (unsigned)(rowoff + col));
                    if (cst < best) { best = cst; bpos = i; }
                }
            }
            int2 res = make_int2(-1, -1);
            if (bpos >= 0 && (long long)best <= S.reject_threshold) {      // :943 (quirk Q3)
                mirror[mo + bpos] = DEAD;
                m--;
                res = make_int2((int)(((unsigned)c << 16) | (unsigned)bpos), best);
            }
            out_r[qc] = res;
            qc++;
            idw = qc < qe ? ids_l[qc - tq0] : IMAX;
        } while (qc < qe && (idw & ID_MASK) < limit);
        m_l[c] = m; qcur_l[c] = qc;
        dry_l[c] = (cdA_l[c] & CAPABLE) ? ORDER_ID2(qc + m, qe) : IMAX;
        if (evals) ev_l[c] += evals;
    };

    const int thr32 = S.reject_threshold > 0x7FFF ? 0x7FFF : (S.reject_threshold < 0 ? -1 : (int)S.reject_threshold);   // costs < 2^15 here
    const int gl = lane & (GRP - 1);              // lane within its group
    const int gw = lane / GRP;                    // group within the wavefront
    const int CW = (C + REPL_WAVES - 1) / REPL_WAVES;     // buckets owned by one wavefront
    const int cw0 = wave * CW, cw1 = min(C, cw0 + CW);
    // ---- MatchFunction in lower-bound rounds
    for (;;) {
        // (A) LB = oldest order that can find its own cluster dry, pc = its cluster (every wavefront computes both)
        int LB, pc;
        {
            int lv = IMAX, lcl = 0;
            for (int c = lane; c < C; c += WAVE) {
                const int v = dry_l[c];
                if (v < lv) { lv = v; lcl = c; }
            }
            LB = wave_min_i32(lv);
            const unsigned long long who = ballot(lv == LB);
            pc = rdlane(lcl, __ffsll((long long)who) - 1);     // order ids are unique: exactly one bucket
        }
        // prefetch for the neighbour search of order LB (position known now: the (m+1)-th pending order of pc)
        int pnode = 0, cj = 0;
        int s0 = 0, s1 = 0;
        if (LB != IMAX) {
            s0 = S.dfs_off[pc]; s1 = S.dfs_off[pc + 1];
            pnode = S.so_pnode[qcur_l[pc] + m_l[pc]];
            const int sx = s0 + wave + lane * REPL_WAVES;      // this wavefront's candidate clusters, lane j = j-th
            if (sx < s1) cj = S.dfs_seq[sx];
        }
        PROF_STAMP(2);
        // (B) all pending orders older than LB: ordinary own-cluster matches (:924-965).  Each wavefront lists the
        //     buckets of its range that have such orders; its eight 8-lane groups take one bucket each.
        for (int cb = cw0; cb < cw1; cb += WAVE) {
            const int cmine = cb + lane;
            bool has = false;
            if (cmine < cw1) {
                const int qc = qcur_l[cmine];
                has = qc < qend_l[cmine] && (ids_l[qc - tq0] & ID_MASK) < LB;
            }
            const unsigned long long hm = ballot(has);
            if (hm == 0) continue;
            if (has) s_wl[wave][popc64(hm & lanemask_lt())] = cmine;
            wave_fence();
            const int nwork = popc64(hm);
            for (int w0 = 0; w0 < nwork; w0 += GRPS_WAVE) {
                bool act = w0 + gw < nwork;
                const int c = act ? s_wl[wave][w0 + gw] : 0;
                int qc = qcur_l[c];
                const int qe = qend_l[c];
                const int cda = cdA_l[c];
                const int nc = cda & 2047;
                // cost of (pickup p, idle entry) = S.blk[boff + p * nc + column]: 32-bit offsets from one uniform base
                const int boff = cdB_l[c] - ((cda >> 11) & 0xFFFF);
                const int mo = moff_l[c], m0 = act ? moff_l[c + 1] - mo : 0;
                int m = m_l[c], evals = 0;
                // longest list among this wavefront's buckets of this step: bounds the (uniform) slot loops
                const bool longlist = m0 > SLOTS * GRP;         // handled after the register-resident lists
                const bool act_all = act;
                act = act && !longlist;
                int mx = longlist ? 0 : m0;
                mx = max(mx, dpp_mov<0x4E, 0xF>(mx, mx)); mx = max(mx, dpp_mov<0x141, 0xF>(mx, mx)); mx = max(mx, dpp_mov<0x140, 0xF>(mx, mx));
                const int mmaxw = max(max(rdlane(mx, 0), rdlane(mx, 16)), max(rdlane(mx, 32), rdlane(mx, 48)));
                if (ballot(act) != 0) {
                    // the whole list sits in this group's registers: up to SLOTS candidates per lane
                    int col[SLOTS];
                    unsigned amask = 0;
#pragma unroll
                    for (int u = 0; u < SLOTS; ++u) {
                        col[u] = DEAD;
                        if (u * GRP < mmaxw) {
                            const int i = u * GRP + gl;
                            const int v = mirror[mo + min(i, max(m0, 1) - 1)];
                            col[u] = i < m0 ? v : DEAD;
                            amask |= (col[u] != DEAD ? 1u : 0u) << u;
                        }
                    }
#ifdef R2DIAG
                    PROF_STAMP_NW(3);
#endif
                    while (ballot(act) != 0) {
                        // up to OB pending orders of the bucket: their cost gathers are in flight together
                        int idw[OB], cst[OB][SLOTS];
#pragma unroll
                        for (int o = 0; o < OB; ++o) idw[o] = (act && qc + o < qe) ? ids_l[qc + o - tq0] : IMAX;
#pragma unroll
                        for (int o = 0; o < OB; ++o) {
                            const bool oo = (idw[o] & ID_MASK) < LB && idw[o] != IMAX;
                            const bool any = ballot(oo) != 0;       // usually one pending order: one row of gathers
                            const int rowoff = boff + (oo ? (idw[o] >> ID_BITS) : 0) * nc;
                            const unsigned take = oo ? amask : 0u;
#pragma unroll
                            for (int u = 0; u < SLOTS; ++u) {
                                cst[o][u] = 0;
                                if (any && u * GRP < mmaxw) {
                                    cst[o][u] = cost_at(blk_b, (unsigned)(((take >> u) & 1u) ? rowoff + col[u] : 0));
                                }
                            }
                        }
#pragma unroll
                        for (int o = 0; o < OB; ++o) {
                            const bool oo = act && idw[o] != IMAX && (idw[o] & ID_MASK) < LB;
                            if (ballot(oo) == 0) break;
                            int key = IMAX;
                            const unsigned take = oo ? amask : 0u;
#pragma unroll
                            for (int u = 0; u < SLOTS; ++u)
                                if (u * GRP < mmaxw) key = min(key, ((take >> u) & 1u) ? (cst[o][u] << 16) | (u * GRP + gl) : IMAX);
                            key = grp_min_i32(key);
                            if (oo) {
                                evals += m;
                                int2 res = make_int2(-1, -1);
                                if (key != IMAX && (key >> 16) <= thr32) {     // :943 (quirk Q3)
                                    const int pos = key & 0xFFFF;
                                    if ((pos & (GRP - 1)) == gl) amask &= ~(1u << (pos / GRP));
                                    if (gl == 0) mirror[mo + pos] = DEAD;
                                    m--;
                                    res = make_int2((int)(((unsigned)c << 16) | (unsigned)pos), key >> 16);
                                }
                                if (gl == 0) out_r[qc] = res;
                                qc++;
                            } else {
                                act = false;         // orders of a bucket are sorted by id: nothing older than LB is left
                            }
                        }
                        if (act) act = qc < qe && (ids_l[qc - tq0] & ID_MASK) < LB;
                    }
#ifdef R2DIAG
                    PROF_STAMP_NW(1);
#endif
                }
                act = act_all && longlist;
                if (ballot(act) != 0) {
                    // long lists: one order at a time, 64 candidates per pass
                    int idw = act ? ids_l[qc - tq0] : IMAX;
                    while (ballot(act) != 0) {
                        int key = IMAX;
                        if (act && m > 0) {
                            const int rowoff = boff + (idw >> ID_BITS) * nc;
                            for (int i0 = 0; i0 < m0; i0 += 8 * GRP) {
                                int cl[8], cst[8], ok[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int i = i0 + u * GRP + gl;
                                    cl[u] = mirror[mo + min(i, m0 - 1)];
                                    ok[u] = (i < m0 ? 1 : 0) & (cl[u] != DEAD ? 1 : 0);
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u)
                                    cst[u] = cost_at(blk_b, (unsigned)(ok[u] ? rowoff + cl[u] : 0));
#pragma unroll
                                for (int u = 0; u < 8; ++u)
                                    key = min(key, ok[u] ? (cst[u] << 16) | (i0 + u * GRP + gl) : IMAX);
                            }
                        }
                        key = grp_min_i32(key);
                        if (act) {
                            evals += m;
                            int2 res = make_int2(-1, -1);
                            if (key != IMAX && (long long)(key >> 16) <= S.reject_threshold) {     // :943 (quirk Q3)
                                if (gl == 0) mirror[mo + (key & 0xFFFF)] = DEAD;
                                m--;
                                res = make_int2((int)(((unsigned)c << 16) | (unsigned)(key & 0xFFFF)), key >> 16);
                            }
                            if (gl == 0) out_r[qc] = res;
                            qc++;
                            idw = qc < qe ? ids_l[qc - tq0] : IMAX;
                            act = qc < qe && (idw & ID_MASK) < LB;
                        }
                    }
                }
                if (w0 + gw < nwork && gl == 0) {
                    m_l[c] = m; qcur_l[c] = qc;
                    dry_l[c] = (cda & CAPABLE) ? ORDER_ID2(qc + m, qe) : IMAX;
                    if (evals) ev_l[c] += evals;
                }
            }
            wave_fence();
        }
        // lower bound of the cost from any node of candidate cluster cj to the dry order's pickup node (byte costs only):
        // lets the scan skip clusters that cannot beat the best vehicle found in the most promising ones
        int lbj = 0;
        if (U8 && S.lbc != nullptr && LB != IMAX && s0 + wave + lane * REPL_WAVES < s1) lbj = (int)S.lbc[(size_t)pnode * C + cj];
        __syncthreads();
        PROF_STAMP(3);
        if (LB == IMAX) break;
        if (m_l[pc] > 0) {
            // not dry after all (an older order of this bucket was rejected by the pickup window, :943, without
            // taking a vehicle): order LB is an ordinary own-cluster match
            if (threadIdx.x == 0) own_match_thread(pc, LB + 1);
            __syncthreads();
            continue;
        }
        // (C) FindServerVehicleFunction for order LB: every wavefront scans its share of the visit sequence.
        //     key = (cost << 16 | visit position, list position << 16 | cluster): lexicographic minimum == the
        //     reference's first strict minimum in visit order, then list order
        const int q = qcur_l[pc];
        const char *crow_b = U8 ? reinterpret_cast<const char *>(S.cost8 + (size_t)pnode * S.N) : reinterpret_cast<const char *>(S.cost + (size_t)pnode * S.N);
        int bhi = IMAX, blo = IMAX;
        for (int jb = 0; s0 + wave + jb * REPL_WAVES < s1; jb += WAVE) {
            if (jb > 0) {                               // visit sequences longer than 64 clusters per wavefront
                const int sx = s0 + wave + (jb + lane) * REPL_WAVES;
                cj = sx < s1 ? S.dfs_seq[sx] : 0;
                lbj = (U8 && S.lbc != nullptr && sx < s1) ? (int)S.lbc[(size_t)pnode * C + cj] : 0;
            }
            const int nj = min(WAVE, (s1 - s0 - wave - jb * REPL_WAVES + REPL_WAVES - 1) / REPL_WAVES);
            int mj = 0, moj = 0, m0j = 0;
            if (lane < nj) {
                mj = m_l[cj];
                moj = moff_l[cj];
                m0j = mj > 0 ? moff_l[cj + 1] - moj : 0;
            }
            {   // :986-991 runs for every visited cluster: its live vehicles count as evaluations of order LB
                const int rs = row_sum_i32(mj);
                const int tot = rdlane(rs, 0) + rdlane(rs, 16) + rdlane(rs, 32) + rdlane(rs, 48);
                if (lane == 0 && tot) atomicAdd(&ev_l[pc], tot);
            }
            // slots = (cluster j of this wavefront, 64-entry chunk b of its list), walked in (j, b) order, eight cost
            // gathers in flight per lane.  key = cost << 16 | j << 9 | b: with the lane as the last tie-break this
            // is (cost, visit position, list position)
            // Two passes over the candidate clusters: first those whose cost bound is within PRUNE_DELTA of the smallest
            // bound - the nearest vehicle is almost always there -, then only the clusters whose bound does not exceed the
            // best cost found (<=: an equal cost in an earlier visit position still wins).  A skipped cluster holds no
            // vehicle that could be the first strict minimum, so the result is unchanged; evaluations were counted above.
            const unsigned long long cand = ballot(m0j > 0);
            const int lbmin = wave_min_i32(m0j > 0 ? lbj : IMAX);
            unsigned long long live = ballot(m0j > 0 && lbj <= lbmin + PRUNE_DELTA);
            const unsigned long long first_pass = live;
            int b = 0, best = IMAX;
            for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                const int bc = wave_min_i32(best) >> 16;            // 32767 when nothing was found
                live = cand & ~first_pass & ballot(lbj <= bc);
                b = 0;
            }
            while (live != 0) {
                // branch-free in three passes, so that the eight LDS reads and then the eight gathers are issued
                // back to back instead of one dependent chain per slot (integer flags and byte offsets on purpose:
                // bool && and 64-bit indexing make the compiler fall back to exec-mask branches)
                int cl[8], cst[8], seq[8], in[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    in[k] = 0; cl[k] = DEAD; seq[k] = 0;
                    if (live != 0) {
                        const int j = __ffsll((long long)live) - 1;
                        const int m0c = rdlane(m0j, j), moc = rdlane(moj, j);
                        const int i = b * WAVE + lane;
                        in[k] = i < m0c ? 1 : 0;
                        cl[k] = mirror[moc + min(i, m0c - 1)];
                        seq[k] = (j << 9) | b;
                        ++b;
                        if (b * WAVE >= m0c) { b = 0; live &= live - 1; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    in[k] &= cl[k] != DEAD ? 1 : 0;
                    cst[k] = cost_at(crow_b, (unsigned)(in[k] ? cl[k] : 0));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    best = min(best, in[k] ? (cst[k] << 16) | seq[k] : IMAX);
            }
            }
            // this batch's winner in global terms
            const int wbest = wave_min_i32(best);
            if (wbest != IMAX) {
                const int wl = __ffsll((long long)ballot(best == wbest)) - 1;       // lowest lane = lowest list position
                const int j = (wbest >> 9) & 63, bb = wbest & 511;
                const int hi = (wbest & ~0xFFFF) | ((jb + j) * REPL_WAVES + wave);
                const int lo = ((bb * WAVE + wl) << 16) | rdlane(cj, j);
                if (hi < bhi || (hi == bhi && lo < blo)) { bhi = hi; blo = lo; }
            }
        }
        if (lane == 0) { s_cand[wave][0] = bhi; s_cand[wave][1] = blo; }
        PROF_STAMP(4);
        __syncthreads();
        if (threadIdx.x == 0) {
            int whi = s_cand[0][0], wlo = s_cand[0][1];
            for (int w = 1; w < REPL_WAVES; ++w) {
                const int h2 = s_cand[w][0], l2 = s_cand[w][1];
                if (h2 < whi || (h2 == whi && l2 < wlo)) { whi = h2; wlo = l2; }
            }
            int2 res = make_int2(-1, -1);
            const int wc = whi >> 16;
            if (whi != IMAX && (long long)wc <= S.reject_threshold) {
                const int wcl = wlo & 0xFFFF, wpos = wlo >> 16;
                mirror[moff_l[wcl] + wpos] = DEAD;
                const int mw = m_l[wcl] - 1;
                m_l[wcl] = mw;
                if (cdA_l[wcl] & CAPABLE) dry_l[wcl] = ORDER_ID2(qcur_l[wcl] + mw, qend_l[wcl]);
                res = make_int2((int)(((unsigned)wcl << 16) | (unsigned)wpos), wc);
            }
            out_r[q] = res;
            qcur_l[pc] = q + 1;
            dry_l[pc] = ORDER_ID2(q + 1, qend_l[pc]);     // m == 0: the very next order is dry again
        }
        __syncthreads();
        PROF_STAMP(5);
#ifdef VDS_PROF
        if (prof && lane == 0) g_prof[(size_t)pwave * PROF_SLOTS + 6] += 1;
#endif
    }
    // ---- resolve the preliminary results: vehicle ids, arrivals (:954-960), counters (the id table is dead now)
    int *rc_l = ids_l;
    for (int i = threadIdx.x; i < RCNT * C; i += REPL_THREADS) rc_l[i] = 0;
    __syncthreads();
    for (int qq = tq0 + (int)threadIdx.x; qq < tq1; qq += 3 * REPL_THREADS) {
        // three orders per thread at a time: their records, then their vehicle-id gathers, then their posts travel together
        int4 rec[3];
        int2 pr[3];
        int veh[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = qq + u * REPL_THREADS;
            rec[u] = make_int4(0, 0, 0, 0); pr[u] = make_int2(-1, -1);
            if (q < tq1) { rec[u] = S.so_rec[q]; pr[u] = out_r[q]; }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            veh[u] = -1;
            if (pr[u].x != -1) {
                const int vc = (int)((unsigned)pr[u].x >> 16), vpos = pr[u].x & 0xFFFF;
                veh[u] = (int)D.idle[((size_t)vc * S.R + r) * S.idle_cap + vpos].x;
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = qq + u * REPL_THREADS;
            if (q >= tq1) continue;
            int *cl = rc_l + (int)((unsigned)rec[u].z >> 16) * RCNT;
            atomicAdd(&cl[CNT_ORDERS], 1);
            if (pr[u].x == -1) {
                atomicAdd(&cl[CNT_REJECTS], 1);
            } else {
                out_r[q] = make_int2(veh[u], pr[u].y);
                post_arrival(S, D, rec[u].z & 0xFFFF, r, t, now, veh[u], rec[u].x, now + pr[u].y + rec[u].w, 0, (int)((unsigned)rec[u].y >> 16));
                atomicAdd(&cl[CNT_WAIT], pr[u].y);
                atomicAdd(&cl[CNT_VALUE], rec[u].w);
            }
        }
    }
    __syncthreads();
    // ---- IdleVehicles.remove (:963), once per bucket: order-preserving compaction (survivors = mirror entries not DEAD)
    for (int c = wave; c < C; c += 4 * REPL_WAVES) {
        // lists of at most 64 entries: four buckets' loads in flight; longer ones chunk by chunk below
        uint2 e4[4];
        bool keep4[4], small4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cu = c + u * REPL_WAVES;
            const int mo = cu < C ? moff_l[cu] : 0, m0 = cu < C ? moff_l[cu + 1] - mo : 0;
            small4[u] = cu < C && m0 <= WAVE && m_l[cu] != m0;
            keep4[u] = small4[u] && lane < m0 && mirror[mo + lane] != DEAD;
            e4[u] = make_uint2(0u, 0u);
            if (keep4[u]) e4[u] = D.idle[((size_t)cu * S.R + r) * S.idle_cap + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long kb = ballot(keep4[u]);
            if (keep4[u]) D.idle[((size_t)(c + u * REPL_WAVES) * S.R + r) * S.idle_cap + popc64(kb & lanemask_lt())] = e4[u];
        }
    }
    for (int c = wave; c < C; c += REPL_WAVES) {
        const int mo = moff_l[c], m0 = moff_l[c + 1] - mo;
        if (m_l[c] == m0 || m0 <= WAVE) continue;
        uint2 *idle = D.idle + ((size_t)c * S.R + r) * S.idle_cap;
        int kept = 0;
        for (int base = 0; base < m0; base += WAVE) {
            const int i = base + lane;
            uint2 e = make_uint2(0u, 0u);
            bool keep = false;
            if (i < m0) {
                keep = mirror[mo + i] != DEAD;
                if (keep) e = idle[i];
            }
            const unsigned long long kb = ballot(keep);
            wave_fence();        // this chunk's sources are read before any of its (lower or equal) targets is written
            if (keep) idle[kept + popc64(kb & lanemask_lt())] = e;
            kept += popc64(kb);
        }
    }
    PROF_STAMP(7);
    // ---- flush: idle counts and this tick's counter deltas
    for (int c = threadIdx.x; c < C; c += REPL_THREADS) {
        const size_t b = (size_t)c * S.R + r;
        D.hdr[b * HDR_WORDS + HDR_IDLE] = m_l[c];
        long long *cnt = D.cnt + b * CNT_WORDS;
#pragma unroll
        for (int w = 0; w < RCNT; ++w) { const int d = rc_l[c * RCNT + w]; if (d) cnt[w] += d; }
        if (ev_l[c]) cnt[CNT_EVALS] += ev_l[c];
        if (arr_l[c]) cnt[CNT_ARRIVALS] += arr_l[c];
    }
#undef ORDER_ID2
}

// ---------------------------------------------------------------------------------------
// k_dfs_walk: second half of the HYBRID neighbour-search tick.  The first half is k_tick_rows in stamp mode:
// cluster-major, four replicas per wavefront, cost block in LDS - UpdateFunction and the own-cluster matching (:924-933) of
// every bucket up to the order that finds the list exhausted, nothing committed, every taken entry stamped with the rank of
// its order.  This kernel, one 256-thread workgroup per replica, then
//   1. loads the stamps of the replica's ~V idle entries into LDS (u16 each: 0xFFFF free, else the rank that took it);
//   2. walks the DRY orders (own cluster exhausted, :936) in id order with ONE wavefront: the visit sequence of
//      FindServerVehicleFunction (:978-996), lane j = j-th visited cluster; the vehicles alive at the order's time are the
//      entries with stamp > rank; evaluations = sum over the visited clusters of the alive count, derived without a scan
//      (list length - own matches before the order - steals so far; own matches are a prefix of the bucket's orders);
//      candidate clusters are pruned by the cost lower bound (Static.lbc); the winner (first strict minimum in visit, then
//      list order) is stolen: stamp = the dry order's rank.  If it had been taken later by an own-cluster order, that
//      bucket's matching is redone from its first order after this one (rare), which may add a dry order with a later rank;
//   3. commits the slot: evaluations of the own-cluster orders from the final stamps, vehicle ids and arrival posts
//      (:954-960) for every order, counters, order-preserving compaction of the lists (:963), headers.
// Preconditions (vds_api dfs_hybrid_ok): the fast kernel's (fast_ok: costs < 2^23 and never above the pickup window, blocks fit
// LDS), one order day per workgroup chunk, < 65535 orders per slot, V < 65536, C <= 2047 nodes per cluster, LDS footprint.
#ifndef WK_THREADS
#define WK_THREADS 256
#endif
#define WK_WAVES (WK_THREADS / WAVE)
#ifndef WK_MIN_WAVES
#define WK_MIN_WAVES 1                      // wavefronts per SIMD the kernel is compiled for (register cap)
#endif
#ifndef WK_PRIO
#define WK_PRIO 0                           // s_setprio of wavefront 0 while it serves the chain of dry orders
#endif
#define WK_FREE 0xFFFFu
#ifndef WK_K
#define WK_K 3
#endif
//                                             candidates kept per scanned dry order (2-4 measure the same; 8 costs 2 % in the extraction loop)
#define WK_NS 32                            // pool of scan records (the second order of a paired scan may wait there for a while):
#define WK_NS_MIN 8                         // Static.walk_pool of them, as many as keep the workgroup within a quarter of a CU's LDS
#define WK_REC (4 * WK_K + 6)               // one scan record (ints): WK_K candidates {cost << 16 | visit index << 8 | 64-entry chunk of the
                                            // list, index of the entry's stamp, its cluster | orders of that cluster before the dry order << 16, list position},
                                            // the number of candidates; and for the first candidate, if an own-cluster order a holds it: a, the
                                            // two entries {cost << 16 | position} a would pick instead, whether those are all it could pick; pad
#ifndef WK_EVA
#define WK_EVA 8                            // visit rows in flight per wavefront in the evaluation pass (a)
#endif
#ifndef WK_REDO_PRE
#define WK_REDO_PRE 1                       // the scan also works out what the holder of its first candidate would pick instead
#endif
#ifndef WK_G
#define WK_G 2                              // dry orders of one bucket that share a scan (consecutive sorted positions), at most
#endif
static_assert(WK_G >= 1 && WK_G <= 3, "dfs_scan is instantiated for 1, 2 and 3 orders per scan");
#ifndef WK_PAIR_SPAN
#define WK_PAIR_SPAN 65536                  // a paired scan takes the bucket's next order only if its rank is at most this far ahead
#endif
#ifndef WK_RES
#define WK_RES 10                           // orders a thread resolves at a time (their loads in flight together)
#endif
#ifndef WK_SLACK
#define WK_SLACK 1                          // second scan pass: clusters whose cost bound is within this of the best cost found
#endif
#ifdef WKDEBUG
#define WKCHK(cond, code, a, b2) do { if (!(cond)) { printf("k_dfs_walk check %d failed: r %d t %d lane %d  %d %d\n", code, (int)blockIdx.x, t, (int)threadIdx.x, (int)(a), (int)(b2)); return; } } while (0)
#else
#define WKCHK(cond, code, a, b2) do { } while (0)
#endif

__host__ __device__ inline size_t dfs_walk_lds_bytes(int C, int V, int mto, int ns = WK_NS, int dn = 0) {
    // (dense layout: the rank tables are read from L2, their LDS holds the entries' node bytes: RankTab)
    const size_t ids = dn ? 0 : (size_t)(mto + 2 > RCNT * C ? mto + 2 : RCNT * C);     // two u16 rank tables, later the resolve counters
    const size_t words = (size_t)(mto + 31) / 32 + 1;
    return ((size_t)8 * C + 1 + ids + (dn ? 3 : 2) * words + (1 + ns) * WK_REC + 4 * WAVE + ((size_t)V + 1) / 2 + (dn ? ((size_t)V + 3) / 4 : 0)) * sizeof(int);
}

template <bool U8>
__device__ __forceinline__ int cost_elem(const char *base, unsigned elem) {
    return U8 ? (int)*reinterpret_cast<const unsigned char *>(base + elem) : *reinterpret_cast<const int *>(base + (elem << 2));
}
template <int JB>
__device__ __forceinline__ unsigned long long pick_mask(const unsigned long long (&m)[JB], int jb) {
    unsigned long long v = m[0];
#pragma unroll
    for (int x = 1; x < JB; ++x) v = jb == x ? m[x] : v;
    return v;
}
template <int JB>
__device__ __forceinline__ int pick_lane(const int (&a)[JB], int jb, int l) {
    int v = rdlane(a[0], l);
#pragma unroll
    for (int x = 1; x < JB; ++x) { const int u = rdlane(a[x], l); v = jb == x ? u : v; }
    return v;
}
__device__ __forceinline__ int lds_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// flags between the wavefronts of a workgroup, all in LDS: acquire / release at workgroup scope on the LDS address space only
// (a flag store after the data stores, a flag load before the data loads).  An unrestricted workgroup-scope release / acquire
// would also wait for the wavefront's outstanding HBM stores - a full round trip per served order.
__device__ __forceinline__ int lds_acquire(const int *p) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return v;
}
__device__ __forceinline__ void lds_release(int *p, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool lds_cas(int *p, int expect, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    const bool ok = __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return ok;
}

// dfs_scan: what one dry order (sorted position q, rank rho, pickup node pnode) would be served with, by ONE wavefront: the
// visit sequence of FindServerVehicleFunction (:978-996) comes precomputed per order (Static.so_vis: j-th visited cluster |
// orders of that cluster before this one << 16; Static.so_lb: the cost lower bound of that cluster), lane (j & 63) of batch
// (j >> 6) = j-th visited cluster; candidate clusters (alive count > 0, derived from the list length, the own matches before
// the order and the steals so far: ls_l = own matches << 16 | steals, one word so that a reader sees a consistent pair) pruned
// by the bound in two passes; eight (cluster, 64-entry chunk) gathers in flight; the WK_K best candidates in the exact order of
// the reference (cost, visit order, list position: first strict minimum), as far as they are KNOWN to be the best (each lane
// keeps its two smallest keys; clusters never scanned cost at least their bound).  The first is always exact.
// The scan may run WHILE the walk serves earlier dry orders: stamps only ever decrease (a steal or a re-pick of a redo chain
// lowers the stamp of the entry it takes) and ls_l is published once per served order, so whatever mixture of states the scan
// reads, the vehicles it considers are a superset of those alive when the order's turn comes, and the list a sorted prefix of
// that superset: its first entry still alive at that time IS the winner (none kept: nothing was alive, the order is rejected).
// DN: the idle lists are the DENSE layout's packed entries {veh << 8 | loc_local} (neighbour search on the dense layout, round 6)
// The slot's rank tables - rank of a sorted position, sorted position of a rank.  Wide layout: u16 copies in LDS (the walk's serving
// wavefront reads them on its critical path).  Dense layout (DN): read where they are - they are static and the same for every replica
// (L2) - and the LDS they took holds the node bytes of the replica's idle entries instead (the scans lose an HBM level each).
template <bool DN>
struct RankTab {
    const unsigned short *rq_l, *qr_l;      // LDS copies (!DN)
    const int *rank_g, *q_g;                // Static.so_rank + tq0, Static.ord_q + tick_off[t] (DN)
    int tq0;
    __device__ __forceinline__ int qr(int rk) const { return DN ? q_g[rk] - tq0 : (int)qr_l[rk]; }      // sorted position (relative to tq0) of a rank
    __device__ __forceinline__ int rq(int i) const { return DN ? rank_g[i] : (int)rq_l[i]; }           // rank of a sorted position (relative)
};
template <bool DN>
__device__ __forceinline__ unsigned idle_loc(const State &D, size_t elem) {
    if (DN) return reinterpret_cast<const unsigned *>(D.idle)[elem] & 0xFFu;
    return D.idle[elem].y & 0xFFFFu;
}
template <bool U8, int JB, int G = 1, bool PRE = (WK_REDO_PRE != 0), bool DN = false>
__device__ __forceinline__ void dfs_scan(const Static &S, const State &D, int r, int q, const int (&rho)[G], const int (&pnode)[G],
                                         const int *m0_l, const int *moff_l, const int *ls_l, const int *cda_l,
                                         const unsigned short *st_l, const RankTab<DN> &RT, const unsigned char *loc_l, int tq0, unsigned *const (&rec)[G], unsigned long long *pacc = nullptr) {
    // G = 2: the dry orders at sorted positions q and q + 1 of ONE bucket (same visit sequence; rho[1] > rho[0], so whatever is
    // alive for the second is alive for the first) share the walk over the candidate lists: one set of stamp reads and node-word
    // loads, a cost gather and a pair of smallest keys per order.  A cluster scanned for either order counts as scanned for both.
    const int lane = lane_id();
#ifdef VDS_PROF
    unsigned long long sc_ts = pacc ? __builtin_amdgcn_s_memtime() : 0ull;
#define SCT(i) do { if (pacc) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - sc_ts; sc_ts = t_; } } while (0)
#else
#define SCT(i) do { } while (0)
#endif
    const char *crow_b[G];
    unsigned ck[G][JB];
    int lbj[G][JB], m0j[JB], moj[JB], cofj[JB];
    unsigned long long cand[G][JB], scanned[JB], live[JB];
#pragma unroll
    for (int o = 0; o < G; ++o) {
        crow_b[o] = U8 ? reinterpret_cast<const char *>(S.cost8 + (size_t)pnode[o] * S.N) : reinterpret_cast<const char *>(S.cost + (size_t)pnode[o] * S.N);
        const unsigned *vis = S.so_vis + (size_t)(q + o) * S.seq_pad;
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) {
            ck[o][jb] = vis[jb * WAVE + lane];
            lbj[o][jb] = (U8 && S.so_lb != nullptr) ? (int)S.so_lb[(size_t)(q + o) * S.seq_pad + jb * WAVE + lane] : 0;
        }
    }
    SCT(0);
    int lbmin[G];
#pragma unroll
    for (int o = 0; o < G; ++o) lbmin[o] = IMAX;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        int alive[G];
#pragma unroll
        for (int o = 0; o < G; ++o) alive[o] = 0;
        m0j[jb] = 0; moj[jb] = 0; cofj[jb] = 0;
        if (ck[0][jb] != 0xFFFFFFFFu) {
            const int c = (int)(ck[0][jb] & 0xFFFFu);
            const int ls = lds_load(&ls_l[c]);
            m0j[jb] = m0_l[c]; moj[jb] = moff_l[c]; cofj[jb] = (cda_l[c] >> 11) & 0xFFFF;
#pragma unroll
            for (int o = 0; o < G; ++o) alive[o] = m0j[jb] - min((int)(ck[o][jb] >> 16), ls >> 16) - (ls & 0xFFFF);
        }
        scanned[jb] = 0ull; live[jb] = 0ull;
#pragma unroll
        for (int o = 0; o < G; ++o) {
            cand[o][jb] = ballot(alive[o] > 0);
            lbmin[o] = min(lbmin[o], alive[o] > 0 ? lbj[o][jb] : IMAX);
        }
    }
#pragma unroll
    for (int o = 0; o < G; ++o) lbmin[o] = wave_min_i32(lbmin[o]);
    SCT(1);
    int b1[G], b2[G], nval[G];
#pragma unroll
    for (int o = 0; o < G; ++o) { b1[o] = IMAX; b2[o] = IMAX; nval[o] = 0; }
    for (int pass = 0; pass < 2; ++pass) {
        int bound[G];
#pragma unroll
        for (int o = 0; o < G; ++o) bound[o] = pass == 0 ? lbmin[o] + PRUNE_DELTA : (wave_min_i32(b1[o]) >> 16) + WK_SLACK;
        bool anylive = false;
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) {
            unsigned long long lv = 0ull;
#pragma unroll
            for (int o = 0; o < G; ++o) lv |= cand[o][jb] & ballot(lbj[o][jb] <= bound[o]);
            live[jb] = lv & ~scanned[jb];
            scanned[jb] |= live[jb];
            anylive |= live[jb] != 0ull;
        }
        int jbc = -1, b = 0;
        unsigned long long cl = 0ull;
        while (anylive) {
            // eight (cluster, 64-entry chunk) slots at a time: their stamps (LDS), the node words of the entries that look
            // alive (HBM), then the cost gathers, then the two smallest keys of the lane.  (Fetching the cluster's segment of
            // the cost row together with the node words and shuffling the cost out of it was measured: the extra loads make
            // every load slower, 19.6 k instead of 15.6 k cycles per scan.)
            int in[G][8], seq[8], sidx[8];
            unsigned yv[8];
            int cof[8];
            size_t ip[8];
            bool any = false;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                seq[k8] = 0; yv[k8] = 0u; cof[k8] = 0; sidx[k8] = 0; ip[k8] = 0;
#pragma unroll
                for (int o = 0; o < G; ++o) in[o][k8] = 0;
                while (cl == 0ull && jbc + 1 < JB) { ++jbc; cl = pick_mask<JB>(live, jbc); b = 0; }
                if (cl != 0ull) {
                    any = true;
                    const int j = __ffsll((long long)cl) - 1;
                    const int cc = (int)((unsigned)pick_lane<JB>(reinterpret_cast<const int (&)[JB]>(ck[0]), jbc, j) & 0xFFFFu);
                    const int m0c = pick_lane<JB>(m0j, jbc, j), moc = pick_lane<JB>(moj, jbc, j);
                    const int i = b * WAVE + lane;
#pragma unroll
                    for (int o = 0; o < G; ++o) in[o][k8] = i < m0c ? 1 : 0;
                    sidx[k8] = moc + min(i, m0c - 1);
                    ip[k8] = ((size_t)cc * S.R + r) * S.idle_cap + i;
                    cof[k8] = pick_lane<JB>(cofj, jbc, j);
                    seq[k8] = (((jbc << 6) | j) << 8) | b;
                    ++b;
                    if (b * WAVE >= m0c) { b = 0; cl &= cl - 1ull; }
                }
            }
            if (!any) break;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const int st = (int)st_l[sidx[k8]];
#pragma unroll
                for (int o = 0; o < G; ++o) in[o][k8] &= (st > rho[o] ? 1 : 0);
            }
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) if (in[0][k8]) yv[k8] = DN ? (unsigned)loc_l[sidx[k8]] : idle_loc<DN>(D, ip[k8]);        // (alive for a later order = alive for the first)
#pragma unroll
            for (int o = 0; o < G; ++o) {
                int cst[8];
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8)
                    cst[k8] = cost_elem<U8>(crow_b[o], (unsigned)(in[o][k8] ? cof[k8] + (int)(yv[k8] & 0xFFFF) : 0));
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int v = in[o][k8] ? (cst[k8] << 16) | seq[k8] : IMAX;
                    nval[o] += in[o][k8];
                    b2[o] = min(b2[o], max(b1[o], v));
                    b1[o] = min(b1[o], v);
                }
            }
#ifdef VDS_PROF
            if (pacc) pacc[5] += 1;
#endif
        }
        SCT(2 + pass);
    }
#pragma unroll
    for (int o = 0; o < G; ++o) {
        // candidates of clusters never scanned cost at least their bound (and could win a tie on visit order)
        int ulb = IMAX;
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) ulb = min(ulb, (((cand[o][jb] & ~scanned[jb]) >> lane) & 1ull) ? lbj[o][jb] : IMAX);
        ulb = wave_min_i32(ulb);
        // the WK_K smallest keys, ascending: lane kk keeps the kk-th (and the lane it came from = its position inside the chunk)
        int nl = 0, npop = 0, mykey = IMAX, mywl = 0;
        for (int kk = 0; kk < WK_K; ++kk) {
            const int m = wave_min_i32(b1[o]);
            if (m == IMAX) break;
            if (kk > 0 && (m >> 16) >= ulb) break;
            const int wl = __ffsll((long long)ballot(b1[o] == m)) - 1;       // lowest lane = lowest list position
            if (lane == kk) { mykey = m; mywl = wl; }
            ++nl;
            bool stop = false;
            if (lane == wl) { b1[o] = b2[o]; b2[o] = IMAX; ++npop; stop = npop == 2 && nval[o] > 2; }   // the lane's third smallest is unknown
            if (ballot(stop)) break;
        }
        {   // ... and each of those lanes resolves its entry: stamp index = start of the cluster's stamps + position; cluster | k
            const int j = (mykey >> 8) & 255, bb = mykey & 255;
            int mo = 0, ckw = 0;
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) {
                const int u = __shfl(moj[jb], j & 63, WAVE), u2 = __shfl((int)ck[o][jb], j & 63, WAVE);
                mo = (j >> 6) == jb ? u : mo; ckw = (j >> 6) == jb ? u2 : ckw;
            }
            if (lane < nl) reinterpret_cast<int4 *>(rec[o])[lane] = make_int4(mykey, mo + bb * WAVE + mywl, ckw, bb * WAVE + mywl);
            // the first candidate is the likely winner.  If an own-cluster order a holds it (taken later than this order), stealing
            // it makes a pick again (k_dfs_walk, redo chain): what a would pick - the two best entries of its cluster alive at ITS
            // time (stamp > a; the same shrinking-set argument makes the first of them still alive the re-pick) - is worked out
            // here, two HBM levels off the walk's critical path.  Lists of more than 64 entries: left to the walk.
            const int sidx0 = rdlane(mo + bb * WAVE + mywl, 0), wcl = rdlane(ckw, 0) & 0xFFFF;
            int a0 = -1, r1 = IMAX, r2 = IMAX, complete = 0;
            if (PRE && nl > 0) {
                const int st0 = (int)st_l[sidx0];
                const int m0w = m0_l[wcl];
                if (st0 != (int)WK_FREE && st0 > rho[o] && m0w <= WAVE) {
                    a0 = st0;
                    const int mow = moff_l[wcl];
                    const int4 cd = S.cdesc[wcl];
                    const int ncw = cda_l[wcl] & 2047;
                    const int pick = S.so_rec[tq0 + RT.qr(a0)].y & 0xFFFF;
                    const bool al = lane < m0w && (int)st_l[mow + min(lane, m0w - 1)] > a0;
                    int key = IMAX;
                    if (al) {
                        const int lo2 = DN ? (int)loc_l[mow + lane] : (int)idle_loc<DN>(D, ((size_t)wcl * S.R + r) * S.idle_cap + lane);
                        key = (cost_elem<U8>(U8 ? reinterpret_cast<const char *>(S.blk8) : reinterpret_cast<const char *>(S.blk),
                                             (unsigned)((U8 ? cd.z : cd.y) + pick * ncw + lo2)) << 16) | lane;
                    }
                    r1 = wave_min_i32(key);
                    r2 = wave_min_i32(key == r1 ? IMAX : key);
                    complete = popc64(ballot(al)) <= 2 ? 1 : 0;
                }
            }
            if (lane == 0) { rec[o][4 * WK_K + 1] = (unsigned)a0; rec[o][4 * WK_K + 2] = (unsigned)r1; rec[o][4 * WK_K + 3] = (unsigned)r2; rec[o][4 * WK_K + 4] = (unsigned)complete; }
        }
        if (lane == 0) rec[o][4 * WK_K] = (unsigned)nl;
    }
    SCT(4);
}
// one order
template <bool U8, int JB, bool DN = false>
__device__ __forceinline__ void dfs_scan(const Static &S, const State &D, int r, int q, int rho, int pnode,
                                         const int *m0_l, const int *moff_l, const int *ls_l, const int *cda_l,
                                         const unsigned short *st_l, const RankTab<DN> &RT, const unsigned char *loc_l, int tq0, unsigned *rec, unsigned long long *pacc = nullptr) {
    const int rho1[1] = {rho}, pn1[1] = {pnode};
    unsigned *const rec1[1] = {rec};
    dfs_scan<U8, JB, 1, (WK_REDO_PRE != 0), DN>(S, D, r, q, rho1, pn1, m0_l, moff_l, ls_l, cda_l, st_l, RT, loc_l, tq0, rec1, pacc);
}

// JB: 64-cluster batches of the longest visit sequence (Static.seq_pad / 64: 1, 2 or 4).
// (Round 5's deferred-acceptance form of the walk - exact, never faster; round 6's form of it with eight scans per wavefront likewise -
// left the library in round 6: profiles/r05/, profiles/r06/gs_experiment.md hold the measurements and the last source.)
// DN: neighbour search on the DENSE layout (round 6; Static.dense_st).  The first half is k_tick_dense in stamp form, which - unlike
//     k_tick_rows' - COMMITS what it does: results with vehicle ids, static arrival slots, counters, the entries' stamps in HBM
//     (State.stamp); nothing is left for a per-replica commit.  This kernel then returns at once unless a searching bucket of the
//     replica ran dry (State.dry); otherwise it loads the stamps, serves the dry orders as before and writes out only what MOVED:
//     the served dry orders and the own-cluster orders whose vehicle was stolen (result, arrival slot / ring post, stamp, counter
//     deltas, the lists' alive counts).  The lists themselves are packed by the next slot's tick (or k_dense_flush).
template <bool U8, int JB, bool DN = false>
__global__ __launch_bounds__(WK_THREADS, WK_MIN_WAVES) void k_dfs_walk(Static S, State D, int t) {
    extern __shared__ int lds_dyn[];
    const char *blk_b = U8 ? reinterpret_cast<const char *>(S.blk8) : reinterpret_cast<const char *>(S.blk);
    const int C = S.C;
    const int mto = S.max_tick_orders;
    int *m0_l = lds_dyn;                  // [C] list length after Update (taken entries still inside); later the final length
    int *moff_l = lds_dyn + C;            // [C+1] start of the cluster's stamps
    int *qend_l = moff_l + C + 1;         // [C]
    int *lm_l = qend_l + C;               // [C] own-cluster matches of the bucket (a prefix of its orders)
    int *sc_l = lm_l + C;                 // [C] vehicles stolen from the cluster so far
    int *ls_l = sc_l + C;                 // [C] lm << 16 | sc as of the last served dry order (what the scanning wavefronts read)
    int *tk_l = ls_l + C;                 // [C] sum over the cluster's steals of min(orders of the bucket before the thief, own matches)
    int *cdA_l = tk_l + C;                // [C] n_c | first cost column << 11 | can search << 30
    int *tab_l = cdA_l + C;               // rank tables (u16), later the resolve counters
    const int ids_n = DN ? 0 : (mto + 2 > RCNT * C ? mto + 2 : RCNT * C);      // (dense layout: no LDS rank tables, no commit counters)
    unsigned short *rq_l = reinterpret_cast<unsigned short *>(tab_l);                 // [mto] rank of sorted position
    unsigned short *qr_l = rq_l + ((mto + 1) & ~1);                                   // [mto] sorted position of rank
    unsigned *dry_bits = reinterpret_cast<unsigned *>(tab_l + ids_n);
    const int nwords = (mto + 31) / 32 + 1;
    unsigned *clm_bits = dry_bits + nwords;                                           // dry orders somebody has claimed for scanning
    unsigned *mov_bits = clm_bits + nwords;                                           // (DN) own-cluster orders whose vehicle was stolen: they picked again
    unsigned *slot_l = mov_bits + (DN ? nwords : 0);                                  // one record, for the scans wavefront 0 does itself
    unsigned *pool_l = slot_l + WK_REC;                                               // WK_NS records, filled by wavefronts 1..3
    const int bmw = (C + 31) / 32;
    const int ns = S.walk_pool;
    int *lg_l = reinterpret_cast<int *>(pool_l + ns * WK_REC);                        // [64][4] the steal log's current chunk
    unsigned short *st_l = reinterpret_cast<unsigned short *>(lg_l + 4 * WAVE);       // [V] stamps
    unsigned char *loc_l = reinterpret_cast<unsigned char *>(st_l + 2 * ((S.V + 1) / 2));      // (DN) [V] node bytes of the idle entries (index inside the cluster)
    __shared__ int s_ev;                  // evaluations of the dry orders
    __shared__ int s_nlog;                // steals (entries of the replica's steal log)
    __shared__ int s_nmid;                // lists of 65 .. 128 entries to compact
#ifdef WKDEBUG
    __shared__ int s_dbg;                 // (make dbg) the dry orders' evaluations counted by the walk itself, against the closed form
#endif
    __shared__ int s_cursor;              // rank of the dry order the walk is at: the scanning wavefronts look for work from there on
    __shared__ int s_done;                // the walk is over
    __shared__ int s_slot[WK_NS];         // pool record s: 0 free, else rank of its dry order << 2 | 1 being filled / 2 ready
    const int r = S.r_lo + (int)blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    if (DN && D.dry[r] == 0) return;          // no searching bucket of this replica ran dry in this slot: k_tick_dense has done it all
    const DayView dv = day_view(S, r);
    if (t >= dv.T) return;
    const int now = dv.now0 + t * S.tick_minutes;
    const int *bkt_off = dv.bkt_off;
    const int tq0 = bkt_off[(size_t)t * C], tq1 = bkt_off[(size_t)(t + 1) * C];
    const int nord = tq1 - tq0;
    int2 *out_r = D.out + (size_t)r * S.Oq - dv.q_base;
    const RankTab<DN> RT{rq_l, qr_l, S.so_rank + tq0, S.ord_q + dv.tick_off[t], tq0};
#ifdef VDS_PROF
    const bool prof = (g_ablate & 128) != 0;
    unsigned long long tprev = prof ? __builtin_amdgcn_s_memtime() : 0ull;
    const int pwave = (int)((blockIdx.x * WK_WAVES + wave) & (PROF_WAVES - 1));
#endif
    // ---- tables
    for (int i0 = threadIdx.x; i0 < (DN ? 0 : nord); i0 += 6 * WK_THREADS) {       // (six loads in flight per thread)
        int rk[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) rk[u] = i0 + u * WK_THREADS < nord ? S.so_rank[tq0 + i0 + u * WK_THREADS] : 0;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (i0 + u * WK_THREADS < nord) { rq_l[i0 + u * WK_THREADS] = (unsigned short)rk[u]; qr_l[rk[u]] = (unsigned short)(i0 + u * WK_THREADS); }
    }
    for (int c = threadIdx.x; c < C; c += WK_THREADS) {
        const int4 cd = S.cdesc[c];
        const bool capable = S.dfs_off[c + 1] > S.dfs_off[c];
        cdA_l[c] = cd.x | (S.cl_off[c] << 11) | (capable ? CAPABLE : 0);
        const int q0 = bkt_off[(size_t)t * C + c], q1 = bkt_off[(size_t)t * C + c + 1];
        int m0 = D.hdr[((size_t)c * S.R + r) * HDR_WORDS + HDR_IDLE];
        if (DN) { const int rw = D.hdr[((size_t)c * S.R + r) * HDR_WORDS + HDR_RAW]; if (rw) m0 = rw - 1; }      // (the raw length: this slot's taken entries are inside)
        const int own = min(q1 - q0, m0);                 // the fast kernel matched while vehicles remained
        m0_l[c] = m0; qend_l[c] = q1; lm_l[c] = own; sc_l[c] = 0; ls_l[c] = own << 16; tk_l[c] = 0;
    }
    for (int w = threadIdx.x; w < nwords; w += WK_THREADS) { dry_bits[w] = 0u; clm_bits[w] = 0u; if (DN) mov_bits[w] = 0u; }
    if (threadIdx.x == 0) { s_ev = 0; s_cursor = 0; s_done = 0; s_nlog = 0; s_nmid = 0; }
    if (threadIdx.x < WK_NS) s_slot[threadIdx.x] = 0;
    __syncthreads();
    PROF_STAMP(24);
    if (wave == 0) {            // exclusive prefix of the list lengths
        int run = 0;
        for (int base = 0; base < C; base += WAVE) {
            const int c = base + lane;
            const int v = c < C ? m0_l[c] : 0;
            int inc = v;
            for (int o = 1; o < WAVE; o <<= 1) { const int u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
            if (c < C) moff_l[c] = run + inc - v;
            run += rdlane(inc, WAVE - 1);
        }
        if (lane == 0) moff_l[C] = run;
    }
    __syncthreads();
    PROF_STAMP(25);
    // ---- stamps: free, or the rank of the own-cluster order the fast kernel gave the entry to (its preliminary result says which)
    if (DN) {
        // (dense layout) the stamps as k_tick_dense wrote them: one list per thread, eight 16-byte pieces (64 stamps) in flight
        for (int c = threadIdx.x; c < C; c += WK_THREADS) {
            const int m0 = m0_l[c], mo = moff_l[c];
            const uint4 *sp4 = reinterpret_cast<const uint4 *>(D.stamp + ((size_t)c * S.R + r) * S.idle_cap);
            for (int i0 = 0; i0 < m0; i0 += 64) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); if (i0 + 8 * u < m0) v[u] = sp4[(i0 >> 3) + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int i = i0 + 8 * u + e;
                        if (i < m0) st_l[mo + i] = (unsigned short)(w4[e >> 1] >> ((e & 1) << 4));
                    }
                }
            }
            // ... and the entries' node bytes (the scans and the re-pick chains then read LDS only, apart from the cost byte)
            const uint4 *ip4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned *>(D.idle) + ((size_t)c * S.R + r) * S.idle_cap);
            for (int i0 = 0; i0 < m0; i0 += 32) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = make_uint4(0u, 0u, 0u, 0u); if (i0 + 4 * u < m0) v[u] = ip4[(i0 >> 2) + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = i0 + 4 * u + e;
                        if (i < m0) loc_l[mo + i] = (unsigned char)(w4[e] & 0xFFu);
                    }
                }
            }
        }
    } else {
    for (int i = threadIdx.x; i < (moff_l[C] + 1) / 2; i += WK_THREADS) reinterpret_cast<unsigned *>(st_l)[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < nord; i0 += 6 * WK_THREADS) {
        int2 pr[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) pr[u] = i0 + u * WK_THREADS < nord ? out_r[tq0 + i0 + u * WK_THREADS] : make_int2(-1, -1);
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (pr[u].x != -1) st_l[moff_l[(unsigned)pr[u].x >> 16] + (pr[u].x & 0xFFFF)] = rq_l[i0 + u * WK_THREADS];
    }
    }
    PROF_STAMP(26);
    // dry orders: everything behind a searching cluster's exhaustion point
    for (int c = threadIdx.x; c < C; c += WK_THREADS) {
        if (!(cdA_l[c] & CAPABLE)) continue;
        const int qa = c == 0 ? tq0 : qend_l[c - 1];
        for (int q = qa + lm_l[c]; q < qend_l[c]; q += 8) {          // (eight ranks in flight: on the dense layout they come from L2)
            int rk8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rk8[u] = q + u < qend_l[c] ? RT.rq(q + u - tq0) : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (rk8[u] >= 0) atomicOr(&dry_bits[rk8[u] >> 5], 1u << (rk8[u] & 31));
        }
    }
    __syncthreads();
    PROF_STAMP(0);
    // ---- the walk.  Wavefront 0 serves the dry orders in id (rank) order: evaluations (the alive counts of the visited clusters
    //      as they stand, closed form), the first candidate of the order's record still alive, the steal, the redo chain.  The
    //      records are made by wavefronts 1..3, which scan the dry orders up to WK_NS ahead of the walk, in rank order, claiming
    //      them through s_cursor (dfs_scan: why a scan against a moving state is exact).  Wavefront 0 scans itself only when
    //      every kept candidate has died, or when a redo made an order dry that the cursor had already passed.
    {
    auto next_dry = [&](int from) -> int {       // (the words ascend with the lanes: the first lane with a bit set holds the minimum)
        for (int w0 = from >> 5; w0 < nwords; w0 += WAVE) {
            const int w = w0 + lane;
            unsigned bits = w < nwords ? (unsigned)lds_load(reinterpret_cast<const int *>(&dry_bits[w])) : 0u;
            if (w == (from >> 5)) bits &= 0xFFFFFFFFu << (from & 31);
            const unsigned long long nz = ballot(bits != 0u);
            if (nz != 0ull) {
                const int l = __ffsll((long long)nz) - 1;
                return (w0 + l) * 32 + __ffs(rdlane((int)bits, l)) - 1;
            }
        }
        return IMAX;
    };
    if (wave == 0) {
        if (WK_PRIO) __builtin_amdgcn_s_setprio(WK_PRIO);
        int nlog = 0;
        int4 *slog = D.slog + (size_t)r * mto;
        int rho = next_dry(0);
#ifdef WKDEBUG
        int dbg_ev = 0;
#endif
#ifdef VDS_PROF
        unsigned long long p_wait = 0, p_chain = 0, p_seg[4] = {0, 0, 0, 0}, p_cnt[4] = {0, 0, 0, 0};
#endif
        while (rho != IMAX) {
            // (the order's sorted position - an LDS round trip - only where a scan by this wavefront needs it)
            auto q_of = [&]() -> int { return tq0 + RT.qr(rho); };
            // the order's record: in the pool (ready, or being filled), still to be claimed (wait), or passed over (scan here)
            const unsigned *rec = slot_l;
            int slot = -1;
#ifdef VDS_PROF
            const unsigned long long p_t0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
            if (lane == 0) lds_release(&s_cursor, rho);
            while (true) {
                // (a scanning wavefront publishes its slot word before it claims the order, and withdraws it if another was first)
                const int sw = lane < ns ? lds_acquire(&s_slot[lane]) : 0;
                const unsigned long long ready = ballot(sw == ((rho << 2) | 2));
                if (ready != 0ull) { slot = __ffsll((long long)ready) - 1; break; }
                if (ballot(sw == ((rho << 2) | 1)) == 0ull) {
                    // not in the pool: claim it for a scan by this wavefront - unless a scanning wavefront has just done so
                    unsigned old = 0u;
                    if (lane == 0) old = atomicOr(&clm_bits[rho >> 5], 1u << (rho & 31));
                    if (!((unsigned)__builtin_amdgcn_readfirstlane((int)old) & (1u << (rho & 31)))) break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef VDS_PROF
            unsigned long long p_ts = 0;
            if (prof) { p_ts = __builtin_amdgcn_s_memtime(); p_wait += p_ts - p_t0; }
#define PSEG(i) do { if (prof) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); p_seg[i] += t_ - p_ts; p_ts = t_; } } while (0)
#else
#define PSEG(i) do { } while (0)
#endif
            if (slot >= 0) {
                rec = pool_l + slot * WK_REC;
            } else {
                const int q = q_of();
                dfs_scan<U8, JB, DN>(S, D, r, q, rho, S.so_pnode[q], m0_l, moff_l, ls_l, cdA_l, st_l, RT, loc_l, tq0, slot_l);
                wg_order();
#ifdef VDS_PROF
                p_cnt[2] += 1;
#endif
            }
            int4 e = make_int4(IMAX, 0, 0, 0);
            if (lane < WK_K) e = reinterpret_cast<const int4 *>(rec)[lane];
            int nl = (int)rec[4 * WK_K];
            int ra0 = (int)rec[4 * WK_K + 1], rr1 = (int)rec[4 * WK_K + 2], rr2 = (int)rec[4 * WK_K + 3], rcomp = (int)rec[4 * WK_K + 4];
            int stv = lane < nl ? (int)st_l[e.y] : -1;
            if (slot >= 0) {
                wg_order();
                if (lane == 0) lds_release(&s_slot[slot], 0);
            }
            unsigned long long okb = ballot(stv > rho);
            if (nl > 0 && okb == 0ull) {           // every kept candidate has been taken since: scan again, on the state as it is
                const int q = q_of();
                dfs_scan<U8, JB, DN>(S, D, r, q, rho, S.so_pnode[q], m0_l, moff_l, ls_l, cdA_l, st_l, RT, loc_l, tq0, slot_l);
                wg_order();
                e = make_int4(IMAX, 0, 0, 0);
                if (lane < WK_K) e = reinterpret_cast<const int4 *>(slot_l)[lane];
                nl = (int)slot_l[4 * WK_K];
                ra0 = (int)slot_l[4 * WK_K + 1]; rr1 = (int)slot_l[4 * WK_K + 2]; rr2 = (int)slot_l[4 * WK_K + 3]; rcomp = (int)slot_l[4 * WK_K + 4];
                stv = lane < nl ? (int)st_l[e.y] : -1;
                okb = ballot(stv > rho);
#ifdef VDS_PROF
                p_cnt[2] += 1;
#endif
            }
            PSEG(0);
#ifdef WKDEBUG
            {   // reference: the alive counts of the visited clusters as they stand now
                const int q = q_of();
                int alive = 0;
                for (int jb = 0; jb < JB; ++jb) {
                    const unsigned v = S.so_vis[(size_t)q * S.seq_pad + jb * WAVE + lane];
                    if (v != 0xFFFFFFFFu) { const int c = (int)(v & 0xFFFFu); const int ls = ls_l[c]; alive += m0_l[c] - min((int)(v >> 16), ls >> 16) - (ls & 0xFFFF); }
                }
                const int rs = row_sum_i32(alive);
                dbg_ev += rdlane(rs, 0) + rdlane(rs, 16) + rdlane(rs, 32) + rdlane(rs, 48);
            }
#endif
            if (okb != 0ull) {                     // (no candidate at all: nothing was alive when the order was scanned - rejected, :973;
                                                   //  the fast kernel has written that result already)
                const int first = __ffsll((long long)okb) - 1;
                const int key = rdlane(e.x, first), idx = rdlane(e.y, first), ckw = rdlane(e.z, first), wpos = rdlane(e.w, first);
                int a = rdlane(stv, first);                 // the winner's stamp: free, or the own-cluster order that took it later
                const int wc = key >> 16;
                if ((long long)wc <= S.reject_threshold) {
                    const int wcl = ckw & 0xFFFF;
                    const int mo = idx - wpos;              // (= moff_l[wcl]: the record carries the position, no LDS round trip)
                    WKCHK(wcl < C && wpos >= 0 && wpos < m0_l[wcl < C ? wcl : 0], 4, wcl, wpos);
                    // the steal: stamp, and a log entry {rank, cluster | k, result} - staged in LDS and written out 64 at a time (an
                    // HBM store per served order would cost this wavefront a round trip at its next register reuse); the
                    // order's result reaches D.out from the log, after the walk
                    if (lane == 0) {
                        st_l[idx] = (unsigned short)rho;
                        int *lg = lg_l + 4 * (nlog & (WAVE - 1));
                        lg[0] = rho; lg[1] = ckw; lg[2] = (int)(((unsigned)wcl << 16) | (unsigned)wpos); lg[3] = wc;
                    }
                    ++nlog;
                    if ((nlog & (WAVE - 1)) == 0)
                        slog[nlog - WAVE + lane] = make_int4(lg_l[4 * lane], lg_l[4 * lane + 1], lg_l[4 * lane + 2], lg_l[4 * lane + 3]);
                    wg_order();
                    // (DN: the log is kept as before - the evaluation pass reads it; the orders' results come from the final stamps)
#ifdef VDS_PROF
                    p_cnt[0] += 1; if (a != (int)WK_FREE) p_cnt[1] += 1;
#endif
                    PSEG(1);
#ifdef VDS_PROF
                    const unsigned long long p_c0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
                    int exhausted = 0;
                    if (a != (int)WK_FREE) {
                        // the stolen vehicle had been taken later by own-cluster order a of wcl: that order picks again among the
                        // entries alive at ITS time (stamp > a); if its new pick had been taken by a later order, that one picks
                        // again, ... until a free entry is taken or the list is exhausted - then the last own match of the bucket
                        // (only it can find nothing) turns dry.  Nobody else's choice changes.
                        const int cda = cdA_l[wcl];
                        const int nc = cda & 2047;
                        const bool capable = (cda & CAPABLE) != 0;
                        const int4 cd = S.cdesc[wcl];
                        const int boff = U8 ? cd.z : cd.y;
                        const int m0 = m0_l[wcl];
                        const size_t ibase = ((size_t)wcl * S.R + r) * S.idle_cap;
                        // the list's node words travel together with the first victim's pickup node (one HBM level, not two), and
                        // serve every step of the chain; lists of more than 64 entries read them step by step
                        // first step from the record, when the scan worked it out for this very holder: the first of its two entries
                        // still alive at a's time is the re-pick; none alive and nothing else existed: the bucket is exhausted
                        bool chain = true;
                        if (first == 0 && ra0 == a) {
                            const int p1 = rr1 != IMAX ? (rr1 & 0xFFFF) : -1, p2 = rr2 != IMAX ? (rr2 & 0xFFFF) : -1;
                            const int s1 = p1 >= 0 ? (int)st_l[mo + p1] : -1, s2 = p2 >= 0 ? (int)st_l[mo + p2] : -1;
                            const int pk = s1 > a ? p1 : (s2 > a ? p2 : -1);
                            if (pk >= 0) {
                                const int y = tq0 + RT.qr(a);
                                const int bst = s1 > a ? s1 : s2;
                                if (lane == 0) {
                                    st_l[mo + pk] = (unsigned short)a;
                                    if (DN) atomicOr(&mov_bits[a >> 5], 1u << (a & 31));
                                    else out_r[y] = make_int2((int)(((unsigned)wcl << 16) | (unsigned)pk), (s1 > a ? rr1 : rr2) >> 16);
                                }
                                wg_order();
                                if (bst == (int)WK_FREE) chain = false; else a = bst;
                            } else if (rcomp) {
                                const int y = tq0 + RT.qr(a);
                                exhausted = 1;
                                if (lane == 0) {
                                    if (capable) atomicOr(&dry_bits[a >> 5], 1u << (a & 31));
                                    if (DN) atomicOr(&mov_bits[a >> 5], 1u << (a & 31));
                                    else out_r[y] = make_int2(-1, -1);
                                }
                                chain = false;
                            }
                        }
                        unsigned yv0 = 0u;
                        if (chain && m0 <= WAVE && lane < m0) yv0 = DN ? (unsigned)loc_l[mo + lane] : idle_loc<DN>(D, ibase + lane);
                        while (chain) {
                            const int y = tq0 + RT.qr(a);
                            const int pick = S.so_rec[y].y & 0xFFFF;
                            int lc = IMAX, lp = -1;
                            if (m0 <= WAVE) {
                                if (lane < m0 && (int)st_l[mo + lane] > a) { lc = cost_elem<U8>(blk_b, (unsigned)(boff + pick * nc + (int)(yv0 & 0xFFFF))); lp = lane; }
                            } else {
                                for (int base = 0; base < m0; base += WAVE) {
                                    const int ii = base + lane;
                                    if (ii < m0 && (int)st_l[mo + ii] > a) {
                                        const int lo2 = DN ? (int)loc_l[mo + ii] : (int)idle_loc<DN>(D, ibase + ii);
                                        const int cst = cost_elem<U8>(blk_b, (unsigned)(boff + pick * nc + lo2));
                                        if (lp < 0 || cst < lc) { lc = cst; lp = ii; }
                                    }
                                }
                            }
                            const int minc = wave_min_i32(lp >= 0 ? lc : IMAX);
                            if (minc == IMAX) {
                                exhausted = 1;
                                if (lane == 0) {
                                    if (capable) atomicOr(&dry_bits[a >> 5], 1u << (a & 31));
                                    if (DN) atomicOr(&mov_bits[a >> 5], 1u << (a & 31));
                                    else out_r[y] = make_int2(-1, -1);      // (rejected, unless the search serves it - then the log says so)
                                }
                                break;
                            }
                            const int minp = wave_min_i32((lp >= 0 && lc == minc) ? lp : IMAX);
                            const int bst = (int)st_l[mo + minp];
                            wg_order();
                            if (lane == 0) {
                                st_l[mo + minp] = (unsigned short)a;
                                if (DN) atomicOr(&mov_bits[a >> 5], 1u << (a & 31));
                                else out_r[y] = make_int2((int)(((unsigned)wcl << 16) | (unsigned)minp), minc);
                            }
                            wg_order();
                            if (bst == (int)WK_FREE) break;
                            a = bst;
                        }
                    }
#ifdef VDS_PROF
                    if (prof) { __builtin_amdgcn_s_waitcnt(0); p_chain += __builtin_amdgcn_s_memtime() - p_c0; }
#endif
                    // own matches << 16 | steals of the cluster: one update once the order is served (what the scans read)
                    wg_order();
                    if (lane == 0) atomicAdd(&ls_l[wcl], 1 - (exhausted << 16));
                }
            }
#ifdef VDS_PROF
            if (prof) p_ts = __builtin_amdgcn_s_memtime();
            p_cnt[3] += 1;
#endif
            wg_order();
            rho = next_dry(rho + 1);
            PSEG(2);
        }
        if (lane < (nlog & (WAVE - 1)))
            slog[(nlog & ~(WAVE - 1)) + lane] = make_int4(lg_l[4 * lane], lg_l[4 * lane + 1], lg_l[4 * lane + 2], lg_l[4 * lane + 3]);
        if (lane == 0) { s_nlog = nlog; lds_release(&s_done, 1); }
        if (WK_PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef WKDEBUG
        if (lane == 0) s_dbg = dbg_ev;
#endif
#ifdef VDS_PROF
        if (prof && lane == 0) {
            g_prof[(size_t)pwave * PROF_SLOTS + 8] += p_wait; g_prof[(size_t)pwave * PROF_SLOTS + 9] += p_chain;
            for (int i = 0; i < 3; ++i) g_prof[(size_t)pwave * PROF_SLOTS + 12 + i] += p_seg[i];
            g_prof[(size_t)pwave * PROF_SLOTS + 2] += p_cnt[0]; g_prof[(size_t)pwave * PROF_SLOTS + 3] += p_cnt[1];
            g_prof[(size_t)pwave * PROF_SLOTS + 4] += p_cnt[2]; g_prof[(size_t)pwave * PROF_SLOTS + 6] += p_cnt[3];
        }
#endif
    } else {
#ifdef VDS_PROF
        unsigned long long p_scan = 0, p_n = 0, p_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        while (true) {
            // the first dry order from the walk's position on that nobody has claimed
            const int from = lds_acquire(&s_cursor);
            int b = IMAX;
            for (int w = (from >> 5) + lane; w < nwords; w += WAVE) {
                unsigned bits = (unsigned)lds_load(reinterpret_cast<const int *>(&dry_bits[w])) & ~(unsigned)lds_load(reinterpret_cast<const int *>(&clm_bits[w]));
                if (w == (from >> 5)) bits &= 0xFFFFFFFFu << (from & 31);
                if (bits != 0u) { b = w * 32 + __ffs((int)bits) - 1; break; }
            }
            b = wave_min_i32(b);
            if (b == IMAX) {
                if (lds_acquire(&s_done)) break;
                __builtin_amdgcn_s_sleep(10);
                continue;
            }
            const int swl = lane < ns ? lds_load(&s_slot[lane]) : -1;
            unsigned long long freem = ballot(swl == 0);
            if (freem == 0ull) { __builtin_amdgcn_s_sleep(10); continue; }
            const int slot = __ffsll((long long)freem) - 1;
            freem &= freem - 1ull;
            int won = 0;
            if (lane == 0 && lds_cas(&s_slot[slot], 0, (b << 2) | 1)) {
                won = (atomicOr(&clm_bits[b >> 5], 1u << (b & 31)) & (1u << (b & 31))) ? 0 : 1;
                if (!won) lds_release(&s_slot[slot], 0);
            }
            won = __builtin_amdgcn_readfirstlane(won);
            if (!won) continue;
            const int q = tq0 + RT.qr(b);
            // the next orders of the same bucket (dry as well: the dry orders of a bucket are its last ones; consecutive sorted
            // positions, ascending ranks) share the scan, up to WK_G of them, pool permitting (one free record is left to the others)
            int rk[WK_G], sl[WK_G], pn[WK_G];
            int g = 1;
            rk[0] = b; sl[0] = slot; pn[0] = S.so_pnode[q];
#pragma unroll
            for (int o = 1; o < WK_G; ++o) { rk[o] = -1; sl[o] = -1; pn[o] = 0; }
            if (WK_G > 1 && q + 1 < tq1 && popc64(freem) >= 2) {
                const int4 ra = S.so_rec[q];
                int bk[WK_G];
#pragma unroll
                for (int o = 1; o < WK_G; ++o) {
                    const int qo = min(q + o, tq1 - 1);
                    bk[o] = (int)((unsigned)S.so_rec[qo].z >> 16);
                    pn[o] = S.so_pnode[qo];
                }
                bool go = true;
#pragma unroll
                for (int o = 1; o < WK_G; ++o) {
                    go = go && q + o < tq1 && bk[o] == (int)((unsigned)ra.z >> 16) && popc64(freem) >= 2;
                    if (go) {
                        const int r2 = RT.rq(q + o - tq0);
                        const int s2 = __ffsll((long long)freem) - 1;
                        int won2 = 0;
                        if (lane == 0 && r2 - b <= WK_PAIR_SPAN && lds_cas(&s_slot[s2], 0, (r2 << 2) | 1)) {
                            won2 = (atomicOr(&clm_bits[r2 >> 5], 1u << (r2 & 31)) & (1u << (r2 & 31))) ? 0 : 1;
                            if (!won2) lds_release(&s_slot[s2], 0);
                        }
                        if (__builtin_amdgcn_readfirstlane(won2)) { rk[o] = r2; sl[o] = s2; freem &= freem - 1ull; g = o + 1; }
                        else go = false;
                    }
                }
            }
#ifdef VDS_PROF
            const unsigned long long p_s0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
#define WK_PACC , prof ? p_acc : nullptr
#else
#define WK_PACC
#endif
#if WK_G >= 3
            if (g == 3) {
                const int rho3[3] = {rk[0], rk[1], rk[2]}, pn3[3] = {pn[0], pn[1], pn[2]};
                unsigned *const rec3[3] = {pool_l + sl[0] * WK_REC, pool_l + sl[1] * WK_REC, pool_l + sl[2] * WK_REC};
                dfs_scan<U8, JB, 3, (WK_REDO_PRE != 0), DN>(S, D, r, q, rho3, pn3, m0_l, moff_l, ls_l, cdA_l, st_l, RT, loc_l, tq0, rec3 WK_PACC);
            } else
#endif
#if WK_G >= 2
            if (g == 2) {
                const int rho2[2] = {rk[0], rk[1]}, pn2[2] = {pn[0], pn[1]};
                unsigned *const rec2[2] = {pool_l + sl[0] * WK_REC, pool_l + sl[1] * WK_REC};
                dfs_scan<U8, JB, 2, (WK_REDO_PRE != 0), DN>(S, D, r, q, rho2, pn2, m0_l, moff_l, ls_l, cdA_l, st_l, RT, loc_l, tq0, rec2 WK_PACC);
            } else
#endif
            {
                dfs_scan<U8, JB, DN>(S, D, r, q, b, pn[0], m0_l, moff_l, ls_l, cdA_l, st_l, RT, loc_l, tq0, pool_l + slot * WK_REC WK_PACC);
            }
            wg_order();
            if (lane == 0) {
#pragma unroll
                for (int o = 0; o < WK_G; ++o) if (o < g) lds_release(&s_slot[sl[o]], (rk[o] << 2) | 2);
            }
#ifdef VDS_PROF
            if (prof && g > 1) p_acc[6] += 1;
#endif
#ifdef VDS_PROF
            if (prof) { p_scan += __builtin_amdgcn_s_memtime() - p_s0; p_n += 1; }
#endif
        }
#ifdef VDS_PROF
        if (prof && lane == 0) {
            g_prof[(size_t)pwave * PROF_SLOTS + 10] += p_scan; g_prof[(size_t)pwave * PROF_SLOTS + 11] += p_n;
            for (int i = 0; i < 8; ++i) g_prof[(size_t)pwave * PROF_SLOTS + 16 + i] += p_acc[i];
        }
#endif
    }
    }   // (the walk)
    __syncthreads();
    PROF_STAMP(1);
    __shared__ int s_dw, s_dv, s_drej;            // (DN) what the moved orders change in the replica's counters: wait, value, rejects
    if constexpr (DN) {
    // ---- dense layout: what MOVED, read off the final stamps.  k_tick_dense has committed every order as if nothing were stolen
    //      (result, arrival slot, counters); the orders whose answer differs now are the dry ones (dry_bits) and the own-cluster
    //      orders that picked again (mov_bits).  (1) every idle entry whose stamp names such an order IS that order's vehicle: result
    //      {vehicle, cost}, arrival (:954-960: its static slot when the arrival stays inside the slot's window, else the ring), the
    //      entry's stamp in HBM (what the next slot's tick - or k_dense_flush - drops), the counters' difference to what was
    //      committed; (2) a moved order that holds nothing is unserved now.
    unsigned *got_bits = clm_bits;
    for (int w = threadIdx.x; w < nwords; w += WK_THREADS) got_bits[w] = 0u;
    if (threadIdx.x == 0) { s_dw = 0; s_dv = 0; s_drej = 0; }
    __syncthreads();
    {
        const unsigned *idle32 = reinterpret_cast<const unsigned *>(D.idle);
        const int total = moff_l[C];
        int dw = 0, dv = 0, drej = 0;
        int i = (int)threadIdx.x;
        while (i < total) {
            int mi[4], msv[4];
            int nm = 0;
            for (; i < total && nm < 4; i += WK_THREADS) {
                const int sv = (int)st_l[i];
                if (sv == (int)WK_FREE) continue;
                if (!(((dry_bits[sv >> 5] | mov_bits[sv >> 5]) >> (sv & 31)) & 1u)) continue;
                mi[nm] = i; msv[nm] = sv; ++nm;
            }
            int mc[4], mpos[4], my[4], mpn[4], mslot[4];
            unsigned ment[4];
            int4 mrec[4];
            int2 mold[4];
            bool mdry[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mc[u] = 0; mpos[u] = 0; my[u] = tq0; mpn[u] = 0; mslot[u] = -1; ment[u] = 0u; mrec[u] = make_int4(0, 0, 0, 0); mold[u] = make_int2(-1, -1); mdry[u] = false;
                if (u < nm) {
                    int lo = 0, hi = C;                          // cluster of stamp index i: the last c with moff_l[c] <= i
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (moff_l[mid] <= mi[u]) lo = mid; else hi = mid; }
                    // (empty clusters share their start with the next one: the LAST cluster starting at or before i holds it)
                    mc[u] = lo; mpos[u] = mi[u] - moff_l[lo];
                    my[u] = tq0 + RT.qr(msv[u]);
                    mdry[u] = (dry_bits[msv[u] >> 5] >> (msv[u] & 31)) & 1u;
                    ment[u] = idle32[((size_t)mc[u] * S.R + r) * S.idle_cap + mpos[u]];
                    mrec[u] = S.so_rec[my[u]];
                    mold[u] = out_r[my[u]];
                    mpn[u] = S.so_pnode[my[u]];
                    mslot[u] = S.so_slot[my[u]];
                }
            }
            int mcst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mcst[u] = 0;
                if (u < nm) {
                    const int cda = cdA_l[mc[u]];
                    const int loc = (int)(ment[u] & 0xFFu);
                    // a dry order that now holds a vehicle of its OWN cluster cannot exist (it searched because that list was
                    // exhausted for it), so dry = the pickup node's row of the matrix, own = the cluster's block
                    if (mdry[u]) {
                        const char *crow = U8 ? reinterpret_cast<const char *>(S.cost8 + (size_t)mpn[u] * S.N) : reinterpret_cast<const char *>(S.cost + (size_t)mpn[u] * S.N);
                        mcst[u] = cost_elem<U8>(crow, (unsigned)(((cda >> 11) & 0xFFFF) + loc));
                    } else {
                        const int4 cd = S.cdesc[mc[u]];
                        mcst[u] = cost_elem<U8>(blk_b, (unsigned)((U8 ? cd.z : cd.y) + (mrec[u].y & 0xFFFF) * (cda & 2047) + loc));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= nm) continue;
                const int c = mc[u], sv = msv[u];
                const int veh = (int)(ment[u] >> 8), cst = mcst[u];
                out_r[my[u]] = make_int2(veh, cst);
                D.stamp[((size_t)c * S.R + r) * S.idle_cap + mpos[u]] = (unsigned short)sv;
                const int rel = cst + mrec[u].w;
                const int d = rel <= 0 ? 1 : ticks_until<true>(S, rel);
                const int dmin = mrec[u].w <= 0 ? 1 : ticks_until<true>(S, mrec[u].w);
                const int dl = (int)((unsigned)mrec[u].y >> 16);
                // (SupplyExpect in place, State.sup: the committed arrival leaves its plane, the new one enters its own - post_arrival bumps
                // the plane itself)
                if (D.sup != nullptr && mold[u].x != -1) {
                    const int relo = mold[u].y + mrec[u].w;
                    atomicSub(&D.sup[sup_index(S.C, S.R, t + (relo <= 0 ? 1 : ticks_until<true>(S, relo)), mrec[u].z & 0xFFFF, r)], 1);
                }
                if (mslot[u] >= 0 && d - dmin <= S.pull_W) {
                    D.arr[arr_index(S.R, mslot[u], r)] = pull_entry(veh, t + d);
                    if (D.sup != nullptr) atomicAdd(&D.sup[sup_index(S.C, S.R, t + d, mrec[u].z & 0xFFFF, r)], 1);
                } else {
                    if (mslot[u] >= 0) D.arr[arr_index(S.R, mslot[u], r)] = pull_reject(t);
                    post_arrival<true, true>(S, D, mrec[u].z & 0xFFFF, r, t, now, veh, mrec[u].x, now + rel, 0, dl);
                }
                if (mold[u].x == -1) { drej -= 1; dw += cst; dv += mrec[u].w; }
                else dw += cst - mold[u].y;
                atomicOr(&got_bits[sv >> 5], 1u << (sv & 31));
            }
        }
        __syncthreads();
        // (2) moved orders without a vehicle: committed as matched by k_tick_dense, unserved now
        for (int rk = (int)threadIdx.x; rk < nord; rk += WK_THREADS) {
            const unsigned w = (dry_bits[rk >> 5] | mov_bits[rk >> 5]) & ~got_bits[rk >> 5];
            if (!((w >> (rk & 31)) & 1u)) continue;
            const int y = tq0 + RT.qr(rk);
            const int2 old = out_r[y];
            if (old.x == -1) continue;              // (a dry order nobody could serve: committed as a reject already)
            const int4 rec = S.so_rec[y];
            const int slot = S.so_slot[y];
            out_r[y] = make_int2(-1, -1);
            if (slot >= 0) D.arr[arr_index(S.R, slot, r)] = pull_reject(t);
            if (D.sup != nullptr) {
                const int relo = old.y + rec.w;
                atomicSub(&D.sup[sup_index(S.C, S.R, t + (relo <= 0 ? 1 : ticks_until<true>(S, relo)), rec.z & 0xFFFF, r)], 1);
            }
            drej += 1; dw -= old.y; dv -= rec.w;
        }
        if (dw) atomicAdd(&s_dw, dw);
        if (dv) atomicAdd(&s_dv, dv);
        if (drej) atomicAdd(&s_drej, drej);
    }
    __syncthreads();
    }
    // ---- evaluations (:986-991 for the dry orders, :924 for the own-cluster ones), after the fact: with the final own matches
    //      lm of every bucket and the steal log {rank of the thief, cluster | orders of that cluster before the thief << 16},
    //        dry order of rank p:  sum over its visited clusters c of  m0_c - min(k_c(p), lm_c) - #steals from c by ranks < p
    //      (the own matches of a bucket are a prefix of its orders, and an exhaustion after p's turn only removes matches p never
    //      counted: min(k, lm at p's turn) = min(k, final lm));
    //        own-cluster orders of bucket c (its first lm_c):  lm*m0 - lm(lm-1)/2 - (steals*lm - tk),  tk = sum over the steals
    //      from c of min(k, lm_c) - a steal that came after k of the bucket's orders is seen by the lm - min(k, lm) matches behind it
    const int nlog = s_nlog;
    const int4 *slog = D.slog + (size_t)r * mto;
    for (int c = threadIdx.x; c < C; c += WK_THREADS) { const int ls = ls_l[c]; lm_l[c] = ls >> 16; sc_l[c] = ls & 0xFFFF; }
    __syncthreads();
    for (int i = threadIdx.x; i < nlog; i += WK_THREADS) {
        const int4 sl = slog[i];
        const int c = sl.y & 0xFFFF;
        atomicAdd(&tk_l[c], min((int)((unsigned)sl.y >> 16), lm_l[c]));
        if (!DN) out_r[tq0 + RT.qr(sl.x)] = make_int2(sl.z, sl.w);          // the served dry order's result
    }
    PROF_STAMP(29);
    {
        int acc = 0;        // per lane; reduced at the end
        // (a) per dry order: sum of m0 - min(k, lm) over its visit sequence; the dry orders are dealt round-robin to the wavefronts
        //     (lane l holds word l of the dry bits of a 64-word block), WK_EVA orders' rows in flight
        for (int wb = 0; wb < nwords; wb += WAVE) {
            const unsigned myw = wb + lane < nwords ? dry_bits[wb + lane] : 0u;
            int inc = __popc(myw);
            const int own = inc;
            for (int o = 1; o < WAVE; o <<= 1) { const int u = __shfl_up(inc, o, WAVE); if (lane >= o) inc += u; }
            const int excl = inc - own, total = rdlane(inc, WAVE - 1);
            for (int k0 = wave; k0 < total; k0 += WK_EVA * WK_WAVES) {
                unsigned vv[WK_EVA][JB];
#pragma unroll
                for (int u = 0; u < WK_EVA; ++u) {
                    const int kx = k0 + u * WK_WAVES;
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) vv[u][jb] = 0xFFFFFFFFu;
                    if (kx < total) {
                        const int hl = __ffsll((long long)ballot(excl <= kx && kx < excl + own)) - 1;      // the lane whose word holds the kx-th dry order
                        unsigned bits = (unsigned)rdlane((int)myw, hl);
                        for (int d = kx - rdlane(excl, hl); d > 0; --d) bits &= bits - 1u;
                        const int rk = (wb + hl) * 32 + __ffs((int)bits) - 1;
                        const unsigned *vis = S.so_vis + (size_t)(tq0 + RT.qr(rk)) * S.seq_pad;
#pragma unroll
                        for (int jb = 0; jb < JB; ++jb) vv[u][jb] = vis[jb * WAVE + lane];
                    }
                }
#pragma unroll
                for (int u = 0; u < WK_EVA; ++u)
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb)
                        if (vv[u][jb] != 0xFFFFFFFFu) { const int c = (int)(vv[u][jb] & 0xFFFFu); acc += m0_l[c] - min((int)(vv[u][jb] >> 16), lm_l[c]); }
            }
        }
        PROF_STAMP(30);
        // (b) the steals each dry order saw: a steal (thief of rank p', victim cluster c') is counted by every dry order of a LATER
        //     rank whose bucket's visit sequence holds c' (Static.vis_bits: the visit sets as a bit matrix).  Lane = log entry;
        //     the searching clusters with dry orders are dealt round-robin to the wavefronts, four membership words in flight;
        //     the dry orders of a bucket are its last ones, their ranks come from LDS (one address for the wavefront)
        for (int i0 = 0; i0 < nlog; i0 += WAVE) {
            int prk = IMAX, pcl = 0;
            if (i0 + lane < nlog) { const int4 s4 = slog[i0 + lane]; prk = s4.x; pcl = s4.y & 0xFFFF; }
            int taken = 0;          // searching clusters with dry orders met so far
            for (int cb = 0; cb < C; cb += WAVE) {
                const int cme = cb + lane;
                int qdm = 0, qem = 0;
                if (cme < C && (cdA_l[cme] & CAPABLE)) { qdm = (cme == 0 ? tq0 : qend_l[cme - 1]) + lm_l[cme]; qem = qend_l[cme]; }
                unsigned long long dm = ballot(qdm < qem);
#ifdef VDS_PROF
                if (prof && wave == 0 && lane == 0 && i0 == 0) g_prof[(size_t)pwave * PROF_SLOTS + 23] += popc64(dm);
#endif
                while (dm != 0ull) {
                    unsigned wd[4];
                    int qd4[4], qe4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        qd4[u] = 0; qe4[u] = 0; wd[u] = 0u;
                        while (dm != 0ull && (taken % WK_WAVES) != wave) { dm &= dm - 1ull; ++taken; }
                        if (dm != 0ull) {
                            const int l = __ffsll((long long)dm) - 1;
                            dm &= dm - 1ull; ++taken;
                            qd4[u] = rdlane(qdm, l); qe4[u] = rdlane(qem, l);
                            wd[u] = S.vis_bits[(size_t)(cb + l) * bmw + (pcl >> 5)];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool member = ((wd[u] >> (pcl & 31)) & 1u) != 0u;
                        for (int qb = qd4[u]; qb < qe4[u]; ++qb) acc -= (member && prk < RT.rq(qb - tq0)) ? 1 : 0;
                    }
                }
            }
        }
        PROF_STAMP(31);
        const int rs = row_sum_i32(acc);
        const int tot = rdlane(rs, 0) + rdlane(rs, 16) + rdlane(rs, 32) + rdlane(rs, 48);
        if (lane == 0 && tot != 0) atomicAdd(&s_ev, tot);
    }
    __syncthreads();
#ifdef WKDEBUG
    if (threadIdx.x == 0 && s_dbg != 0x7FFFFFFF && s_ev != s_dbg) printf("k_dfs_walk evals: r %d t %d post-hoc %d walk %d nlog %d\n", r, t, s_ev, s_dbg, nlog);
#endif
    if constexpr (DN) {
        // dense layout: the counters' difference to what k_tick_dense committed (it counted every bucket as if nothing were stolen:
        // min(k, m0) own matches, evaluations lm0 * m0 - lm0 (lm0 - 1) / 2), the alive counts of the lists that lost more than that,
        // and HDR_RAW for a list k_tick_dense had left compact
        int dev = 0;
        for (int c = threadIdx.x; c < C; c += WK_THREADS) {
            const int nm = lm_l[c], m0 = m0_l[c], sc = sc_l[c];
            const int kc = qend_l[c] - (c == 0 ? tq0 : qend_l[c - 1]);
            const int lm0 = min(kc, m0);
            dev += (nm * m0 - (nm * (nm - 1)) / 2 - (sc * nm - tk_l[c])) - (lm0 * m0 - (lm0 * (lm0 - 1)) / 2);
            if (nm != lm0 || sc != 0) {
                const size_t b = (size_t)c * S.R + r;
                D.hdr[b * HDR_WORDS + HDR_IDLE] = m0 - nm - sc;
                D.hdr[b * HDR_WORDS + HDR_RAW] = m0 + 1;
            }
        }
        if (dev) atomicAdd(&s_ev, dev);
        __syncthreads();
        if (threadIdx.x == 0) {
            long long *cnt = D.cnt + (size_t)r * CNT_WORDS;         // (cluster 0's row: the counters are summed over the clusters on read)
            if (s_drej) cnt[CNT_REJECTS] += s_drej;
            if (s_dw) cnt[CNT_WAIT] += s_dw;
            if (s_dv) cnt[CNT_VALUE] += s_dv;
            if (s_ev) cnt[CNT_EVALS] += s_ev;
            D.dry[r] = 0;
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += WK_THREADS) {
        const int nm = lm_l[c], m0 = m0_l[c], sc = sc_l[c];
        const int ev = nm * m0 - (nm * (nm - 1)) / 2 - (sc * nm - tk_l[c]);
        lm_l[c] = ev; sc_l[c] = m0 - nm - sc;                          // (lm / sc are dead now: reused as evaluations / final length)
    }
    __syncthreads();
    PROF_STAMP(5);
    // ---- commit: vehicle ids, arrivals (:954-960), counters (the rank tables are dead now)
    int *rc_l = tab_l;
    for (int i = threadIdx.x; i < RCNT * C; i += WK_THREADS) rc_l[i] = 0;
    __syncthreads();
    for (int qq = tq0 + (int)threadIdx.x; qq < tq1; qq += WK_RES * WK_THREADS) {
        int4 rec[WK_RES];
        int2 pr[WK_RES];
        int veh[WK_RES];
#pragma unroll
        for (int u = 0; u < WK_RES; ++u) {
            const int q = qq + u * WK_THREADS;
            rec[u] = make_int4(0, 0, 0, 0); pr[u] = make_int2(-1, -1);
            if (q < tq1) { rec[u] = S.so_rec[q]; pr[u] = out_r[q]; }
        }
        // ring positions of the arrivals: the counter atomics travel together with the vehicle-id gathers (they need the order's
        // record and its wait, not the vehicle), so the entries' stores are the third - not the fourth ... (2 + orders)-th - HBM
        // level of this pass.  rp: -1 no near arrival (unmatched, or the far inbox: post_arrival), else position | slots ahead << 16
        int rp[WK_RES];
#pragma unroll
        for (int u = 0; u < WK_RES; ++u) {
            veh[u] = -1; rp[u] = -1;
            if (pr[u].x != -1) {
                const int vc = (int)((unsigned)pr[u].x >> 16), vpos = pr[u].x & 0xFFFF;
#ifdef WKDEBUG
                if (vc >= C || vpos >= m0_l[vc < C ? vc : 0]) { printf("walk resolve: r %d t %d q %d (rank %d) pr %x %d vc %d vpos %d\n", r, t, qq + u * WK_THREADS - tq0, (int)S.so_rank[qq + u * WK_THREADS], pr[u].x, pr[u].y, vc, vpos); pr[u].x = -1; continue; }
#endif
                veh[u] = (int)D.idle[((size_t)vc * S.R + r) * S.idle_cap + vpos].x;
                const int rel = pr[u].y + rec[u].w;
                const int d = rel <= 0 ? 1 : ticks_until<false>(S, rel);
                if (d < S.H) {
                    const size_t i = ((size_t)((t + d) & (S.H - 1)) * S.C + (rec[u].z & 0xFFFF)) * S.R + r;
                    rp[u] = (atomicAdd(&D.ring_cnt[i], 0x10001) & 0xFFFF) | (d << 16);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WK_RES; ++u) {
            const int q = qq + u * WK_THREADS;
            if (q >= tq1) continue;
            int *cl = rc_l + (int)((unsigned)rec[u].z >> 16) * RCNT;
            atomicAdd(&cl[CNT_ORDERS], 1);
            if (pr[u].x == -1) {
                atomicAdd(&cl[CNT_REJECTS], 1);
            } else {
                out_r[q] = make_int2(veh[u], pr[u].y);
                const int dl = (int)((unsigned)rec[u].y >> 16);
                if (rp[u] >= 0) {
                    const int pos = rp[u] & 0xFFFF;
                    const size_t i = ((size_t)((t + (rp[u] >> 16)) & (S.H - 1)) * S.C + (rec[u].z & 0xFFFF)) * S.R + r;
                    if (pos >= S.ring_cap) atomicOr(&D.err[0], ERR_RING_CAP);
                    else D.ring[i * S.ring_cap + pos] = make_int4(veh[u], rec[u].x, now + pr[u].y + rec[u].w, meta_pack(t, 0, dl));
                } else post_arrival(S, D, rec[u].z & 0xFFFF, r, t, now, veh[u], rec[u].x, now + pr[u].y + rec[u].w, 0, dl);
                atomicAdd(&cl[CNT_WAIT], pr[u].y);
                atomicAdd(&cl[CNT_VALUE], rec[u].w);
            }
        }
    }
    // lists of 65 .. 128 entries that lost something: their clusters, for the batched compaction below (tk_l is dead)
    for (int c = threadIdx.x; c < C; c += WK_THREADS)
        if (m0_l[c] > WAVE && m0_l[c] <= 2 * WAVE && sc_l[c] != m0_l[c]) tk_l[atomicAdd(&s_nmid, 1)] = c;
    __syncthreads();
    PROF_STAMP(27);
    // ---- IdleVehicles.remove (:963), once per bucket: order-preserving compaction (survivors = free entries, node words clean)
    // (software-pipelined - the next batch's loads issued before this batch's stores, two batches of 8 - measured: 37.25 vs 36.88 ms
    // per day, the pass is bound by the memory system's throughput, not by its round trips: not kept)
    for (int c = wave; c < C; c += 12 * WK_WAVES) {        // (twelve lists in flight per wavefront)
        uint2 e4[12];
        bool keep4[12], small4[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int cu = c + u * WK_WAVES;
            const int mo = cu < C ? moff_l[cu] : 0, m0 = cu < C ? m0_l[cu] : 0;
            small4[u] = cu < C && m0 <= WAVE && sc_l[cu] != m0;
            keep4[u] = small4[u] && lane < m0 && (unsigned)st_l[mo + min(lane, max(m0, 1) - 1)] == WK_FREE;
            e4[u] = make_uint2(0u, 0u);
            if (keep4[u]) e4[u] = D.idle[((size_t)cu * S.R + r) * S.idle_cap + lane];
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const unsigned long long kb = ballot(keep4[u]);
            if (keep4[u]) D.idle[((size_t)(c + u * WK_WAVES) * S.R + r) * S.idle_cap + popc64(kb & lanemask_lt())] = e4[u];
        }
    }
    {   // lists of 65 .. 128 entries: six in flight per wavefront, both chunks of a list read before either is written
        const int nmid = s_nmid;
        for (int i0 = wave * 6; i0 < nmid; i0 += 6 * WK_WAVES) {
            uint2 e6[6][2];
            bool k6[6][2];
            int c6[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                c6[u] = i0 + u < nmid ? tk_l[i0 + u] : -1;
                const int cu = max(c6[u], 0);
                const int mo = moff_l[cu], m0 = m0_l[cu];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = h * WAVE + lane;
                    k6[u][h] = c6[u] >= 0 && i < m0 && (unsigned)st_l[mo + min(i, m0 - 1)] == WK_FREE;
                    e6[u][h] = make_uint2(0u, 0u);
                    if (k6[u][h]) e6[u][h] = D.idle[((size_t)cu * S.R + r) * S.idle_cap + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const unsigned long long kb0 = ballot(k6[u][0]), kb1 = ballot(k6[u][1]);
                uint2 *idle = D.idle + ((size_t)max(c6[u], 0) * S.R + r) * S.idle_cap;
                if (k6[u][0]) idle[popc64(kb0 & lanemask_lt())] = e6[u][0];
                if (k6[u][1]) idle[popc64(kb0) + popc64(kb1 & lanemask_lt())] = e6[u][1];
            }
        }
    }
    PROF_STAMP(28);
    {   // the lists of more than 128 entries that lost something (rare): found by a vote, dealt round-robin to the wavefronts
        int met = 0;
        for (int cb = 0; cb < C; cb += WAVE) {
            const int cme = cb + lane;
            unsigned long long lm = ballot(cme < C && m0_l[min(cme, C - 1)] > 2 * WAVE && sc_l[min(cme, C - 1)] != m0_l[min(cme, C - 1)]);
            for (; lm != 0ull; lm &= lm - 1ull, ++met) {
                if ((met % WK_WAVES) != wave) continue;
                const int c = cb + __ffsll((long long)lm) - 1;
                const int mo = moff_l[c], m0 = m0_l[c];
                uint2 *idle = D.idle + ((size_t)c * S.R + r) * S.idle_cap;
                int kept = 0;
                for (int base = 0; base < m0; base += 4 * WAVE) {      // four chunks read before the first is written (targets never lie ahead)
                    uint2 e4[4];
                    bool k4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = base + u * WAVE + lane;
                        e4[u] = make_uint2(0u, 0u);
                        k4[u] = i < m0 && (unsigned)st_l[mo + min(i, m0 - 1)] == WK_FREE;
                        if (k4[u]) e4[u] = idle[i];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned long long kb = ballot(k4[u]);
                        if (k4[u]) idle[kept + popc64(kb & lanemask_lt())] = e4[u];
                        kept += popc64(kb);
                    }
                }
            }
        }
    }
    PROF_STAMP(7);
    // ---- flush: list lengths and this tick's counter deltas (arrivals were counted by the fast kernel)
    for (int c = threadIdx.x; c < C; c += WK_THREADS) {
        const size_t b = (size_t)c * S.R + r;
        D.hdr[b * HDR_WORDS + HDR_IDLE] = sc_l[c];
        long long *cnt = D.cnt + b * CNT_WORDS;
#pragma unroll
        for (int w = 0; w < RCNT; ++w) { const int d = rc_l[c * RCNT + w]; if (d) cnt[w] += d; }
        const int ev = lm_l[c] + (c == 0 ? s_ev : 0);
        if (ev) cnt[CNT_EVALS] += ev;
    }
}

// ---------------------------------------------------------------------------------------
// k_build_vis: the per-order visit rows of the hybrid tick, built on the device at vds_load_orders* (round 5; the host loop - 100 MB
// of rows at configs[3], then their upload - was 56 of the 62 ms of a Reload, :130-212).  Thread (sorted order q, visit slot j):
//   so_vis[q][j] = j-th cluster c' of the visit sequence of q's pickup cluster (FindServerVehicleFunction :978-996, Static.dfs_seq)
//                  | orders of bucket (q's slot, c') with a smaller id << 16   (ranks ascend with the sorted position inside a bucket:
//                  a binary search over Static.so_rank), 0xFFFFFFFF past the end of the sequence;
//   so_lb[q][j]  = Static.lbc[pickup node of q][c'] (byte costs), 255 past the end.
// so_bkt0[q]: index into bkt_off of (q's day, q's slot, cluster 0).
__global__ __launch_bounds__(256) void k_build_vis(Static S, const int *so_bkt0, unsigned *so_vis, unsigned char *so_lb, long long n_orders) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long q = i / S.seq_pad;
    const int j = (int)(i % S.seq_pad);
    if (q >= n_orders) return;
    const int pc = (int)((unsigned)S.so_rec[q].z >> 16);
    const int s0 = S.dfs_off[pc], n = S.dfs_off[pc + 1] - s0;
    unsigned v = 0xFFFFFFFFu;
    unsigned char lb = 255;
    if (j < n) {
        const int c = S.dfs_seq[s0 + j];
        const int b0 = so_bkt0[q];
        int lo = S.bkt_off[b0 + c];
        int hi = S.bkt_off[b0 + c + 1];
        const int rk = S.so_rank[q];
        const int first = lo;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (S.so_rank[mid] < rk) lo = mid + 1; else hi = mid; }
        v = (unsigned)c | ((unsigned)(lo - first) << 16);
        if (so_lb) lb = S.lbc[(size_t)S.so_pnode[q] * S.C + c];
    }
    so_vis[i] = v;
    if (so_lb) so_lb[i] = lb;
}

// ---------------------------------------------------------------------------------------
// launchers (called from vds_api.hip)
// ---------------------------------------------------------------------------------------
void launch_build_vis(const Static &S, const int *so_bkt0, unsigned *so_vis, unsigned char *so_lb, long long n_orders, hipStream_t st) {
    const long long total = n_orders * S.seq_pad;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_build_vis, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S, so_bkt0, so_vis, so_lb, n_orders);
}

void launch_hybrid_rows(const Static &S, const State &D, int t, int lds_ints, hipStream_t st, int r_lo, int r_n);      // vds_tick.hip

void launch_tick_replica2(const Static &S, const State &D, int t, hipStream_t st) {
    const size_t lds = replica2_lds_ints(S.C, S.V, S.max_tick_orders) * sizeof(int);
    if (S.u8_ok) hipLaunchKernelGGL(k_tick_replica2<true>, dim3(S.R), dim3(REPL_THREADS), lds, st, S, D, t);
    else hipLaunchKernelGGL(k_tick_replica2<false>, dim3(S.R), dim3(REPL_THREADS), lds, st, S, D, t);
}

size_t dfs_walk_lds(const Static &S) { return dfs_walk_lds_bytes(S.C, S.V, S.max_tick_orders, S.walk_pool > 0 ? S.walk_pool : WK_NS_MIN, S.dense_st); }
// the record pool: WK_NS records, fewer (down to WK_NS_MIN) when that keeps the walk's workgroup within 40 KB - four per CU
int dfs_walk_pool(const Static &S) {
    int ns = WK_NS;
    while (ns > WK_NS_MIN && dfs_walk_lds_bytes(S.C, S.V, S.max_tick_orders, ns, S.dense_st) + 128 > 40 * 1024) --ns;
    return ns;
}

static void emit_walk(const Emit &e, void (*k)(Static, State, int), dim3 grid, dim3 block, size_t lds, Static S, State D, int t) {
    if (!e.graph) { hipLaunchKernelGGL(k, grid, block, lds, e.st, S, D, t); return; }
    void *args[3] = {&S, &D, &t};
    hipKernelNodeParams p{};
    p.func = reinterpret_cast<void *>(k); p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds; p.kernelParams = args; p.extra = nullptr;
    *e.err = hipGraphAddKernelNode(e.node, e.graph, e.deps, e.ndeps, &p);
}

// second half of the hybrid neighbour-search tick for the replicas [r_lo, r_lo + r_n) (r_n = 0: all).  vds_run launches the tick
// per GROUP of replicas as parallel branches of the day graph, so that the stamp-mode kernel of one group (VALU-bound) runs
// under the walk of another (a per-replica dependency chain that leaves the CUs mostly idle).
void emit_hybrid_walk(const Emit &e, const Static &S0, const State &D, int t, int r_lo, int r_n) {
    Static S = S0;
    S.r_lo = r_lo;
    const dim3 grid(r_n > 0 ? r_n : S.R);
    if (S.dense_st) {
        // the dense layout's walk (byte costs: vds_api.hip grants the stamp form only then)
        emit_walk(e, S.seq_pad <= 64 ? k_dfs_walk<true, 1, true> : (S.seq_pad <= 128 ? k_dfs_walk<true, 2, true> : k_dfs_walk<true, 4, true>), grid, dim3(WK_THREADS), dfs_walk_lds(S), S, D, t);
        return;
    }
    if (S.u8_ok) emit_walk(e, S.seq_pad <= 64 ? k_dfs_walk<true, 1> : (S.seq_pad <= 128 ? k_dfs_walk<true, 2> : k_dfs_walk<true, 4>), grid, dim3(WK_THREADS), dfs_walk_lds(S), S, D, t);
    else emit_walk(e, S.seq_pad <= 64 ? k_dfs_walk<false, 1> : (S.seq_pad <= 128 ? k_dfs_walk<false, 2> : k_dfs_walk<false, 4>), grid, dim3(WK_THREADS), dfs_walk_lds(S), S, D, t);
}

void launch_hybrid_walk(const Static &S, const State &D, int t, hipStream_t st, int r_lo, int r_n) {
    Emit e; e.st = st;
    emit_hybrid_walk(e, S, D, t, r_lo, r_n);
}

void launch_tick_hybrid(const Static &S, const State &D, int t, int lds_ints, hipStream_t st) {
    launch_hybrid_rows(S, D, t, lds_ints, st, 0, 0);
    launch_hybrid_walk(S, D, t, st, 0, 0);
}

void launch_match_dfs(const Static &S, const State &D, int t, hipStream_t st) {
    hipLaunchKernelGGL(k_match_dfs, dim3(S.R), dim3(64), 0, st, S, D, t);
}

}  // namespace vds
