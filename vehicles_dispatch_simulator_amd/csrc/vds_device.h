// Device-side data model of libvds (gfx950 / CDNA4 only).
//
// Everything is Struct-of-Arrays in HBM, laid out CLUSTER-major: the unit of parallel work
// is one (cluster, replica) "bucket"; a workgroup owns one cluster for a run of consecutive
// replicas, so the cluster's cost block sits in LDS once and every bucket access is a
// contiguous, coalesced segment.
//
//   hdr      [C][R][8]           int32  bucket header (HDR_* below)
//   cnt      [C][R][8]           int64  bucket-owned partial counters (no atomics; reduced on read)
//   idle     [C][R][idle_cap]    {veh, loc_local}   Cluster.IdleVehicles, kept in LIST ORDER
//   ring     [H][C][R][ring_cap] arrival entries indexed by ARRIVAL TICK mod H: the in-flight part of
//                                Cluster.VehiclesArrivetime.  A tick only touches the slot that is
//                                due; other buckets append with one atomic on ring_cnt.
//   ring_cnt [H][C][R]           low 16 bits: entries in the slot, high 16: those carrying an order
//                                (SupplyExpect of the previous tick reads the high half)
//   fl/inbox                     "far" entries whose arrival lies >= H ticks ahead (rare): owner-private
//                                list + double-buffered inbox, migrated into the ring when they get near
//   out      [R][Oq] {veh, wait} per-order result, indexed by bucket-sorted order position q
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vds {

enum {
    HDR_IDLE = 0,       // len(IdleVehicles)
    HDR_IDLE_PRE = 1,   // PerMatchIdleVehicles of the last tick
    HDR_ORDERS = 2,     // len(Cluster.Orders) of the last tick
    HDR_FL = 3,         // far list fill
    HDR_INBOX0 = 4,     // far inbox fill, parity 0 (atomic target of other buckets)
    HDR_INBOX1 = 5,     // far inbox fill, parity 1
    HDR_WORDS = 8
};

enum {
    CNT_ORDERS = 0, CNT_REJECTS = 1, CNT_WAIT = 2, CNT_VALUE = 3,
    CNT_EVALS = 4, CNT_ARRIVALS = 5, CNT_DISPATCH = 6, CNT_DISPATCH_COST = 7, CNT_WORDS = 8
};

// sticky device error bits (err[0])
enum { ERR_IDLE_CAP = 1, ERR_FL_CAP = 2, ERR_INBOX_CAP = 4, ERR_DISPATCH = 8, ERR_RING_CAP = 16 };

// arrival entry: x = vehicle, y = id (order id, or dispatch sequence number), z = arrival minute,
// w = meta = insert_tick << 16 | is_dispatch << 15 | dest_local.
// Dict insertion order == ascending key (insert_tick, is_dispatch, id).
__host__ __device__ inline int meta_pack(int tick, int is_dispatch, int dest_local) {
    return (tick << 16) | (is_dispatch << 15) | dest_local;
}
__host__ __device__ inline int meta_dest(int w) { return w & 0x7FFF; }
__host__ __device__ inline int meta_is_dispatch(int w) { return (w >> 15) & 1; }
__host__ __device__ inline unsigned long long entry_key(int id, int w) {
    return ((unsigned long long)((unsigned)w >> 15) << 32) | (unsigned)id;
}

// sorted order record: x = order id, y = pick_local | dest_local << 16,
// z = dest_cluster | pick_cluster << 16, w = OrderValue
// One order day.  A handle holds n_days of them (vds_load_order_days); replica r replays day replica_day[r] - the
// reference's one Simulation == one city == its own Orders (simulator.py:325-342).  The per-day tables are concatenated;
// bkt_off / tick_off / ord_q hold ABSOLUTE positions (into so_rec / ord_q), results live at out[r][q - q_base].
struct DayDesc {
    int bkt_base;         // start of the day's [T*C + 1] slice of bkt_off
    int tick_base;        // start of the day's [T + 1] slice of tick_off
    int now0;             // RealExpTime at tick 0 (:1037)
    int T;                // iterations of `while self.RealExpTime <= EndTime` (:1048) of this day: the city stands still afterwards
    int q_base;           // first sorted-order position of the day
    int Oq;               // processed orders of the day
    int max_tick_orders;
    int pad;
};
struct DayView { const int *bkt_off; const int *tick_off; int now0, T, q_base; };

struct Static {
    int N, C, V, R, Oq, T;           // R: replicas the tables hold; Oq: result slots per replica (max over days); T: longest day
    int R_ext;                       // replicas the caller sees (vds_config.replicas); R >= R_ext when the replicas are stored regrouped
    const int *int2ext;              // [R] stored ("internal") replica -> the caller's replica index, -1 for a padding replica that
                                     // never runs; null: identity.  Order days per replica with a map that mixes days inside aligned
                                     // groups of 16 replicas: the library stores the replicas grouped by day (vds_api.hip)
    int n_days;                      // 1: one order stream shared by every replica (the fast path of k_tick_rows)
    const int *rperm;                // [rslots] k_tick_rows row slot -> replica, grouped by order day in groups of 16 (-1 padding); null: identity
    int rslots;
    int chunk_days;                  // n_days > 1 and every aligned group of 16 replicas (one k_tick_rows workgroup) replays one day
    const DayDesc *day;              // [n_days]
    const int *replica_day;          // [R]
    const int4 *replica_desc;        // [R] {bkt_base, now0, T, q_base} of the replica's day: one load instead of replica_day -> day[]
    int tick_minutes, now0;          // RealExpTime at tick 0
    unsigned tick_magic;             // ceil(2^32 / tick_minutes): n / tick == mulhi(n, magic) for 0 <= n < tick_div_limit
    int tick_div_limit;              // (0: always divide)
    long long reject_threshold;
    int idle_cap, fl_cap, in_cap;    // fl_cap / in_cap: far list / far inbox
    int H, ring_cap;                 // arrival ring: H ticks (power of two) x ring_cap entries
    int fast_ok;                     // costs fit the packed (cost << 7 | pos) fast kernel and never exceed the reject threshold
    int window_live;                 // some cost exceeds the reject threshold: `cost > PICKUPTIMEWINDOW` (:943) can really reject
    int max_nc;
    const int *cost;                 // [N*N] rows = node, COLUMNS cluster-contiguous: column cl_off[c] + loc_local
    const int *node2cluster;         // [N]
    const int *node_local;           // [N]
    const int *cl_off;               // [C+1]
    const int *cl_nodes;             // [sum n_c]
    const long long *blk_off;        // [C+1] (16-byte aligned starts)
    const int *blk;                  // blk[c][p][l] = cost[node_p * N + node_l], each block padded to 4 ints
    const int4 *cdesc;               // [C] {n_c, blk_off, blk8_off, 0}: one load per workgroup
    const int *corder;               // [C] clusters by descending size: heavy workgroups are dispatched first
    const int4 *cdesc_ord;           // [C] {n_c, blk_off, cluster, blk8_off} in that order: one dependent load less per workgroup
    const unsigned char *blk8;       // the same cost blocks as bytes (16-byte aligned starts) when every cost is <= 255
    const unsigned char *cost8;      // byte copy of `cost` (same column order) under the same condition
    int u8_ok;
    const unsigned char *lbc;        // [N][C] min over the nodes n of cluster c of cost8[p*N + col(n)]: lower bound of RoadCost(vehicle in c, pickup p)
                                     // (neighbour-search mode with byte costs; nullptr otherwise)
    const int *dfs_off;              // [C+1]
    const int *dfs_seq;              // visit sequence excluding the start cluster
    const int4 *so_rec;              // [Oq]
    const int *bkt_off;              // [T*C + 1]
    const int *tick_off;             // [T+1] into ord_q
    const int *ord_q;                // [Oq] processed orders in id order -> q
    const int *so_pnode;             // [Oq] pickup NODE of each sorted order (row of the cost matrix the DFS scans)
    int walk_pool;                   // scan records in k_dfs_walk's LDS pool (8..32)
    const unsigned *so_vis;          // [Oq][seq_pad] j-th cluster of the sorted order's visit sequence | orders of that cluster of the same slot with a smaller id << 16 (0xFFFFFFFF past the end; nullable)
    const unsigned char *so_lb;      // [Oq][seq_pad] lbc[pickup node of the sorted order][j-th cluster of its visit sequence] (nullable)
    int seq_pad;                     // longest visit sequence, rounded up to a multiple of 64
    const int *so_rank;              // [Oq] rank of the sorted position inside its slot, in id order (= cursor order of :912-973)
    int max_tick_orders;             // most orders processed in one tick
    int r_lo;                        // hybrid tick launched for a GROUP of replicas (vds_run: groups on streams, the stamp-mode k_tick_rows
                                     // of one group under the k_dfs_walk of another): first replica of the launch, a multiple of 16
    // ---- layout T ("lanes" tick, k_tick_lanes in vds_lanes.hip): lane = replica.  Per-replica tables are transposed inside
    // groups of 64 replicas so that the 64 lanes of a wavefront - the same cluster in 64 consecutive replicas - touch
    // consecutive addresses:
    //   idle  u32 {veh << 8 | loc_local}   [C][G][idle_cap / 4][64][4]   (entry e of replica r: quad e / 4, column r % 64, element e % 4)
    //   ring  int4 (unchanged entry)       [H][C][G][ring_cap][64]
    //   out   int2 {veh, wait}             [Oq][64 G]
    // hdr / cnt / ring_cnt / fl / inbox keep their [C][R] layouts.
    int layoutT;                     // 1: the tables above are in layout T
    int G;                           // groups of 64 replicas = ceil(R / 64)
    const int4 *cdesc_lanes;         // [C] {n_c | log2(lanes per bucket) << 16, offset into blk8s, cluster, row stride n_c + 1}, heaviest first
    const int2 *lane_blocks;         // [lane_nblocks] {index into cdesc_lanes, wavefront index inside the cluster}
    int lane_nblocks;
    const unsigned char *blk8s;      // per-cluster byte cost blocks with row stride n_c + 1; column n_c holds 0xFF (the cost of a taken / absent entry)
    int lane_loc_slots, lane_key_slots;   // per-lane LDS capacities: idle entries / arrivals per lane (x lanes per bucket)
    int lane_force_slow;             // testing: every wavefront takes the table-free slow path
    int lane_ablate;                 // timing experiments only (results INVALID when non-zero): bit0 no counter atomics, bit1 no result
                                     // stores, bit2 no arrival posts, bit3 no list write-back, bit4 no header store
};

// where a host-side launcher puts its kernel: on a stream, or as a kernel node of an explicitly built hipGraph
struct Emit {
    hipStream_t st = nullptr;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t *deps = nullptr;
    size_t ndeps = 0;
    hipGraphNode_t *node = nullptr;
    hipError_t *err = nullptr;
};

struct State {
    int *hdr;
    long long *cnt;
    uint2 *idle;
    int4 *ring;
    int *ring_cnt;
    int4 *fl;
    int4 *inbox;
    int4 *slog;          // [R][max_tick_orders] steal log of the hybrid neighbour-search tick: {rank of the thief, cluster | orders of that cluster before the thief << 16, the thief's result}
    int2 *out;
    int *err;
    int *work;   // [2] deferred-bucket counters by tick parity, then [2][C*R] bucket indices
};

// ---- layout T addressing (host and device)
// idle entry e of (c, r): ((unsigned *)D.idle)[idleT_base + (e >> 2) * 256 + (e & 3)]
__host__ __device__ inline size_t idleT_base(const Static &S, int c, int r) {
    return ((((size_t)c * S.G + (size_t)(r >> 6)) * (size_t)(S.idle_cap >> 2)) * 64 + (size_t)(r & 63)) * 4;
}
__host__ __device__ inline size_t idleT_elem(int e) { return (size_t)(e >> 2) * 256 + (size_t)(e & 3); }
// ring entry i of (slot, c, r): D.ring[ringT_base + i * 64]
__host__ __device__ inline size_t ringT_base(const Static &S, int slot, int c, int r) {
    return ((((size_t)slot * S.C + c) * S.G + (size_t)(r >> 6)) * (size_t)S.ring_cap) * 64 + (size_t)(r & 63);
}
__host__ __device__ inline unsigned idleT_pack(unsigned veh, unsigned loc) { return (veh << 8) | (loc & 0xFFu); }

// One (cluster, replica) idle list in either layout: what the kernels shared by both layouts (reset, dispatch) go through.
struct IdleRef {
    uint2 *p;        // layout 0: {veh, loc_local} entries, contiguous
    unsigned *q;     // layout T (non-null): packed entries, see Static
    __device__ __forceinline__ uint2 get(int e) const {
        if (!q) return p[e];
        const unsigned v = q[idleT_elem(e)];
        return make_uint2(v >> 8, v & 0xFFu);
    }
    __device__ __forceinline__ void set(int e, uint2 v) const {
        if (!q) p[e] = v;
        else q[idleT_elem(e)] = idleT_pack(v.x, v.y);
    }
};
__device__ __forceinline__ IdleRef idle_ref(const Static &S, const State &D, int c, int r) {
    IdleRef f;
    if (S.layoutT) { f.p = nullptr; f.q = reinterpret_cast<unsigned *>(D.idle) + idleT_base(S, c, r); }
    else { f.p = D.idle + ((size_t)c * S.R + r) * S.idle_cap; f.q = nullptr; }
    return f;
}

__device__ __forceinline__ DayView day_view(const Static &S, int r) {
    if (S.n_days <= 1) return DayView{S.bkt_off, S.tick_off, S.now0, S.T, 0};
    const DayDesc d = S.day[S.replica_day[r]];
    return DayView{S.bkt_off + d.bkt_base, S.tick_off + d.tick_base, d.now0, d.T, d.q_base};
}
// results of replica r, indexed by ABSOLUTE sorted-order position q
__device__ __forceinline__ int2 *out_row(const Static &S, const State &D, int r) {
    const int qb = S.n_days <= 1 ? 0 : S.day[S.replica_day[r]].q_base;
    return D.out + (size_t)r * S.Oq - qb;
}

}  // namespace vds
