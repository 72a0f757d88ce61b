// Device-side data model of libvds (gfx950 / CDNA4 only).
//
// Everything is Struct-of-Arrays in HBM, laid out CLUSTER-major: the unit of parallel work
// is one (cluster, replica) "bucket"; a workgroup owns one cluster for a run of consecutive
// replicas, so the cluster's cost block sits in LDS once and every bucket access is a
// contiguous, coalesced segment.
//
//   hdr      [C][R][16]          int32  bucket header (HDR_* below) + 32-bit partial counters of the dense fast path
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
// The plain tick (no neighbour search) runs on the DENSE variant of idle / ring (4-byte / 8-byte entries): see "Dense layout".
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
    HDR_RAW = 6,        // dense neighbour-search tick (Static.dense_st): 0, or 1 + the list's RAW length - entries taken during the
                        // last slot (stamp != free, State.stamp) are still inside, positions [0, raw); HDR_IDLE counts the ones alive.
                        // Every other writer of a header leaves 0 here ("the list is compact") and needs to know nothing about it
    // words 8 .. 13: 32-bit partial counters (CNT_ORDERS .. CNT_ARRIVALS) of the dense tick's byte-cost fast path (round 5): the bucket's
    // header and its counters are ONE 64-byte record - one read request and one write request per bucket and tick instead of two
    // each (header 32 B + counters 64 B).  Per tick a bucket adds at most 64 orders x 256 candidates / 254 minutes, a day has at most
    // 65 535 slots: the sums stay below 2^31.  Every other path adds to the 64-bit table `cnt`; k_reduce_counters sums both.
    HDR_CNT32 = 8,
    HDR_WORDS = 16
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

struct State;
struct Static {
    int N, C, V, R, Oq, T;           // R: replicas the tables hold; Oq: result slots per replica (max over days); T: longest day
    int R_ext;                       // replicas the caller sees (vds_config.replicas); R >= R_ext when the replicas are stored regrouped
    const int *int2ext;              // [R] stored ("internal") replica -> the caller's replica index, -1 for a padding replica that
                                     // never runs; null: identity.  Order days per replica with a map that mixes days inside aligned
                                     // groups of 16 replicas: the library stores the replicas grouped by day (vds_api.hip)
    int n_days;                      // 1: one order stream shared by every replica (the fast path of k_tick_rows)
    const int *rperm;                // [rslots] k_tick_rows row slot -> replica, grouped by order day in groups of 16 (-1 padding); null: identity
    int rslots;
    int chunk_days;                  // n_days > 1 and every aligned group of row_gran replicas (one workgroup of the fast kernels) replays one day
    int row_gran;                    // 16; 8 / 4 on the dense layout when the days come in aligned groups of eight / four replicas (k_tick_dense with 8- / 4-row
                                     // workgroups, round 5: eight replicas per day no longer mean one order stream per row)
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
    const unsigned *vis_bits;        // [C][(C + 31) / 32] bit c' of row c: cluster c' is in the visit sequence of c (k_dfs_walk's evaluation pass)
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
    // ---- dense layout (the plain tick k_tick_dense, vds_tick_dense.hip; see "Dense layout" below)
    int dense;                       // 1: idle / ring hold packed entries
    int dense_lpr;                   // lanes per replica of k_tick_dense: 16 / 8 / 4 (4 / 8 / 16 replicas per wavefront)
    int dense_tab, dense_keys;       // idle entries / arrivals per bucket its fast path takes (<= 128 / 64; smaller: tests)
    int dense_force_slow;            // testing: every bucket takes dense_bucket_slow
    const Static *self_dev;          // device-resident copy of this struct and of the State next to it (vds_api.hip keeps them current):
    const struct State *state_dev;   // what the rarely taken slow path of k_tick_dense reads instead of by-value kernel arguments
    // static arrival slots ("pull", see below): 1 = order-carrying arrivals go through D.arr instead of the ring
    int pull, pull_W, pull_hmax;     // W: most slots an arrival can lie behind its earliest slot a0; hmax: longest trip in slots (dmin)
    int arr_slots;                   // slots of the longest day: D.arr is [arr_slots][R] (arr_index).  (A replica-major table for day
                                     // mode 2 - a row's candidate slots as one run - was built and measured in round 4: 19.92 vs 19.88 ms
                                     // per day with 128 days, 22.53 vs 22.54 with 1024: no difference, not kept)
    const int *so_slot;              // [Oq] per sorted order: slot index in its day's D.arr rows (-1: ring / far path)
    const int2 *d_rec;               // per slot, sorted by (destination cluster, a0, id): {dense_key(insert tick, 0, id), a0 | dest_local << 16 | dmin << 24}
    const int *d_first;              // per day [(TA + 1) x C]: first slot (absolute d_rec position) of cluster c with a0 >= a
    const int4 *replica_desc2;       // [R] {d_first base, d_rec base, TA, slots} of the replica's day
    int ring_min_on;                 // (dense) State.ring_min exists: arrival minutes of dispatched vehicles, for the container views
    const int *blk32s;               // (dense, costs beyond a byte) per-cluster int cost blocks with row stride n_c + 1, column n_c = DENSE_DEAD_COST
    const unsigned char *blk8s;      // (dense, byte costs <= 254) the same as bytes, column n_c = 0xFF: the loc byte of a taken / absent
                                     // entry names that column, so the entry loses every comparison without a test in the match loop
    const int4 *cdesc_dense;         // [C] {n_c, offset into blk8s / blk32s, cluster, 0}, heaviest cluster first
    int dense_st;                    // 1: neighbour search on the dense layout (round 6): k_tick_dense in STAMP form + k_dfs_walk reading dense lists
                                     //    (taken entries stay in the list, stamped with the rank of their order, until the NEXT slot's tick drops
                                     //    them while it loads the list; dry orders are counted per replica in State.dry)
    const int4 *tdesc;               // (dense, one shared order day) [T][C][2] in cdesc_dense's cluster order: {first sorted order, orders, first
                                     // candidate slot, candidate slots} of the (slot, cluster) bucket - one scalar load next to cdesc_dense's
                                     // instead of four that depend on it (the workgroup's prologue is a chain of dependent loads)
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
    int4 *ring;          // wide layout: int4 entries; dense layout: uint2 {veh << 8 | dest_local, key} (ring2())
    int *ring_cnt;
    int4 *fl;
    int4 *inbox;
    int4 *slog;          // [R][max_tick_orders] steal log of the hybrid neighbour-search tick: {rank of the thief, cluster | orders of that cluster before the thief << 16, the thief's result}
    int2 *out;
    int *err;
    int *work;   // [2] deferred-bucket counters by tick parity, then [2][C*R] bucket indices
    unsigned *arr;       // dense layout with static arrival slots: [slots of the longest day][R] pull_entry / pull_reject (arr_index)
    unsigned short *stamp; // dense neighbour-search tick: [C][R][idle_cap] per idle entry 0xFFFF (alive) or the rank - position in id order inside
                         // the slot - of the order that took it during the last slot; valid for positions below the list's raw length
    int *slow_tick;      // dense layout: [T] buckets that left k_tick_dense's fast path, per slot of the running episode (what the per-slot
                         // choice between its two forms is made from: vds_api.hip adapt_dense)
    int *dry;            // dense neighbour-search tick: [R] buckets of the replica whose orders outran the list in a searching cluster
                         // during this slot (k_tick_dense adds, k_dfs_walk reads and clears: 0 = nothing to walk)
    int *sup;            // dense layout: [VDS_SUP_PLANES][C][R] SupplyExpect kept in place (:880-891) - plane a & 31 counts the order-carrying vehicles that
                         // will arrive in (cluster, replica) at slot a: bumped by a no-return atomic where the arrival is posted (match time), cleared
                         // by the bucket's own tick at slot a.  After the tick of slot t, plane (t + 1) & 31 IS SupplyExpect of slot t
    int *sup_slot;       // [1] (t + 1) & 31 of the last tick launched: which plane a device-side policy reads (vds_supply_inplace)
    int *ring_min;       // dense layout: [H][C][R][ring_cap] arrival minute of a DISPATCHED vehicle's entry (order-carrying entries: recomputed
                         // from the order's result on the read side); written by the dispatch kernels only, never read by a tick
};

// ---- Dense layout (Static.dense): what the plain tick (no neighbour search) runs on when V < 2^24, every cluster has at most
// 255 nodes, H <= 32 and order ids / dispatch sequence numbers of one slot stay below 2^25 - otherwise the wide layout above.
//   idle   u32  [C][R][idle_cap]      {veh << 8 | loc_local}
//   ring   uint2 [H][C][R][ring_cap]  {veh << 8 | dest_local, key}, key = (insert_tick & 63) << 26 | is_dispatch << 25 | id
//          (id: Order.ID, or the dispatch's sequence number inside its slot).  Dict insertion order = ascending
//          (insert_tick, is_dispatch, id); the entries of one arrival slot were inserted during the last H <= 32 ticks, so
//          key - ((t - 32) << 26) (mod 2^32) orders them (dense_key_rel).
//   ring_min int [H][C][R][ring_cap]  arrival minute, DISPATCH entries only
// hdr / cnt / ring_cnt / fl / inbox / out keep their layouts (far entries stay int4 {veh, id, arrive, meta}).
// Static arrival slots ("pull").  An order's arrival slot is its (static) processing slot + d, d = ceil((PickupWaitTime + OrderValue) /
// slot length) (:954-960): only the wait - at most the largest cost inside the pickup cluster - is dynamic, so the arrival lies in
// [a0, a0 + W] with a0 = processing slot + d(wait 0) static.  Every processed order owns ONE u32 per replica in D.arr (arr_index),
// slots sorted by (destination cluster, a0, id): the matching bucket stores pull_entry(veh, arrival slot) (pull_reject: no
// vehicle) with a plain store - no atomic, no position to wait for, 16 consecutive replicas = one 64-byte line - and the
// destination bucket of slot t reads the entries of its orders with a0 in [t - W, t] (a contiguous range, known from static
// tables) and takes those whose arrival-slot byte is t's (every candidate was processed less than 128 slots ago).  Dict insertion order = the orders' (insert tick, id) keys (static, d_rec).  Dispatched
// vehicles (hooks) and orders whose trip may outlive the ring horizon keep the ring / far path.  Used when W <= DENSE_PULL_WMAX.
#define DENSE_PULL_WMAX 3
#define VDS_SUP_PLANES 32           // arrival slots ahead the supply planes hold (>= the ring horizon of the dense layout, H <= 32)
__host__ __device__ inline size_t sup_index(int C, int R, int slot, int c, int r) { return ((size_t)(slot & (VDS_SUP_PLANES - 1)) * C + c) * R + r; }
// D.arr is [arr_slots][R] u32: the candidates of a bucket are consecutive slots, a workgroup (16 / 32 consecutive replicas) reads
// and writes 64 / 128 contiguous bytes per slot, and the 4 KB of one slot's R = 1024 replicas are read by the cluster's workgroups
// at about the same time.  (Blocked by 32 replicas - [R / 32][arr_slots][32], every workgroup streaming its own contiguous run of
// n slots - was built and measured in round 4: configs[1] 7.05 vs 6.92 ms per day on one box, 16 days 8.67 vs 8.84: not kept.)
__host__ __device__ inline size_t arr_index(int R, int slot, int r) { return (size_t)slot * (size_t)R + (size_t)r; }
__host__ __device__ inline unsigned pull_entry(int veh, int arrival_slot) { return ((unsigned)veh << 8) | ((unsigned)arrival_slot & 0xFFu); }
// rejected at slot t: a byte no slot in (t, t + 128) has
__host__ __device__ inline unsigned pull_reject(int t) { return 0xFFFFFF00u | ((unsigned)(t + 128) & 0xFFu); }
__host__ __device__ inline bool pull_is_reject(unsigned e) { return (e >> 8) == 0xFFFFFFu; }
// slots between `now_slot` and the entry's arrival (<= 0: arrived), for entries processed at most 127 slots before now_slot
__host__ __device__ inline int pull_slots_ahead(unsigned e, int now_slot) { return (int)(signed char)((e & 0xFFu) - ((unsigned)now_slot & 0xFFu)); }
#define DENSE_DEAD_COST 0x00FFFFFF      // cost of the dead column of an int block: (cost << 7 | pos) stays a positive int
#define DENSE_ID_BITS 25
__host__ __device__ inline unsigned dense_pack(unsigned veh, unsigned loc) { return (veh << 8) | (loc & 0xFFu); }
__host__ __device__ inline unsigned dense_key(int tick, int is_dispatch, int id) {
    return ((unsigned)(tick & 63) << 26) | ((unsigned)(is_dispatch & 1) << 25) | ((unsigned)id & ((1u << DENSE_ID_BITS) - 1));
}
// comparable form of a key read at tick t (insert ticks in (t - 32, t])
__host__ __device__ inline unsigned dense_key_rel(unsigned key, int t) { return key - ((unsigned)((t - 32) & 63) << 26); }
__host__ __device__ inline int dense_key_is_dispatch(unsigned key) { return (int)((key >> 25) & 1u); }
__host__ __device__ inline int dense_key_id(unsigned key) { return (int)(key & ((1u << DENSE_ID_BITS) - 1)); }
// insert tick of an entry seen at tick t_now (the slot it sits in is due after t_now, it was inserted at most H - 1 ticks ago)
__host__ __device__ inline int dense_key_tick(unsigned key, int t_now) { return t_now - (int)(((unsigned)t_now - (key >> 26)) & 63u); }

// One (cluster, replica) idle list in either layout: what the kernels shared by both layouts (reset, dispatch) go through.
struct IdleRef {
    uint2 *p;        // wide layout: {veh, loc_local} entries
    unsigned *q;     // dense layout (non-null): packed entries
    __device__ __forceinline__ uint2 get(int e) const {
        if (!q) return p[e];
        const unsigned v = q[e];
        return make_uint2(v >> 8, v & 0xFFu);
    }
    __device__ __forceinline__ void set(int e, uint2 v) const {
        if (!q) p[e] = v;
        else q[e] = dense_pack(v.x, v.y);
    }
};
__device__ __forceinline__ IdleRef idle_ref(const Static &S, const State &D, int c, int r) {
    IdleRef f;
    const size_t off = ((size_t)c * S.R + r) * S.idle_cap;
    if (S.dense) { f.p = nullptr; f.q = reinterpret_cast<unsigned *>(D.idle) + off; }
    else { f.p = D.idle + off; f.q = nullptr; }
    return f;
}
__device__ __forceinline__ uint2 *ring2(const State &D) { return reinterpret_cast<uint2 *>(D.ring); }

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
