// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (and the TCC_EA0_* request counters behind them) on gfx950 against KNOWN byte
// counts in the access patterns of k_tick_dense / k_tick_rows ("a calibrated byte count").  MI355X_MICROARCH.md (HBM
// section) calibrates only wide coalesced streaming reads (FETCH_SIZE reports half the bytes) and says every other width, and
// WRITE_SIZE, must be calibrated in one's own pattern.  Every kernel below moves a byte count that is known in closed form; the
// driver prints it as `useful` (bytes the lanes ask for), `sectors` (distinct 32-byte sectors touched x 32) and `lines` (distinct
// 128-byte lines x 128), one line per kernel; profiles/ubench/bytes_calib.py joins that with the per-kernel counter means of
//   rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ TCC_EA0_RDREQ_32B | TCC_EA0_WRREQ TCC_EA0_WRREQ_64B  (separate passes).
// All buffers are far larger than the 32 MiB of L2 and every byte is touched once per launch: what the L2 asks the fabric for is
// what the kernel asks for, rounded to the request granularity - the thing being calibrated.
//   hipcc --offload-arch=gfx950 -O3 -o bytes_calib bytes_calib.hip && ./bytes_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// (a) wide coalesced streaming read / write: 16 B per lane, consecutive lanes consecutive addresses (the guide's case)
__global__ __launch_bounds__(256) void cal_stream_read16(const uint4 *p, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void cal_stream_write16(uint4 *p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4(v, v + 1, v + 2, v + 3);
}
// (b) the idle-list read of a row group: rows of `cap` u32 ([C][R][idle_cap] u32: row stride cap * 4 bytes), LPR lanes read the first
// m entries of their row as 16-byte chunks (lane l: entries 4l .. 4l+3), rows of one workgroup are consecutive
template <int LPR, int M>
__global__ __launch_bounds__(256) void cal_row_read(const unsigned *idle, int rows, int cap, unsigned *sink) {
    const int m = M;
    const int lg = threadIdx.x % LPR;
    unsigned acc = 0;
    for (int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < rows; row += gridDim.x * (256 / LPR)) {
        for (int e = lg * 4; e < m; e += LPR * 4) { const uint4 v = *reinterpret_cast<const uint4 *>(idle + (size_t)row * cap + e); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
// (c) the same as a write-back of the first m entries
template <int LPR, int M>
__global__ __launch_bounds__(256) void cal_row_write(unsigned *idle, int rows, int cap, unsigned v) {
    const int m = M;
    const int lg = threadIdx.x % LPR;
    for (int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < rows; row += gridDim.x * (256 / LPR))
        for (int e = lg * 4; e < m; e += LPR * 4) *reinterpret_cast<uint4 *>(idle + (size_t)row * cap + e) = make_uint4(v, v, v, v + e);
}
// (d) static arrival slots D.arr[slot][R]: 16 consecutive replicas x 4 B = one 64-byte segment per (slot, replica chunk), slots
// scattered (every (slot, chunk) pair touched exactly once per launch: a permutation through an odd multiplier)
__global__ __launch_bounds__(256) void cal_seg64_store(unsigned *arr, unsigned nseg, unsigned mult, unsigned v) {
    for (unsigned s = blockIdx.x * 16 + threadIdx.x / 16; s < nseg; s += gridDim.x * 16) {
        const unsigned seg = (unsigned)(((unsigned long long)s * mult) % nseg);
        arr[(size_t)seg * 16 + (threadIdx.x & 15)] = v + s;
    }
}
__global__ __launch_bounds__(256) void cal_seg64_load(const unsigned *arr, unsigned nseg, unsigned mult, unsigned *sink) {
    unsigned acc = 0;
    for (unsigned s = blockIdx.x * 16 + threadIdx.x / 16; s < nseg; s += gridDim.x * 16) {
        const unsigned seg = (unsigned)(((unsigned long long)s * mult) % nseg);
        acc ^= arr[(size_t)seg * 16 + (threadIdx.x & 15)];
    }
    if (acc == 0x12345678u) *sink = acc;
}
// (e) ring posts of the wide / dense layouts: one lane = one post = an atomic add on a counter [slots][R] (16 consecutive replicas
// of one slot: one 64-byte segment of counters) + an entry store of EB bytes at ring[(slot * R + r) * ring_cap + pos], slots scattered
template <int EB>
__global__ __launch_bounds__(256) void cal_post(int *cnt, char *ring, unsigned nseg, unsigned mult, int ring_cap, unsigned v) {
    for (unsigned s = blockIdx.x * 16 + threadIdx.x / 16; s < nseg; s += gridDim.x * 16) {
        const unsigned seg = (unsigned)(((unsigned long long)s * mult) % nseg);
        const size_t i = (size_t)seg * 16 + (threadIdx.x & 15);
        const int pos = atomicAdd(&cnt[i], 0x10001) & 3;
        if (EB == 16) *reinterpret_cast<uint4 *>(ring + (i * ring_cap + pos) * 16) = make_uint4(v, v, v, v + s);
        else *reinterpret_cast<uint2 *>(ring + (i * ring_cap + pos) * 8) = make_uint2(v, v + s);
    }
}
// (f) bucket headers [C][R][8] int32: a 16-byte load + a 12-byte store per bucket, buckets consecutive (32-byte stride)
__global__ __launch_bounds__(256) void cal_hdr_rw(int *hdr, size_t buckets, int v) {
    for (size_t b = (size_t)blockIdx.x * 64 + threadIdx.x / 4; b < buckets; b += (size_t)gridDim.x * 64) {
        const int lg = threadIdx.x & 3;
        const int4 h = *reinterpret_cast<const int4 *>(hdr + b * 8);
        if (lg < 3) hdr[b * 8 + lg] = h.x + v + lg;
    }
}
// (g) bucket counters [C][R][8] int64: 8 lanes read-modify-write 8 B each, buckets consecutive (64-byte stride)
__global__ __launch_bounds__(256) void cal_cnt_rmw(long long *cnt, size_t buckets, int v) {
    for (size_t b = (size_t)blockIdx.x * 32 + threadIdx.x / 8; b < buckets; b += (size_t)gridDim.x * 32) {
        const int lg = threadIdx.x & 7;
        cnt[b * 8 + lg] += v + lg;
    }
}
// (h) per-order results out[R][Oq] int2: the LPR lanes of a row store 8 B each, k consecutive orders of the row (k * 8 bytes
// contiguous), rows Oq * 8 bytes apart
template <int LPR>
__global__ __launch_bounds__(256) void cal_out_store(int2 *out, int rows, size_t Oq, int q0, int k, int v) {
    const int lg = threadIdx.x % LPR;
    for (int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < rows; row += gridDim.x * (256 / LPR))
        for (int j = lg; j < k; j += LPR) out[(size_t)row * Oq + q0 + j] = make_int2(v, j);
}

static hipEvent_t e0, e1;
template <typename F>
static void run(const char *name, int reps, double useful_rd, double sect_rd, double line_rd, double useful_wr, double sect_wr, double line_wr, F &&launch) {
    launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel\": \"%s\", \"launches\": %d, \"ms\": %.4f, \"read\": {\"useful\": %.0f, \"sectors\": %.0f, \"lines\": %.0f}, \"write\": {\"useful\": %.0f, \"sectors\": %.0f, \"lines\": %.0f}}\n",
           name, reps + 1, ms / reps, useful_rd, sect_rd, line_rd, useful_wr, sect_wr, line_wr);
}

int main() {
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int reps = 4;
    unsigned *sink; CHK(hipMalloc(&sink, 64));
    const size_t big = (size_t)1 << 30;                      // 1 GiB work buffer
    char *buf; CHK(hipMalloc(&buf, big)); CHK(hipMemset(buf, 1, big));
    const int grid = 256 * 8;
    // (a)
    run("cal_stream_read16", reps, (double)big, (double)big, (double)big, 0, 0, 0, [&] { cal_stream_read16<<<grid, 256>>>((const uint4 *)buf, big / 16, sink); });
    run("cal_stream_write16", reps, 0, 0, 0, (double)big, (double)big, (double)big, [&] { cal_stream_write16<<<grid, 256>>>((uint4 *)buf, big / 16, 7u); });
    // (b) (c): rows of cap = 160 entries (640 B stride, configs[1]'s idle_cap region of sizes), m = 16 / 48 / 112 entries
    const int cap = 160;
    const int rows = (int)(big / (cap * 4));
#define ROWCASE(M) { \
        const double use = (double)rows * M * 4, sect = (double)rows * ((M * 4 + 31) / 32) * 32, line = (double)rows * ((M * 4 + 127) / 128) * 128; \
        run("cal_row_read<16, " #M ">", reps, use, sect, line, 0, 0, 0, [&] { cal_row_read<16, M><<<grid, 256>>>((const unsigned *)buf, rows, cap, sink); }); \
        run("cal_row_read<8, " #M ">", reps, use, sect, line, 0, 0, 0, [&] { cal_row_read<8, M><<<grid, 256>>>((const unsigned *)buf, rows, cap, sink); }); \
        run("cal_row_write<16, " #M ">", reps, 0, 0, 0, use, sect, line, [&] { cal_row_write<16, M><<<grid, 256>>>((unsigned *)buf, rows, cap, 3u); }); \
        run("cal_row_write<8, " #M ">", reps, 0, 0, 0, use, sect, line, [&] { cal_row_write<8, M><<<grid, 256>>>((unsigned *)buf, rows, cap, 3u); }); }
    ROWCASE(16) ROWCASE(48) ROWCASE(112)
    // (d) 64-byte segments: 2^24 of them = 1 GiB
    {
        const unsigned nseg = (unsigned)(big / 64);
        const double use = (double)nseg * 64, line = (double)nseg * 128;       // two segments share a line: each line is touched twice, at different times
        run("cal_seg64_store", reps, 0, 0, 0, use, use, line / 2, [&] { cal_seg64_store<<<grid, 256>>>((unsigned *)buf, nseg, 2654435761u, 5u); });
        run("cal_seg64_load", reps, use, use, line / 2, 0, 0, 0, [&] { cal_seg64_load<<<grid, 256>>>((const unsigned *)buf, nseg, 2654435761u, sink); });
    }
    // (e) posts: counters 64 MiB (2^20 segments of 16), ring_cap 4 entries: 16-byte entries -> ring 1 GiB, 8-byte entries -> 512 MiB
    {
        const unsigned nseg = 1u << 20;
        int *cnt; CHK(hipMalloc(&cnt, (size_t)nseg * 64)); CHK(hipMemset(cnt, 0, (size_t)nseg * 64));
        const double posts = (double)nseg * 16;
        // per post: atomic = 4 B read + 4 B written (one 64-byte segment per 16 posts); entry = EB bytes in its own 32-byte sector
        run("cal_post<16>", reps, posts * 4, (double)nseg * 64, (double)nseg * 128, posts * (4 + 16), (double)nseg * 64 + posts * 32, (double)nseg * 128 + posts * 128,
            [&] { cal_post<16><<<grid, 256>>>(cnt, buf, nseg, 2654435761u, 4, 9u); });
        run("cal_post<8>", reps, posts * 4, (double)nseg * 64, (double)nseg * 128, posts * (4 + 8), (double)nseg * 64 + posts * 32, (double)nseg * 128 + posts * 128,
            [&] { cal_post<8><<<grid, 256>>>(cnt, buf, nseg, 2654435761u, 4, 9u); });
        CHK(hipFree(cnt));
    }
    // (f) (g)
    {
        const size_t buckets = big / 32;
        run("cal_hdr_rw", reps, (double)buckets * 16, (double)buckets * 32, (double)big, (double)buckets * 12, (double)buckets * 32, (double)big,
            [&] { cal_hdr_rw<<<grid, 256>>>((int *)buf, buckets, 1); });
        const size_t b2 = big / 64;
        run("cal_cnt_rmw", reps, (double)big, (double)big, (double)big, (double)big, (double)big, (double)big, [&] { cal_cnt_rmw<<<grid, 256>>>((long long *)buf, b2, 1); });
    }
    // (h) results: 7 consecutive orders per row and launch (56 B at a moving 64-byte aligned offset), rows 8 KiB apart
    {
        const size_t Oq = 1024;                              // 8 KiB per row -> 131 072 rows in the buffer
        const int rws = (int)(big / (Oq * 8));
        int q0 = 0;
        const int k = 7;
        run("cal_out_store<16>", reps, 0, 0, 0, (double)rws * k * 8, (double)rws * 64, (double)rws * 128,
            [&] { cal_out_store<16><<<grid, 256>>>((int2 *)buf, rws, Oq, (q0 = (q0 + 8) % 1000), k, 1); });
    }
    CHK(hipFree(buf));
    return 0;
}
