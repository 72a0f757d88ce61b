// Micro-benchmark: issue rate of the integer VALU / DPP / LDS-crossbar instructions k_tick_rows is made of,
// on gfx950.  Answers "how many cycles does one wave64 VALU instruction occupy its SIMD" - the constant behind
// the VALU-issue floor quoted in DESIGN.md / bench.py (the microarchitecture guide says 2 cycles for v_fma_f32).
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITERS 2048
#define UNROLL 32

template <int KIND>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * (i + 1);
    __shared__ unsigned char lds[4096];
    if (KIND == 4) { for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (unsigned char)(i * 7); __syncthreads(); }
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int &x = a[u & 7];
            const int y = a[(u + 3) & 7];
            if (KIND == 0) asm volatile("v_min_i32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 1) asm volatile("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(x) : "v"(x), "v"(y));
            if (KIND == 2) asm volatile("v_min_i32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x) : "v"(y), "v"(x), "0"(x));
            if (KIND == 3) x = __builtin_amdgcn_ds_bpermute((y & 63) << 2, x);
            if (KIND == 4) x += lds[(x + y) & 4095];
            if (KIND == 5) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345678) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int waves_per_simd) {
    int *out;
    hipMalloc(&out, 4096);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * N_ITERS * UNROLL;
    const double clk = p.clockRate * 1e3;              // Hz (max clock)
    printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.0f MHz max clock)\n", name, waves_per_simd, ms,
           ms * 1e-3 * clk / instr_per_simd, clk / 1e6);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_min_i32", w);
        run<1>("v_lshl_or_b32", w);
        run<2>("v_min_i32_dpp", w);
        run<5>("v_add_u32", w);
        run<3>("ds_bpermute_b32", w);
        run<4>("ds_read_u8+add", w);
    }
    return 0;
}
