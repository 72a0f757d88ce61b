/* vds_debug.h - test / instrumentation hooks exported by libvds.so next to the drop-in boundary of vds.h.
 *
 * NOT part of the boundary: nothing here has a counterpart in the reference.  These entry points exist for the parity
 * tests (tests/), the guarded and instrumented builds (make canary / make prof / make dbg) and the measurement scripts
 * under profiles/.  They are declared so that every symbol libvds.so exports is declared somewhere
 * (tests/test_abi_symbols.py checks include/vds.h + this file against `nm -D`).
 *
 * vds_debug_dense changes which of the library's own (result-identical) kernel variants runs; vds_debug_ablate makes
 * results INVALID (timing only) - neither belongs in a product build's call path.
 */
#ifndef VDS_DEBUG_H
#define VDS_DEBUG_H

#include "vds.h"

#ifdef __cplusplus
extern "C" {
#endif

/* dense tick (k_tick_dense) variants, effective from the next vds_load_orders* on: lanes per replica (8 / 16; 0 = default),
 * idle entries / arrivals per bucket its fast path takes (0 = 128 or 256 / 64), force_slow bit 0: every bucket takes the slow
 * path, bit 1: arrival ring instead of static arrival slots.  Results do not depend on any of them (tests/test_gpu_parity.py). */
int vds_debug_dense(vds_handle *h, int32_t lanes_per_replica, int32_t tab, int32_t keys, int32_t force_slow);

/* timing-only ablation switches of the tick kernels (instrumented build; profiles/ablate.py).  Non-zero flags: results INVALID. */
int vds_debug_ablate(vds_handle *h, int32_t flags);

/* instrumented build (make prof): 32 section cycle counters of the tick kernels (profiles/sections*.py) */
int vds_debug_read_prof(vds_handle *h, uint64_t *out32);

/* instrumented build: launch spans of k_tick_dense, [2 chains][256 slots]{first wavefront in, last out} on the 100 MHz
 * s_memrealtime clock; out1024 may be null (reset only) */
int vds_debug_read_span(vds_handle *h, uint64_t *out1024, int32_t reset);

/* the 16 words of the device error block: [0] sticky error bits, [2] buckets off the fast path, [4..15] why (instrumented build) */
int vds_debug_read_err(vds_handle *h, int32_t *out16);

/* guarded build (make canary): device tables of this process whose guard zones were written (0 in every other build);
 * a negative VDS_E* code if the check itself failed */
int vds_debug_check_guards(vds_handle *h);

/* guarded build only: damages one guard zone on purpose (the test of the check); VDS_EINVAL in other builds */
int vds_debug_poke_guard(vds_handle *h);

/* executable day graphs parked in the process-wide pool right now (vds_run, tests/test_gpu_run_groups.py) */
int vds_debug_graph_pool_size(void);

/* k_tick_dense's per-slot form as the handle stands (vds_api.hip adapt_dense): out[t] = 1 where slot t runs 16 lanes per replica with
 * 256-entry tables whatever the base form, 0 where it runs the base form; *n_out = slots written (0: no per-slot choice was made). */
int vds_debug_tick_forms(vds_handle *h, uint8_t *out, int32_t cap, int32_t *n_out);

/* DPP primitives of the kernels on nwaves x 64 int32 values (tests/test_gpu_primitives.py): out_wave [nwaves] wavefront
 * minima; out_rowmin / out_rowsum / out_rowscan [nwaves * 64] per-lane 16-lane-row minimum, row sum and inclusive row scan */
int vds_selftest_dpp(vds_handle *h, const int32_t *in, int32_t *out_wave, int32_t *out_rowmin, int32_t *out_rowsum,
                     int32_t *out_rowscan, int32_t nwaves);

#ifdef __cplusplus
}
#endif
#endif
