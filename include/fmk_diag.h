/* fmk_diag.h -- test and tooling entry points of finmlkit_amd.  NOT part of the drop-in C ABI (include/fmk.h) and not used by any
 * product path (the Python modules of finmlkit_amd never bind them).
 *
 *  (1) probes: built into their own library finmlkit_amd/lib/libfmk_diag.so (csrc/fmk_diag.hip), which links against libfmk_hip.so
 *      only for the context type -- bandwidth / latency calibration kernels, the host-to-device rate of the box, a generator of
 *      full-mantissa trade sizes for bench.py's second cfg-4 timing;
 *  (2) counters: what a product kernel of libfmk_hip.so did on its last call (which tier, how many bars were redone); they read
 *      internal state and therefore live in libfmk_hip.so, but are declared here, not in fmk.h.
 */
#ifndef FMK_DIAG_H
#define FMK_DIAG_H
#include "fmk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) probes: libfmk_diag.so ---------------------------------------------------------- */
/* Read-only streaming bandwidth probe (tools/readbw.py): calibrates the HBM ceiling quoted in DESIGN.md.
 * variant 0: 16 B loads per lane, 1: 8 B loads per lane, 2: 4 B loads per lane, 3: 8 B STORES per lane (the buffer is
 * overwritten) -- the last three calibrate FETCH_SIZE / WRITE_SIZE for the access widths the reducers use
 * (tools/pmc_calibrate.py).  Not used by any product path. */
int fmk_diag_read_bandwidth(fmk_ctx *ctx, const void *d_buf, size_t bytes, int variant, int blocks_per_cu,
                            double *elapsed_ms);
/* float32 amounts with a full random 24-bit mantissa in [2^-7, 2) (sums inexact in every order, like real trade sizes): the
 * second cfg-4 timing of bench.py.  Not used by any product path. */
int fmk_diag_fill_amounts_dev(fmk_ctx *ctx, uint64_t seed, int64_t n, float *d_amount);
/* a one-thread marker kernel (k_diag_marker) in the stream: tools/cfgprof.py brackets its measured region with which = 1 / 2 */
int fmk_diag_marker_dev(fmk_ctx *ctx, int which);
/* Two columns read in lock-step (tools/placement.py): 8 B elements of d_a8 and 4 B elements of d_b4 at the same index.
 * pattern 0: flat grid-stride; 1: each wave streams `seg` contiguous elements of both, then jumps by the number of waves
 * (the one-wave-per-bar walk of the reducers); 2: as 1, d_a8 only; 3: as 1, up to 16 rows of both columns requested before any is used.  Not used by any product path. */
int fmk_diag_read_two_streams(fmk_ctx *ctx, const void *d_a8, const void *d_b4, int64_t n, int pattern, int seg,
                              int blocks_per_cu, double *elapsed_ms);
/* Dependent-access latency of one wave (tools/hoplat.py): `hops` hops of `loads` coalesced 512 B rows, the next address
 * depending on the data read; shader cycles per hop.  Not used by any product path. */
int fmk_diag_hop_latency(fmk_ctx *ctx, const void *d_buf, int64_t n, int64_t stride, int loads, int hops,
                         double *cycles_per_hop, double *elapsed_ms);

/* Price / amount / side read by one wave per `seg`-tick segment in 512-tick tiles with lane l owning r = 1, 2, 4 or 8 CONSECUTIVE
 * ticks (tools/ownedread.py): r = 1 is the reducers' chunk layout, the others read 16-byte vectors per lane.  Not used by any product path. */
int fmk_diag_read_owned(fmk_ctx *ctx, const double *d_price, const float *d_amount, const signed char *d_side, int64_t n, int seg,
                        int r, int blocks_per_cu, double *elapsed_ms);

/* host-to-device rate of this box for one buffer, GB/s, best of three: mode 1 = hipMemcpy from pinned memory (the link's ceiling),
 * 0 = hipMemcpy from pageable memory, 2 = fmk_h2d_columns from pageable memory */
int fmk_diag_h2d_rate(fmk_ctx *ctx, size_t bytes, int mode, double *gbps);

/* ---- (2) counters of the last call: in libfmk_hip.so -------------------------------------- */
/* Which tier the last fmk_cusum_bar_indexer[_dev] call of this process took (tests): *tier 1 = the chain walk of
 * fmk_cusum_chain.hip, 0 = the fixed point; chunks the walk opened; its status (0 done, 1 budget, 2 uncertain decision,
 * 3 non-finite return, -1 not tried).  Not used by any product path. */
int fmk_diag_cusum_last(int64_t *tier, int64_t *opened, int64_t *status);
/* Which path answered the last fmk_dollar_bar_indexer[_dev] call of this process: 0 the closed form alone (every decision certain),
 * 1 closed form + exact tier, 2 the same on a stream with increments >= threshold (block trades: the stretch walk of
 * fmk_dollar_exact.hip), 3 the serial walk, 4 a cached result. */
int fmk_diag_dollar_last(int64_t *path);
/* order-flow redo since the last call: {(bar, column) pairs redone in tick order, 512-term tiles walked, tiles added term by term,
 * pairs of column 0 .. 6 (buy / sell volume, buy / sell dollars, spread, signed volume, signed dollars)} */
int fmk_diag_dir_redo(fmk_ctx *ctx, int64_t *out10);
/* the last one-pass cfg-4 sizing call (csrc/fmk_fused.h): bars it handed to the footprint class kernels / to k_bar_dir / entries of
 * the tick-order redo list (float32 ties of the dollar columns and mean_spread) */
int fmk_diag_fused_last(fmk_ctx *ctx, int64_t *n_fp_list, int64_t *n_dir_list, int64_t *n_redo);
/* the device's exp (csrc/fmk_exp.h: glibc's, restated) of n doubles on the device -- tests compare it with the host's exp() */
int fmk_diag_exp_dev(fmk_ctx *ctx, const double *d_x, int64_t n, double *d_out);
/* bars of the last fmk_comp_bar_footprints_fill_median_dev call whose median took the generic selection (bracket miss) */
int fmk_diag_fp_median_fallbacks(fmk_ctx *ctx, int64_t *count);
/* ... and into how many LATER segments that walk split the two sides' chains (0: each side walked in one piece); a segment
 * starts at a chunk boundary from which the side's state provably does not depend on earlier ticks (k_cc_sync).  *rate: the
 * estimate the tier was chosen by (512-tick sub-blocks per chunk and side of the leading chunks with a certain close; -1: none). */
int fmk_diag_cusum_segments(int64_t *segments, double *rate);
/* last CUSUM call: 1 if the one-pass form (csrc/fmk_cusum_onepass.h) answered, its fix-up launches, the chunks that had not merged
   within the first launch's limit, the number of chunks */
int fmk_diag_cusum_onepass(int64_t *used, int64_t *fix_launches, int64_t *pending_first, int64_t *chunks);

#ifdef __cplusplus
}
#endif
#endif
