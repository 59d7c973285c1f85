/*
 * ldpc_erasure_amd.h -- C ABI of libldpc_erasure_amd.so: MI355X (gfx950) GF(256) LDPC / Reed-Solomon
 * erasure decoding, the hot path of chadac8j/LDPC_Erasure_Codes.
 *
 * Every entry point names the reference interface it replaces (paths relative to /root/reference).
 * Plain C: int return (0 = OK, negative = LDPC_AMD_E*), no exceptions, no torch types.  A failed
 * decode of an individual frame is NOT an API error: it is reported per frame in residual[]/status[].
 *
 * Threading: a context is thread-compatible (one stream per context, no global mutable state); use one
 * context per host thread / per GPU.  All work is enqueued on the context's stream; calls taking host
 * pointers synchronise before returning, calls with LDPC_AMD_DEVICE_PTRS are asynchronous until
 * ldpc_amd_synchronize() -- so H2D / decode / D2H of successive batches can overlap the way the three
 * FPGA kernels of OpenCL/device/ldpc_erasure_decoder_top.cl:57-158 do.
 *
 * Data layouts (all little endian, densely packed):
 *   sym    [nframes][n][S]  uint8   symbol j of frame f is a packet of S bytes; every byte is one GF(256)
 *                                   element.  S = 1 is the Matlab model (one element per symbol,
 *                                   Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:4), S = 1024 the FPGA
 *                                   packet of OpenCL/host/src/main.cpp:42-47 (128 x u64).  S is 1 or a multiple of 16.
 *   erased [nframes][n]     uint8   non-zero = erased (Matlab: value -1, ...Decoder.m:9; FPGA: is_erasure,
 *                                   main.cpp:46).  The payload of an erased symbol is ignored.
 *   out    [nframes][n][S]  uint8   the full length-n word, as Matlab returns it (...Decoder.m:129);
 *                                   symbols that stay unknown are written as 0.
 */
#ifndef LDPC_ERASURE_AMD_H
#define LDPC_ERASURE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_AMD_OK 0
#define LDPC_AMD_EINVAL (-1)   /* bad argument                                   */
#define LDPC_AMD_EHIP (-2)     /* HIP runtime error (see ldpc_amd_last_error)     */
#define LDPC_AMD_ENOMEM (-3)   /* device allocation failed                        */
#define LDPC_AMD_ENOCODE (-4)  /* unknown code / RS handle or code index          */
#define LDPC_AMD_EUNSUP (-5)   /* shape not supported by the kernels              */

/* flags */
#define LDPC_AMD_DEVICE_PTRS 1u /* every data pointer of the call is a device pointer; call is async */
#define LDPC_AMD_INPLACE 2u     /* ldpc_amd_decode_batch with out == sym (device pointers, S >= 16): received symbols are
                                   left where they are and only the erased ones are written -- about half the HBM traffic.
                                   The reference returns a copy (Matlab value semantics, separate FPGA output buffer); this
                                   is an extension for callers that own the frame buffer. */

/* per-frame status[] values */
#define LDPC_AMD_ST_MP_DONE 0      /* message passing recovered everything                            */
#define LDPC_AMD_ST_ML_SOLVED 1    /* residual erasures solved by the ML (Gaussian elimination) stage */
#define LDPC_AMD_ST_ML_RANKDEF 2   /* ML stage hit an empty pivot column: like the reference
                                      (...Decoder.m:87-90,127) the partially reduced rhs is written   */
#define LDPC_AMD_ST_ML_SKIPPED 3   /* residual erasures left: ML disabled, or more erasures than checks */

typedef struct ldpc_amd_ctx ldpc_amd_ctx;

/* ---- life-cycle -------------------------------------------------------------------------------
 * ldpc_amd_init    replaces init_opencl()  (OpenCL/host/src/main.cpp:111,439-544): selects the device,
 *                  creates the stream (the reference: context + 3 queues + program + kernels + buffers).
 * ldpc_amd_cleanup replaces cleanup()      (main.cpp:113,668-691). */
int ldpc_amd_init(int device_ordinal, ldpc_amd_ctx **ctx);
void ldpc_amd_cleanup(ldpc_amd_ctx *ctx);
/* Last error text of this context (ctx == NULL: of the last failed ldpc_amd_init on this thread).
 * Replaces checkError(status, "...") of AOCLUtils (main.cpp:493,508,569), which prints and exits. */
const char *ldpc_amd_last_error(const ldpc_amd_ctx *ctx);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. */
int ldpc_amd_set_stream(ldpc_amd_ctx *ctx, void *hip_stream);
/* Waits for the context's stream.  Also the point where a violated internal assumption reported by a kernel of an earlier
 * asynchronous call (e.g. dynamic LDS not at LDS address 0) comes back as LDPC_AMD_EHIP instead of silently wrong bytes. */
int ldpc_amd_synchronize(ldpc_amd_ctx *ctx);
/* Tuning / diagnostic knobs of this context (the table of DESIGN.md's appendix: "SCATTER_B", "ML_SOLVE", "HOST_PIPELINE", ... with
 * or without the LDPC_AMD_ prefix, case-insensitive; value NULL or "" restores the shipped default).  The environment variables
 * LDPC_AMD_<KEY> give the initial values and are read ONCE, inside ldpc_amd_init -- no call of this library looks at the process
 * environment afterwards, so a multi-threaded host can change a knob of one context without racing another (the role of the
 * reference's command-line options, OpenCL/host/src/main.cpp:157-170,217-246).  A knob changes how a batch is decoded, never a
 * byte of the result.  LDPC_AMD_EINVAL for an unknown key or a value outside the knob's range. */
int ldpc_amd_configure(ldpc_amd_ctx *ctx, const char *key, const char *value);
/* The knobs of this context that differ from the shipped defaults, as "NAME=value NAME=value" (word-valued knobs as 0 / 1) into
 * buf (at most cap - 1 characters + NUL; buf may be NULL); returns the length of the full text: 0 = everything at its default.  For
 * measurement scripts that want to print what they actually measured (bench.py writes it into its detail file). */
int ldpc_amd_knobs(ldpc_amd_ctx *ctx, char *buf, int cap);

/* ---- code ROM ----------------------------------------------------------------------------------
 * ldpc_amd_code_params: the row ldpc_params[code_ind][0..5] = {n, k, firstRow, lastRow, RS_n, RS_k}
 * (OpenCL/device/LDPC_Vlist_data.h:10-14, OpenCL/host/inc/Main_LDPC_header.h:10-14).  code_ind 0 and 1
 * are the reference's; 2 = (4000,2000) from Matlab/n4000_k2000_no6cycles_triangleForm.mat; 3 = (4080,3060)
 * (synthesised, only if built in).  firstRow/lastRow index the built-in table of this library. */
int ldpc_amd_code_params(int code_ind, int params[6]);
/* Loads a built-in code (the copy loop of OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:38-43).
 * coef_seed != 0: GF(256) coefficients drawn per non-zero from include/ldpc_erasure_amd_synth.h
 * (rule of Matlab/ErasureCodes_NonBinaryLDPCSim.m:51-58); coef_seed == 0: all coefficients 1 (binary H).
 * Returns a code handle >= 0 or a negative error. */
int ldpc_amd_load_builtin_code(ldpc_amd_ctx *ctx, int code_ind, uint64_t coef_seed);
/* Registers a custom code from 0-based CSR (what Matlab/ErasureCodes_NonBinaryLDPCSim.m:91-107 builds as
 * Vlist / Vlist_val): row_ptr[n-k+1], cols ascending per row, coefs 1..255.  Returns a code handle. */
int ldpc_amd_register_code(ldpc_amd_ctx *ctx, int n, int k, const uint32_t *row_ptr, const uint16_t *cols,
                           const uint8_t *coefs);
int ldpc_amd_code_info(ldpc_amd_ctx *ctx, int code, int *n, int *k, int *nnz);
/* The static schedules of the systematic encoder of a code handle: info[0] dependency levels of the parity triangle, [1] groups of
 * the level-collapsed schedule (0: the code has none), [2] accumulators its steps pull in all, [3] scatter entries left, [4] the
 * longest pull list, [5] 1 = the last packet-mode ldpc_amd_encode_batch of this context ran the grouped schedule.  Diagnostic
 * (tests, bench.py); no reference counterpart. */
int ldpc_amd_encode_info(ldpc_amd_ctx *ctx, int code, int info[6]);
/* Copies the host CSR of a code handle back (cols/coefs may be NULL). */
int ldpc_amd_code_csr(ldpc_amd_ctx *ctx, int code, uint32_t *row_ptr, uint16_t *cols, uint8_t *coefs);

/* ---- hot path: batched hybrid MP + ML erasure decode ---------------------------------------------
 * Batched equivalent of
 *   [Msg, iterations] = My_LDPC_HybridML_NonBinary_Erasure_Decoder(recv_vec_val, Vlist, Clist, H_sparse,
 *                                                                  n, k, GF_add, GF_mult, GF_inv)
 * (Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:4), one call per frame in the reference
 * (Matlab/ErasureCodes_NonBinaryLDPCSim.m:218), and -- with a binary code handle, do_ml = 0 -- of
 * My_LDPC_Erasure_Decoder (Matlab/My_LDPC_Erasure_Decoder.m:3) and of the FPGA kernel
 * ldpc_erasure_decoder(num_iter, code_ind) (OpenCL/device/ldpc_erasure_decoder.cl:24).
 *   max_sweeps  itenum (reference constants: 10 hybrid ...Decoder.m:13, 50 binary My_LDPC_Erasure_Decoder.m:10)
 *   do_ml       do_ML_decode (...Decoder.m:6)
 *   sweeps[f]   iterations (2nd Matlab output)          -- may be NULL
 *   residual[f] erasures left after the MP sweeps (num_cur_erasures, ...Decoder.m:51) -- may be NULL
 *   status[f]   LDPC_AMD_ST_*                           -- may be NULL */
int ldpc_amd_decode_batch(ldpc_amd_ctx *ctx, int code, int S, int64_t nframes, const uint8_t *sym,
                          const uint8_t *erased, int max_sweeps, int do_ml, uint8_t *out, int32_t *sweeps,
                          int32_t *residual, int32_t *status, unsigned flags);

/* Systematic encoder (the step before the path): source[nframes][k][S] -> codeword[nframes][n][S].
 * Replaces Matlab/ErasureCodes_NonBinaryLDPCSim.m:173-182 and OpenCL/device/ldpc_erasure_encoder.cl:45-94. */
int ldpc_amd_encode_batch(ldpc_amd_ctx *ctx, int code, int S, int64_t nframes, const uint8_t *source,
                          uint8_t *codeword, unsigned flags);

/* ---- Reed-Solomon comparator ---------------------------------------------------------------------
 * ldpc_amd_rs_create builds the systematic generator of Matlab/Test_My_RS_Decode.m:22,30-37
 * (G(row,col) = alpha^(row*col), G <- inv(G(:,1:k)) G) and returns an RS handle.
 * ldpc_amd_rs_decode_batch is the batched  Msg = My_RS_Decode(recv_vec_ind, recv_vec_gf256_val, m, n, k,
 * Prim_poly, G, log_lookup)  (Matlab/My_RS_Decode.m:14; table version
 * Matlab/My_RS_Decode_Optimize_With_GFTables.m:15):
 *   recv_idx [nblocks][k]     uint16  0-BASED strictly ascending positions (< n) of the first k received symbols
 *                                     (Matlab/ReedSolomonErasureCodes.m:80-81).  Host pointers: a violation returns
 *                                     LDPC_AMD_EINVAL; LDPC_AMD_DEVICE_PTRS: the offending block decodes to zeros and is
 *                                     counted (ldpc_amd_rs_bad_blocks).
 *   recv_val [nblocks][k][S]  uint8   their values;   msg [nblocks][k][S] the recovered source block. */
int ldpc_amd_rs_create(ldpc_amd_ctx *ctx, int n, int k);
int ldpc_amd_rs_generator(ldpc_amd_ctx *ctx, int rs, uint8_t *g /* [k][n], host */);
int ldpc_amd_rs_encode_batch(ldpc_amd_ctx *ctx, int rs, int S, int64_t nblocks, const uint8_t *source,
                             uint8_t *codeword, unsigned flags);
int ldpc_amd_rs_decode_batch(ldpc_amd_ctx *ctx, int rs, int S, int64_t nblocks, const uint16_t *recv_idx,
                             const uint8_t *recv_val, uint8_t *msg, unsigned flags);
/* Number of blocks of the LAST ldpc_amd_rs_decode_batch on this context whose recv_idx was malformed (a position >= n, or
 * not strictly ascending) and which were therefore decoded to all zeros.  Synchronises the context's stream.  The reference
 * has no failure signalling at this point (Matlab/My_RS_Decode_Optimize_With_GFTables.m:95-97 is an empty branch; its caller
 * guarantees the precondition, Matlab/ReedSolomonErasureCodes.m:80-81); with LDPC_AMD_DEVICE_PTRS this is how a caller tells an
 * all-zero message from a refused block. */
int ldpc_amd_rs_bad_blocks(ldpc_amd_ctx *ctx, long long *count);

/* ---- synthetic source / channel on the device (the role of the FPGA's data_in kernel,
 * OpenCL/device/ldpc_erasure_decoder_top.cl:57-120; streams of include/ldpc_erasure_amd_synth.h).
 * Device pointers only. */
int ldpc_amd_synth_source(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int k, int S,
                          uint8_t *d_source);
int ldpc_amd_synth_erasures_uniform(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int n,
                                    double per, uint8_t *d_erased);

/* Gilbert-Elliott bursty channel of Matlab/Bursty_Error_Channel_Model_Generator.m:12-47 (good state erases w.p.
 * alpha, bad w.p. beta, P(good->bad) = 0.1/good_transition_bias, P(bad->good) = 0.1), the chain state carried across
 * symbols and frames from global symbol 0 (Matlab/ErasureCodes_NonBinaryLDPCSim.m:163,191-198). */
int ldpc_amd_synth_erasures_bursty(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int n,
                                   double alpha, double beta, double good_transition_bias, uint8_t *d_erased);

/* ---- drop-in for the three OpenCL kernels of the FPGA harness ------------------------------------
 * The reference host sets 6 + 2 + 3 kernel arguments and enqueues three tasks
 * (OpenCL/host/src/main.cpp:578-604, 617-626).  These three calls take the same scalars in the same order.
 * symbol_type is the reference's AoS packet (main.cpp:44-47, 1032 bytes with natural alignment). */
#define LDPC_AMD_SYM_LEN 128
typedef struct {
    unsigned long symbol[LDPC_AMD_SYM_LEN];
    unsigned char is_erasure;
} ldpc_amd_symbol_type;
typedef struct { /* error_type, OpenCL/device/ldpc_erasure_decoder_top.cl:46-49 */
    int num_LDPC_errors;
    int num_RS_errors;
} ldpc_amd_error_type;
/* data_in(global symbol_type*, ushort nldpc, int seed, int PER_numerator_div_64, int code_ind, long numFrames)
 * (ldpc_erasure_decoder_top.cl:58-65): the source of numFrames*n erasure flags from the kernel's own generator
 * (threefry4x32-20, key {1, seed}, 32-bit counter incremented per symbol, erased iff (rv & 0x3F) < PER_numerator,
 * :74-110), payload all-zero (the all-zero codeword, :77-82). data_in may be NULL (the FPGA kernel never reads it).
 * Like the FPGA kernel -- a frame loop feeding a channel -- the call materialises nothing: it arms the source, and the
 * decoder call draws the stream chunk by chunk (memory O(chunk), so N_T = 1e6 ... 2e8 of the paper's Table I runs). */
int ldpc_amd_data_in(ldpc_amd_ctx *ctx, const ldpc_amd_symbol_type *data_in, unsigned short nldpc, int seed,
                     int PER_numerator_div_64, int code_ind, long numFrames);
/* The same source armed at frame `firstFrame` of its stream: the run covers frames [firstFrame, firstFrame + numFrames) of what
 * ldpc_amd_data_in(..., firstFrame + numFrames) would draw -- one rank's shard of a multi-device run
 * (ldpc_amd_group_fpga_run, include/ldpc_erasure_amd_multi.h).  No FPGA counterpart (one device there). */
int ldpc_amd_data_in_at(ldpc_amd_ctx *ctx, const ldpc_amd_symbol_type *data_in, unsigned short nldpc, int seed,
                        int PER_numerator_div_64, int code_ind, long numFrames, long firstFrame);
/* ldpc_erasure_decoder(short num_iter, int code_ind) (ldpc_erasure_decoder_perf_tests.cl:30): decodes the
 * frames of the last ldpc_amd_data_in with the binary packet-XOR message-passing decoder, frame loop like :52-238:
 * chunk by chunk  source -> decode -> running counters {num_frame_errors, num_RS_frame_errors} (:46-47,70-80,229-236).
 * Asynchronous on the context's stream; ldpc_amd_data_out synchronises. */
int ldpc_amd_ldpc_erasure_decoder(ldpc_amd_ctx *ctx, short num_iter, int code_ind);
/* The reference holds two bodies for that kernel.  ldpc_amd_ldpc_erasure_decoder follows the one the top-level design
 * includes (ldpc_erasure_decoder_top.cl:161 -> ldpc_erasure_decoder.cl:24-105: num_iter in-order sweeps over all
 * checks).  This entry point follows the other one, ldpc_erasure_decoder_perf_tests.cl:30-236, whose argument list is
 * the one the host passes (main.cpp:593-596): two copies of the frame, checks [0,m/2) swept on the first and
 * [m/2,m) on the second, merged after every iteration, stop when num_current_correct == k -- as written, including the
 * premature stops that rule allows (merged parity symbols are counted too, :180-201). */
int ldpc_amd_ldpc_erasure_decoder_perf_tests(ldpc_amd_ctx *ctx, short num_iter, int code_ind);
/* Per-frame results of the last of the two decoder calls (host pointers, either may be NULL): systematic symbols
 * still erased (:213-220, the frame-error criterion) and iterations run.  No FPGA counterpart (the FPGA streams only
 * the running error counters); exists so that tests can compare frame by frame.  Kept for runs of up to 2^22 frames
 * (LDPC_AMD_EUNSUP beyond); LDPC_AMD_EINVAL unless a decoder call ran over exactly the last data_in's frames. */
int ldpc_amd_fpga_frame_stats(ldpc_amd_ctx *ctx, long numFrames, int32_t *residual_sys, int32_t *iterations);
/* data_out(global symbol_type*, int code_ind, long numFrames) (ldpc_erasure_decoder_top.cl:124-127): waits for the
 * run and collects the frame-error counters (ERROR_STAT); data_out, if not NULL, receives the first k symbols of the
 * last frame.  LDPC_AMD_EINVAL unless a decoder call ran since the last data_in with the same code_ind / numFrames. */
int ldpc_amd_data_out(ldpc_amd_ctx *ctx, ldpc_amd_symbol_type *data_out, int code_ind, long numFrames,
                      ldpc_amd_error_type *stats);

/* ---- measurement -------------------------------------------------------------------------------------
 * The reference times its run with OpenCL event profiling (CL_QUEUE_PROFILING_ENABLE, main.cpp:515; getStartEndTime on
 * the data_out event, :652).  With profiling on, every kernel launch of the decode path is bracketed by HIP events on
 * the context's stream; ldpc_amd_get_profile synchronises, returns the summed device time (ms) and launch count per
 * kernel kind since the last call, and resets the counters. */
#define LDPC_AMD_PROF_PEEL 0   /* ldpc_peel_kernel  (schedule; + apply when S = 1) */
#define LDPC_AMD_PROF_APPLY 1  /* packet kernel(s): ldpc_scatter_kernel (tier 1) + ldpc_scatter_big_kernel (tier 2), or ldpc_apply_kernel */
#define LDPC_AMD_PROF_ML 2     /* ldpc_ml_kernel + ldpc_ml_solve_kernel (Gaussian elimination on residual frames) */
#define LDPC_AMD_PROF_APPLY_TIER2 3 /* the tier-2 launch of the packet kernel alone (already inside LDPC_AMD_PROF_APPLY) */
#define LDPC_AMD_PROF_ML_SOLVE 4    /* ldpc_ml_solve_kernel alone (already inside LDPC_AMD_PROF_ML) */
#define LDPC_AMD_PROF_KINDS 5
/* enable: 0 off; 1 one bracket per kind of a call (PEEL, APPLY, ML); 2 additionally the nested brackets APPLY_TIER2 and ML_SOLVE
 * (two more event pairs per call: use it for break-downs, not inside a timed region that is quoted as throughput). */
int ldpc_amd_set_profiling(ldpc_amd_ctx *ctx, int enable);
int ldpc_amd_get_profile(ldpc_amd_ctx *ctx, double ms[LDPC_AMD_PROF_KINDS], int64_t launches[LDPC_AMD_PROF_KINDS]);
/* Name (as rocprofv3 prints it, e.g. "ldpc_scatter_kernel<16, 2, true, 8, false>") of the kernel instantiation the LAST
 * launch of that kind used -- the launch plan depends on code, S and batch, so a report must not hard-code it.
 * "" when no kernel of the kind was launched yet. */
const char *ldpc_amd_profile_kernel_name(ldpc_amd_ctx *ctx, int kind);
/* Launch plan of the last ldpc_amd_decode_batch (of its last chunk): info[0] frames (= wavefronts) per peel workgroup,
 * [1] frames resident per CU, [2] 1 = code tables read from global memory instead of LDS, [3] peel LDS bytes per workgroup,
 * [4] peel LDS bytes per frame, [5] bytes of every row per packet-kernel workgroup (0: no packet kernel), [6] accumulator
 * cap of tier 1, [7] 1 = two tiers.  Diagnostic (tools/tune_s1.py); no reference counterpart. */
int ldpc_amd_last_plan(ldpc_amd_ctx *ctx, int info[8]);
/* What the ML stage of the last ldpc_amd_decode_batch (of its last chunk) did, read back from the stage's device counters (the call
 * waits for the context's stream): stats[0] residual frames handed to the stage, [1] of them solved through the fast path's
 * schedules (packet mode, csrc/ml_pi.inc; 0 at S = 1), [2] of those flagged by the consistency test and factored again in the
 * reference's elimination order (received symbols that were not a codeword), [3] frames whose schedule did not fit the arena.
 * One-batch lag of the adaptive skip (knob ML_PI_ADAPTIVE, default on): a context whose previous packet batch had NO residual frame
 * skips the fast path's launches for the next batch, so the first residual-heavy batch after a clean one reports [1] = 0 (all its
 * frames are factored exactly -- same bytes, slower) and the batch after that has the fast path again.  "Previous" means the last
 * batch whose counters had reached the host when this one was launched (an un-awaited copy), so the lag is one batch or more.
 * Diagnostic (tests, bench.py); no reference counterpart. */
int ldpc_amd_ml_stats(ldpc_amd_ctx *ctx, long long stats[4]);

/* ---- diagnostics ----------------------------------------------------------------------------------- */
/* Device self-test of the GF(256) primitives (packed multiply vs. table) -> 0 when all 65536 products and
 * 255 inverses agree with the host tables. */
int ldpc_amd_selftest(ldpc_amd_ctx *ctx);
/* Device-to-device copy of `bytes` (16-byte multiple, device pointers) with the streaming loads/stores of the packet
 * kernel: `reps` single launches of each of five launch shapes after a warm-up; *ms_per_copy = the BEST single launch (the
 * figure is quoted as what a linear read + write copy reaches on this box).  The measured copy rate of the box is what
 * SURVEY.md 8(d) asks to be quoted next to the nominal HBM peak; no reference counterpart. */
int ldpc_amd_copy_probe(ldpc_amd_ctx *ctx, const void *src, void *dst, uint64_t bytes, int reps, double *ms_per_copy);
/* Host copy of the GF tables the kernels use: mult[256*256], inv[256] (inv[0] = 0), either may be NULL. */
int ldpc_amd_gf_tables(uint8_t *mult, uint8_t *inv);
const char *ldpc_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_ERASURE_AMD_H */
