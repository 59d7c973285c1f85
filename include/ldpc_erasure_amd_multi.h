/* ldpc_erasure_amd_multi.h -- the multi-device layer of libldpc_erasure_amd.so, below Python: one context and one host
 * thread per device, frames sharded with no data-path exchange, ONE gather at the end of a job.
 *
 * Reference facts this layer rests on (paths relative to /root/reference):
 *   frames are independent        one decoder call per frame, Matlab/ErasureCodes_NonBinaryLDPCSim.m:218;
 *                                 the FPGA decoder's per-frame loop, OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:52
 *   host life-cycle               init_opencl() / run() / cleanup() over ONE device, OpenCL/host/src/main.cpp:266-310 (the
 *                                 reference host opens device 0 only, :470-476); a group is N such life-cycles side by side
 *   host stays C                  BASELINE.json north_star: "the host stays C and calls hand-written CDNA4 HIP through a thin C-ABI"
 *
 * Plain C: int return (0 = OK, negative = LDPC_AMD_E*), no exceptions, no torch types.  The shard arithmetic is the one of
 * ldpc_erasure_codes_amd/sharding.py (contiguous blocks in rank order; the first nframes % nranks ranks hold one frame more), so
 * rank order IS frame order.  The final gather moves the per-frame status words (12 bytes per frame) -- and, on request, the
 * decoded frames -- to rank 0's device with hipMemcpyPeerAsync (xGMI between the GPUs of a node; a device-to-device copy when
 * several ranks share one device, which is how the layer is tested on a one-GPU box).
 */
#ifndef LDPC_ERASURE_AMD_MULTI_H
#define LDPC_ERASURE_AMD_MULTI_H

#include "ldpc_erasure_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ldpc_amd_group ldpc_amd_group;

/* Contiguous block of `rank` out of `nranks`: frames [*first, *first + *count). */
void ldpc_amd_shard_frames(int64_t nframes, int nranks, int rank, int64_t *first, int64_t *count);

/* nranks contexts, rank r on device devices[r] (devices == NULL: r modulo the number of HIP devices; ordinals may repeat).
 * Replaces N x init_opencl() (main.cpp:439).  ldpc_amd_group_destroy replaces N x cleanup() (main.cpp:668). */
int ldpc_amd_group_create(int nranks, const int *devices, ldpc_amd_group **group);
void ldpc_amd_group_destroy(ldpc_amd_group *group);
int ldpc_amd_group_size(const ldpc_amd_group *group);
int ldpc_amd_group_device(const ldpc_amd_group *group, int rank);
ldpc_amd_ctx *ldpc_amd_group_ctx(ldpc_amd_group *group, int rank);
/* Last error of a group call (the failing rank's context text included). */
const char *ldpc_amd_group_last_error(const ldpc_amd_group *group);

/* The same code on every rank (code tables are replicated: <= 40 KB per device).  Returns a GROUP code handle (>= 0; the
 * group keeps the handle the code got on every rank's context) or a negative error; the group_decode_* calls take group handles. */
int ldpc_amd_group_load_builtin_code(ldpc_amd_group *group, int code_ind, uint64_t coef_seed);
int ldpc_amd_group_register_code(ldpc_amd_group *group, int n, int k, const uint32_t *row_ptr, const uint16_t *cols, const uint8_t *coefs);

/* ldpc_amd_decode_batch over the group, HOST pointers (same arrays, same meaning): rank r decodes its block on its device from
 * its own host thread -- upload, decode and download of the ranks overlap -- and writes its results into the caller's arrays in
 * place.  No device-side gather is needed on this path: the download IS the gather. */
int ldpc_amd_group_decode_batch(ldpc_amd_group *group, int code, int S, int64_t nframes, const uint8_t *sym, const uint8_t *erased,
                                int max_sweeps, int do_ml, uint8_t *out, int32_t *sweeps, int32_t *residual, int32_t *status);

/* Device-resident form.  Rank r's frames are already on its device: sym[r] / erased[r] / out[r] / words[r] are DEVICE pointers on
 * device(r) holding count_r = shard(nframes, r) frames (words[r]: int32 [3][count_r] = sweeps, residual, status rows).
 * Every rank decodes from its own host thread; then the status words are gathered to rank 0's device:
 *   gathered_words   device pointer on device(0), int32 [3][nframes], or NULL (no gather)
 *   gathered_out     device pointer on device(0), [nframes][n][S], or NULL (outputs stay sharded: the consumer reads them in place)
 * with hipMemcpyPeerAsync on every rank's own stream.  decode_ms / gather_ms (may be NULL): wall clock of the two phases, the
 * slowest rank's.  Asynchronous work is complete when the call returns. */
int ldpc_amd_group_decode_resident(ldpc_amd_group *group, int code, int S, int64_t nframes, const uint8_t *const *sym,
                                   const uint8_t *const *erased, int max_sweeps, int do_ml, uint8_t *const *out, int32_t *const *words,
                                   int32_t *gathered_words, uint8_t *gathered_out, double *decode_ms, double *gather_ms);

/* The FPGA harness trio -- data_in / ldpc_erasure_decoder / data_out with the reference's argument lists (main.cpp:578-604) --
 * sharded: rank r draws and decodes frames [first_r, first_r + count_r) of the SAME erasure stream (the stream is a pure function
 * of seed and symbol index, so the union over the ranks is the single-device run frame for frame), and the two error counters are
 * summed.  perf_tests_body != 0 selects ldpc_amd_ldpc_erasure_decoder_perf_tests. */
int ldpc_amd_group_fpga_run(ldpc_amd_group *group, unsigned short nldpc, int seed, int PER_numerator_div_64, int code_ind,
                            long numFrames, short num_iter, int perf_tests_body, ldpc_amd_error_type *total);

/* Weak-scaling throughput run, all in C (what `ldpc_erasure_decoder_host -g N -b F` prints): every rank synthesises F frames of
 * S-byte packets of built-in code code_ind on its device (uniform erasures at rate per, frame indices continuing across the ranks),
 * encodes them, decodes them `steps` times (one untimed warm-up), the status words are gathered to rank 0's device and every rank
 * compares its output with its codewords on the device.  result[0] frames/s of the whole group (decode phase), [1] decode ms per
 * step (slowest rank), [2] gather ms, [3] 1.0 if every frame decoded to its codeword on every rank and the gathered status words
 * equal the ranks' own, else 0.0. */
int ldpc_amd_group_bench_resident(ldpc_amd_group *group, int code_ind, uint64_t coef_seed, int S, int64_t frames_per_rank, double per,
                                  int max_sweeps, int steps, double result[4]);

#ifdef __cplusplus
}
#endif
#endif
