/*
 * oracle.h -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (ldpc_erasure_codes_amd/, include/ldpc_erasure_amd.h) never links, imports or calls it.
 *
 * Every function restates one reference file, line by line, with 0-based C indices where Matlab is
 * 1-based; the citation is on each prototype.  Paths are relative to /root/reference.
 *
 * Pinning status (see DESIGN.md "Oracle"): the reference is Matlab (cannot run here: no
 * Matlab/Octave) plus an Intel-FPGA OpenCL host (needs AOCLUtils + aoc, absent), so there is no
 * oracle/_ref build.  The reference ships NO input/output vectors for its decoders.  What pins this
 * restatement is: (1) the complete GF(256) add/mult/inv tables of
 * Matlab/GF_256_add_mult_inv_tables.mat (tests/golden/gf256_tables_ref.npz), (2) the three H
 * matrices and the OpenCL code ROM (tests/golden/code_rom_ref.npz), (3) the paper's (6,3) worked
 * example, (4) an independent 1-based Python transliteration of the same .m files
 * (tests/matlab_literal.py) that must agree bit for bit, and (5) round-trip / linearity properties.
 * Decoder control flow therefore has no reference-produced golden vector: "decoder I/O parity
 * unpinned by reference vectors; arithmetic and code tables pinned".
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GF(256) tables: Matlab/Build_GF256_Lookup_Tables.m:21-67 (poly [1 0 1 1 1 0 0 0 1] = 0x171,
 *      Matlab/ErasureCodes_NonBinaryLDPCSim.m:70) ------------------------------------------------- */
typedef struct {
    uint8_t add[256][256];  /* GF_add_lookup(a+1,b+1)  */
    uint8_t mult[256][256]; /* GF_mult_lookup(a+1,b+1) */
    uint8_t inv[255];       /* GF_inv_lookup(x), x = 1..255 -> inv[x-1] (no +1: ...Decoder.m:47,100) */
    uint8_t antilog[256];   /* gf_log_inv(ii): [0]=0, [1]=1, [2]=alpha, ...  (Build...m:21-29)        */
    int16_t log[256];       /* log_lookup(x+1): log[0] = -1 stands for -inf    (Build...m:31-32)      */
} oracle_gf;

/* Fills *t.  prim_poly is the integer form of the polynomial (369 = 0x171 for the reference). */
void oracle_gf_build(oracle_gf *t, int prim_poly);
const oracle_gf *oracle_gf_default(void); /* poly 0x171, built once */

/* ---- code container: Vlist / Vlist_val as built at Matlab/ErasureCodes_NonBinaryLDPCSim.m:91-107 -- */
typedef struct {
    int n, k, m, nnz, width;  /* width = max row degree + 1 */
    int *vlist;               /* m x width, row-major: [deg, col_1 .. col_deg, 0 ...], cols 1-BASED ascending */
    int *vlist_val;           /* m x width: [deg, H_nb(row, col_1) ...]                                       */
} oracle_code;

/* Builds the container from a 0-based CSR (row_ptr[m+1], cols ascending per row, coefs 1..255).
 * Returns NULL on malformed input. */
oracle_code *oracle_code_create(int n, int k, const uint32_t *row_ptr, const uint16_t *cols,
                                const uint8_t *coefs);
void oracle_code_destroy(oracle_code *c);

/* ---- a8: systematic encoder, Matlab/ErasureCodes_NonBinaryLDPCSim.m:174-182 -------------------- */
/* source[k] -> codeword[n] (codeword[0..k) = source). */
void oracle_ldpc_encode(const oracle_code *c, const uint8_t *source, uint8_t *codeword);
/* Same statements on S independent byte lanes: source[k*S] -> codeword[n*S]. */
void oracle_ldpc_encode_packets(const oracle_code *c, int S, const uint8_t *source, uint8_t *codeword);

/* ---- a1-a4: Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:4-130 --------------------------
 * recv[n]: 0..255 or -1 (erasure, ...Decoder.m:9).  itenum: sweep cap (reference constant 10, :13).
 * do_ml: reference constant 1 (:6).  msg[n] <- y_current (:129), *iterations <- itestep (:130).
 * info (may be NULL): [0] erasures left after the MP loop (:51), [1] 1 if the ML block ran (:61),
 *                     [2] dont_do_jordan (:83-90).
 * Returns 0, or -2 when the ML block would index past rhs in Matlab (num_erasures > n-k, :127);
 * in that case the MP result is returned and the erasures stay -1. */
int oracle_ldpc_hybridml_nonbinary_decode(const oracle_code *c, const int16_t *recv, int itenum,
                                          int do_ml, int16_t *msg, int *iterations, int *info);

/* Packet form used by the S > 1 parity tests: lane l of the S-byte symbols is decoded exactly as an
 * independent S = 1 frame with the same erasure pattern.  sym[n*S],
 * erased[n] (0/1), out[n*S]; unrecovered symbols come back as 0 with out_erased[j] = 1. */
int oracle_ldpc_hybridml_nonbinary_decode_packets(const oracle_code *c, int S, const uint8_t *sym,
                                                  const uint8_t *erased, int itenum, int do_ml,
                                                  uint8_t *out, uint8_t *out_erased, int *iterations,
                                                  int *info);

/* ---- a10: binary siblings ----------------------------------------------------------------------
 * Matlab/My_LDPC_Erasure_Decoder.m:3-50 (MP only; reference itenum = 50, :10). Values 0/1 or -1. */
int oracle_ldpc_binary_mp_decode(const oracle_code *c, const int16_t *recv, int itenum, int16_t *msg,
                                 int *iterations);
/* Matlab/My_LDPC_HybridML_Erasure_Decoder.m:3-90 (MP + GF(2) elimination; itenum = 10, :9). */
int oracle_ldpc_binary_hybridml_decode(const oracle_code *c, const int16_t *recv, int itenum,
                                       int16_t *msg, int *iterations, int *info);

/* ---- a5: Reed-Solomon -------------------------------------------------------------------------- */
/* Generator of Matlab/Test_My_RS_Decode.m:22,30-37: G(row,col) = alpha^(row*col) (1-based, alpha = 2),
 * then G = inv(G(1:k,1:k)) * G.  g[k*n] row-major.  Returns 0, -1 if the k x k block is singular. */
int oracle_rs_generator(int n, int k, uint8_t *g);
/* codeword = source * G  (Test_My_RS_Decode.m:48) */
void oracle_rs_encode(int n, int k, const uint8_t *g, const uint8_t *source, uint8_t *codeword);
/* Matlab/My_RS_Decode_Optimize_With_GFTables.m:15-118.  recv_ind[k]: 0-BASED ascending positions of the
 * first k received symbols (ReedSolomonErasureCodes.m:80-81), recv_val[k] their values.  msg[k].
 * Returns 0; 1 when the matrix was found rank deficient (:95-97, reference does nothing). */
int oracle_rs_decode(int n, int k, const uint8_t *g, const uint16_t *recv_ind, const uint8_t *recv_val,
                     uint8_t *msg);

/* ---- a9: channel models ------------------------------------------------------------------------ */
/* One step of Matlab/Bursty_Error_Channel_Model_Generator.m:12-47.  rand_num / state_rand_num are
 * the two uniform draws of :25-26.  Returns error_out; *next_state as in :27-47. */
int oracle_bursty_channel_step(int current_state, double alpha, double beta, double good_transition_bias,
                               double rand_num, double state_rand_num, int *next_state);

/* ---- synthetic inputs (include/ldpc_erasure_amd_synth.h), so that CPU and GPU see the same frames */
void oracle_synth_coefs(uint64_t seed, int nnz, uint8_t *coefs);
void oracle_synth_source(uint64_t seed, int64_t frame0, int nframes, int k, int S, uint8_t *src);
void oracle_synth_erasures_uniform(uint64_t seed, int64_t frame0, int nframes, int n, double per,
                                   uint8_t *erased);
/* Gilbert-Elliott over the symbol stream of frames [frame0, frame0+nframes): the chain state is
 * carried across symbols and frames (ErasureCodes_NonBinaryLDPCSim.m:163,192), so frame0 > 0 replays
 * the chain from global symbol 0. */
void oracle_synth_erasures_bursty(uint64_t seed, int64_t frame0, int nframes, int n, double alpha,
                                  double beta, double good_transition_bias, uint8_t *erased);

/* Erasure flags of the FPGA source kernel data_in (OpenCL/device/ldpc_erasure_decoder_top.cl:57-120): threefry4x32
 * with key {1, seed} and a counter incremented before every symbol (:74-75,96-97), erased iff (rv & 0x3F) <
 * PER_numerator_div_64 (:105).  count = numFrames * n symbols, frames concatenated. */
void oracle_fpga_data_in_erasures(int seed, int per_numerator_div_64, int64_t count, uint8_t *erased);
/* One frame through the FPGA decoder, OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:56-236: two copies of the
 * codeword, parity checks [0, m/2) swept in order on the first copy and [m/2, m) on the second (:98-176), the copies
 * merged after every iteration (:178-201), stop when num_current_correct == k (:205-207).  One 64-bit word per
 * symbol stands for the 128-word payload (every word goes through the same XORs).  is_erasure / payload are updated
 * in place to the state of the first copy after the loop; returns the number of the first k symbols still erased
 * (:214-220, a frame error when > 0); *iterations = iter_ind. */
int oracle_fpga_perf_decoder_frame(const oracle_code *c, int num_iter, uint8_t *is_erasure, uint64_t *payload,
                                   int *iterations);
void oracle_threefry4x32_20(const uint32_t ctr[4], const uint32_t key[4], uint32_t out[4]);

/* ---- batch helpers for bench.py's cpu_baseline leg (single thread each; callers may fork) ------- */
/* Decodes nframes S=1 frames laid out like the GPU ABI (sym[nframes*n], erased[nframes*n]). */
int oracle_ldpc_decode_batch_s1(const oracle_code *c, int nframes, const uint8_t *sym,
                                const uint8_t *erased, int itenum, int do_ml, uint8_t *out,
                                int32_t *sweeps, int32_t *residual, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
