// internal.h -- shared between the C-ABI translation unit and the HIP kernel translation units.
// Not installed; the public interface is include/ldpc_erasure_amd.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/ldpc_erasure_amd.h"

namespace ldpc_amd {

// ---------------------------------------------------------------------------------------------
// GF(2^8), primitive polynomial x^8+x^6+x^5+x^4+1 = 0x171 (369).  The reference fixes it at
// Matlab/ErasureCodes_NonBinaryLDPCSim.m:70 ([1 0 1 1 1 0 0 0 1], MSB first) and ships the resulting
// tables in Matlab/GF_256_add_mult_inv_tables.mat; tests compare ldpc_amd_gf_tables() with that file.
// ---------------------------------------------------------------------------------------------
constexpr int kPrimPoly = 0x171;

struct GfHost {
    uint8_t log[256];   // log[0] unused (0)
    uint8_t exp[512];   // exp[i] = alpha^(i mod 255)
    uint8_t inv[256];   // inv[0] = 0
    uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
};
const GfHost &gf_host();

// Per-coefficient byte-permute tables for the packed multiply (see gf256_dev.h): 8 dwords per coefficient.
void build_mul3_tables(uint32_t *tab /* [256*8] */);

// ---------------------------------------------------------------------------------------------
// Device view of one LDPC code
// ---------------------------------------------------------------------------------------------
struct DevCode {
    int n, k, m, nnz;
    int maxdeg;   // largest row degree
    int maxcoldeg;  // largest column degree (width of the per-frame padded source->target edge lists)
    int degpad;   // template bucket the kernels are instantiated for (8, 14, 16 or 24) >= maxdeg
    int mpad;     // m rounded up to a multiple of 64
    const uint32_t *row_ptr;  // [m+1]
    const uint32_t *edges;    // [nnz]  col | coef << 16 | log(coef) << 24   (CSR order, ascending cols)
    const uint16_t *ell_col;  // [degpad][mpad]  transposed padded rows, 0xFFFF = none
    const uint8_t *ell_logc;  // [degpad][mpad]  log(coef)
    const uint8_t *ell_coef;  // [degpad][mpad]  coef
    const uint32_t *ell_pk;   // [degpad][mpad]  col | log(coef) << 16 (S = 1 kernel: one LDS read per neighbour)
    const uint16_t *rx_off;   // [mpad / 64][degpad][64]  col * 2, padding n * 2: byte offset of the neighbour's 16-bit key word (peel_relax.inc;
                              // chunk-major: the entries of a check are a compile-time stride apart, the lanes of a chunk read consecutive ones)
    const uint8_t *rx_logc;   // [mpad / 64][degpad][64]  log(coef), same order
    const uint8_t *enc_invc;  // [m]  inverse of the coefficient each static encode step divides by
    const uint32_t *cell;     // [n][1 << cdw_shift]  column lists: check | coef << 16, 0xFFFFFFFF = none
    int cdw_shift;            // log2 of the padded column-list width (>= maxcoldeg)
    // static encode schedule (all parity symbols erased): rows sorted by dependency level
    const uint32_t *enc_steps;   // [m] row | (k+row) << 16
    const uint16_t *enc_lvlend;  // [enc_nlevels+1], [L] = end offset of level L, [0] = 0
    int enc_nlevels;
    const uint32_t *enc_src;     // [n][1 << cdw_shift] static symbol -> (slot * 128 | coef << 24) lists of the encoder (LDS offset of the slot's 128-byte piece)
    const uint16_t *enc_order;   // [k] source symbols ordered by the number of checks they feed (most first; index order within)
    const uint32_t *enc_lst;     // [enc_lst_n] the enc_src lists of the parity symbols, compact, in schedule order (level phase of the encoder)
    const uint16_t *enc_lst_off; // [m + 1] step -> first word of its list
    int enc_lst_n;               // 0: not available (more than 65534 words)
    // Grouped static encode schedule (round 4; DESIGN.md section 4.2 "encoder: levels collapsed offline").  The parity triangle's
    // dependency levels (27 for (2040,1530), 100 for (4080,3060)) are a chain of barriers in which the workgroup streams nothing.
    // The schedule is static per code, so the host collapses consecutive levels into GROUPS: a step of a group does not wait for
    // the in-group parities it depends on but PULLS their raw accumulators with composite coefficients
    //     val_r = inv_r * (acc_r ^ XOR_a c_ra * acc_a),   c_ra = sum over the dependency paths a -> r of prod (h * inv)
    // (GF(256) products are associative and distributive: the same bytes), and an in-group parity is not scattered into in-group
    // accumulators.  One barrier per group; a step pulls at most ENC_CAP accumulators.
    const uint32_t *encg_steps;   // [m] row | (k+row) << 16, sorted by group
    const uint16_t *encg_lvlend;  // [encg_nlevels + 1] end offsets of the groups
    const uint8_t *encg_invc;     // [m]
    const uint32_t *encg_src;     // [n][1 << cdw_shift] like enc_src, slots of the grouped order, in-group parity edges left out
    const uint32_t *encg_ent;     // [encg_ent_n] per step: its pull entries, then its scatter entries (slot * 128 | coef << 24)
    const uint16_t *encg_ent_off; // [m + 1]
    const uint8_t *encg_npull;    // [m] how many of a step's entries are pulls
    int encg_nlevels;             // number of groups, 0: no grouped schedule
    int encg_ent_n;
};

struct HostCode {
    int n = 0, k = 0, m = 0, nnz = 0, maxdeg = 0;
    std::vector<uint32_t> row_ptr;
    std::vector<uint16_t> cols;
    std::vector<uint8_t> coefs;
    DevCode dev{};
    int enc_info[5] = {0, 0, 0, 0, 0};   // levels of the plain static schedule, groups, pull entries, scatter entries, longest pull list
    std::vector<void *> allocs;
};

struct HostRs {
    int n = 0, k = 0;
    std::vector<uint8_t> g;  // [k][n] systematic generator
    uint8_t *d_g = nullptr;  // device copy, [k][n]
    uint8_t *d_pt = nullptr; // device copy of the parity part transposed: [n-k][k]
};

// Tuning / diagnostic knobs.  The defaults are what is shipped.  They are read from the environment
// (LDPC_AMD_<NAME>) ONCE, by ldpc_amd_init, and can be changed per context with ldpc_amd_configure: nothing on the call path
// looks at the process environment (a C-ABI library must not depend on getenv in a multi-threaded host).
struct Knobs {
    int apply_gather = 0;        // APPLY=gather: packet kernel in gather form (A/B baseline)
    int scatter_b = 256;         // SCATTER_B: bytes of every row per packet-kernel workgroup (256, 128, 64)
    int scatter_tiers = 2;       // SCATTER_TIERS: 1 = single tier
    int scatter_nt = 1;          // SCATTER_NT: non-temporal row loads / stores
    int scatter_xcd = 1;         // SCATTER_XCD: slices of a frame placed on one XCD
    int scatter_dyn = 1;         // SCATTER_DYN: 1 LDS counter, 0 fixed stride, 2 compacted list, 3 sorted list, 4 list sorted inside 64-row windows
    int scatter_r = 2;           // SCATTER_R: row pieces in flight per lane group, tier 1 (1, 2, 4)
    int scatter_r2 = 3;          // SCATTER_R2: ... tier 2 (2, 3, 4)
    int scatter_lists = 0;       // SCATTER_LISTS: the relaxation's schedules carry the steps' column lists (set-up of the packet kernel: one global round trip
                                 // instead of two).  Measured: packet kernels unchanged (3.143 / 4.327 / 6.439 against 3.138 / 4.334 / 6.442 ms), peel +0.005...0.06: off
    int scatter_pairs = 1;       // SCATTER_PAIRS: the relaxation's schedules group the levels in pairs (one barrier per pair, second halves pull); 0: plain levels
    int scatter_xl = 1;          // SCATTER_XL: the level phase's lists are translated to accumulator addresses at set-up (0: inside every level)
    int scatter_t2b = 256;       // SCATTER_T2B: bytes of every row per TIER-2 workgroup (256: one workgroup per CU; 128: two)
    int peel_wpb = 0;            // PEEL_WPB: frames per peel workgroup (0 = auto)
    int peel_gt = -1;            // PEEL_GT: S = 1 kernel reads the code tables from global memory (-1 = auto)
    int enc_persist = 1;         // ENC_PERSIST: packet encoder as persistent workgroups (tables and lists set up once per workgroup); 0: one workgroup per (frame, slice)
    int scatter_t2p = 2;         // SCATTER_T2P: tier 2 -- consecutive pieces of a frame per work item (1, 2, 4, 8): the frame's set-up is done once per item
    int scatter_t2p_force = 0;   // SCATTER_T2P_FORCE: 1 = SCATTER_T2P pieces per item whatever the length of the tier-2 list (tests)
    int peel_relax = 1;          // PEEL_RELAX: S = 1 decode (and the pattern-only runs) by time-stamp relaxation (peel_relax.inc); 0: the serial per-solve loop
    int ml_solve = 1;            // ML_SOLVE: 1 solve schedules + solve kernel, 0 solve inside the ML kernel, 2 emit only (diagnostic)
    int ml_dbg = 0;              // ML_DBG: diagnostic build only
    int ml_solve_b = 128;        // ML_SOLVE_B: bytes of every row per solve-kernel workgroup
    long long ml_arena_words = 0;   // ML_ARENA_WORDS: size of the schedule arena in 64-bit words (0 = auto)
    int ml_threads = 0;          // ML_THREADS: threads of the ML kernel's workgroup (0 = 1024 / ML_PACK)
    int ml_pi = 1;               // ML_PI: packets -- the fast path (ml_pi.inc: peel on + inactivation) before the exact elimination.  1: verified
                                 // (frames that fail the consistency test are redone exactly: the reference's bytes on ANY input); 2: not
                                 // verified (the reference's bytes when the received symbols are a codeword with erasures); 0: exact only
    int ml_pi_adaptive = 1;      // ML_PI_ADAPTIVE: skip the fast path's launches for a batch when the previous batch had no residual frame
    int ml_pi_imax = 256;        // ML_PI_IMAX: fast path -- frames that need more inactivations than this go to the exact elimination
    int ml_pi_wgs = 0;           // ML_PI_WGS: fast path -- at most this many workgroups (0: one per CU's LDS share)
    int ml_pi_waves = 4;         // ML_PI_WAVES: fast path -- at most this many wavefronts (frames) per workgroup
    int ml_pi_lds = 160;         // ML_PI_LDS: KB of LDS per fast-path workgroup (160: one workgroup of up to four frames per CU)
    int ml_overlap_prio = 0;     // ML_OVERLAP_PRIO: 1 = that second stream has the lowest priority (measured slower: 5.67 against 5.45 ms on cfg 3)
    int ml_overlap = 2;          // ML_OVERLAP: packets -- the factorisation runs on a second stream beside the packet kernel; 0 = behind it
    int ml_pack = 2;             // ML_PACK: ML-kernel workgroups per CU (1, 2, 3, 4): 1024 / P threads and 160 KB / P of LDS each
    int enc_clist = 1;           // ENC_CLIST: encoder -- the level phase reads the parity symbols' lists from LDS (compact copy); 0: from global memory
    int enc_b = 128;             // ENC_B: encoder piece size (128: two workgroups per CU; 256: the decoder's plan)
    int enc_list = 0;            // ENC_LIST: encoder streams the source rows in the order of their column degree
    int enc_group = 1;           // ENC_GROUP: encoder runs the grouped (level-collapsed) static schedule when the code has one; 0: level by level
    int enc_cap = 8;             // ENC_CAP: read when a code is REGISTERED -- a step of the grouped schedule pulls at most this many accumulators (0: no grouped schedule)
    int rs_generic = 0;          // RS=generic: RS decode always through the generic LDS kernel
    int rs_vw = 0;               // RS_VW: dwords per lane of the packet RS kernel (0 = auto = 1; 2 and 4 where S allows)
    int host_pipeline = 1;       // HOST_PIPELINE: chunked upload / compute / download pipeline for large host buffers
    long fpga_chunk = 65536;     // FPGA_CHUNK: frames per chunk of the streamed FPGA-harness run
    long chunk_s1 = 65536;       // CHUNK_S1: S = 1 decode -- frames per launch of a long batch (packets always 16384: their schedules are per-frame workspaces).
                                 // Every launch drains the device once: 65536 (4080,3060) frames 2.233 ms in four launches, 2.082 in one (tools/ab_chunk_s1.py)
};
// key: "LDPC_AMD_SCATTER_B" or "SCATTER_B" (case-insensitive); value nullptr or "" restores the default.  0 = OK, -1 = unknown key / bad value.
int knob_set(Knobs &k, const char *key, const char *value);
bool knobs_from_env(Knobs &k, std::string *rejected);   // false: *rejected = "LDPC_AMD_X=value" of the first value a knob refused

// Growable device scratch
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace ldpc_amd

struct ldpc_amd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t aux_ml = nullptr;                      // packets: the ML factorisation beside the packet kernel (created on first use)
    hipEvent_t ml_events[2] = {nullptr, nullptr};      // fork / join of that stream
    hipStream_t aux_in = nullptr, aux_out = nullptr;   // host-pointer pipeline: H2D / D2H streams (created on first use)
    hipEvent_t pipe_events[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // ... and its events (created once)
    std::string err;
    std::vector<ldpc_amd::HostCode *> codes;
    std::vector<ldpc_amd::HostRs *> rs;
    // workspaces
    ldpc_amd::Scratch sched;    // per-frame schedules (packet path)
    ldpc_amd::Scratch schedlists;  // ... the steps' column lists in schedule order (relaxation schedules)
    ldpc_amd::Scratch schedpull;   // ... their per-step records when the levels come in pairs (16 bytes per step)
    ldpc_amd::Scratch mlws;     // ML stage matrices
    ldpc_amd::Scratch mlstate;  // residual erasure masks of the frames handed to the ML stage
    ldpc_amd::Scratch mlops;    // packets: arena of the ML stage's solve schedules
    ldpc_amd::Scratch mlrec;    // packets: [ML-list slot][8] u32 schedule records
    ldpc_amd::Scratch mllist;   // [1 + nframes] int32: count, frame ids
    ldpc_amd::Scratch biglist;  // [1 + nframes] int32: frames with many steps (scatter tier 2)
    ldpc_amd::Scratch encctr;   // persistent encoder: ring of (item counter, workgroups done) pairs, self-resetting
    unsigned enc_launches = 0;
    ldpc_amd::Scratch stage_in, stage_er, stage_out, stage_i32;  // host-pointer staging
    bool ml_head_valid = false;                   // a packet-mode ML stage has copied its demand there at least once
    unsigned long long *ml_head_host = nullptr;   // pinned: arena words the last packet-mode ML stage asked for
    int *dev_err_host = nullptr;   // pinned, device-visible: error bits set by kernels (check_device_error turns them into LDPC_AMD_EHIP)
    size_t ml_arena_words = 0;  // current size of the solve-schedule arena (64-bit words), grown on demand
    void *pin = nullptr;        // pinned host bounce block of the small-call path
    size_t pin_cap = 0;
    ldpc_amd::Scratch rsws;
    ldpc_amd::Scratch rsbad;    // int: malformed blocks of the last RS decode (decoded to zeros)
    // FPGA-harness emulation state (ldpc_amd_data_in / _ldpc_erasure_decoder / _data_out)
    // The run is streamed in chunks like the FPGA's frame loop (ldpc_erasure_decoder_perf_tests.cl:52): fpga_erased holds the
    // flags of ONE chunk, fpga_stats the two running counters (+ per-frame results: of the whole run when it is short
    // enough to keep them for ldpc_amd_fpga_frame_stats, else of one chunk).
    ldpc_amd::Scratch fpga_erased, fpga_stats;
    long fpga_frames = -1;        // numFrames of the last ldpc_amd_data_in, -1 = none
    long fpga_first = 0;          // first frame of the run in the source's stream (ldpc_amd_data_in_at: a shard of a multi-device run)
    long fpga_decoded = -1;       // numFrames the last decoder call ran over, -1 = no decoder call since data_in
    bool fpga_kept = false;       // per-frame results of the whole run are in fpga_stats
    int fpga_code_ind = -1;
    int fpga_per64 = 0;
    int fpga_seed = 0;
    int fpga_binary_code[4] = {-1, -1, -1, -1};
    int sm_count = 256;
    ldpc_amd::Knobs knobs;
    // profiling (ldpc_amd_set_profiling): event pairs per kernel kind
    int profiling = 0;          // 0 off, 1 the three kinds of a call, 2 + the nested brackets (tier 2 alone, solve kernel alone)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[LDPC_AMD_PROF_KINDS];
    std::vector<hipEvent_t> prof_pool;
    std::string prof_names[LDPC_AMD_PROF_KINDS];   // template instantiation the last launch of each kind used
    int last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // launch plan of the last decode (ldpc_amd_last_plan)
    int last_enc_grouped = 0;                      // the last packet-mode encode ran the grouped static schedule (ldpc_amd_encode_info)
};

namespace ldpc_amd {

// ---- launchers implemented in the .hip files (all asynchronous on ctx->stream) -----------------
struct DecodeArgs {
    DevCode code;
    int S;
    int64_t nframes;
    const uint8_t *sym;     // [nframes][in_rows][S]
    const uint8_t *erased;  // [nframes][n] or nullptr (encode mode: rows >= in_rows are erased)
    int in_rows;            // n (decode) or k (encode)
    int max_sweeps;
    int do_ml;
    uint8_t *out;           // [nframes][n][S]
    int32_t *sweeps, *residual, *status;  // may be nullptr
    int flags_only = 0;                   // peel only (no data movement): FPGA-harness statistics
    int32_t *residual_sys = nullptr;      // [nframes] unknown symbols among the first k, or nullptr
    int inplace = 0;                      // out == sym: received rows are not rewritten
};

hipError_t upload_constants(hipStream_t s);
int launch_decode(ldpc_amd_ctx *ctx, const DecodeArgs &a);
int launch_encode(ldpc_amd_ctx *ctx, const DevCode &code, int S, int64_t nframes, const uint8_t *src, uint8_t *cw);
int launch_fpga_halves(ldpc_amd_ctx *ctx, const DevCode &code, int64_t nframes, const uint8_t *erased, int num_iter,
                       int32_t *residual_sys, int32_t *iterations);
int launch_selftest(ldpc_amd_ctx *ctx);
int launch_copy_probe(ldpc_amd_ctx *ctx, const uint8_t *src, uint8_t *dst, uint64_t bytes, int reps, double *ms);
int launch_synth_source(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int k, int S, uint8_t *d);
int launch_synth_erasures(ldpc_amd_ctx *ctx, uint64_t seed, uint32_t stream_id, int64_t first, int64_t count,
                          uint64_t thresh, uint8_t *d);
int launch_synth_bursty(ldpc_amd_ctx *ctx, uint64_t seed, int64_t first, int64_t count, double alpha, double beta,
                        double bias, uint8_t *d);
int launch_synth_fpga(ldpc_amd_ctx *ctx, uint32_t seed, uint64_t first, int64_t count, int per64, uint8_t *d);
int launch_rs_decode(ldpc_amd_ctx *ctx, const HostRs &rs, int S, int64_t nblocks, const uint16_t *idx,
                     const uint8_t *val, uint8_t *msg);
int launch_rs_encode(ldpc_amd_ctx *ctx, const HostRs &rs, int S, int64_t nblocks, const uint8_t *src, uint8_t *cw);
int launch_fpga_stats(ldpc_amd_ctx *ctx, const DevCode &code, int rs_n, int rs_k, int64_t nframes,
                      const uint8_t *erased0, const int32_t *residual_k, unsigned long long *stats /* [2], accumulated */);

int scratch_reserve(ldpc_amd_ctx *ctx, Scratch &s, size_t bytes);
// Brackets one kernel launch with events when profiling is on (no-ops otherwise).
hipEvent_t prof_begin(ldpc_amd_ctx *ctx, int level = 1);
void prof_end(ldpc_amd_ctx *ctx, int kind, hipEvent_t start);
int set_error(ldpc_amd_ctx *ctx, int code, const char *fmt, ...);
// After a synchronisation: LDPC_AMD_EHIP (and the word cleared) if a kernel of this context reported a violated assumption.
int check_device_error(ldpc_amd_ctx *ctx);

#define LDPC_HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ldpc_amd::set_error((ctx), LDPC_AMD_EHIP, "%s failed: %s (%s:%d)", #expr,      \
                                       hipGetErrorString(e__), __FILE__, __LINE__);               \
    } while (0)

}  // namespace ldpc_amd
