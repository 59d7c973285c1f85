// kernels.hip -- the single device translation unit of libldpc_erasure_amd.so (gfx950 / CDNA4 only).
//
// Hot path of the reference:
//   a1  in-order message-passing sweeps   Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:13-59
//   a2  rhs build                         ...Decoder.m:63-82
//   a3  forward elimination               ...Decoder.m:85-115
//   a4  back substitution + write-back    ...Decoder.m:117-129
//   a5  RS erasure decode                 Matlab/My_RS_Decode_Optimize_With_GFTables.m:15-118   (rs_kernels.inc)
//   a8  systematic encoder                Matlab/ErasureCodes_NonBinaryLDPCSim.m:173-182
//
// Design (DESIGN.md has the long version):
//   * The erasure PATTERN of a frame fixes which check solves which symbol and in what order; the symbol
//     VALUES only ride along.  So every frame is first "peeled" on its erasure flags alone by one wavefront
//     (exact emulation of the reference's sequential Gauss-Seidel sweep: 64 checks per step, one lane per
//     check, wave ballot/ffs to pick the next check with exactly one unknown, readlane to broadcast the
//     solved symbol).  The result is a list of (check, symbol) steps tagged with a dependency level.
//   * S = 1 (the Matlab model): the same wavefront then applies the steps level by level, one lane per
//     step, on the codeword held in LDS, with GF(256) log/antilog tables in LDS.
//   * S >= 16 (packets): a second kernel streams the S-byte rows: every wavefront handles one step at a
//     time with a wave-uniform coefficient, 16 bytes per lane, v_perm_b32 multiply (gf256_dev.h).
//   * Residual frames are compacted into a list and solved by the ML kernel (exact pivot order of the
//     reference, including its behaviour on rank-deficient systems).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "internal.h"
#include "gf256_dev.h"
#include "../../include/ldpc_erasure_amd_synth.h"

namespace ldpc_amd {

namespace {

constexpr int kWave = 64;
using us2 = unsigned short __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// Orders LDS traffic between the lanes of ONE wavefront (hardware executes a wave's LDS ops in order;
// this only stops the compiler from moving accesses across the point).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// residual frames are filed under size classes (ml_list header: count, class counts; then frame ids, then the class lists)
constexpr int kMlClasses = 16, kMlHdr = 32;
// bits of the context's device-error word (pinned host memory the kernels write to directly; read by check_device_error)
constexpr int kDevErrLdsBase = 1;   // a kernel that encodes absolute LDS addresses found its dynamic LDS not at address 0
constexpr int kDevErrRelaxCap = 2;  // a loop of ldpc_peel_relax_kernel ran into its safety cap (never in a correct run)

// Diagnostic build only (-DLDPC_AMD_STAMPS, tools/stamp_peel.py): per-phase cycle sums of the peel kernel go to a
// buffer nothing else reads.  The product build contains no stamp.
#ifdef LDPC_AMD_STAMPS
__device__ unsigned long long g_peel_stamps[56];   // [0..15] peel / packet kernel, [16..31] ML kernel, [32..39] ML solve kernel, [40..55] ML fast path
#define LDPC_STAMP(i)                                                                     \
    do {                                                                                  \
        const unsigned long long t__ = __builtin_amdgcn_s_memtime();                      \
        if (lane_id() == 0) atomicAdd(&g_peel_stamps[i], t__ - stamp_prev);               \
        stamp_prev = __builtin_amdgcn_s_memtime();                                        \
    } while (0)
#define LDPC_STAMP_INIT unsigned long long stamp_prev = __builtin_amdgcn_s_memtime()
#else
#define LDPC_STAMP(i) do { } while (0)
#define LDPC_STAMP_INIT do { } while (0)
#endif

// =================================================================================================
// a1: peeling of one frame by one wavefront -- exact in-order (Gauss-Seidel) sweep semantics
// =================================================================================================
// Reference loop (...Decoder.m:21-59): for every sweep, for ii = 1..m IN ORDER: count the erased
// neighbours of check ii in the CURRENT state; if exactly one, solve it (visible to check ii+1 at once).
// After the sweep stop if no erasure is left among ALL n symbols.
//
// Emulation: 64 consecutive checks at a time, lane = check.  cnt / xs (xor of erased neighbour ids) /
// ml (max level of known neighbours) are computed from the state array at chunk start and then patched
// for lanes ABOVE a solving lane only -- exactly the visibility rule of the sequential loop.
//   st[j]   : 0xFFFF erased and unknown, otherwise the dependency level of symbol j (0 = received)
//   steps[] : (check | symbol << 16) in solve order; slvl[] the step's level (1 + max level of its inputs)
// EllT: uint16_t (neighbour ids) or uint32_t (id | log(coef) << 16, the table of the S = 1 kernel)
template <int MAXDEG, typename EllT>
__device__ __forceinline__ void peel_wave(const EllT *ell_col, int mpad, uint16_t *st, uint32_t *steps,
                                          uint16_t *slvl, int max_sweeps, int &nsteps, int &remaining,
                                          int &sweeps, int &maxlvl)
{
    const int lane = lane_id();
    const int nchunks = mpad >> 6;
    nsteps = 0; sweeps = 0; maxlvl = 0;
    while (sweeps < max_sweeps) {               // :21  while (stopsig==0) && (itestep<itenum)
        sweeps++;                               // :23
        const int solved_before = nsteps;
        for (int ch = 0; ch < nchunks && remaining > 0; ch++) {  // :27 (rows past the last erasure change nothing)
            const int row = (ch << 6) + lane;
            // two rounds of independent LDS reads: the check's neighbour ids, then their states
            uint32_t c[MAXDEG], sv[MAXDEG];
#pragma unroll
            for (int t = 0; t < MAXDEG; t++) c[t] = (uint32_t)ell_col[t * mpad + row] & 0xFFFFu;
#pragma unroll
            for (int t = 0; t < MAXDEG; t++) sv[t] = st[c[t] == 0xFFFFu ? 0u : c[t]];
            int cnt = 0;
            uint32_t xs = 0, ml = 0;
            us2 cp[MAXDEG / 2];  // neighbour ids packed in pairs for the membership test below (0xFFFF = none)
#pragma unroll
            for (int t = 0; t < MAXDEG; t++) {  // :30-35
                const bool valid = c[t] != 0xFFFFu;
                const bool er = valid && sv[t] == 0xFFFFu;
                cnt += er ? 1 : 0;
                xs ^= er ? c[t] : 0u;
                ml = max(ml, (valid && !er) ? sv[t] : 0u);
                if (t & 1) cp[t >> 1] = us2{(unsigned short)c[t - 1], (unsigned short)c[t]};
            }
            uint64_t elig = ~0ull;
            for (;;) {
                const uint64_t cand = __ballot(cnt == 1) & elig;  // :37 num_erasures == 1
                if (cand == 0) break;
                const int l = __ffsll((long long)cand) - 1;        // first such check in row order
                const uint32_t e = __builtin_amdgcn_readlane(xs, l);
                const uint32_t lv = __builtin_amdgcn_readlane(ml, l) + 1u;
                if (lane == 0) {
                    st[e] = (uint16_t)lv;                          // :47 y_current(erasure_ind) = ...
                    steps[nsteps] = (uint32_t)((ch << 6) + l) | (e << 16);
                    slvl[nsteps] = (uint16_t)lv;
                }
                nsteps++; remaining--;
                maxlvl = max(maxlvl, (int)lv);
                elig = (l == 63) ? 0ull : (~0ull << (l + 1));
                // does this check contain e?  min over (id ^ e) of the packed pairs is 0 in the matching half
                const us2 ee = us2{(unsigned short)e, (unsigned short)e};
                us2 mn = us2{(unsigned short)0xFFFFu, (unsigned short)0xFFFFu};
#pragma unroll
                for (int t = 0; t < MAXDEG / 2; t++) mn = __builtin_elementwise_min(mn, (us2)(cp[t] ^ ee));
                const bool hit = (mn.x == 0) || (mn.y == 0);
                if (hit && lane > l) { cnt--; xs ^= e; ml = max(ml, lv); }
                if (lane == l) cnt = 0;
            }
            wave_sync();
        }
        if (remaining == 0) break;              // :51-54
        // a sweep that solved nothing leaves the state as it was: every further sweep repeats it, the reference runs
        // them all and returns itestep = itenum -- same result without running them
        if (nsteps == solved_before) { sweeps = max_sweeps; break; }
    }
}

// Counting sort of the steps by level (wavefront-local, LDS).  On return sorted[] holds the steps grouped
// by level and lvlend[L] (L = 0..maxlvl) is the end offset of level L (lvlend[0] = 0).
__device__ __forceinline__ void sort_steps_by_level(const uint32_t *steps, const uint16_t *slvl, int nsteps,
                                                    int maxlvl, uint32_t *lvlend, uint32_t *sorted)
{
    const int lane = lane_id();
    for (int i = lane; i <= maxlvl; i += kWave) lvlend[i] = 0;
    wave_sync();
    for (int i = lane; i < nsteps; i += kWave) atomicAdd(&lvlend[slvl[i]], 1u);
    wave_sync();
    uint32_t carry = 0;
    for (int base = 0; base <= maxlvl; base += kWave) {
        const int idx = base + lane;
        const uint32_t v = (idx <= maxlvl) ? lvlend[idx] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (idx <= maxlvl) lvlend[idx] = incl - v + carry;  // exclusive prefix = start of level idx
        carry += __shfl(incl, 63);
    }
    wave_sync();
    for (int i = lane; i < nsteps; i += kWave) {
        const uint32_t pos = atomicAdd(&lvlend[slvl[i]], 1u);  // start -> end while filling
        sorted[pos] = steps[i];
    }
    wave_sync();
}

// =================================================================================================
// Kernel A: peel (+ apply when S == 1).  One wavefront per frame, WPB frames per workgroup.
// =================================================================================================
struct PeelLds {        // byte offsets into dynamic LDS
    int ell_col;        // u16 [degpad][mpad]
    int ell_logc;       // u8  [degpad][mpad]      (S == 1 only)
    int lg, ex;         // u8 [256], u8 [512]      (S == 1 only)
    int wave0;          // first per-wave region
    int wave_stride;
    // inside a per-wave region
    int st;             // u16 [n]  later reused as u32 lvlend[maxlvl+1]
    int y;              // u8  [n]                 (S == 1 only)
    int steps;          // u32 [m]
    int slvl;           // u16 [m]
    int sorted;         // u32 [m]
    int total;
};

struct PeelArgs {
    DevCode code;
    PeelLds lds;
    int64_t nframes;
    const uint8_t *sym;     // S == 1: [nframes][in_rows]
    const uint8_t *erased;  // [nframes][n] or nullptr
    int in_rows;
    int max_sweeps, do_ml;
    uint8_t *out;           // S == 1: [nframes][n]
    int32_t *sweeps, *residual, *status;
    int32_t *residual_sys;  // unknown symbols among the first k (FPGA frame-error criterion), or nullptr
    // packet path: schedule output
    uint32_t *sched_hdr;    // [nframes][2]  nsteps, maxlvl
    uint32_t *sched_steps;  // [nframes][m]
    uint16_t *sched_lvlend; // [nframes][m+1]
    uint8_t *sched_invc;    // [nframes][m]  inverse of the coefficient step i divides by (packet kernel)
    int32_t *big_list;      // [0] = count, [1] = tier 2's work counter, [2..] frames with more than tcap steps (scatter tier 2), or nullptr
    int tcap;
    // ML hand-off
    int32_t *ml_list;       // [0] = count, [1..kMlClasses] = counts of the size classes, [kMlHdr + slot] = frame id,
                            // [kMlHdr + nframes + k nframes + i] = slot of the i-th frame of size class k (largest residual first)
    uint8_t *ml_state;      // [slot][n]  1 = still erased
};

// GT ("global tables"): the code tables are read from global memory (they are shared by all frames and stay in the
// vector L1 / L2) instead of being staged in LDS; the LDS then holds only per-frame state, which puts more frames on a
// CU.  Chosen by the host for long S = 1 batches (launch_decode).
template <int MAXDEG, bool FUSED_S1, bool GT>
__global__ __launch_bounds__(1024) void ldpc_peel_kernel(PeelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevCode &cd = a.code;
    const int n = cd.n, m = cd.m, mpad = cd.mpad;
    const int lane = lane_id(), wave = wave_id();
    const int wpb = (int)(blockDim.x >> 6);

    // ---- code tables -> LDS (shared by the frames of this workgroup)
    // S = 1: one 32-bit word per neighbour (id | log(coef) << 16); packets: the 16-bit ids only
    using EllT = typename std::conditional<FUSED_S1, uint32_t, uint16_t>::type;
    const EllT *gsrc = FUSED_S1 ? reinterpret_cast<const EllT *>(cd.ell_pk) : reinterpret_cast<const EllT *>(cd.ell_col);
    const EllT *ell_col = GT ? gsrc : reinterpret_cast<const EllT *>(smem + a.lds.ell_col);
    if (!GT) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(gsrc);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smem + a.lds.ell_col);
        const int words = (MAXDEG * mpad * (int)sizeof(EllT)) >> 2;
        for (int i = (int)threadIdx.x; i < words; i += (int)blockDim.x) dst[i] = src[i];
    }
    uint8_t *lg = smem + a.lds.lg;
    uint8_t *ex = smem + a.lds.ex;
    if (FUSED_S1) {
        for (int i = (int)threadIdx.x; i < 256; i += (int)blockDim.x) lg[i] = c_log[i];
        for (int i = (int)threadIdx.x; i < 512; i += (int)blockDim.x) ex[i] = c_exp[i];
    }
    __syncthreads();

    const int64_t f = (int64_t)blockIdx.x * wpb + wave;
    if (f >= a.nframes) return;  // no workgroup barrier below this point
    LDPC_STAMP_INIT;

    unsigned char *wbase = smem + a.lds.wave0 + wave * a.lds.wave_stride;
    uint16_t *st = reinterpret_cast<uint16_t *>(wbase + a.lds.st);
    uint32_t *steps = reinterpret_cast<uint32_t *>(wbase + a.lds.steps);
    uint16_t *slvl = reinterpret_cast<uint16_t *>(wbase + a.lds.slvl);
    uint32_t *sorted = reinterpret_cast<uint32_t *>(wbase + a.lds.sorted);

    // ---- load the frame's erasure flags (the symbols of an S == 1 frame are loaded after the peel, see below)
    int remaining = 0;
    const uint8_t *er = a.erased ? a.erased + f * n : nullptr;
    const uint8_t *sy = FUSED_S1 ? a.sym + f * a.in_rows : nullptr;
    const bool fast = ((n & 7) == 0) && ((a.in_rows & 7) == 0) && ((reinterpret_cast<uintptr_t>(er) & 7) == 0) &&
                      ((reinterpret_cast<uintptr_t>(sy) & 7) == 0);
    {
        if (fast) {
            // 8 flags per 64-bit load, 4 independent loads in flight per lane
            constexpr int U = 4;
            const int nq = n >> 3;
            const uint64_t *er64 = reinterpret_cast<const uint64_t *>(er);
            int cnt = 0;
            for (int q0 = 0; q0 < nq; q0 += kWave * U) {
                uint64_t ew[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = q0 + u * kWave + lane;
                    ew[u] = (er && q < nq) ? er64[q] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = q0 + u * kWave + lane;
                    if (q < nq) {
                        uint32_t s4[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int b = 0; b < 8; b++) {
                            const bool e = er ? (((ew[u] >> (8 * b)) & 0xFFull) != 0) : (q * 8 + b >= a.in_rows);
                            if (e) { s4[b >> 1] |= 0xFFFFu << (16 * (b & 1)); cnt++; }
                        }
                        *reinterpret_cast<U4 *>(st + q * 8) = U4{s4[0], s4[1], s4[2], s4[3]};
                    }
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
            remaining = cnt;
        } else {
            for (int j0 = 0; j0 < n; j0 += kWave) {
                const int j = j0 + lane;
                bool e = false;
                if (j < n) {
                    e = er ? (er[j] != 0) : (j >= a.in_rows);
                    st[j] = e ? (uint16_t)0xFFFFu : (uint16_t)0;
                }
                remaining += __popcll(__ballot(e));
            }
        }
    }
    wave_sync();

    LDPC_STAMP(0);  // frame load
    int nsteps, sweeps, maxlvl;
    peel_wave<MAXDEG, EllT>(ell_col, mpad, st, steps, slvl, a.max_sweeps, nsteps, remaining, sweeps, maxlvl);
    LDPC_STAMP(1);  // peeling sweeps

    // ---- per-frame results + hand-off of residual frames to the ML stage
    int stcode = LDPC_AMD_ST_MP_DONE;
    if (remaining > 0) {
        stcode = LDPC_AMD_ST_ML_SKIPPED;
        if (a.do_ml && remaining <= m) {   // ...Decoder.m:61; E > n-k cannot be written back (:127)
            int slot = 0;
            if (lane == 0) {
                slot = atomicAdd(&a.ml_list[0], 1);
                a.ml_list[kMlHdr + slot] = (int32_t)f;
                // size classes by residual count: the ML kernel and the solve kernel hand out the large systems first
                // (their run time is the slowest workgroup's; the last class handed out holds the smallest systems)
                const int k_ = kMlClasses - 1 - min(kMlClasses - 1, remaining * kMlClasses / m);
                a.ml_list[kMlHdr + a.nframes + (int64_t)k_ * a.nframes + atomicAdd(&a.ml_list[1 + k_], 1)] = slot;
            }
            slot = (int)uniform((uint32_t)slot);
            uint8_t *ms = a.ml_state + (int64_t)slot * n;
            for (int j = lane; j < n; j += kWave) ms[j] = (st[j] == 0xFFFFu) ? 1 : 0;
        }
    }
    if (a.residual_sys) {
        int rs = 0;
        for (int j0 = 0; j0 < cd.k; j0 += kWave) {
            const int j = j0 + lane;
            rs += __popcll(__ballot(j < cd.k && st[j] == 0xFFFFu));
        }
        if (lane == 0) a.residual_sys[f] = rs;
    }
    if (lane == 0) {
        if (a.sweeps) a.sweeps[f] = sweeps;      // :130 iterations = itestep
        if (a.residual) a.residual[f] = remaining;
        if (a.status) a.status[f] = stcode;
    }
    wave_sync();

    // ---- group the steps by dependency level (st is dead from here on: reuse it for the level offsets)
    uint32_t *lvlend = reinterpret_cast<uint32_t *>(st);
    LDPC_STAMP(2);  // status words / ML hand-off
    sort_steps_by_level(steps, slvl, nsteps, maxlvl, lvlend, sorted);
    LDPC_STAMP(3);  // level sort

    if (!FUSED_S1) {
        if (!a.sched_hdr) return;  // flags-only run
        if (lane == 0) {
            a.sched_hdr[2 * f] = (uint32_t)nsteps;
            a.sched_hdr[2 * f + 1] = (uint32_t)maxlvl;
            if (a.big_list && nsteps > a.tcap) a.big_list[2 + atomicAdd(&a.big_list[0], 1)] = (int32_t)f;
        }
        uint32_t *gs = a.sched_steps + f * m;
        uint8_t *gi = a.sched_invc + f * m;
        for (int i = lane; i < nsteps; i += kWave) {
            const uint32_t step = sorted[i];
            gs[i] = step;
            // the coefficient of the solved symbol in its check (:47 divides by it): found once here, not by every
            // slice workgroup of the packet kernel
            const uint32_t row = step & 0xFFFFu, t = step >> 16;
            int pos = 0;
#pragma unroll
            for (int tt = 0; tt < MAXDEG; tt++) pos = (((uint32_t)ell_col[tt * mpad + row] & 0xFFFFu) == t) ? tt : pos;
            gi[i] = c_inv[cd.ell_coef[(size_t)pos * mpad + row]];
        }
        uint16_t *gl = a.sched_lvlend + f * (m + 1);
        for (int i = lane; i <= maxlvl; i += kWave) gl[i] = (uint16_t)lvlend[i];
        LDPC_STAMP(4);  // schedule write-out
        return;
    }

    // ---- S == 1: the codeword goes into the LDS space of the (now dead) unsorted step list
    uint8_t *y = reinterpret_cast<uint8_t *>(steps);
    if (fast) {
        constexpr int U = 4;
        const int nq = n >> 3;
        const uint64_t *er64 = reinterpret_cast<const uint64_t *>(er);
        const uint64_t *sy64 = reinterpret_cast<const uint64_t *>(sy);
        for (int q0 = 0; q0 < nq; q0 += kWave * U) {
            uint64_t ew[U], sw[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u * kWave + lane;
                ew[u] = (er && q < nq) ? er64[q] : 0ull;
                sw[u] = (q < nq && q * 8 < a.in_rows) ? sy64[q] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = q0 + u * kWave + lane;
                if (q < nq) {
                    uint64_t keep = 0;  // 0xFF in every byte that was received
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const bool e = er ? (((ew[u] >> (8 * b)) & 0xFFull) != 0) : (q * 8 + b >= a.in_rows);
                        if (!e) keep |= 0xFFull << (8 * b);
                    }
                    *reinterpret_cast<uint64_t *>(y + q * 8) = sw[u] & keep;
                }
            }
        }
    } else {
        for (int j = lane; j < n; j += kWave) {
            const bool e = er ? (er[j] != 0) : (j >= a.in_rows);
            y[j] = (e || j >= a.in_rows) ? (uint8_t)0 : sy[j];
        }
    }
    wave_sync();

    // ---- S == 1: apply the steps on the LDS-resident codeword, one lane per step, level by level.
    //      y(e) = inv(H(i,e)) * sum_{j != e} H(i,j) y(j)      (...Decoder.m:39-47)
    for (int L = 1; L <= maxlvl; L++) {
        const int s0 = (int)lvlend[L - 1], s1 = (int)lvlend[L];
        for (int base = s0; base < s1; base += kWave) {
            const int i = base + lane;
            if (i < s1) {
                const uint32_t step = sorted[i];
                const int row = (int)(step & 0xFFFFu);
                const uint32_t tgt = step >> 16;
                // four rounds of independent LDS reads instead of MAXDEG dependent chains
                uint32_t c[MAXDEG], lc[MAXDEG], v[MAXDEG], lv[MAXDEG];
#pragma unroll
                for (int t = 0; t < MAXDEG; t++) {
                    const uint32_t w = (uint32_t)ell_col[t * mpad + row];
                    c[t] = w & 0xFFFFu;
                    lc[t] = w >> 16;   // (0 for the 16-bit table of the packet instantiation, which never gets here)
                }
#pragma unroll
                for (int t = 0; t < MAXDEG; t++) v[t] = y[c[t] == 0xFFFFu ? 0u : c[t]];
#pragma unroll
                for (int t = 0; t < MAXDEG; t++) lv[t] = lg[v[t]];
                uint32_t sum = 0, lce = 0;
#pragma unroll
                for (int t = 0; t < MAXDEG; t++) {
                    const uint32_t p = ex[lv[t] + lc[t]];
                    if (c[t] == tgt) lce = lc[t];
                    else if (c[t] != 0xFFFFu && v[t]) sum ^= p;
                }
                y[tgt] = sum ? ex[lg[sum] + 255u - lce] : (uint8_t)0;
            }
            wave_sync();
        }
    }

    LDPC_STAMP(6);  // S = 1 apply
    // ---- write Msg (...Decoder.m:129); unknown symbols are 0
    uint8_t *o = a.out + f * n;
    if ((n & 3) == 0) {
        const uint32_t *yw = reinterpret_cast<const uint32_t *>(y);
        uint32_t *ow = reinterpret_cast<uint32_t *>(o);
        for (int i = lane; i < (n >> 2); i += kWave) ow[i] = yw[i];
    } else {
        for (int j = lane; j < n; j += kWave) o[j] = y[j];
    }
    LDPC_STAMP(7);  // output store
}

// =================================================================================================
// Kernel B (packets, S a multiple of 16): copy the received rows and execute the frame's steps.
// One workgroup per frame; one wavefront executes one step at a time:
//     out[e] = inv(h_ie) * XOR_{j != e} h_ij * out[j]           16 bytes per lane, coefficient wave-uniform
// =================================================================================================
struct ApplyArgs {
    DevCode code;
    int S;
    int64_t nframes;
    const uint8_t *sym;      // [nframes][in_rows][S]
    const uint8_t *erased;   // [nframes][n] or nullptr (encode)
    int in_rows;
    uint8_t *out;            // [nframes][n][S]
    const uint32_t *sched_hdr;     // per frame, or nullptr for the static encode schedule
    const uint32_t *sched_steps;
    const uint16_t *sched_lvlend;
};

__global__ __launch_bounds__(512) void ldpc_apply_kernel(ApplyArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevCode &cd = a.code;
    const int n = cd.n, m = cd.m, S = a.S;
    const int64_t f = blockIdx.x;
    const int lane = lane_id(), wave = wave_id();
    const int nw = (int)(blockDim.x >> 6);

    uint32_t *steps = reinterpret_cast<uint32_t *>(smem);
    uint16_t *lvlend = reinterpret_cast<uint16_t *>(smem + (size_t)m * 4);
    int nsteps, nlev;
    if (a.sched_hdr) {
        nsteps = (int)a.sched_hdr[2 * f];
        nlev = (int)(a.sched_hdr[2 * f + 1] & 0x7FFFFFFFu);
        const uint32_t *gs = a.sched_steps + f * m;
        const uint16_t *gl = a.sched_lvlend + f * (m + 1);
        for (int i = (int)threadIdx.x; i < nsteps; i += (int)blockDim.x) steps[i] = gs[i];
        for (int i = (int)threadIdx.x; i <= nlev; i += (int)blockDim.x) lvlend[i] = gl[i];
    } else {
        nsteps = m;
        nlev = cd.enc_nlevels;
        for (int i = (int)threadIdx.x; i < nsteps; i += (int)blockDim.x) steps[i] = cd.enc_steps[i];
        for (int i = (int)threadIdx.x; i <= nlev; i += (int)blockDim.x) lvlend[i] = cd.enc_lvlend[i];
    }

    // ---- copy phase: received rows in -> out, erased rows zero-filled (unknown symbols read as 0)
    const uint8_t *fin = a.sym + f * (int64_t)a.in_rows * S;
    uint8_t *fout = a.out + f * (int64_t)n * S;
    const uint8_t *er = a.erased ? a.erased + f * n : nullptr;
    {
        const int64_t units = (int64_t)n * S / 16;
        const U4 zero = {0, 0, 0, 0};
        for (int64_t u = threadIdx.x; u < units; u += blockDim.x) {
            const int row = (int)((u * 16) / S);
            const bool e = er ? (er[row] != 0) : (row >= a.in_rows);
            U4 v = zero;
            if (!e) v = *reinterpret_cast<const U4 *>(fin + u * 16);
            *reinterpret_cast<U4 *>(fout + u * 16) = v;
        }
    }
    __syncthreads();

    // ---- steps, level by level
    const int pieces = (S + 1023) / 1024;
    for (int L = 1; L <= nlev; L++) {
        const int s0 = lvlend[L - 1], s1 = lvlend[L];
        for (int s = s0 + wave; s < s1; s += nw) {
            const uint32_t step = uniform(steps[s]);
            const uint32_t row = step & 0xFFFFu, tgt = step >> 16;
            const uint32_t e0 = cd.row_ptr[row], e1 = cd.row_ptr[row + 1];
            for (int p = 0; p < pieces; p++) {
                const int off = p * 1024 + lane * 16;
                const bool active = off < S;
                U4 acc = {0, 0, 0, 0};
                uint32_t ctgt = 1;
                for (uint32_t e = e0; e < e1; e++) {
                    const uint32_t ed = cd.edges[e];
                    const uint32_t col = ed & 0xFFFFu, c = (ed >> 16) & 0xFFu;
                    if (col == tgt) { ctgt = c; continue; }
                    if (active) {
                        const U4 v = *reinterpret_cast<const U4 *>(fout + (int64_t)col * S + off);
                        gfmac16(acc, load_multab(c), v);
                    }
                }
                if (active) {
                    const U4 r = gfmul16(load_multab(uniform(c_inv[ctgt])), acc);
                    *reinterpret_cast<U4 *>(fout + (int64_t)tgt * S + off) = r;
                }
            }
        }
        __syncthreads();
    }
}

// =================================================================================================
// Kernel B' (packets, the HBM-bound kernel): scatter form of the same arithmetic.
//
// Kernel B gathers: every step re-reads its ~12 source rows, so a row is fetched ~2.2 times (once for the
// copy, ~1.3 times as a source) and the fetches are too far apart in time to hit in L2 / Infinity Cache.
// Here every received row is read from HBM exactly ONCE: it is written to `out` and, while still in
// registers, multiplied into the accumulators of the steps it feeds (the per-frame transposed lists the peel
// kernel emits; a variant that walked H's static column lists inside this kernel instead measured 25 % slower).  The accumulators of all steps of the frame slice live in LDS (T x B bytes, B = 16*LPR
// bytes of every row per workgroup; 4 slices x 256 B for S = 1024 on the (2040,1530) code = 130 KB), updated
// with ds_xor_b64.  HBM traffic per frame: 0.9 nS read + nS write -- the algorithmic minimum for an
// out-of-place decode.  Solved symbols are finalised level by level (acc * inv(h)), written out and scattered
// on to the later steps that use them.
//
// Coefficients differ between the 64/LPR row pieces a wavefront handles at once, so the multiply tables come
// from an 8 KB LDS copy (two ds_reads per edge) instead of scalar registers.
// =================================================================================================
struct ScatterArgs {
    DevCode code;
    int S, nslices;
    int64_t nframes;
    const uint8_t *sym;
    const uint8_t *erased;
    uint8_t *out;
    const uint32_t *sched_hdr;
    const uint32_t *sched_steps;
    const uint16_t *sched_lvlend;
    const uint8_t *sched_invc;
    int in_rows;              // rows per input frame: n (decode) or k (encode: rows >= k are the unknowns)
    int static_sched;         // encode: the code's static schedule / lists are used for every frame
    int inplace;              // out == sym: received rows stay where they are, only erased rows are written
    int xcd_map;              // place the slices of a frame on one XCD
    int dyn_rows;             // streaming phase: row batches handed out through an LDS counter
    int tcap;                 // tier 1 handles frames with at most tcap steps (its LDS holds tcap accumulators)
    int nslots;               // accumulator slots in this launch's LDS (tcap in tier 1, m in tier 2 / encode)
    int32_t *big_list;        // tier 2: [0] = count, [1] = work counter, [2..] ids of the frames with more steps; nullptr in tier 1
    int lds_acc, lds_tgt, lds_invc, lds_lvlend, lds_rowctr, lds_mt, lds_soc, lds_chk;
    int lds_soc_bytes;        // size of the row-kind / row-list region
    int enc_list;             // encode: stream the source rows in DevCode::enc_order
    int enc_clist;            // encode: the level phase reads DevCode::enc_lst from LDS (copied over the dead row tables at lds_soc)
    const uint32_t *sched_lists;  // [nframes][m][cdw] the steps' column lists in schedule order (peel_relax.inc mode 2), or nullptr
    const uint32_t *sched_pull;   // [nframes][m][4] pairs: level of the step, two pulled accumulators (slot | coef << 24), spare (peel_relax.inc mode 2)
    int pairs;                // the schedules' levels come in groups of two (first / second half): one barrier per group, second halves pull
    int t2_pieces;            // tier 2: consecutive pieces of a frame per work item (the set-up is shared); must divide nslices
    int t2_force;             // ... also when the list is short (tests; by default short lists keep one piece per item)
    int xl_setup;             // level-phase lists translated (check -> accumulator address) at set-up instead of inside every level
    int enc_group;            // encode: the grouped static schedule (DevCode::encg_*: levels collapsed offline, steps pull in-group accumulators)
    int *err;                 // pinned host word (ldpc_amd_ctx::dev_err_host): a kernel whose assumptions do not hold reports here
    int dbg;                  // diagnostic build only (-DLDPC_AMD_MLDBG): 32768 = tier 1 also takes the frames with more than tcap steps,
                              // cut off at tcap steps (WRONG bytes: prices the first pass of a level-split tier 2)
};

__device__ __forceinline__ MulTab lds_multab(const uint32_t *mt, uint32_t c)
{
    const U4 q = *reinterpret_cast<const U4 *>(mt + c * 8);
    MulTab t;
    t.t0 = q.x; t.t1 = q.y; t.t2 = q.z; t.t3 = q.w; t.t4 = mt[c * 8 + 4];
    return t;
}

// acc(16 bytes) ^= v goes to LDS as two ds_xor_b64.  Interleaved layout (a lane's 16 bytes contiguous): lanes 8..15 of every
// 16-lane group issue their halves in the opposite order, so that one instruction touches all 32 LDS banks exactly once per
// 16 lanes (lds_xor16_at below).
// Split-half layout of a 16-bytes-per-lane accumulator slice (G lanes x 16 B): the G low halves first (8 B per lane), then
// the G high halves.  One ds_xor_b64 then covers 8 G contiguous bytes -- every bank once per 16 lanes -- without the per-lane
// choice of which half goes first (four v_cndmask_b32 per accumulate in the interleaved layout).  a_lo / a_hi: LDS byte addresses.
__device__ __forceinline__ void lds_xor16_split(uint32_t a_lo, uint32_t a_hi, const U4 &v)
{
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned long long lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    const unsigned long long hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
    __hip_atomic_fetch_xor(reinterpret_cast<lds_u64 *>((uintptr_t)a_lo), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_xor(reinterpret_cast<lds_u64 *>((uintptr_t)a_hi), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ U4 lds_read16_split(const unsigned char *slice, int gl, int half_bytes)
{
    const unsigned long long lo = *reinterpret_cast<const unsigned long long *>(slice + gl * 8);
    const unsigned long long hi = *reinterpret_cast<const unsigned long long *>(slice + half_bytes + gl * 8);
    return U4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
}
// multiply table of a coefficient from its LDS byte address
__device__ __forceinline__ MulTab lds_multab_at(uint32_t addr)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const u32x4 q = *reinterpret_cast<const lds_u32x4 *>((uintptr_t)addr);
    MulTab t;
    t.t0 = q.x; t.t1 = q.y; t.t2 = q.z; t.t3 = q.w; t.t4 = *reinterpret_cast<const lds_u32 *>((uintptr_t)(addr + 16));
    return t;
}

// the same with the two 8-byte targets given as LDS byte addresses (first instruction -> a1, second -> a2; a1 is the lane's
// half h): no pointer arithmetic on the way to the ds_xor_b64
__device__ __forceinline__ void lds_xor16_at(uint32_t a1, uint32_t a2, const U4 &v, int h)
{
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned long long lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    const unsigned long long hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
    __hip_atomic_fetch_xor(reinterpret_cast<lds_u64 *>((uintptr_t)a1), h ? hi : lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_xor(reinterpret_cast<lds_u64 *>((uintptr_t)a2), h ? lo : hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool NT>
__device__ __forceinline__ U4 stream_load16(const uint8_t *p)
{
    if (NT) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        U4 v;
        v.x = __builtin_nontemporal_load(q); v.y = __builtin_nontemporal_load(q + 1);
        v.z = __builtin_nontemporal_load(q + 2); v.w = __builtin_nontemporal_load(q + 3);
        return v;
    }
    return *reinterpret_cast<const U4 *>(p);
}

template <bool NT>
__device__ __forceinline__ void stream_store16(uint8_t *p, const U4 &v)
{
    if (NT) {
        uint32_t *q = reinterpret_cast<uint32_t *>(p);
        __builtin_nontemporal_store(v.x, q); __builtin_nontemporal_store(v.y, q + 1);
        __builtin_nontemporal_store(v.z, q + 2); __builtin_nontemporal_store(v.w, q + 3);
    } else {
        *reinterpret_cast<U4 *>(p) = v;
    }
}

// PERSIST (encoder only: ldpc_scatter_static_kernel): a workgroup encodes many (frame, slice) items, and everything its LDS holds that
// depends on the CODE alone -- multiply tables, the schedule's tables, the level phase's lists -- is set up by the first item (warm = false)
// and kept; a warm item zeroes its accumulators and starts streaming.
// WARM without PERSIST (decoder, tier 2: ldpc_scatter_big_kernel with several pieces per work item): the item before was another piece of
// the SAME frame -- everything the set-up builds from the frame's schedule and erasure flags (step tables, check -> slot table, row kinds,
// translated lists, pull records) is in place; only the accumulators and the row counter start over.
template <int LPR, int R, bool NT, bool INPLACE, int WPE = 4, bool PERSIST = false, bool WARM = false>
__device__ __forceinline__ void scatter_frame(const ScatterArgs &a, unsigned char *smem, const int64_t f, const int sl)
{
    constexpr bool warm = WARM;
    // (the persistent form is the encoder's: the decoder's paths fold away in its instantiations)
    const bool is_static = PERSIST ? true : (a.static_sched != 0);
    constexpr int RPW = 64 / LPR;              // row pieces per wave instruction
    constexpr int KQ = (16 + LPR - 1) / LPR;   // edge words held per lane (maxcoldeg <= 16)
    constexpr int B = 16 * LPR;                // bytes of every row handled by this workgroup (R pieces in flight per lane)
    const DevCode &cd = a.code;
    const int n = cd.n, S = a.S, cdw = cd.maxcoldeg;
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int lane = lane_id(), wave = wave_id(), nw = nthr >> 6;
    const int g = lane / LPR, gl = lane % LPR, h = (lane >> 3) & 1;
    const int gbase = lane & ~(LPR - 1);

    // LDS map of the packet kernels (scatter_set_lds): the 8 KB of multiply tables FIRST, the accumulators behind them, the small
    // tables behind those.  With the tables at LDS address 0 an edge word (coef << 24 | LDS address of the accumulator slice) gives
    // the table address with ONE shift (coef * 32 = word >> 19: LDS addresses stay below 2^18) and the two accumulate addresses
    // with one and-or each.
    constexpr int kMtOff = 0, kAccOff = 8192;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    // The dynamic LDS of these kernels starts at LDS address 0: they must NOT declare static __shared__ variables (edge words
    // carry absolute LDS addresses).  Checked, not just assumed, and REPORTED: the frame is left undecoded and the context's
    // device-error word is set, which the next synchronising call returns as LDPC_AMD_EHIP (a bare return here
    // would hand back LDPC_AMD_OK with wrong bytes).
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8 *)(smem);
    if (lds0 != 0u) {
        if (tid == 0 && a.err) atomicOr(a.err, kDevErrLdsBase);
        return;
    }
    __builtin_assume(lds0 == 0u);
    const uint32_t accbase = (uint32_t)kAccOff;
    unsigned char *acc = smem + kAccOff;
    uint16_t *tgt = reinterpret_cast<uint16_t *>(smem + a.lds_tgt);
    uint8_t *invc = smem + a.lds_invc;
    uint16_t *lvlend = reinterpret_cast<uint16_t *>(smem + a.lds_lvlend);
    uint32_t *mt = reinterpret_cast<uint32_t *>(smem + kMtOff);
    uint8_t *rk = smem + a.lds_soc;  // row kinds, [n]
    uint16_t *soc = reinterpret_cast<uint16_t *>(smem + a.lds_chk);  // check -> slot of the step that uses it, 0xFFFF

    LDPC_STAMP_INIT;
#ifdef LDPC_AMD_MLDBG
    const int nsteps = is_static ? cd.m : ((a.dbg & 32768) ? min((int)a.sched_hdr[2 * f], a.nslots) : (int)a.sched_hdr[2 * f]);
#else
    const int nsteps = is_static ? cd.m : (int)a.sched_hdr[2 * f];
#endif
    const bool grouped = is_static && a.enc_group;
    const uint32_t hdr1 = is_static ? 0u : a.sched_hdr[2 * f + 1];   // levels | bit 31: they come in pairs (peel_relax.inc mode 2)
    const int nlev = is_static ? (grouped ? cd.encg_nlevels : cd.enc_nlevels) : (int)(hdr1 & 0x7FFFFFFFu);
    const uint32_t *gs = is_static ? (grouped ? cd.encg_steps : cd.enc_steps) : a.sched_steps + f * cd.m;
    const uint16_t *gle = is_static ? (grouped ? cd.encg_lvlend : cd.enc_lvlend) : a.sched_lvlend + f * (cd.m + 1);
    const uint8_t *gic = is_static ? (grouped ? cd.encg_invc : cd.enc_invc) : a.sched_invc + f * cd.m;
    const uint8_t *erf = a.erased ? a.erased + f * (int64_t)n : nullptr;
    // Set-up costs one global-memory latency: everything that comes from global memory (this thread's steps, erasure
    // flags, level offsets, multiply tables) is requested first, the LDS is initialised while the loads are in flight.
    constexpr int SPT = 2, EPT = 4;             // steps / symbols per thread held in registers (m <= 2048, n <= 4096)
    uint32_t stp[SPT];
    uint32_t siv[SPT], erv[EPT];
#pragma unroll
    for (int u = 0; u < SPT; u++) {
        const int s = tid + u * nthr;
        stp[u] = (s < nsteps && !warm) ? gs[s] : 0u;
        siv[u] = (s < nsteps && !warm) ? (uint32_t)gic[s] : 0u;
    }
#pragma unroll
    for (int u = 0; u < EPT; u++) {
        const int j = tid + u * nthr;
        erv[u] = (j < n && !warm) ? (erf ? (uint32_t)erf[j] : (j >= a.in_rows ? 1u : 0u)) : 0u;
    }
    // The column lists of the symbols solved in phase B go to the LDS left over behind the accumulators of this
    // frame (nsteps of nslots used), so that phase B issues no global load: a load behind the phase's row stores
    // would wait for those stores (one memory counter), once per level.  Frames without room keep the global lists.
    // Paired levels (a.pairs: the schedule comes from peel_relax.inc mode 2): the steps of a group's second half pull the raw accumulators
    // of their first-half inputs; their pull entries (two words) and every step's level (u16) go behind the lists when there is room,
    // else the frame runs its levels one by one (they are a valid levelling on their own) without pulls and exclusions.
    const bool pairs_f = !is_static && a.pairs && (hdr1 >> 31) && a.xl_setup && (int64_t)nsteps * (B + 4 * cdw + 10) + 16 <= (int64_t)a.nslots * B;
    const bool lds_lists = !is_static && (pairs_f || (int64_t)nsteps * (B + 4 * cdw) <= (int64_t)a.nslots * B);
    uint32_t *slist = reinterpret_cast<uint32_t *>(acc + (size_t)nsteps * B);   // [nsteps][cdw]
    uint32_t *plist = slist + (size_t)nsteps * cdw;                              // [nsteps][2] pull entries (pairs_f)
    uint16_t *slev = reinterpret_cast<uint16_t *>(plist + (size_t)nsteps * 2);   // [nsteps] level of the step (pairs_f)
    const uint32_t *gpl = (pairs_f ? a.sched_pull : nullptr);
    uint32_t prc[SPT][3];   // this thread's step records (level, two pulls), requested with the other set-up loads
#pragma unroll
    for (int u = 0; u < SPT; u++) {
        const int s = tid + u * nthr;
        prc[u][0] = prc[u][1] = prc[u][2] = 0xFFFFFFFFu;
        if (pairs_f && s < nsteps && !warm) {
            const uint32_t *r_ = gpl + ((size_t)f * cd.m + s) * 4;
            prc[u][0] = r_[0]; prc[u][1] = r_[1]; prc[u][2] = r_[2];
        }
    }
    constexpr int LPT = 2;                      // list words per thread held in registers
    uint32_t lw[LPT];
    if (lds_lists && !warm) {
#pragma unroll
        for (int u = 0; u < LPT; u++) {
            const int e = tid + u * nthr;
            lw[u] = 0xFFFFFFFFu;
            if (e < nsteps * cdw) {
                if (a.sched_lists) {   // (the peel kernel laid the lists out in schedule order: no dependent step -> symbol -> list load)
                    lw[u] = a.sched_lists[(size_t)f * cd.m * cdw + e];
                } else {
                    const int s = e / cdw, idx = e - s * cdw;
                    lw[u] = cd.cell[((int64_t)(gs[s] >> 16) << cd.cdw_shift) + idx];
                }
            }
        }
    }
    if (!warm) {
    for (int i = tid; i < 2048; i += nthr) mt[i] = c_mul3[i];
#ifdef LDPC_AMD_MLDBG
    for (int i = tid; i <= nlev; i += nthr) lvlend[i] = (uint16_t)min((int)gle[i], nsteps);
#else
    for (int i = tid; i <= nlev; i += nthr) lvlend[i] = gle[i];
#endif
    }
    if (!is_static && !warm)   // (the encoder's lists are in slot form already: it has no check -> slot table)
        for (int i = tid; i < (cd.m + 1) / 2; i += nthr) reinterpret_cast<uint32_t *>(soc)[i] = 0xFFFFFFFFu;
    if (tid < (warm ? 1 : 2)) reinterpret_cast<int *>(smem + a.lds_rowctr)[tid] = 0;   // [0] row-batch counter of the streaming phase, [1] received rows
    for (int i = tid; i < nsteps * LPR; i += nthr) reinterpret_cast<U4 *>(acc)[i] = U4{0, 0, 0, 0};
    // row kinds: 1 received, 2 erased and never solved (written as 0), 0 erased and solved in phase B (set below)
    // (PERSIST: the encoder's source rows are all received -- its stream does not consult the table, whose place the lists of the level phase keep)
    if (!PERSIST && !warm) {
#pragma unroll
    for (int u = 0; u < EPT; u++) {
        const int j = tid + u * nthr;
        if (j < n) rk[j] = erv[u] ? (uint8_t)2 : (uint8_t)1;
    }
    for (int j = tid + EPT * nthr; j < n; j += nthr)
        rk[j] = (erf ? (erf[j] != 0) : (j >= a.in_rows)) ? (uint8_t)2 : (uint8_t)1;
    }
    __syncthreads();
    auto put_step = [&](int s, uint32_t step, uint32_t iv) {
        const uint32_t row = step & 0xFFFFu, t = step >> 16;
        tgt[s] = (uint16_t)t;
        if (!is_static) soc[row] = (uint16_t)s;
        invc[s] = (uint8_t)iv;
        if (!PERSIST) rk[t] = 0;
    };
    auto put_pull = [&](int s, uint32_t lv, uint32_t p0, uint32_t p1) {
        // a pull entry leaves as (LDS address of the pulled accumulator slice | coef << 24), like the scatter entries
        slev[s] = (uint16_t)lv;
        plist[2 * s] = p0 == 0xFFFFFFFFu ? p0 : (((uint32_t)kAccOff + (p0 & 0x00FFFFFFu) * (uint32_t)B) | (p0 & 0xFF000000u));
        plist[2 * s + 1] = p1 == 0xFFFFFFFFu ? p1 : (((uint32_t)kAccOff + (p1 & 0x00FFFFFFu) * (uint32_t)B) | (p1 & 0xFF000000u));
    };
    if (!warm) {
#pragma unroll
    for (int u = 0; u < SPT; u++) {
        const int s = tid + u * nthr;
        if (s < nsteps) put_step(s, stp[u], siv[u]);
        if (pairs_f && s < nsteps) put_pull(s, prc[u][0], prc[u][1], prc[u][2]);
    }
    for (int s = tid + SPT * nthr; s < nsteps; s += nthr) {
        put_step(s, gs[s], gic[s]);
        if (pairs_f) { const uint32_t *r_ = gpl + ((size_t)f * cd.m + s) * 4; put_pull(s, r_[0], r_[1], r_[2]); }
    }
    }
    if (lds_lists && !warm) {
        // The lists go to LDS TRANSLATED (check -> LDS address of the step's accumulator slice | coef << 24, the form the turns of
        // `scatter` consume; the list's own step left out): the two dependent look-ups of the translation then happen once, here,
        // for all steps at a time, instead of inside every level's chain of LDS round trips (round 4).
        if (a.xl_setup) __syncthreads();   // the check -> slot table is complete
        auto xl = [&](uint32_t w, uint32_t own) -> uint32_t {
            if (!a.xl_setup) return w;
            const uint32_t s_ = soc[w == 0xFFFFFFFFu ? 0u : (w & 0xFFFFu)];
            bool keep = w != 0xFFFFFFFFu && s_ != 0xFFFFu && s_ != own;
            if (pairs_f) {   // a first half (odd level) does not scatter into the second half of its own group: those steps pull instead
                const uint32_t lo = slev[own];
                const uint32_t x0 = lvlend[lo], x1 = lvlend[min((int)lo + 1, nlev)];
                keep = keep && !((lo & 1u) && s_ >= x0 && s_ < x1);
            }
            return keep ? (((uint32_t)kAccOff + s_ * (uint32_t)B) | ((w & 0x00FF0000u) << 8)) : 0xFFFFFFFFu;
        };
#pragma unroll
        for (int u = 0; u < LPT; u++) {
            const int e = tid + u * nthr;
            if (e < nsteps * cdw) slist[e] = xl(lw[u], (uint32_t)(e / cdw));
        }
        for (int e = tid + LPT * nthr; e < nsteps * cdw; e += nthr) {
            const int s = e / cdw, idx = e - s * cdw;
            slist[e] = xl(a.sched_lists ? a.sched_lists[(size_t)f * cd.m * cdw + e] : cd.cell[((int64_t)(gs[s] >> 16) << cd.cdw_shift) + idx], (uint32_t)s);
        }
    }
    __syncthreads();

    // List mode (a.dyn_rows == 2): the received rows are compacted into a list, so that the streaming loop below has
    // no conditional memory operation -- every lane group loads (and stores) a row in every pass -- and the compiler
    // can count the outstanding operations instead of waiting for all of them.  The list overwrites the row kinds,
    // which are pulled into registers first.  Rows that were erased and are never solved are zeroed here.
    uint16_t *rlist = reinterpret_cast<uint16_t *>(rk);
    int nrecv = 0;
    // Sorted list mode (a.dyn_rows == 3): the list is additionally ordered by the number of accumulators a row feeds
    // (its column-list entries whose check is used by a step of this frame), most first.  The four row pieces a wavefront
    // handles at once then have the same number of edges, so no lane group idles while another finishes its row (with
    // rows in index order a pass takes max-over-four turns: 2.4 for 1.3 edges per row at 10 % erasures), and the rows
    // that feed nothing come last and skip the multiply set-up altogether.
    const bool sorted_mode = a.dyn_rows == 3 && n <= EPT * nthr && !is_static && cdw <= 16;
    // Windowed sorted list (a.dyn_rows == 4): the same ordering INSIDE windows of 64 consecutive rows (the rows one wavefront
    // of the set-up holds), windows in index order: the pieces of a pass still take about the same number of turns, and a pass
    // stays inside a 64 KB stretch of the frame instead of hopping all over it (what the global order lost to).
    const bool win_mode = a.dyn_rows == 4 && n <= EPT * nthr && !is_static && cdw <= 16;
    // Encoder (static schedule): the list is the code's source symbols in the order of their column degree, prepared by the
    // host (DevCode::enc_order) -- same effect as the sorted mode, no per-frame work.  Measured slower than index order (the
    // list look-ups and the lost DRAM locality cost more than the balanced turns save): only with LDPC_AMD_ENC_LIST=1.
    const bool static_list = !PERSIST && is_static && a.enc_list && a.lds_soc_bytes >= 2 * a.in_rows;
    const bool list_mode = ((a.dyn_rows == 2 || sorted_mode || win_mode) && n <= EPT * nthr && !is_static) || static_list;
    if (static_list) {
        for (int i = tid; i < a.in_rows; i += nthr) rlist[i] = cd.enc_order[i];
        __syncthreads();
        nrecv = a.in_rows;
    } else if (list_mode) {
        int *nrecv_p = reinterpret_cast<int *>(smem + a.lds_rowctr) + 1;
        int *bins = reinterpret_cast<int *>(smem + a.lds_rowctr) + 4;   // [0..16] rows per edge count, [17..33] fill pointers
        uint32_t kd[EPT], ec[EPT];
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int j = tid + u * nthr;
            kd[u] = (j < n) ? (uint32_t)rk[j] : 0u;
            ec[u] = 0;
        }
        if (sorted_mode || win_mode) {
            if (tid < 34 && sorted_mode) bins[tid] = 0;
#pragma unroll
            for (int u = 0; u < EPT; u++) {
                const int j = tid + u * nthr;
                if (kd[u] == 1u) {
                    uint32_t c = 0;
                    for (int idx = 0; idx < cdw; idx++) {
                        const uint32_t w = cd.cell[((uint32_t)j << cd.cdw_shift) + (uint32_t)idx];
                        c += (w != 0xFFFFFFFFu && soc[w & 0xFFFFu] != 0xFFFFu) ? 1u : 0u;
                    }
                    ec[u] = c;
                }
            }
        }
        if (win_mode) {   // received rows per window (window u * nw + wave = the rows this wavefront holds in slot u)
#pragma unroll
            for (int u = 0; u < EPT; u++) {
                const int c = __popcll(__ballot(kd[u] == 1u));
                if (lane == 0) bins[u * nw + wave] = c;
            }
        }
        __syncthreads();
        uint8_t *fz = a.out + f * (int64_t)n * S + (int64_t)sl * B;
        if (sorted_mode) {
#pragma unroll
            for (int u = 0; u < EPT; u++)
                for (int c = 0; c <= cdw; c++) {
                    const uint64_t mask = __ballot(kd[u] == 1u && ec[u] == (uint32_t)c);
                    if (lane == 0 && mask) atomicAdd(&bins[c], __popcll(mask));
                }
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int c = cdw; c >= 0; c--) { bins[17 + c] = run; run += bins[c]; }
                *nrecv_p = run;
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int j = tid + u * nthr;
            const bool recv = kd[u] == 1u;
            if (win_mode) {
                // start of this window in the list = received rows of the windows before it (at most 64 windows: one per lane)
                const int w = u * nw + wave;
                int base = (lane < w) ? bins[lane] : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) base += __shfl_xor(base, d);
                for (int c = cdw; c >= 0; c--) {   // most edges first inside the window
                    const bool mine = recv && ec[u] == (uint32_t)c;
                    const uint64_t mask = __ballot(mine);
                    if (mine) rlist[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)j;
                    base += __popcll(mask);
                }
                if (u == EPT - 1 && wave == nw - 1 && lane == 0) *nrecv_p = base;   // the last window ends the list
            } else if (sorted_mode) {
                for (int c = 0; c <= cdw; c++) {
                    const bool mine = recv && ec[u] == (uint32_t)c;
                    const uint64_t mask = __ballot(mine);
                    int base = 0;
                    if (lane == 0 && mask) base = atomicAdd(&bins[17 + c], __popcll(mask));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (mine) rlist[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)j;
                }
            } else {
                const uint64_t mask = __ballot(recv);
                int base = 0;
                if (lane == 0 && mask) base = atomicAdd(nrecv_p, __popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (recv) rlist[base + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)j;
            }
            if (kd[u] == 2u)
                for (int q = 0; q < LPR; q++) stream_store16<NT>(fz + (int64_t)j * S + q * 16, U4{0, 0, 0, 0});
        }
        __syncthreads();
        nrecv = *nrecv_p;
    }
    LDPC_STAMP(12);  // scatter: set-up
    // Uniform 64-bit bases + 32-bit per-lane offsets (a frame slice spans n * S < 4 GB): the row addresses then cost one
    // 32-bit multiply-add instead of 64-bit multiplies -- the streaming loop is bound by vector ALU issue, not by waits.
    const uint8_t *fin0 = a.sym + f * (int64_t)a.in_rows * S + (int64_t)sl * B;
    uint8_t *fout0 = a.out + f * (int64_t)n * S + (int64_t)sl * B;
    const uint32_t lo16 = (uint32_t)gl * 16u, S32 = (uint32_t)S;
    // (24-bit multiply: full rate, where the 32-bit one the compiler picked -- v_mad_u64_u32 -- runs at a quarter; j < 2^16 and
    // S < 2^24 are checked by the host's plan)
    auto in_row = [&](int j) { return fin0 + (__umul24((uint32_t)j, S32) + lo16); };
    auto out_row = [&](int j) { return fout0 + (__umul24((uint32_t)j, S32) + lo16); };
    // H's static column lists (check | coef << 16); the encoder's are already in (slot | coef << 16) form
    const uint32_t *spad = is_static ? (grouped ? cd.encg_src : cd.enc_src) : cd.cell;
    const bool translate = !is_static;

    // multiplies v into the accumulators of the steps that symbol j feeds: ew = the symbol's list, entry t held by
    // lane (t % LPR) of the group, 0xFFFFFFFF = no entry.  Every group walks the set bits of its own validity mask.
    // (entries are in the form to_slots leaves them in: LDS address of the accumulator slice | coef << 24; a lane's own two
    // 8-byte halves -- split-half layout -- are OR-ed in: the slice address is a multiple of B, the lane's part is below B)
    // LPR == 16 (256-byte pieces): split-half layout, no per-lane choice.  Narrower pieces put two or more lane groups -- different
    // accumulators -- into every 16 lanes, and their 8 LPR contiguous bytes would meet on the same banks: those keep the interleaved
    // layout in which lanes 8-15 of every 16 issue their halves in the opposite order (measured: encoder 4.08 -> 4.24 ms with the
    // split layout at LPR = 8).
    constexpr bool kSplit = LPR == 16;
    const uint32_t lane_a = kSplit ? (uint32_t)(gl * 8) : (uint32_t)(gl * 16 + h * 8);
    const uint32_t lane_b = kSplit ? (uint32_t)(B / 2 + gl * 8) : (uint32_t)(gl * 16 + (1 - h) * 8);
    auto scatter = [&](const U4 &v, const uint32_t (&ew)[KQ]) {
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            if (q * LPR >= cdw) break;
            uint32_t gm = (uint32_t)(__ballot(ew[q] != 0xFFFFFFFFu) >> gbase) & (uint32_t)((1ull << LPR) - 1ull);
            while (__any(gm != 0)) {
                const bool go = gm != 0;
                const int u = gm ? __builtin_ctz(gm) : 0;   // (gm == 0: lane 0 of the group -- such lanes drop what they read; no poison from cttz(0))
                gm &= gm - 1u;
                const uint32_t ed = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + u) << 2, (int)ew[q]);
                if (go) {
                    const U4 prod = gfmul16(lds_multab_at(ed >> 19), v);   // coef * 32: the tables start the LDS
                    // (two different masks -- both clear the coefficient byte, addresses are below 2^18 -- so that each address is
                    // one v_and_or_b32 instead of a shared v_and_b32 plus two v_or_b32)
                    if (kSplit) lds_xor16_split((ed & 0x00FFFFFFu) | lane_a, (ed & 0x007FFFFFu) | lane_b, prod);
                    else lds_xor16_at((ed & 0x00FFFFFFu) | lane_a, (ed & 0x007FFFFFu) | lane_b, prod, h);
                }
            }
        }
    };

    // A symbol's column-list word (check | coef << 16) becomes (slot | coef << 16) if the check is used by a step of
    // this frame -- other than the step that solves the symbol itself (own) -- and 0xFFFFFFFF otherwise: one LDS
    // look-up per lane, no per-frame lists in HBM.
    // The entry leaves as (byte offset of the step's accumulator slice | coef << 24), what the turns of `scatter` consume.
    auto to_slots = [&](uint32_t (&ew)[KQ], uint32_t own) {
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            const uint32_t w = ew[q];
            if (!translate) {   // encoder: the static lists hold (slot * 128 | coef << 24): the form of its default 128-byte pieces
                const uint32_t so = (B >= 128) ? (w & 0x00FFFFFFu) * (uint32_t)(B / 128 > 0 ? B / 128 : 1) : (w & 0x00FFFFFFu) / (uint32_t)(B < 128 ? 128 / B : 1);
                ew[q] = (w != 0xFFFFFFFFu) ? ((accbase + so) | (w & 0xFF000000u)) : 0xFFFFFFFFu;
                continue;
            }
            const uint32_t s = soc[w == 0xFFFFFFFFu ? 0u : (w & 0xFFFFu)];
            ew[q] = (w != 0xFFFFFFFFu && s != 0xFFFFu && s != own) ? ((accbase + s * (uint32_t)B) | ((w & 0x00FF0000u) << 8)) : 0xFFFFFFFFu;
        }
    };

    // ---- phase A: stream the received rows once.  Software pipelined: batch i+1's loads are issued before batch
    //      i's stores (vmcnt retires in order and counts stores -- loads issued behind stores would wait for them),
    //      and the row kinds come from LDS so that no global load sits between a wave and its row loads.
    struct RowBatch {
        U4 v[R];
        uint32_t ew[R][KQ];
        int kind[R];  // 0 skip (erased and solved later, or past the end), 1 received row, 2 erased and never solved
        int row[R];   // list mode: the row a lane group holds
    };
    auto fetch_list = [&](int i0, RowBatch &b) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int idx = i0 + r * RPW + g;
            const bool valid = idx < nrecv;
            const int j = (int)rlist[valid ? idx : nrecv - 1];   // past the end: the last row again (same bytes, same place)
            b.kind[r] = valid ? 1 : 0;
            b.row[r] = j;
            b.v[r] = stream_load16<NT>(in_row(j));
#pragma unroll
            for (int q = 0; q < KQ; q++) {
                const int e = gl + q * LPR;
                const uint32_t w = spad[((uint32_t)j << cd.cdw_shift) + (uint32_t)(e < cdw ? e : cdw - 1)];
                b.ew[r][q] = (valid && e < cdw) ? w : 0xFFFFFFFFu;
            }
        }
    };
    // rows the streaming phase walks: all n for the decoder; the encoder's parity rows (j >= k) are all produced by the level
    // phase, so its stream ends at k (a quarter of the (2040,1530) rows)
    const int nstream = is_static ? a.in_rows : n;
    auto fetch = [&](int j0, RowBatch &b) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = j0 + r * RPW + g;
            const int kd = (j < nstream) ? (PERSIST ? 1 : (int)rk[j]) : 0;
            b.kind[r] = kd;
            b.v[r] = U4{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < KQ; q++) b.ew[r][q] = 0xFFFFFFFFu;
            if (kd == 1) {
                b.v[r] = stream_load16<NT>(in_row(j));
#pragma unroll
                for (int q = 0; q < KQ; q++) {
                    const int idx = gl + q * LPR;
                    if (idx < cdw) b.ew[r][q] = spad[((uint32_t)j << cd.cdw_shift) + (uint32_t)idx];
                }
            }
        }
    };
    {
        // Row batches are handed out through an LDS counter (first come, first served) instead of a fixed stride: the
        // waves of a workgroup do different amounts of accumulator work per row, and the workgroup cannot enter the
        // level phase before its slowest wave is through.  LDPC_AMD_SCATTER_DYN=0 restores the fixed stride.
        int *rowctr = reinterpret_cast<int *>(smem + a.lds_rowctr);
        const int step = R * RPW;
        RowBatch cur, nxt;
        if (list_mode) {
            int i0 = nrecv, i1 = 0;
            if (nrecv > 0) {
                if (lane == 0) i0 = atomicAdd(rowctr, step);
                i0 = __builtin_amdgcn_readfirstlane(i0);
            }
            if (i0 < nrecv) {
                auto process = [&](const RowBatch &b) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!INPLACE) stream_store16<NT>(out_row(b.row[r]), b.v[r]);
                        uint32_t ew[KQ];
#pragma unroll
                        for (int q = 0; q < KQ; q++) ew[q] = b.ew[r][q];
                        to_slots(ew, 0xFFFFu);
                        scatter(b.v[r], ew);
                    }
                };
                fetch_list(i0, cur);
                for (;;) {   // the body has no conditional memory operation: next batch's loads, this batch's stores
                    if (lane == 0) i1 = atomicAdd(rowctr, step);
                    i1 = __builtin_amdgcn_readfirstlane(i1);
                    if (i1 >= nrecv) break;
                    fetch_list(i1, nxt);
                    process(cur);
                    cur = nxt;
                }
                process(cur);
            }
        } else if (a.dyn_rows) {
            int j0 = 0, j1 = 0;
            if (lane == 0) j0 = atomicAdd(rowctr, step);
            j0 = __builtin_amdgcn_readfirstlane(j0);
            fetch(j0, cur);
            while (j0 < nstream) {
                if (lane == 0) j1 = atomicAdd(rowctr, step);
                j1 = __builtin_amdgcn_readfirstlane(j1);
                fetch(j1, nxt);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int j = j0 + r * RPW + g;
                    if (cur.kind[r] == 2 || (cur.kind[r] == 1 && !INPLACE)) stream_store16<NT>(out_row(j), cur.v[r]);
                    to_slots(cur.ew[r], 0xFFFFu);
                    scatter(cur.v[r], cur.ew[r]);
                }
                cur = nxt;
                j0 = j1;
            }
        } else {
            const int stride = nw * step;
            int j0 = wave * step;
            fetch(j0, cur);
            for (; j0 < nstream; j0 += stride) {
                fetch(j0 + stride, nxt);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int j = j0 + r * RPW + g;
                    if (cur.kind[r] == 2 || (cur.kind[r] == 1 && !INPLACE)) stream_store16<NT>(out_row(j), cur.v[r]);
                    to_slots(cur.ew[r], 0xFFFFu);
                    scatter(cur.v[r], cur.ew[r]);
                }
                cur = nxt;
            }
        }
    }
    LDPC_STAMP(13);  // scatter: streaming phase
    __syncthreads();
    LDPC_STAMP(14);  // scatter: wait for the slowest wave of the stream

    // ---- phase B: finalise the solved symbols level by level and scatter them on.  Two instantiations: with the
    //      lists in LDS the loop contains no global load, so nothing in it waits on the memory counter (a load would
    //      wait for the row stores issued before it, once per level); the other one reads the lists from global memory
    //      and requests a wave's first list of the next level before the barrier.
    // (Encoder: the lists of the parity symbols, compact -- DevCode::enc_lst -- are copied over the row tables the streaming phase
    // is done with; round 3: its level phase was 59 % of a workgroup's time with one global list load per level, tools/stamp_encode.py.)
    const uint32_t *clist = reinterpret_cast<const uint32_t *>(smem + a.lds_soc);
    const uint16_t *coff = reinterpret_cast<const uint16_t *>(smem + a.lds_soc + 4 * cd.enc_lst_n);
    // Grouped schedule: per step its pull entries, then its scatter entries, 3 bytes each in LDS (slot u16 | coef u8: the 4-byte
    // words would not fit beside all m accumulators in half a CU's LDS), one offset array and the pull counts.
    const int gent = cd.encg_ent_n, gent2 = (gent + 7) & ~7;
    const uint16_t *g_slot = reinterpret_cast<const uint16_t *>(smem + a.lds_soc);
    const uint8_t *g_coef = smem + a.lds_soc + 2 * gent2;
    const uint16_t *g_off = reinterpret_cast<const uint16_t *>(smem + a.lds_soc + 3 * gent2);
    const uint8_t *g_np = smem + a.lds_soc + 3 * gent2 + 2 * ((cd.m + 8) & ~7);
    if (is_static && warm) {
        // (the lists are in place since the workgroup's first item)
    } else if (grouped) {
        uint16_t *ws = reinterpret_cast<uint16_t *>(smem + a.lds_soc);
        uint8_t *wc = smem + a.lds_soc + 2 * gent2;
        uint16_t *wo = reinterpret_cast<uint16_t *>(smem + a.lds_soc + 3 * gent2);
        uint8_t *wn = smem + a.lds_soc + 3 * gent2 + 2 * ((cd.m + 8) & ~7);
        for (int i = tid; i < gent; i += nthr) {
            const uint32_t w = cd.encg_ent[i];
            ws[i] = (uint16_t)((w & 0x00FFFFFFu) >> 7);   // slot (the words hold slot * 128)
            wc[i] = (uint8_t)(w >> 24);
        }
        for (int i = tid; i <= nsteps; i += nthr) wo[i] = cd.encg_ent_off[i];
        for (int i = tid; i < nsteps; i += nthr) wn[i] = cd.encg_npull[i];
        __syncthreads();
    } else if (is_static && a.enc_clist) {
        uint32_t *cw_ = reinterpret_cast<uint32_t *>(smem + a.lds_soc);
        uint16_t *co_ = reinterpret_cast<uint16_t *>(smem + a.lds_soc + 4 * cd.enc_lst_n);
        for (int i = tid; i < cd.enc_lst_n; i += nthr) cw_[i] = cd.enc_lst[i];
        for (int i = tid; i <= nsteps; i += nthr) co_[i] = cd.enc_lst_off[i];
        __syncthreads();
    }
    auto phase_b = [&](auto lds_tag) {
        constexpr int MODE = decltype(lds_tag)::value;   // 0: lists from global memory, 1: padded lists in LDS, 2: compact lists in LDS
        constexpr bool LL = MODE != 0;
        auto load_list = [&](int s, int s1, uint32_t (&ew)[KQ]) {
#pragma unroll
            for (int q = 0; q < KQ; q++) ew[q] = 0xFFFFFFFFu;
            if (s < s1) {
                if (MODE == 2) {
                    const uint32_t o0 = coff[s], o1 = coff[s + 1];
#pragma unroll
                    for (int q = 0; q < KQ; q++) {
                        const uint32_t idx = o0 + (uint32_t)(gl + q * LPR);
                        if (idx < o1) ew[q] = clist[idx];
                    }
                    return;
                }
                const int t = tgt[s];
#pragma unroll
                for (int q = 0; q < KQ; q++) {
                    const int idx = gl + q * LPR;
                    if (idx < cdw) ew[q] = MODE == 1 ? slist[s * cdw + idx] : spad[((uint32_t)t << cd.cdw_shift) + (uint32_t)idx];
                }
            }
        };
        uint32_t ewn[KQ];
        if (!LL) load_list((nlev >= 1 ? (int)lvlend[0] : 0) + wave * RPW + g, nlev >= 1 ? (int)lvlend[1] : 0, ewn);
        const bool paired = MODE == 1 && pairs_f;   // groups of two levels behind one barrier; the second half pulls (see pairs_f)
#ifdef LDPC_AMD_MLDBG
        // diagnostic (WRONG bytes, timing only; tools/bound_cfg3.py): bit 64 = every step of the frame as ONE level, bit 128 = the
        // levels taken in pairs: what collapsing the decoder's levels the way the encoder's are collapsed could gain at most
        const int lstep = (a.dbg & 64) ? (nlev > 0 ? nlev : 1) : (((a.dbg & 128) || paired) ? 2 : 1);
#else
        const int lstep = paired ? 2 : 1;
#endif
        for (int L = 1; L <= nlev; L += lstep) {
            const int s0 = lvlend[L - 1], s1 = lvlend[min(L + lstep - 1, nlev)];
            const int smid = lvlend[L];   // (paired: the group's second half starts here)
            for (int sb = s0 + wave * RPW; sb < s1; sb += nw * RPW) {
                const int s = sb + g;
                U4 val = {0, 0, 0, 0};
                uint32_t ew[KQ];
                if (!LL && sb == s0 + wave * RPW) {
#pragma unroll
                    for (int q = 0; q < KQ; q++) ew[q] = ewn[q];
                } else {
                    load_list(s, s1, ew);
                }
                if (s < s1) {
                    const int t = tgt[s];
                    U4 a16 = kSplit ? lds_read16_split(acc + (size_t)s * B, gl, B / 2)
                                    : *reinterpret_cast<const U4 *>(acc + (size_t)s * B + gl * 16);
                    if (paired && s >= smid) {
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const uint32_t pe = plist[2 * s + q];
                            if (pe != 0xFFFFFFFFu) {
                                const unsigned char *src_ = smem + (pe & 0x00FFFFFFu);
                                const U4 sv_ = kSplit ? lds_read16_split(src_, gl, B / 2) : *reinterpret_cast<const U4 *>(src_ + gl * 16);
                                gfmac16(a16, lds_multab_at(pe >> 19), sv_);
                            }
                        }
                    }
                    val = gfmul16(lds_multab(mt, invc[s]), a16);
                    stream_store16<NT>(out_row(t), val);
                }
                if (MODE != 1 || !a.xl_setup) to_slots(ew, (s < s1) ? (uint32_t)s : 0xFFFFu);   // (MODE 1: translated at set-up)
                scatter(val, ew);
            }
            if (!LL && L < nlev) load_list((int)lvlend[L] + wave * RPW + g, (int)lvlend[L + 1], ewn);
            __syncthreads();
        }
    };
    // Grouped static schedule (encoder; DevCode::encg_*): a group of collapsed levels per barrier.  A step adds the raw accumulators
    // of its in-group ancestors (composite coefficients) to its own before the division -- no step of a group waits for another.
    // (U steps per lane group and trip: two steps carried through the chain of LDS round trips at once measured SLOWER on the
    // (4080,3060) encoder -- 5.44 against 5.14 ms per 2048 frames, same box -- so one.)
    auto phase_g = [&]() {
        constexpr int U = 1;
        for (int L = 1; L <= nlev; L++) {
            const int s0 = lvlend[L - 1], s1 = lvlend[L];
            for (int sb = s0 + wave * RPW; sb < s1; sb += U * nw * RPW) {
                bool on[U];
                uint32_t su[U], o0[U], o1[U], np[U], npmax = 0;
                U4 a16[U], val[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    su[u] = (uint32_t)(sb + u * nw * RPW + g);
                    on[u] = (int)su[u] < s1;
                    o0[u] = on[u] ? g_off[su[u]] : 0u; o1[u] = on[u] ? g_off[su[u] + 1] : 0u; np[u] = on[u] ? g_np[su[u]] : 0u;
                    a16[u] = U4{0, 0, 0, 0}; val[u] = U4{0, 0, 0, 0};
                    if (on[u]) a16[u] = kSplit ? lds_read16_split(acc + (size_t)su[u] * B, gl, B / 2)
                                               : *reinterpret_cast<const U4 *>(acc + (size_t)su[u] * B + gl * 16);
                    npmax = max(npmax, np[u]);
                }
                // The pulls PQ at a time: their list entries first, then the pulled accumulators and the coefficient tables, then the
                // multiply-accumulates -- three LDS round trips per PQ pulls instead of three per pull (an entry past the end of the
                // step's list pulls the step's own accumulator with coefficient 0: no branch in the batch).
                constexpr int PQ = WPE >= 8 ? 2 : 4;
                for (uint32_t i = 0; __any(i < npmax); i += PQ) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        uint32_t sl_[PQ], cf[PQ];
#pragma unroll
                        for (int q = 0; q < PQ; q++) {
                            const bool v = i + (uint32_t)q < np[u];
                            const uint32_t e = o0[u] + (v ? i + (uint32_t)q : 0u);
                            sl_[q] = g_slot[e]; cf[q] = g_coef[e];
                            if (!v) { sl_[q] = on[u] ? su[u] : 0u; cf[q] = 0u; }
                        }
                        U4 src[PQ];
                        MulTab tb[PQ];
#pragma unroll
                        for (int q = 0; q < PQ; q++) {
                            src[q] = kSplit ? lds_read16_split(acc + (size_t)sl_[q] * B, gl, B / 2)
                                            : *reinterpret_cast<const U4 *>(acc + (size_t)sl_[q] * B + gl * 16);
                            tb[q] = lds_multab(mt, cf[q]);
                        }
#pragma unroll
                        for (int q = 0; q < PQ; q++) gfmac16(a16[u], tb[q], src[q]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (on[u]) {
                        val[u] = gfmul16(lds_multab(mt, invc[su[u]]), a16[u]);
                        stream_store16<NT>(out_row(tgt[su[u]]), val[u]);
                    }
                uint32_t ew[U][KQ];
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int q = 0; q < KQ; q++) {
                        const uint32_t idx = o0[u] + np[u] + (uint32_t)(gl + q * LPR);
                        ew[u][q] = idx < o1[u] ? ((accbase + (uint32_t)g_slot[idx] * (uint32_t)B) | ((uint32_t)g_coef[idx] << 24)) : 0xFFFFFFFFu;
                    }
#pragma unroll
                for (int u = 0; u < U; u++) scatter(val[u], ew[u]);
            }
            __syncthreads();
        }
    };
    if (grouped) phase_g();
    else if (is_static && a.enc_clist) phase_b(std::integral_constant<int, 2>{});
    else if (lds_lists) phase_b(std::integral_constant<int, 1>{});
    else phase_b(std::integral_constant<int, 0>{});
    LDPC_STAMP(15);  // scatter: level phase
}

// Tier 1: one workgroup per (frame, slice); frames with more than tcap steps are left to tier 2.
// WPE = waves per SIMD the register allocation must allow (8 -> two 1024-thread workgroups per CU).
template <int LPR, int R, bool NT, int WPE, bool INPLACE>
__global__ __launch_bounds__(1024, WPE) void ldpc_scatter_kernel(ScatterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware placement (LDPC_AMD_SCATTER_XCD=0 switches it off): workgroups are dealt round-robin over the 8 XCDs, so
    // block b = 8 i + x handling slice i % nslices of frame (i / nslices) * 8 + x keeps the slices of a frame -- the
    // four 256-byte pieces of every 1 KB row, and the frame's schedule -- on one XCD at about the same time.
    // Speed only (3 % on the 4096-frame batch).
    int64_t f;
    int sl;
    if (a.xcd_map && (a.nframes & 7) == 0) {
        const int64_t x = blockIdx.x & 7, i = blockIdx.x >> 3;
        f = (i / a.nslices) * 8 + x;
        sl = (int)(i % a.nslices);
    } else {
        f = blockIdx.x / a.nslices;
        sl = (int)(blockIdx.x % a.nslices);
    }
#ifdef LDPC_AMD_MLDBG
    if (!a.static_sched && (int)a.sched_hdr[2 * f] > a.tcap && !(a.dbg & 32768)) return;
#else
    if (!a.static_sched && (int)a.sched_hdr[2 * f] > a.tcap) return;
#endif
    scatter_frame<LPR, R, NT, INPLACE, WPE>(a, smem, f, sl);
}

// Tier 2: the few frames with many steps (LDS sized for m accumulators), grid-stride over the compacted list.
template <int LPR, int R, bool NT, bool INPLACE, int WPE = 4>
__global__ __launch_bounds__(1024, WPE) void ldpc_scatter_big_kernel(ScatterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // work items (frame, slice) are handed out through a device counter, first come first served: their cost varies with the
    // frame's steps and levels, and with ~50 items per workgroup a fixed stride leaves the slowest workgroup a few items behind
    // A work item is t2_pieces consecutive pieces of ONE frame (1, 2 or all of them): the first sets the frame's tables up, the others find
    // them in place (scatter_frame<..., WARM>) -- the set-up was 9.7 % of a tier-2 workgroup's time on the cfg 3 batch, once per piece.
    // (few frames in the list -- the (4080,3060) batch has four -- keep one piece per item: the kernel then lasts one item, not one frame)
    int P = a.t2_pieces > 1 ? a.t2_pieces : 1;
    while (P > 1 && !a.t2_force && (int64_t)a.big_list[0] * (a.nslices / P) < (int64_t)8 * gridDim.x) P >>= 1;
    const int G = a.nslices / P;
    const int items = a.big_list[0] * G;
    int *slot = reinterpret_cast<int *>(smem + a.lds_rowctr) + 2;   // (ints 0 / 1 of the region are re-initialised by every frame)
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) *slot = atomicAdd(&a.big_list[1], 1);
        __syncthreads();
        const int it = *slot;
        if (it >= items) break;
        const int64_t f = a.big_list[2 + it / G];
        const int p0 = (it % G) * P;
        scatter_frame<LPR, R, NT, INPLACE, WPE>(a, smem, f, p0);
        for (int q = 1; q < P; q++) {
            __syncthreads();   // the last level of the piece before is through with the accumulators
            scatter_frame<LPR, R, NT, INPLACE, WPE, false, true>(a, smem, f, p0 + q);
        }
    }
}

// Encoder, persistent form: the schedule is the CODE's, so a workgroup sets its tables up once and then encodes (frame, slice) items
// handed out through a device counter (first come, first served: with a fixed stride the slowest CU's workgroups finish 7 % late).
// The counter resets itself: the last workgroup to find it exhausted zeroes it for the next launch of this context.
template <int LPR, int R, bool NT, int WPE>
__global__ __launch_bounds__(1024, WPE) void ldpc_scatter_static_kernel(ScatterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int items = (int)(a.nframes * a.nslices);
    int *slot = reinterpret_cast<int *>(smem + a.lds_rowctr) + 2;   // (ints 0 / 1 of the region are re-initialised by every item)
    auto next_item = [&]() -> int {
        __syncthreads();   // (also: the last level of the item before is through with the accumulators)
        if (threadIdx.x == 0) *slot = atomicAdd(&a.big_list[0], 1);
        __syncthreads();
        return *slot;
    };
    // (the first item, which sets the tables up, and the warm ones are two instantiations: a run-time flag kept both sets of values alive
    // across the loop -- 108 spilled registers against 12 of the one-item kernel)
    int it = next_item();
    if (it < items) {
        scatter_frame<LPR, R, NT, false, WPE, true, false>(a, smem, it / a.nslices, it % a.nslices);
        while ((it = next_item()) < items) scatter_frame<LPR, R, NT, false, WPE, true, true>(a, smem, it / a.nslices, it % a.nslices);
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&a.big_list[1], 1) == (int)gridDim.x - 1) {   // every workgroup has made its last request
            a.big_list[0] = 0; a.big_list[1] = 0;
            __threadfence();
        }
    }
}

#include "ml_kernel.inc"
#include "ml_pi.inc"
#define RELAX_LOG(i) c_log[i]
#define RELAX_EXP(i) c_exp[i]
#include "peel_relax.inc"

// =================================================================================================
// Synthetic inputs (role of the FPGA data_in kernel, OpenCL/device/ldpc_erasure_decoder_top.cl:57-120)
// =================================================================================================
__global__ void synth_bytes_kernel(uint64_t seed, uint32_t stream, uint64_t base, uint64_t count, uint8_t *dst)
{
    // 4 bytes per thread
    const uint64_t words = (count + 3) / 4;
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) {
            const uint64_t i = w * 4 + b;
            if (i < count) v |= (uint32_t)ldpc_synth_byte(seed, stream, base + i) << (8 * b);
        }
        if (w * 4 + 3 < count) reinterpret_cast<uint32_t *>(dst)[w] = v;
        else for (int b = 0; b < 4 && w * 4 + b < count; b++) dst[w * 4 + b] = (uint8_t)(v >> (8 * b));
    }
}

__global__ void synth_bernoulli_kernel(uint64_t seed, uint32_t stream, uint64_t base, uint64_t count,
                                       uint64_t thresh, uint8_t *dst)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = (uint8_t)ldpc_synth_bernoulli(seed, stream, base + i, thresh);
}

// Gilbert-Elliott channel (Matlab/Bursty_Error_Channel_Model_Generator.m:12-47) as a parallel scan.  The chain is
// sequential in the reference (state carried across symbols and frames, ErasureCodes_NonBinaryLDPCSim.m:163,192);
// a step is a function {good, bad} -> {good, bad} fixed by its transition draw, and functions compose
// associatively: pass 1 composes 64 steps per thread and 256 threads per block, pass 2 scans the block functions,
// pass 3 replays every segment from its now known entry state and writes the flags.
// A function is 2 bits: bit s = next state when entered in state s.
struct GeParams {
    uint64_t seed, count, first;   // symbols [0, count) are simulated, flags of [first, count) are written
    uint64_t ta, tb, t10, t01;     // thresholds: erase in good / bad state, good->bad, bad->good
};
constexpr int kGeSeg = 64;

__device__ __forceinline__ uint32_t ge_step_fn(const GeParams &p, uint64_t i)
{
    const uint64_t r2 = ldpc_synth_u32(p.seed, LDPC_SYNTH_STREAM_BURST_S, i);
    const uint32_t f0 = (r2 < p.t10) ? 1u : 0u;   // in good: go bad?
    const uint32_t f1 = (r2 < p.t01) ? 0u : 1u;   // in bad: go good?
    return f0 | (f1 << 1);
}
__device__ __forceinline__ uint32_t ge_compose(uint32_t first, uint32_t then)
{   // (then o first)(s) = then(first(s))
    const uint32_t a = (then >> (first & 1u)) & 1u, b = (then >> ((first >> 1) & 1u)) & 1u;
    return a | (b << 1);
}

__global__ __launch_bounds__(256) void ge_block_fn_kernel(GeParams p, uint32_t *blockfn)
{
    __shared__ uint32_t fn[256];
    const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * kGeSeg;
    uint32_t f = 2u;  // identity: 0 -> 0, 1 -> 1
    for (int t = 0; t < kGeSeg; t++)
        if (base + t < p.count) f = ge_compose(f, ge_step_fn(p, base + t));
    fn[threadIdx.x] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t g = 2u;
        for (int t = 0; t < 256; t++) g = ge_compose(g, fn[t]);
        blockfn[blockIdx.x] = g;
    }
}

__global__ void ge_scan_kernel(uint32_t nblocks, const uint32_t *blockfn, uint8_t *entry)
{
    if (blockIdx.x || threadIdx.x) return;
    uint32_t s = 0;  // next_state = 0 before the first symbol (ErasureCodes_NonBinaryLDPCSim.m:163)
    for (uint32_t b = 0; b < nblocks; b++) {
        entry[b] = (uint8_t)s;
        s = (blockfn[b] >> s) & 1u;
    }
}

__global__ __launch_bounds__(256) void ge_write_kernel(GeParams p, const uint8_t *entry, uint8_t *dst)
{
    __shared__ uint32_t fn[256];
    __shared__ uint8_t st0[256];
    const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * kGeSeg;
    uint32_t f = 2u;
    for (int t = 0; t < kGeSeg; t++)
        if (base + t < p.count) f = ge_compose(f, ge_step_fn(p, base + t));
    fn[threadIdx.x] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = entry[blockIdx.x];
        for (int t = 0; t < 256; t++) { st0[t] = (uint8_t)s; s = (fn[t] >> s) & 1u; }
    }
    __syncthreads();
    uint32_t s = st0[threadIdx.x];
    for (int t = 0; t < kGeSeg; t++) {
        const uint64_t i = base + t;
        if (i >= p.count) break;
        if (i >= p.first) {
            const uint64_t r1 = ldpc_synth_u32(p.seed, LDPC_SYNTH_STREAM_BURST_E, i);
            dst[i - p.first] = (uint8_t)(r1 < (s ? p.tb : p.ta));
        }
        s = (ge_step_fn(p, i) >> s) & 1u;
    }
}

// erasure flags of the FPGA source kernel: threefry4x32-20, key {1, seed}, counter = symbol index + 1
// (OpenCL/device/ldpc_erasure_decoder_top.cl:74-75,96-110; rule restated in include/ldpc_erasure_amd_synth.h)
// `first` = index of dst[0] in the run's symbol stream: a chunk of a long run continues the stream where the previous
// chunk stopped (the counter is 32 bits wide in the reference and wraps after 2^32 symbols; ldpc_fpga_erased keeps that).
__global__ void synth_fpga_kernel(uint32_t seed, uint64_t first, uint64_t count, int per64, uint8_t *dst)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = (uint8_t)ldpc_fpga_erased(seed, first + i, per64);
}

// =================================================================================================
// Self-test of the packed multiply against the log/antilog tables
// =================================================================================================
__global__ void selftest_kernel(int *bad)
{
    const uint32_t c = blockIdx.x;  // 0..255
    const MulTab t = load_multab(c);
    const uint32_t x = threadIdx.x; // 0..255
    const uint32_t packed = x | ((x ^ 0x5Au) << 8) | ((255u - x) << 16) | (((x * 7u) & 0xFFu) << 24);
    const uint32_t got = gfmul4(t, packed);
    uint32_t want = 0;
    for (int b = 0; b < 4; b++) {
        const uint32_t xb = (packed >> (8 * b)) & 0xFFu;
        want |= gfmul_log(c_log, c_exp, c, xb) << (8 * b);
    }
    if (got != want) atomicAdd(bad, 1);
    if (x == 0 && c != 0) {
        if (gfmul_log(c_log, c_exp, c, c_inv[c]) != 1u) atomicAdd(bad, 1);
    }
}

// Device copy probe: the same 16-byte non-temporal loads/stores the scatter kernel uses, nothing else.  bench.py runs
// it in the same process to quote the copy rate this box reaches next to the nominal HBM peak.
__global__ __launch_bounds__(1024) void copy_probe_kernel(const uint8_t *src, uint8_t *dst, uint64_t chunks)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < chunks; i += (uint64_t)gridDim.x * blockDim.x)
        stream_store16<true>(dst + i * 16, stream_load16<true>(src + i * 16));
}

#include "rs_kernels.inc"
#include "fpga_kernels.inc"

}  // namespace

// =================================================================================================
// Host side: constants, LDS layouts, launches
// =================================================================================================
// Every kernel instantiation that uses dynamic LDS is allowed the full 160 KB ONCE per device (process-wide kernel
// attribute): a per-launch attribute sized for one code / S could be shrunk by another context between set and launch.
static hipError_t allow_max_lds(const void *fn)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({dev, fn});
    return e;
}

hipError_t upload_constants(hipStream_t s)
{
    // built once (magic static); uploads of concurrent ldpc_amd_init calls then all send the same bytes
    static const std::vector<uint32_t> tab_v = [] { std::vector<uint32_t> t(256 * 8); build_mul3_tables(t.data()); return t; }();
    const uint32_t *tab = tab_v.data();
    const GfHost &g = gf_host();
    hipError_t e;
    if ((e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_mul3), tab, 256 * 8 * sizeof(uint32_t), 0, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_log), g.log, 256, 0, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_exp), g.exp, 512, 0, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_inv), g.inv, 256, 0, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

static PeelLds make_peel_lds(const DevCode &cd, bool fused, int wpb, bool gt = false)
{
    PeelLds L{};
    int off = 0;
    L.ell_col = off; if (!gt) off += align_up(cd.degpad * cd.mpad * (fused ? 4 : 2), 16);   // S = 1: id | log(coef) << 16
    L.ell_logc = off;
    L.lg = off; if (fused) off += 256;
    L.ex = off; if (fused) off += 512;
    L.wave0 = off;
    int w = 0;
    // st (u16 per symbol) dies after the peel: its space then holds the level offsets and the sorted steps;
    // steps (unsorted) dies after the sort: for S == 1 its space then holds the codeword
    const int lvl_bytes = align_up(4 * (cd.m + 2), 16);
    L.st = w; w += std::max(align_up(2 * cd.n, 16), lvl_bytes + align_up(4 * cd.m, 16));
    L.sorted = L.st + lvl_bytes;
    L.y = 0;
    L.steps = w; w += align_up(std::max(4 * cd.m, fused ? cd.n : 0), 16);
    L.slvl = w; w += align_up(2 * cd.m, 16);
    L.wave_stride = w;
    L.total = off + wpb * w;
    return L;
}

template <bool FUSED, bool GT = false>
static hipError_t launch_peel_t(const PeelArgs &a, int wpb, hipStream_t s)
{
    const int grid = (int)((a.nframes + wpb - 1) / wpb);
    const dim3 g(grid), b(wpb * 64);
    const size_t lds = (size_t)a.lds.total;
#define LDPC_PEEL_CASE(D)                                                                                  \
    case D: {                                                                                              \
        auto kfn = ldpc_peel_kernel<D, FUSED, GT>;                                                         \
        hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kfn));                                 \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL(kfn, g, b, lds, s, a);                                                          \
        return hipGetLastError();                                                                          \
    }
    switch (a.code.degpad) {
        LDPC_PEEL_CASE(8)
        LDPC_PEEL_CASE(14)
        LDPC_PEEL_CASE(16)
        LDPC_PEEL_CASE(24)
    }
#undef LDPC_PEEL_CASE
    return hipErrorInvalidValue;
}

static const int kLdsMax = 160 * 1024;

// ---- scatter kernel launch plan ---------------------------------------------------------------------------
struct ScatterPlan {
    int lpr = 0;        // 16-byte lanes per row piece (B = 16 * lpr bytes of every row per workgroup), 0 = unusable
    int nslices = 0;    // S / B
    int tcap = 0;       // tier 1 handles frames with <= tcap steps
    bool two_tier = false;
    int lds1 = 0, lds2 = 0;                       // dynamic LDS bytes of tier 1 / tier 2
    int o_tgt = 0, o_invc = 0, o_lvl = 0, o_ctr = 0, o_mt = 0, o_soc = 0, o_chk = 0;  // offsets behind the accumulators (relative)
    int soc_bytes = 0;
};

static int scatter_tail_bytes(const DevCode &cd, ScatterPlan &p)
{
    int off = 0;
    p.o_tgt = off; off += align_up(2 * cd.m, 16);
    p.o_invc = off; off += align_up(cd.m, 16);
    p.o_lvl = off; off += align_up(2 * (cd.m + 2), 16);
    p.o_ctr = off; off += 288;  // row-batch counter of the streaming phase, received-row count, bins of the sorted list / window counts
    p.o_mt = off;               // (the multiply tables start the LDS: counted by scatter_lds_bytes, not here)
    p.o_soc = off; off += align_up(2 * cd.n, 16);  // row kinds (u8), later the list of received rows (u16)
    p.soc_bytes = align_up(2 * cd.n, 16);
    p.o_chk = off; off += align_up(2 * (cd.m + 2), 16);  // check -> slot
    return off;
}

static ScatterPlan plan_scatter(const Knobs &kn, const DevCode &cd, int S)
{
    ScatterPlan p;
    if ((uint64_t)cd.n * (uint64_t)S >= (1ull << 32) || S >= (1 << 24) || cd.n >= (1 << 16)) return p;   // the kernel addresses a frame with 32-bit offsets (24-bit multiplies)
    int B = kn.scatter_b;  // A/B knob: bytes of every row per workgroup
    while (B > 16 && (S % B) != 0) B >>= 1;
    const int tail = scatter_tail_bytes(cd, p) + 8192;   // + the multiply tables in front of the accumulators
    while (B > 16 && cd.m * B + tail > 156 * 1024) B >>= 1;
    if (cd.m * B + tail > kLdsMax) return p;
    p.lpr = B / 16;
    p.nslices = S / B;
    p.lds2 = cd.m * B + tail;
    // Two workgroups per CU (each half of the 160 KB) hide one workgroup's set-up and level phase behind the
    // other's streaming phase: possible when the accumulators of the typical frame fit in half the LDS.
    const bool allow = kn.scatter_tiers != 1;
    const int half = kLdsMax / 2;
    int tcap = (half - tail) / B;
    if (allow && p.lpr >= 8 && tcap >= cd.m / 4 && tcap < cd.m) {
        p.two_tier = true;
        p.tcap = tcap;
        p.lds1 = tcap * B + tail;
    } else {
        p.tcap = cd.m;
        p.lds1 = p.lds2;
    }
    return p;
}

static void scatter_set_lds(ScatterArgs &sa, const ScatterPlan &p, int nacc)
{
    // [0, 8192) multiply tables | [8192, ...) accumulators | small tables   (kMtOff / kAccOff of scatter_frame)
    const int base = 8192 + align_up(nacc * 16 * p.lpr, 16);
    sa.lds_acc = 8192;
    sa.lds_tgt = base + p.o_tgt; sa.lds_invc = base + p.o_invc; sa.lds_lvlend = base + p.o_lvl;
    sa.lds_rowctr = base + p.o_ctr; sa.lds_mt = 0; sa.lds_soc = base + p.o_soc; sa.lds_chk = base + p.o_chk;
    sa.lds_soc_bytes = p.soc_bytes;
}

template <int LPR, int R>
static int launch_scatter_lpr(ldpc_amd_ctx *ctx, const ScatterPlan &p, ScatterArgs sa, int32_t *big_list)
{
    constexpr int THREADS = (LPR >= 8) ? 1024 : (LPR >= 2 ? 512 : 256);
    const Knobs &kn = ctx->knobs;
    const bool nt = kn.scatter_nt != 0;
    sa.dyn_rows = kn.scatter_dyn;
    sa.xcd_map = kn.scatter_xcd;  // measured: 3.12 vs 3.22 ms once the set-up was shortened; =0 switches it off
    const dim3 grid((unsigned)(sa.nframes * sa.nslices));
    // tier 1
    sa.tcap = p.tcap; sa.nslots = p.tcap; sa.big_list = nullptr;
    scatter_set_lds(sa, p, p.tcap);
    if constexpr (LPR >= 8 && R == 2) {   // (the shipped R; the SCATTER_R variants keep the one-item kernel: a third of the instantiations)
        // Encoder: the persistent form (ENC_PERSIST; not with ENC_LIST, whose row list lives where the kept lists do)
        if (sa.static_sched && kn.enc_persist != 0 && !sa.enc_list && !sa.inplace && (sa.enc_group || sa.enc_clist)) {
            const int per_cu = std::max(1, std::min(p.two_tier ? 2 : 1, kLdsMax / std::max(1, p.lds1)));
            const dim3 gp((unsigned)std::min<int64_t>((int64_t)grid.x, (int64_t)ctx->sm_count * per_cu));
            // the item counter: self-resetting, so zeroed once (synchronously: whatever stream the context is moved to later sees it);
            // consecutive launches take consecutive counters of a ring, so two encodes in flight (a caller that changed the context's
            // stream without waiting) do not share one
            constexpr int kEncCtrs = 64;
            if (!ctx->encctr.p) {
                int rc_e;
                if ((rc_e = scratch_reserve(ctx, ctx->encctr, (size_t)kEncCtrs * 64))) return rc_e;
                LDPC_HIP_TRY(ctx, hipMemset(ctx->encctr.p, 0, (size_t)kEncCtrs * 64));
            }
            sa.big_list = (int32_t *)ctx->encctr.p + 16 * (ctx->enc_launches++ % kEncCtrs);
            char nm[96];
            snprintf(nm, sizeof(nm), "ldpc_scatter_static_kernel<%d, %d, %s, %d>", LPR, R, nt ? "true" : "false", p.two_tier ? 8 : 4);
            ctx->prof_names[LDPC_AMD_PROF_APPLY] = nm;
#define LDPC_SCATTER_PS(NTV, WPE)                                                                            \
    {                                                                                                        \
        auto kfn = ldpc_scatter_static_kernel<LPR, R, NTV, WPE>;                                             \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));                               \
        hipLaunchKernelGGL(kfn, gp, dim3(THREADS), (size_t)p.lds1, ctx->stream, sa);                         \
    }
            if (p.two_tier) { if (nt) LDPC_SCATTER_PS(true, 8) else LDPC_SCATTER_PS(false, 8) }
            else { if (nt) LDPC_SCATTER_PS(true, 4) else LDPC_SCATTER_PS(false, 4) }
#undef LDPC_SCATTER_PS
            LDPC_HIP_TRY(ctx, hipGetLastError());
            return LDPC_AMD_OK;
        }
    }
#define LDPC_SCATTER_T1(NTV, WPE, IPV)                                                                         \
    {                                                                                                        \
        auto kfn = ldpc_scatter_kernel<LPR, R, NTV, WPE, IPV>;                                                    \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));                               \
        hipLaunchKernelGGL(kfn, grid, dim3(THREADS), (size_t)p.lds1, ctx->stream, sa);                       \
    }
    const bool ip = sa.inplace != 0;
    {
        char nm[96];
        snprintf(nm, sizeof(nm), "ldpc_scatter_kernel<%d, %d, %s, %d, %s>", LPR, R, nt ? "true" : "false", p.two_tier ? 8 : 4,
                 ip ? "true" : "false");
        ctx->prof_names[LDPC_AMD_PROF_APPLY] = nm;
    }
    if (p.two_tier) {
        if (ip) { if (nt) LDPC_SCATTER_T1(true, 8, true) else LDPC_SCATTER_T1(false, 8, true) }
        else { if (nt) LDPC_SCATTER_T1(true, 8, false) else LDPC_SCATTER_T1(false, 8, false) }
    } else {
        if (ip) { if (nt) LDPC_SCATTER_T1(true, 4, true) else LDPC_SCATTER_T1(false, 4, true) }
        else { if (nt) LDPC_SCATTER_T1(true, 4, false) else LDPC_SCATTER_T1(false, 4, false) }
    }
#undef LDPC_SCATTER_T1
    LDPC_HIP_TRY(ctx, hipGetLastError());
    if (p.two_tier && big_list && !(sa.dbg & 32768)) {
        sa.tcap = sa.code.m; sa.nslots = sa.code.m; sa.big_list = big_list;
        scatter_set_lds(sa, p, sa.code.m);
        {   // pieces of a frame per work item (SCATTER_T2P; the list modes of the stream rebuild their row list per piece: one piece)
            int tp = kn.scatter_dyn <= 1 ? std::max(1, kn.scatter_t2p) : 1;
            while (tp > 1 && (sa.nslices % tp) != 0) tp >>= 1;
            sa.t2_pieces = tp; sa.t2_force = kn.scatter_t2p_force;
        }
        const dim3 g2((unsigned)std::min<int64_t>(sa.nframes * sa.nslices, ctx->sm_count));
        // tier 2 runs one workgroup per CU (4 waves per SIMD, 128 VGPRs each): more row pieces in flight per lane group make up
        // for part of the missing occupancy.  Round 2: two / three / four pieces 4.50 / 4.28 / 4.24 ms on cfg 3 (four shipped, 160 B
        // of spills per lane); round 3, with six vector instructions fewer per edge turn: 4.01 / 3.95 / 4.08 -- three ship (no spills)
        const int r2 = LPR == 16 ? kn.scatter_r2 : 0;
#define LDPC_SCATTER_T2_R(RV, NTV, IPV)                                                                      \
    {                                                                                                        \
        auto kfn = ldpc_scatter_big_kernel<LPR, RV, NTV, IPV>;                                               \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));                               \
        hipLaunchKernelGGL(kfn, g2, dim3(THREADS), (size_t)p.lds2, ctx->stream, sa);                         \
    }
#define LDPC_SCATTER_T2(NTV, IPV)                                                                            \
    if (r2 == 4) LDPC_SCATTER_T2_R((LPR == 16 ? 4 : R), NTV, IPV)                                            \
    else if (r2 == 3) LDPC_SCATTER_T2_R((LPR == 16 ? 3 : R), NTV, IPV)                                       \
    else LDPC_SCATTER_T2_R(R, NTV, IPV)
        {
            char nm[96];
            snprintf(nm, sizeof(nm), "ldpc_scatter_big_kernel<%d, %d, %s, %s>", LPR, (LPR == 16 && (r2 == 3 || r2 == 4)) ? r2 : R, nt ? "true" : "false",
                     ip ? "true" : "false");
            ctx->prof_names[LDPC_AMD_PROF_APPLY_TIER2] = nm;
        }
        hipEvent_t ev2 = prof_begin(ctx, 2);
        // SCATTER_T2B = 128: tier 2 with 128-byte pieces -- all m accumulators then take half a CU's LDS, so TWO tier-2 workgroups
        // share a CU (one streams while the other sets up / runs its levels), at twice the per-byte instruction count of an edge turn
        const int tail_b = p.lds2 - sa.code.m * 16 * p.lpr;   // tables + small arrays of the plan
        const bool t2_128 = LPR == 16 && kn.scatter_t2b == 128 && (sa.S % 128) == 0 && sa.code.m * 128 + tail_b <= kLdsMax / 2;
        if (t2_128) {
            ScatterPlan p2 = p;
            p2.lpr = 8; p2.nslices = sa.S / 128; p2.lds2 = sa.code.m * 128 + tail_b;
            sa.nslices = p2.nslices;
            while (sa.t2_pieces > 1 && (sa.nslices % sa.t2_pieces) != 0) sa.t2_pieces >>= 1;
            scatter_set_lds(sa, p2, sa.code.m);
            const dim3 g3((unsigned)std::min<int64_t>(sa.nframes * sa.nslices, (int64_t)ctx->sm_count * 2));
            ctx->prof_names[LDPC_AMD_PROF_APPLY_TIER2] = std::string("ldpc_scatter_big_kernel<8, 2, ") + (nt ? "true" : "false") + ", " + (ip ? "true" : "false") + ", 8>";
#define LDPC_SCATTER_T2B(NTV, IPV)                                                                           \
    {                                                                                                        \
        auto kfn = ldpc_scatter_big_kernel<8, 2, NTV, IPV, 8>;                                               \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));                               \
        hipLaunchKernelGGL(kfn, g3, dim3(1024), (size_t)p2.lds2, ctx->stream, sa);                           \
    }
            if (ip) { if (nt) LDPC_SCATTER_T2B(true, true) else LDPC_SCATTER_T2B(false, true) }
            else { if (nt) LDPC_SCATTER_T2B(true, false) else LDPC_SCATTER_T2B(false, false) }
#undef LDPC_SCATTER_T2B
        } else
        if (ip) { if (nt) LDPC_SCATTER_T2(true, true) else LDPC_SCATTER_T2(false, true) }
        else { if (nt) LDPC_SCATTER_T2(true, false) else LDPC_SCATTER_T2(false, false) }
#undef LDPC_SCATTER_T2_R
#undef LDPC_SCATTER_T2
        prof_end(ctx, LDPC_AMD_PROF_APPLY_TIER2, ev2);
        LDPC_HIP_TRY(ctx, hipGetLastError());
    }
    return LDPC_AMD_OK;
}

static int launch_scatter(ldpc_amd_ctx *ctx, const ScatterPlan &p, const ScatterArgs &sa, int32_t *big_list)
{
    const int rr = ctx->knobs.scatter_r;
    switch (p.lpr) {
        case 16: return rr == 4 ? launch_scatter_lpr<16, 4>(ctx, p, sa, big_list) : (rr == 1 ? launch_scatter_lpr<16, 1>(ctx, p, sa, big_list) : launch_scatter_lpr<16, 2>(ctx, p, sa, big_list));
        case 8: return rr == 4 ? launch_scatter_lpr<8, 4>(ctx, p, sa, big_list) : launch_scatter_lpr<8, 2>(ctx, p, sa, big_list);
        case 4: return launch_scatter_lpr<4, 1>(ctx, p, sa, big_list);
        case 2: return launch_scatter_lpr<2, 1>(ctx, p, sa, big_list);
        case 1: return launch_scatter_lpr<1, 1>(ctx, p, sa, big_list);
    }
    return set_error(ctx, LDPC_AMD_EUNSUP, "scatter: bad plan");
}

int launch_decode(ldpc_amd_ctx *ctx, const DecodeArgs &d)
{
    const DevCode &cd = d.code;
    if (d.nframes <= 0) return LDPC_AMD_OK;
    const bool fused = (d.S == 1) && !d.flags_only;
    if (!fused && !d.flags_only && (d.S % 16) != 0) return set_error(ctx, LDPC_AMD_EUNSUP, "S must be 1 or a multiple of 16 (got %d)", d.S);
    if (d.max_sweeps < 1) return set_error(ctx, LDPC_AMD_EINVAL, "max_sweeps must be >= 1");

    // bound the per-call workspaces: long batches are processed in chunks of frames
    const int64_t kChunk = fused ? (int64_t)ctx->knobs.chunk_s1 : 16384;
    if (d.nframes > kChunk) {
        for (int64_t f0 = 0; f0 < d.nframes; f0 += kChunk) {
            DecodeArgs c = d;
            c.nframes = std::min(kChunk, d.nframes - f0);
            if (d.sym) c.sym = d.sym + f0 * (int64_t)d.in_rows * d.S;
            if (d.erased) c.erased = d.erased + f0 * cd.n;
            if (d.out) c.out = d.out + f0 * (int64_t)cd.n * d.S;
            if (d.sweeps) c.sweeps = d.sweeps + f0;
            if (d.residual) c.residual = d.residual + f0;
            if (d.status) c.status = d.status + f0;
            if (d.residual_sys) c.residual_sys = d.residual_sys + f0;
            int rcc = launch_decode(ctx, c);
            if (rcc) return rcc;
        }
        return LDPC_AMD_OK;
    }

    // packet path: scatter kernel (rows read once, accumulators in LDS) unless the code's columns are too
    // heavy for the padded per-source lists, or LDPC_AMD_APPLY=gather asks for the gather kernel (A/B runs)
    const Knobs &kn = ctx->knobs;
    const bool want_gather = kn.apply_gather != 0;
    ScatterPlan plan{};
    if (!fused && !d.flags_only && !want_gather && cd.maxcoldeg <= 16) plan = plan_scatter(kn, cd, d.S);
    const bool use_scatter = plan.lpr > 0;
    if (d.inplace && !use_scatter) return set_error(ctx, LDPC_AMD_EUNSUP, "in-place decode needs the scatter kernel");

    // workgroup shape: the peel is latency bound (serial solve chain per frame), so pick the frames-per-workgroup
    // that puts the most wavefronts on a CU within its 160 KB of LDS (the code tables are shared by a workgroup)
    int wpb = 1;
    bool gt = false;
    PeelLds L = make_peel_lds(cd, fused, 1);
    if (L.total > kLdsMax) return set_error(ctx, LDPC_AMD_EUNSUP, "code too large for LDS (%d bytes)", L.total);
    {
        int best = 0;
        const bool env_w = kn.peel_wpb > 0;   // diagnostic: cap the wavefronts (= frames) per workgroup
        const int wcap = env_w ? std::max(1, std::min(16, kn.peel_wpb)) : 16;
        for (int w = 1; w <= wcap; w++) {
            const PeelLds t = make_peel_lds(cd, fused, w);
            if (t.total > kLdsMax) break;
            const int waves = env_w ? w : std::min(32, (kLdsMax / t.total) * w);
            if (waves > best) { best = waves; wpb = w; L = t; }
        }
        // S = 1, long batch: with the code tables left in global memory more frames fit on a CU.  Worth it when the
        // batch is several rounds deep anyway (a single round is latency bound and prefers the LDS tables).
        if (fused && !env_w && kn.peel_gt != 0) {
            int bestg = 0, wg = 1;
            PeelLds Lg = L;
            for (int w = 1; w <= 16; w++) {
                const PeelLds t = make_peel_lds(cd, fused, w, true);
                if (t.total > kLdsMax) break;
                const int waves = std::min(32, (kLdsMax / t.total) * w);
                if (waves > bestg) { bestg = waves; wg = w; Lg = t; }
            }
            const bool deep = d.nframes >= (int64_t)3 * best * ctx->sm_count;
            // measured: (4080,3060) 7 -> 11 frames per CU: -18 %, (4000,2000) 5 -> 8: -15 %, (2040,1530) 16 -> 21: +12 % (slower)
            if (kn.peel_gt == 1 || (deep && bestg * 20 >= best * 29)) { gt = true; wpb = wg; L = Lg; }
        }
    }

    ctx->last_plan[0] = wpb; ctx->last_plan[1] = std::min(32, (kLdsMax / L.total) * wpb); ctx->last_plan[2] = gt ? 1 : 0;
    ctx->last_plan[3] = L.total; ctx->last_plan[4] = L.wave_stride; ctx->last_plan[5] = use_scatter ? plan.lpr * 16 : 0;
    ctx->last_plan[6] = use_scatter ? plan.tcap : 0; ctx->last_plan[7] = use_scatter && plan.two_tier ? 1 : 0;
    const int64_t nf = d.nframes;
    int rc;
    if ((rc = scratch_reserve(ctx, ctx->mllist, sizeof(int32_t) * ((size_t)(1 + kMlClasses) * nf + kMlHdr)))) return rc;
    if (d.do_ml && (rc = scratch_reserve(ctx, ctx->mlstate, (size_t)nf * cd.n))) return rc;
    LDPC_HIP_TRY(ctx, hipMemsetAsync(ctx->mllist.p, 0, kMlHdr * sizeof(int32_t), ctx->stream));
    if (use_scatter && plan.two_tier) {
        if ((rc = scratch_reserve(ctx, ctx->biglist, sizeof(int32_t) * (size_t)(nf + 2)))) return rc;
        LDPC_HIP_TRY(ctx, hipMemsetAsync(ctx->biglist.p, 0, 2 * sizeof(int32_t), ctx->stream));   // [0] count, [1] tier 2's work counter
    }

    // ---- the ML stage (a2-a4): prepared before the packet kernel is launched, because in packet mode its pattern-only part --
    //      the fast path and the factorisation of what the fast path leaves -- can run beside the packet kernel (second stream)
    MlArgs ma{};
    PiArgs pi{};
    int pi_nw = 0, pi_total = 0, grid = 0, total = 0, ml_threads = 0, solve_b = 0;
    bool ml_overlap = false, ml_prepared = false, ml_front_done = false;
    hipEvent_t ml_ev = nullptr;
    auto ml_prepare = [&]() -> int {
        int rc;
        ma.code = cd; ma.S = d.S;
        ma.Spad = fused ? 16 : d.S;
        ma.ml_list = (const int32_t *)ctx->mllist.p; ma.ml_state = (const uint8_t *)ctx->mlstate.p; ma.nframes = nf;
        ma.out = d.out; ma.status = d.status;
        const int maxrow = align_up(cd.m, 16) + 32;
        ma.maxrow = maxrow;
        int off = 0;
        // rlist[3][m] u32; colmap[n] u16 is only alive while rlist[1..2] are not, and shares their bytes
        ma.lds_rlist = off; ma.lds_colmap = off + 4 * cd.m;
        // (packets: the same bytes later hold the per-level histogram, 2 m + 4 words, and the slot map, m halfwords)
        off += align_up(std::max(std::max(12 * cd.m, 4 * cd.m + 2 * cd.n), 4 * (2 * cd.m + 4) + 2 * cd.m + 16), 16);
        ma.lds_elist = off; off += align_up(2 * cd.m, 16);
        ma.lds_colv = off; off += align_up(3 * cd.mpad, 16);
        ma.lds_perm = off; off += align_up(2 * cd.m + 2, 16);
        ma.lds_iperm = off; off += align_up(2 * cd.m, 16);
        ma.lds_orow = off; off += align_up(2 * cd.m, 16);
        ma.lds_plog = off; off += align_up(cd.m, 16);
        ma.lds_mt = off; off += 8192;
        ma.lds_lg = off; off += 256;
        ma.lds_ex = off; off += 1024;
        ma.lds_misc = off; off += 128 + 4 * kMlClasses;
        ma.lds_lvl = off; off += align_up(4 * (cd.m + 2), 16);
        ma.lds_A = off;
        if (off > kLdsMax) return set_error(ctx, LDPC_AMD_EUNSUP, "ML stage: LDS need %d bytes", off);
        if (cd.m > 4096) return set_error(ctx, LDPC_AMD_EUNSUP, "ML stage: more than 4096 checks (%d)", cd.m);   // 12-bit row fields of the pivot key
        // ML_PACK = P workgroups per CU (1024 / P threads, 160 KB / P of LDS each): the factorisation is bound by the instructions
        // all wavefronts of a workgroup issue per column, not by the work in a column, so P small workgroups -- P systems per CU,
        // their matrices in the global scratch (L2) when they do not fit the LDS share -- issue P times fewer of them per system
        const int pack = std::max(1, std::min(4, kn.ml_pack));
        total = std::max(off + 256, (kLdsMax / pack) & ~255);
        if (total > kLdsMax) return set_error(ctx, LDPC_AMD_EUNSUP, "ML stage: LDS need %d bytes", total);
        ma.capA = (total - off) & ~15;
        grid = (int)std::min<int64_t>(nf, (int64_t)ctx->sm_count * pack);
        const size_t perA = (size_t)cd.m * maxrow, perR = fused ? 0 : (size_t)cd.m * d.S;
        if ((rc = scratch_reserve(ctx, ctx->mlws, (perA + perR) * grid + 256))) return rc;
        // work counters: [0] frame hand-out counter, [2..3] arena bump pointer (u64), [4] task counter of the solve kernel.  They
        // sit in the free tail of the residual list's header (ints 18..23), which launch_decode zeroes with that header: one
        // memset per call less (a call is launch-bound at S = 1)
        // [1] frames deferred by the launch beside the packet kernel, [5] frame hand-out counter of the fast path (ldpc_ml_pi_kernel)
        // header ints 24..27: verification failures of the fast path, hand-out counters of the two launches that redo those frames,
        // frames the fast path emitted (read back by ldpc_amd_ml_stats with [0] and [19] = ma.work[1], the deferred frames)
        static_assert(kMlHdr == 32 && 1 + kMlClasses <= 17, "ml_list header: no room for the work counters");
        ma.work = (int32_t *)ctx->mllist.p + 18;
        ma.work2 = (int32_t *)ctx->mllist.p + 17;   // hand-out counter of the launch that solves the deferred frames (mode 2)
        ma.nfail = (int32_t *)ctx->mllist.p + 24;   // frames whose fast-path solution failed the consistency check (not codewords)
        ma.work3 = (int32_t *)ctx->mllist.p + 25;   // hand-out counter of the launch that factors them again, exactly (mode 3)
        ma.wsA = (uint8_t *)ctx->mlws.p + 256;
        ma.wsR = ma.wsA + perA * grid;
        // packets: the ML kernel factors every residual system on bytes and emits a solve schedule (64-bit ops grouped by
        // dependency level) into an arena; ldpc_ml_solve_kernel then runs the schedules on LDS-resident row slices.
        // Frames whose schedule does not fit the arena are solved inside the ML kernel (same bytes, slower).
        ma.use_solve = (!fused && kn.ml_solve != 0) ? 1 : 0;
        ma.dbg = kn.ml_dbg;
        if (ma.use_solve) {
            solve_b = kn.ml_solve_b;   // A/B knob
            while (solve_b > 16 && (d.S % solve_b) != 0) solve_b >>= 1;
            const int tail_sv = align_up(4 * (2 * cd.m + 6), 16) + 8192 + 16 + 4 * kMlClasses;
            while (solve_b > 16 && cd.m * solve_b + tail_sv > 79 * 1024) solve_b >>= 1;
            if (cd.m * solve_b + tail_sv > kLdsMax || cd.n > 65535) ma.use_solve = 0;
        }
        ma.solve_b = solve_b;
        if (ma.use_solve) {
            // The arena follows DEMAND, not the batch size: 16 MB to start with (2 M words); the words the previous call asked for
            // come back through a pinned host word (copied behind every call, never waited for), and when they exceeded three quarters of
            // the arena it grows to twice that demand, at most 1 GB.  A call that overflows is still correct -- the frames
            // that do not fit are solved inside the ML kernel -- so a steady workload is at full speed from its second or third
            // batch, and a batch message passing completes pins 16 MB instead of 64 KB per frame.
            size_t words = std::max<size_t>(ctx->ml_arena_words, (size_t)1 << 21);
            if (ctx->ml_head_host) {
                const unsigned long long need = *ctx->ml_head_host;   // demand of an earlier call (whatever has landed)
                if (need > words / 4 * 3) words = std::min<size_t>(std::max<size_t>(words, (size_t)need * 2), (size_t)1 << 27);
            }
            if (kn.ml_arena_words >= 1024) words = (size_t)kn.ml_arena_words;   // test knob: a small arena makes some frames fall back
            ctx->ml_arena_words = kn.ml_arena_words >= 1024 ? ctx->ml_arena_words : words;
            if ((rc = scratch_reserve(ctx, ctx->mlops, words * 8)) || (rc = scratch_reserve(ctx, ctx->mlrec, (size_t)nf * 32))) return rc;
            ma.ops = (unsigned long long *)ctx->mlops.p; ma.ops_cap = words;
            ma.ops_head = (unsigned long long *)(ma.work + 2);
            ma.rec = (uint32_t *)ctx->mlrec.p;
        } else if (!fused) {
            if ((rc = scratch_reserve(ctx, ctx->mlrec, (size_t)nf * 32))) return rc;
            ma.rec = (uint32_t *)ctx->mlrec.p;   // the fall-back flag is written in either case
        }
        ml_threads = kn.ml_threads > 0 ? kn.ml_threads : std::max(256, (1024 / pack) & ~63);
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(ldpc_ml_kernel)));
        ctx->prof_names[LDPC_AMD_PROF_ML] = "ldpc_ml_kernel";
        // Packets: the fast path first (ml_pi.inc: peel on, inactivate, small dense system; a wavefront per frame).  It emits
        // the schedules of the full-rank frames; ldpc_ml_kernel then factors what it left (rank-deficient frames, rec[6] = 0).
        // (A context whose LAST packet batch had no residual frame at all -- its arena demand, back through the pinned host word, was
        // zero -- skips the fast path's four extra launches for this batch: 23 -> 8.5 us of empty launches per step on a workload
        // message passing completes.  Whatever does reach the ML stage then is factored exactly; the next batch has the fast path again.)
        const bool ml_quiet = kn.ml_pi_adaptive && ctx->ml_head_host && ctx->ml_head_valid && *ctx->ml_head_host == 0;
        if (ma.use_solve && kn.ml_pi != 0 && !ml_quiet) {
            int shared = 8192 + 256 + 1024 + 64;   // multiply tables, log, antilog, size-class counts
            pi.lds_mt = 0; pi.lds_lg = 8192; pi.lds_ex = 8192 + 256;
            pi.lds_edges = pi.lds_rowptr = -1;
            const int rows_bytes = align_up(4 * cd.nnz, 16) + align_up(4 * (cd.m + 1), 16);
            if (rows_bytes <= 40 * 1024) {   // the code's rows in LDS (every frame walks them a few times); else from global memory
                pi.lds_edges = shared; pi.lds_rowptr = shared + align_up(4 * cd.nnz, 16);
                shared += rows_bytes;
            }
            int o = 0;
            pi.o_vinfo = o; o += align_up(2 * cd.n, 16);
            pi.o_cnt = o; o += align_up(4 * (cd.m + 264), 16);
            pi.o_lvl = o; o += align_up(4 * cd.m, 16);
            pi.o_ustate = o; o += align_up(2 * cd.m, 16);
            pi.o_uvar = o; o += align_up(2 * cd.m, 16);
            pi.o_stepc = o; o += align_up(2 * cd.m, 16);
            pi.o_stepi = o; o += align_up(2 * cd.m, 16);
            pi.o_slvl = o; o += align_up(2 * cd.m, 16);
            pi.o_sginv = o; o += align_up(cd.m, 16);
            pi.o_queue = o; o += align_up(2 * cd.m, 16);
            pi.o_candpiv = o; o += align_up(2 * cd.m, 16);
            pi.o_inact = o; o += 512;
            pi.o_sel = o; o += 512;
            pi.o_sellvl = o; o += 512;
            pi.o_dlog = o; o += 256;
            pi.o_misc = o; o += 16 + 512;   // queue tail; the winners of a batch
            pi.o_av = o;
            const int budget = std::max(32, std::min(160, kn.ml_pi_lds)) * 1024 - shared;
            const int av_want = std::min(cd.m * 32, 16 * 1024);
            pi_nw = std::max(1, std::min(std::max(1, std::min(4, kn.ml_pi_waves)), budget / (o + av_want)));
            const int per_wave = (budget / pi_nw) & ~15;
            if (per_wave >= o + 1024 && cd.maxdeg <= kWave && cd.maxcoldeg <= 16 && cd.m < 0x8000 && cd.n < 0xFFFF) {
                pi.av_bytes = per_wave - o;
                pi.lds_wave0 = shared; pi.lds_wave_stride = per_wave;
                pi_total = shared + pi_nw * per_wave;
            } else {
                pi_nw = 0;
            }
        }
        ma.use_pi = pi_nw > 0 ? 1 : 0;
        if (pi_nw > 0) {
            pi.code = cd; pi.ml_list = ma.ml_list; pi.nframes = nf; pi.ml_state = ma.ml_state; pi.work = ma.work + 5;
            pi.status = d.status; pi.ops = ma.ops; pi.ops_cap = ma.ops_cap; pi.ops_head = ma.ops_head; pi.rec = ma.rec;
            pi.solve_b = solve_b;
            pi.verify = kn.ml_pi == 1 ? 1 : 0;
            pi.ndone = (int32_t *)ctx->mllist.p + 27;   // frames the fast path emitted (ldpc_amd_ml_stats)
            pi.imax = kn.ml_pi_imax;
            LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(ldpc_ml_pi_kernel)));
        }
        // The factorisation needs the erasure pattern only: in packet mode (schedules, not payload) it runs on a second stream
        // beside the packet kernel.  A frame whose schedule does not fit the arena is DEFERRED there (rec[5] = 2, ma.work[1] counts
        // them) and solved by a second launch behind the packet kernel -- its in-kernel solve reads the payload.
        // (Only with the fast path: beside the packet kernel the factorisation of EVERY residual frame takes as much from the packet
        // kernel as it saves -- 6.87 against 6.94 ms on cfg 3 -- while the few frames the fast path leaves disappear behind it.)
        ml_overlap = ma.use_solve && use_scatter && pi_nw > 0 && kn.ml_overlap != 0;
        if (ml_overlap) {
            if (!ctx->aux_ml) {   // lowest priority: its kernels fill the gaps the packet kernel leaves, not the other way round
                int lo = 0, hi = 0;
                LDPC_HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
                LDPC_HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->aux_ml, hipStreamNonBlocking, kn.ml_overlap_prio ? lo : 0));
            }
            for (hipEvent_t &e : ctx->ml_events)
                if (!e) LDPC_HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        return LDPC_AMD_OK;
    };
    // fast path + factorisation (pattern only in packet mode)
    auto ml_front = [&]() -> int {
        int rc;
        (void)rc;
        // (profiling: LDPC_AMD_PROF_ML is the stage's share of the MAIN stream.  With the fast path and the factorisation both beside
        // the packet kernel that is ml_back alone; with ML_OVERLAP=1 it comes in two pieces, i.e. two launches of the kind per call)
        ml_ev = (ml_overlap && kn.ml_overlap == 2) ? nullptr : prof_begin(ctx);
        hipStream_t st = ctx->stream;
        auto fork = [&]() -> int {
            prof_end(ctx, LDPC_AMD_PROF_ML, ml_ev);
            ml_ev = nullptr;
            LDPC_HIP_TRY(ctx, hipEventRecord(ctx->ml_events[0], ctx->stream));
            LDPC_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_ml, ctx->ml_events[0], 0));
            st = ctx->aux_ml;
            return LDPC_AMD_OK;
        };
        if (ml_overlap && kn.ml_overlap == 2 && (rc = fork())) return rc;   // =2: the fast path beside the packet kernel too
        if (pi_nw > 0) {
            const int wgs_per_cu = std::max(1, kLdsMax / pi_total);
            int pgrid = (int)std::min<int64_t>((nf + pi_nw - 1) / pi_nw, (int64_t)ctx->sm_count * wgs_per_cu);
            if (kn.ml_pi_wgs > 0) pgrid = std::min(pgrid, kn.ml_pi_wgs);
            hipLaunchKernelGGL(ldpc_ml_pi_kernel, dim3(pgrid), dim3(64 * pi_nw), (size_t)pi_total, st, pi);
            LDPC_HIP_TRY(ctx, hipGetLastError());
        }
        if (ml_overlap && kn.ml_overlap != 2 && (rc = fork())) return rc;
        ma.mode = ml_overlap ? 1 : 0;
        hipLaunchKernelGGL(ldpc_ml_kernel, dim3(grid), dim3(ml_threads), (size_t)total, st, ma);
        LDPC_HIP_TRY(ctx, hipGetLastError());
        if (ml_overlap) LDPC_HIP_TRY(ctx, hipEventRecord(ctx->ml_events[1], ctx->aux_ml));
        return LDPC_AMD_OK;
    };
    // what needs the payload: deferred frames, the solve kernel
    auto ml_back = [&]() -> int {
        if (ml_overlap) {
            ml_ev = prof_begin(ctx);
            LDPC_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ml_events[1], 0));
            ma.mode = 2;   // the deferred frames only (none, normally: the kernel leaves after one load)
            hipLaunchKernelGGL(ldpc_ml_kernel, dim3(grid), dim3(ml_threads), (size_t)total, ctx->stream, ma);
            LDPC_HIP_TRY(ctx, hipGetLastError());
        }
        if (ma.use_solve && kn.ml_solve != 2) {   // =2: diagnostic, schedules emitted but not run (timing of the factor part)
            MlSolveArgs sv{};
            sv.code = cd; sv.S = d.S; sv.nslices = d.S / solve_b; sv.nframes = nf; sv.ml_list = ma.ml_list; sv.rec = ma.rec; sv.ops = ma.ops;
            sv.out = d.out; sv.work = ma.work + 4;
            sv.dbg = ma.dbg; sv.err = ctx->dev_err_host;
            sv.nfail = ma.nfail; sv.round = 1;
            int o = 8192 + cd.m * solve_b;   // multiply tables first (kMlSlot0), then the slots
            sv.lds_tab = o; o += align_up(4 * (2 * cd.m + 6), 16);
            sv.lds_mt = 0;
            sv.lds_misc = o; o += 16 + 4 * kMlClasses;
            const int per_cu = std::max(1, std::min(4, kLdsMax / o));
            const dim3 sg((unsigned)std::min<int64_t>(nf * sv.nslices, (int64_t)ctx->sm_count * per_cu));
#define LDPC_ML_SOLVE(LPRV)                                                                                   \
    {                                                                                                        \
        auto sfn = ldpc_ml_solve_kernel<LPRV>;                                                               \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(sfn)));                               \
        hipLaunchKernelGGL(sfn, sg, dim3(512), (size_t)o, ctx->stream, sv);                                  \
    }
            {
                char nm[64];
                snprintf(nm, sizeof(nm), "ldpc_ml_solve_kernel<%d>", solve_b / 16);
                ctx->prof_names[LDPC_AMD_PROF_ML_SOLVE] = nm;
            }
            hipEvent_t evs = prof_begin(ctx, 2);
            switch (solve_b) {
                case 128: LDPC_ML_SOLVE(8) break;
                case 64: LDPC_ML_SOLVE(4) break;
                case 32: LDPC_ML_SOLVE(2) break;
                default: LDPC_ML_SOLVE(1) break;
            }
            prof_end(ctx, LDPC_AMD_PROF_ML_SOLVE, evs);
            LDPC_HIP_TRY(ctx, hipGetLastError());
            if (pi_nw > 0 && pi.verify) {
                // Verified fast path: a frame whose received symbols are not a codeword makes the residual system INCONSISTENT, and
                // then the bytes depend on which equations a solver uses.  The fast-path schedules therefore also evaluate the
                // equations they did not use; the solve kernel flags a frame with a non-zero one (rec[7], nfail), and these two
                // launches -- which leave after one load when nothing was flagged -- redo such frames in the reference's order.
                ma.mode = 3;
                hipLaunchKernelGGL(ldpc_ml_kernel, dim3(grid), dim3(ml_threads), (size_t)total, ctx->stream, ma);
                LDPC_HIP_TRY(ctx, hipGetLastError());
                sv.round = 2; sv.work = (int32_t *)ctx->mllist.p + 26;
                switch (solve_b) {
                    case 128: LDPC_ML_SOLVE(8) break;
                    case 64: LDPC_ML_SOLVE(4) break;
                    case 32: LDPC_ML_SOLVE(2) break;
                    default: LDPC_ML_SOLVE(1) break;
                }
                LDPC_HIP_TRY(ctx, hipGetLastError());
            }
#undef LDPC_ML_SOLVE
        }
        prof_end(ctx, LDPC_AMD_PROF_ML, ml_ev);
        if (ma.use_solve) {   // this call's arena demand -> pinned host word, read by a later call (no wait here)
            if (!ctx->ml_head_host) {
                if (hipHostMalloc((void **)&ctx->ml_head_host, 64, hipHostMallocDefault) != hipSuccess) ctx->ml_head_host = nullptr;
                else *ctx->ml_head_host = 0;
            }
            if (ctx->ml_head_host) {
                LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->ml_head_host, ma.ops_head, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
                ctx->ml_head_valid = true;
            }
        }
        return LDPC_AMD_OK;
    };

    PeelArgs pa{};
    pa.code = cd; pa.lds = L; pa.nframes = nf; pa.sym = d.sym; pa.erased = d.erased; pa.in_rows = d.in_rows;
    pa.max_sweeps = d.max_sweeps; pa.do_ml = d.do_ml; pa.out = d.out;
    pa.sweeps = d.sweeps; pa.residual = d.residual; pa.status = d.status; pa.residual_sys = d.residual_sys;
    pa.ml_list = (int32_t *)ctx->mllist.p; pa.ml_state = (uint8_t *)ctx->mlstate.p;

    // ---- exact time-stamp relaxation instead of the serial per-solve loop (peel_relax.inc) when the keys fit 16 bits: S = 1 decode
    //      (mode 0), the pattern-only runs (mode 1), the packet path's schedules (mode 2).  The encoder's one-sweep chain and
    //      everything else keep ldpc_peel_kernel.  Returns 1: launched, 0: not applicable, < 0: error.
    uint32_t *pa_lists = nullptr;  // packets: the steps' column lists in schedule order (peel_relax.inc mode 2 -> packet kernel set-up)
    bool pa_lists_on = false;
    uint32_t *pa_pull = nullptr;   // packets: per-step records of the paired-level schedules (peel_relax.inc mode 2 -> packet kernel)
    int pa_pairs = 0;
    int logM = 0;
    while ((1 << logM) < cd.mpad) logM++;
    const bool relax_ok = kn.peel_relax != 0 && d.erased != nullptr && d.in_rows == cd.n && cd.degpad <= 16 && cd.n <= 32767 &&
                          d.max_sweeps <= 62 && ((long)(d.max_sweeps + 1) << logM) <= 65535;
    auto relax_launch = [&](int mode) -> int {
        if (!relax_ok) return 0;
        RelaxArgs ra{};
        ra.n = cd.n; ra.k = cd.k; ra.m = cd.m; ra.mpad = cd.mpad; ra.logM = logM;
        ra.rx_off = cd.rx_off; ra.ell_logc = cd.rx_logc; ra.ell_coef = cd.ell_coef;
        ra.nframes = nf; ra.sym = d.sym; ra.erased = d.erased; ra.max_sweeps = d.max_sweeps; ra.do_ml = mode == 1 ? 0 : d.do_ml;
        ra.out = d.out; ra.sweeps = d.sweeps; ra.residual = d.residual; ra.status = d.status; ra.residual_sys = d.residual_sys;
        ra.ml_list = (int32_t *)ctx->mllist.p; ra.ml_state = (uint8_t *)ctx->mlstate.p; ra.err = ctx->dev_err_host;
        ra.sched_hdr = pa.sched_hdr; ra.sched_steps = pa.sched_steps; ra.sched_lvlend = pa.sched_lvlend; ra.sched_invc = pa.sched_invc;
        ra.big_list = pa.big_list; ra.tcap = pa.tcap;
        ra.sched_pull = pa_pull; ra.pairs = (mode == 2 && pa_pull) ? 1 : 0;
        ra.sched_lists = mode == 2 ? pa_lists : nullptr; ra.cell = cd.cell; ra.cdw = cd.maxcoldeg; ra.cdw_shift = cd.cdw_shift;
        // LDS plan: per frame its keys / values, solver and order lists (packets: + the level histogram); the code tables once per
        // workgroup (or from global memory).  Frames per CU = workgroups per CU x wavefronts per workgroup; the tables in LDS for short
        // batches (a single round is latency bound), in global memory when the batch is deep and that puts >= 1.3x more frames on a CU
        // (measured: (2040,1530) 59 -> 70 M frames/s, (4080,3060) 23 -> 27 M on 65536 frames; 4096 frames: 53 M with the LDS copy, 34 M without).
        auto plan = [&](bool gt_, int w_, RelaxLds &Lr) {
            int off = 0;
            Lr.off16 = off; if (!gt_) off += align_up(2 * cd.degpad * cd.mpad, 16);
            Lr.logc8 = off; if (!gt_ && mode == 0) off += align_up(cd.degpad * cd.mpad, 16);
            Lr.lg = off; off += 256;
            Lr.ex = off; off += 512;
            Lr.wave0 = off;
            int w = 0;
            Lr.key = w; w += align_up(2 * (cd.n + 1), 16);
            Lr.fire = w; w += align_up(2 * cd.mpad, 16);
            Lr.order = w; w += align_up(2 * cd.mpad, 16);
            // (mode 2: the per-sweep counts of the time sort live at the start of the level histogram, which is not in use yet then --
            // with them apart the (2040,1530) frame state is 9472 bytes and only 15 frames fit a CU: a 4096-frame batch needs 16)
            Lr.cnt = w; if (mode != 2) w += 256;
            Lr.hist = w; if (mode == 2) w += align_up(std::max(4 * (cd.m + 2), 256), 16);
            Lr.dep = w; if (mode == 2) w += align_up(cd.mpad, 16);
            Lr.sinv = w; if (mode == 2) w += align_up(cd.mpad, 16);
            Lr.wave_stride = w;
            Lr.total = off + w_ * w;
        };
        auto best_plan = [&](bool gt_, int &w_best, RelaxLds &L_best) -> int {
            int best_f = 0, best_score = 0;
            const int wcap = kn.peel_wpb > 0 ? std::min(16, kn.peel_wpb) : 16;
            for (int w_ = 1; w_ <= wcap; w_++) {
                RelaxLds t;
                plan(gt_, w_, t);
                if (t.total > kLdsMax) break;
                const int wgs = kLdsMax / t.total;
                const int frames = kn.peel_wpb > 0 ? w_ : std::min(32, wgs * w_);
                // without tables to stage, two (or more) smaller workgroups per CU beat one large one holding a frame more
                // ((4080,3060), 65536 frames: 6 x 2 frames 2.20 ms, 13 x 1 2.30 ms): a tenth of a bonus
                const int score = frames * ((gt_ && wgs >= 2 && kn.peel_wpb <= 0) ? 11 : 10);
                if (score >= best_score) { best_score = score; best_f = frames; w_best = w_; L_best = t; }
            }
            return best_f;
        };
        int w_l = 1, w_g = 1;
        RelaxLds L_l{}, L_g{};
        const int f_l = best_plan(false, w_l, L_l), f_g = best_plan(true, w_g, L_g);
        if (f_l <= 0 && f_g <= 0) return 0;
        const bool deep = nf >= (int64_t)3 * std::max(f_l, 1) * ctx->sm_count;
        const bool use_gt = kn.peel_gt == 1 || f_l == 0 || (kn.peel_gt != 0 && deep && f_g * 10 >= f_l * 13);
        const int wr = use_gt ? w_g : w_l;
        ra.lds = use_gt ? L_g : L_l;
        ctx->last_plan[0] = wr; ctx->last_plan[1] = use_gt ? f_g : f_l; ctx->last_plan[2] = use_gt ? 1 : 0;
        ctx->last_plan[3] = ra.lds.total; ctx->last_plan[4] = ra.lds.wave_stride;
        const dim3 g((unsigned)((nf + wr - 1) / wr)), b((unsigned)(wr * 64));
        char nm[96];
        snprintf(nm, sizeof(nm), "ldpc_peel_relax_kernel<%d, %s, %d>", cd.degpad, use_gt ? "true" : "false", mode);
        ctx->prof_names[LDPC_AMD_PROF_PEEL] = nm;
        hipEvent_t ev = prof_begin(ctx);
        bool launched = false;
#define LDPC_RELAX_CASE(D, G, MD)                                                                                        \
    if (!launched && cd.degpad == D && use_gt == G && mode == MD) {                                                      \
        auto kfn = ldpc_peel_relax_kernel<D, G, MD>;                                                                     \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));                                           \
        hipLaunchKernelGGL(kfn, g, b, (size_t)ra.lds.total, ctx->stream, ra);                                            \
        launched = true;                                                                                                 \
    }
#define LDPC_RELAX_DEG(D)                                                                                                \
    LDPC_RELAX_CASE(D, false, 0) LDPC_RELAX_CASE(D, true, 0) LDPC_RELAX_CASE(D, false, 1) LDPC_RELAX_CASE(D, true, 1)    \
    LDPC_RELAX_CASE(D, false, 2) LDPC_RELAX_CASE(D, true, 2)
        LDPC_RELAX_DEG(8) LDPC_RELAX_DEG(14) LDPC_RELAX_DEG(16)
#undef LDPC_RELAX_DEG
#undef LDPC_RELAX_CASE
        LDPC_HIP_TRY(ctx, hipGetLastError());
        prof_end(ctx, LDPC_AMD_PROF_PEEL, ev);
        return launched ? 1 : 0;
    };
    if (fused || d.flags_only) {
        const int rl = relax_launch(d.flags_only ? 1 : 0);
        if (rl < 0) return rl;
        if (rl > 0) {
            if (d.flags_only) return LDPC_AMD_OK;
            if (d.do_ml) {
                if ((rc = ml_prepare())) return rc;
                if ((rc = ml_front())) return rc;
                if ((rc = ml_back())) return rc;
            }
            return LDPC_AMD_OK;
        }
    }
    if (d.flags_only) {
        LDPC_HIP_TRY(ctx, launch_peel_t<false>(pa, wpb, ctx->stream));
        return LDPC_AMD_OK;
    }
    {
        char nm[96];
        snprintf(nm, sizeof(nm), "ldpc_peel_kernel<%d, %s, %s>", cd.degpad, fused ? "true" : "false", gt ? "true" : "false");
        ctx->prof_names[LDPC_AMD_PROF_PEEL] = nm;
    }
    if (fused) {
        hipEvent_t ev = prof_begin(ctx);
        if (gt) { LDPC_HIP_TRY(ctx, (launch_peel_t<true, true>(pa, wpb, ctx->stream))); }
        else { LDPC_HIP_TRY(ctx, launch_peel_t<true>(pa, wpb, ctx->stream)); }
        prof_end(ctx, LDPC_AMD_PROF_PEEL, ev);
    } else {
        const size_t hdr = (size_t)nf * 2 * 4, st = (size_t)nf * cd.m * 4, le = (size_t)nf * (cd.m + 1) * 2, iv = (size_t)nf * cd.m;
        const size_t o1 = (hdr + 255) & ~(size_t)255, o2 = (o1 + st + 255) & ~(size_t)255, o3 = (o2 + le + 255) & ~(size_t)255;
        if ((rc = scratch_reserve(ctx, ctx->sched, o3 + iv))) return rc;
        unsigned char *base = (unsigned char *)ctx->sched.p;
        pa.sched_hdr = (uint32_t *)base; pa.sched_steps = (uint32_t *)(base + o1); pa.sched_lvlend = (uint16_t *)(base + o2);
        pa.sched_invc = base + o3;
        if (use_scatter) {
            pa.tcap = plan.tcap;
            pa.big_list = plan.two_tier ? (int32_t *)ctx->biglist.p : nullptr;
        }
        hipEvent_t ev = nullptr;
        if (use_scatter && relax_ok && kn.scatter_pairs != 0) {   // paired levels: 16 bytes per step for the records the packet kernel reads at set-up
            if ((rc = scratch_reserve(ctx, ctx->schedpull, (size_t)nf * cd.m * 16))) return rc;
            pa_pull = (uint32_t *)ctx->schedpull.p;
        }
        if (use_scatter && relax_ok && kn.scatter_lists != 0) {
            if ((rc = scratch_reserve(ctx, ctx->schedlists, (size_t)nf * cd.m * cd.maxcoldeg * 4))) return rc;
            pa_lists = (uint32_t *)ctx->schedlists.p;
        }
        const int rl = relax_launch(2);   // the schedules by relaxation when its keys fit (else, and with PEEL_RELAX=0: the serial loop)
        if (rl < 0) return rl;
        pa_pairs = (rl > 0 && pa_pull) ? 1 : 0;
        pa_lists_on = rl > 0 && pa_lists;
        if (rl == 0) {
            ev = prof_begin(ctx);
            LDPC_HIP_TRY(ctx, launch_peel_t<false>(pa, wpb, ctx->stream));
            prof_end(ctx, LDPC_AMD_PROF_PEEL, ev);
        }

        if (d.do_ml) {
            if ((rc = ml_prepare())) return rc;
            ml_prepared = true;
            if (ml_overlap) {
                if ((rc = ml_front())) return rc;
                ml_front_done = true;
            }
        }
        if (use_scatter) {
            ScatterArgs sa{};
            sa.code = cd; sa.S = d.S; sa.nslices = plan.nslices; sa.nframes = nf; sa.sym = d.sym; sa.erased = d.erased; sa.out = d.out;
            sa.in_rows = cd.n; sa.static_sched = 0; sa.inplace = d.inplace;
            sa.dbg = kn.ml_dbg; sa.err = ctx->dev_err_host; sa.xl_setup = kn.scatter_xl;
            sa.sched_pull = pa_pull; sa.pairs = pa_pairs; sa.sched_lists = pa_lists_on ? pa_lists : nullptr;
            sa.sched_hdr = pa.sched_hdr; sa.sched_steps = pa.sched_steps; sa.sched_lvlend = pa.sched_lvlend;
            sa.sched_invc = pa.sched_invc;
            ev = prof_begin(ctx);
            if ((rc = launch_scatter(ctx, plan, sa, (int32_t *)ctx->biglist.p))) return rc;
            prof_end(ctx, LDPC_AMD_PROF_APPLY, ev);
        } else {
        ApplyArgs aa{};
        aa.code = cd; aa.S = d.S; aa.nframes = nf; aa.sym = d.sym; aa.erased = d.erased; aa.in_rows = d.in_rows; aa.out = d.out;
        aa.sched_hdr = pa.sched_hdr; aa.sched_steps = pa.sched_steps; aa.sched_lvlend = pa.sched_lvlend;
        const size_t lds = (size_t)cd.m * 4 + (size_t)(cd.m + 2) * 2;
        ctx->prof_names[LDPC_AMD_PROF_APPLY] = "ldpc_apply_kernel";
        ev = prof_begin(ctx);
        hipLaunchKernelGGL(ldpc_apply_kernel, dim3((unsigned)nf), dim3(512), lds, ctx->stream, aa);
        LDPC_HIP_TRY(ctx, hipGetLastError());
        prof_end(ctx, LDPC_AMD_PROF_APPLY, ev);
        }
    }

    if (d.do_ml) {
        if (!ml_prepared && (rc = ml_prepare())) return rc;
        if (!ml_front_done && (rc = ml_front())) return rc;
        if ((rc = ml_back())) return rc;
    }
    return LDPC_AMD_OK;
}

int launch_encode(ldpc_amd_ctx *ctx, const DevCode &cd, int S, int64_t nframes, const uint8_t *src, uint8_t *cw)
{
    if (nframes <= 0) return LDPC_AMD_OK;
    if (S == 1) {
        // all parity symbols erased: one in-order sweep of the peeling decoder IS the encoder
        // (row i has exactly one unknown, column k+i, once rows < i are done: triangle form).
        DecodeArgs d{};
        d.code = cd; d.S = 1; d.nframes = nframes; d.sym = src; d.erased = nullptr; d.in_rows = cd.k;
        d.max_sweeps = 1; d.do_ml = 0; d.out = cw;
        return launch_decode(ctx, d);
    }
    if (S % 16) return set_error(ctx, LDPC_AMD_EUNSUP, "S must be 1 or a multiple of 16 (got %d)", S);
    const Knobs &kn = ctx->knobs;
    if (!kn.apply_gather && cd.maxcoldeg <= 16) {
        // scatter form with the static schedule: source rows read once, all m accumulators in LDS
        ScatterPlan plan = plan_scatter(kn, cd, S);
        if (plan.lpr > 0) {
            plan.two_tier = false; plan.tcap = cd.m; plan.lds1 = plan.lds2;
            // The encoder needs all m accumulators (every check is a step), which at 256-byte row pieces fills the LDS with
            // ONE workgroup per CU.  With 128-byte pieces and the tables the static schedule does not need left out (check ->
            // slot table, received-row list) two workgroups fit, and one streams while the other runs its 27 levels.
            const int eb = kn.enc_b;
            // grouped static schedule (levels collapsed offline): used when the code has one and its lists fit the LDS plan below
            bool grouped = kn.enc_group != 0 && cd.encg_nlevels > 0;
            const int nlev_plan = grouped ? cd.encg_nlevels : cd.enc_nlevels;
            const int gent2 = (cd.encg_ent_n + 7) & ~7;
            const int need_g = 3 * gent2 + 2 * ((cd.m + 8) & ~7) + ((cd.m + 15) & ~15);
            if (plan.lpr == 16 && eb == 128 && (S % 128) == 0) {
                ScatterPlan q = plan;
                int off = 0;
                q.o_tgt = off; off += align_up(2 * cd.m, 16);
                q.o_invc = off; off += align_up(cd.m, 16);
                q.o_lvl = off; off += align_up(2 * (nlev_plan + 2), 16);   // (level / group offsets: as many as this schedule has)
                q.o_ctr = off; off += 288;
                q.o_mt = off;                // (the tables sit in front of the accumulators: added to the total below)
                q.o_soc = off; off += align_up(2 * cd.k, 16) >= cd.n ? align_up(2 * cd.k, 16) : align_up(cd.n, 16);   // row kinds (u8), then the source-row list (u16 [k])
                q.soc_bytes = align_up(2 * cd.k, 16) >= cd.n ? align_up(2 * cd.k, 16) : align_up(cd.n, 16);
                q.o_chk = off;                               // unused in static mode
                if (8192 + cd.m * 128 + off <= kLdsMax / 2) {
                    q.lpr = 8; q.nslices = S / 128; q.lds1 = q.lds2 = 8192 + cd.m * 128 + off;
                    q.two_tier = true;    // (only selects the 8-waves-per-SIMD instantiation; tcap = m: no frame goes to tier 2)
                    plan = q;
                }
            }
            // the compact lists of the parity symbols for the level phase: over the row tables (dead by then), which end the layout --
            // the allocation grows by what they need beyond those tables when that still fits
            int enc_clist = 0;
            const int soc_off = 8192 + align_up(cd.m * 16 * plan.lpr, 16) + plan.o_soc;
            const int limit = (plan.two_tier && plan.lpr == 8) ? kLdsMax / 2 : kLdsMax;
            if (grouped) {
                if (soc_off + need_g <= limit) plan.lds1 = plan.lds2 = std::max(plan.lds1, soc_off + need_g);
                else grouped = false;   // (the plan's level table was sized for the groups: still enough? no -- re-plan below)
            }
            if (!grouped && kn.enc_group != 0 && cd.encg_nlevels > 0 && nlev_plan != cd.enc_nlevels) {
                // the lists did not fit: the level table of the plan above was sized for the groups, the plain schedule has more levels
                const int saved = ctx->knobs.enc_group;
                ctx->knobs.enc_group = 0;
                const int rc_ = launch_encode(ctx, cd, S, nframes, src, cw);
                ctx->knobs.enc_group = saved;
                return rc_;
            }
            if (!grouped && kn.enc_clist && cd.enc_lst_n > 0) {
                const int need = 4 * cd.enc_lst_n + align_up(2 * (cd.m + 1), 16);
                if (soc_off + need <= limit) {
                    enc_clist = 1;
                    plan.lds1 = plan.lds2 = std::max(plan.lds1, soc_off + need);
                }
            }
            ScatterArgs sa{};
            sa.code = cd; sa.S = S; sa.nslices = plan.nslices; sa.nframes = nframes; sa.sym = src; sa.erased = nullptr; sa.out = cw;
            sa.in_rows = cd.k; sa.static_sched = 1; sa.enc_clist = enc_clist; sa.err = ctx->dev_err_host;
            sa.enc_group = grouped ? 1 : 0;
            ctx->last_enc_grouped = sa.enc_group;
            sa.enc_list = kn.enc_list;   // measured slower (4.62 vs 4.14 ms): off unless asked for
            return launch_scatter(ctx, plan, sa, nullptr);
        }
    }
    ApplyArgs aa{};
    aa.code = cd; aa.S = S; aa.nframes = nframes; aa.sym = src; aa.erased = nullptr; aa.in_rows = cd.k; aa.out = cw;
    const size_t lds = (size_t)cd.m * 4 + (size_t)(cd.m + 2) * 2;
    hipLaunchKernelGGL(ldpc_apply_kernel, dim3((unsigned)nframes), dim3(512), lds, ctx->stream, aa);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_rs_decode(ldpc_amd_ctx *ctx, const HostRs &rs, int S, int64_t nblocks, const uint16_t *idx,
                     const uint8_t *val, uint8_t *msg)
{
    if (nblocks <= 0) return LDPC_AMD_OK;
    const int R = rs.n - rs.k;
    int rc0;
    RsArgs a{};
    a.n = rs.n; a.k = rs.k; a.S = S; a.nblocks = nblocks; a.pt = rs.d_pt; a.recv_idx = idx; a.recv_val = val; a.msg = msg;
    int off = 0;
    a.lds_idx = off; off += align_up(2 * rs.k, 16);
    a.lds_pres = off; off += align_up(rs.k, 16);
    a.lds_ulist = off; off += align_up(rs.k, 16);
    a.lds_M = off; off += align_up(2 * R * R, 16);
    a.lds_b = off; off += align_up(R, 16);
    a.lds_lg = off; off += 256;
    a.lds_ex = off; off += 512;
    a.lds_misc = off; off += 16;
    if (off > kLdsMax) return set_error(ctx, LDPC_AMD_EUNSUP, "RS decode: LDS need %d bytes", off);
    const bool rs_generic = ctx->knobs.rs_generic != 0;
    // malformed blocks (positions not ascending / >= n) are decoded to zeros and counted (ldpc_amd_rs_bad_blocks)
    if ((rc0 = scratch_reserve(ctx, ctx->rsbad, 64))) return rc0;
    a.bad = (int *)ctx->rsbad.p;
    LDPC_HIP_TRY(ctx, hipMemsetAsync(a.bad, 0, sizeof(int), ctx->stream));
    if (S >= 256 && (S % 256) == 0 && R <= 32 && rs.k <= 256 && !rs_generic) {
        // packets: one wavefront per (block, slice), M^-1 in registers, rows streamed once, accumulators in registers
        // dwords per lane: 1 (a wavefront per 256-byte slice of the block, 32 accumulator registers, four wavefronts per SIMD) is
        // the fastest -- 21.5 ms per 60 k blocks of 1 KB packets against 22.2 (2 dwords) and 24.9 (4 dwords: 256 registers, two
        // wavefronts per SIMD cannot keep the vector ALU issuing); RS_VW = 2 / 4 are A/B knobs
        int vw = 1;
        if (ctx->knobs.rs_vw == 4 && (S % 1024) == 0) vw = 4;
        else if (ctx->knobs.rs_vw == 2 && (S % 512) == 0) vw = 2;
        const int nslices = S / (256 * vw);
        RsPkLds L{};
        int o = 0;
        L.pt = o; o += align_up(R * rs.k, 16);
        L.lg = o; o += 256;
        L.mtl = o; o += 8192;
        L.wave0 = o; L.wstride = 512 + 256 + 32;
        const int nw = 4;
        o += nw * L.wstride;
        const int64_t items = nblocks * nslices;
        const int grid = (int)std::min<int64_t>((items + nw - 1) / nw, (int64_t)ctx->sm_count * 96);
#define LDPC_RS_PK(VWV, WPSV)                                                                                \
    {                                                                                                        \
        auto kfn = rs_decode_packets_kernel<VWV, WPSV>;                                                      \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));                               \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * nw), (size_t)o, ctx->stream, a, L, nslices);          \
    }
        // (one dword per lane compiled for four wavefronts per SIMD: 19.9 ms; for six / eight -- 80 / 64 registers, the Gauss-Jordan
        // part spills -- 31.0 / 26.6 ms)
        if (vw == 4) LDPC_RS_PK(4, 2) else if (vw == 2) LDPC_RS_PK(2, 4) else LDPC_RS_PK(1, 4)
#undef LDPC_RS_PK
        LDPC_HIP_TRY(ctx, hipGetLastError());
        return LDPC_AMD_OK;
    }
    if (S == 1 && R <= 32 && rs.k <= 256 && !rs_generic) {
        // one wavefront per block, system in registers
        RsFastLds L{};
        int o = 0;
        L.lgp = o; o += align_up(R * rs.k, 16);
        L.lg16 = o; o += 512;
        L.ex = o; o += 1024;
        L.mtl = o; o += 8192;
        L.wave0 = o; L.wstride = 1024 + 64;
        const int nw = 4;
        o += nw * L.wstride;
        const int wgs = std::max(1, std::min(8, kLdsMax / o));
        const int grid = (int)std::min<int64_t>((nblocks + nw - 1) / nw, (int64_t)ctx->sm_count * wgs);
        hipLaunchKernelGGL(rs_decode_s1_kernel, dim3(grid), dim3(64 * nw), (size_t)o, ctx->stream, a, L);
        LDPC_HIP_TRY(ctx, hipGetLastError());
        return LDPC_AMD_OK;
    }
    const int threads = (S == 1) ? 64 : 256;
    int grid = (int)std::min<int64_t>(nblocks, (int64_t)ctx->sm_count * (S == 1 ? 16 : 4));
    if (S != 1) {
        int rc = scratch_reserve(ctx, ctx->rsws, (size_t)grid * R * S);
        if (rc) return rc;
        a.ws = (uint8_t *)ctx->rsws.p;
    }
    auto kfn = rs_decode_kernel;
    LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn)));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(threads), (size_t)off, ctx->stream, a);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_rs_encode(ldpc_amd_ctx *ctx, const HostRs &rs, int S, int64_t nblocks, const uint8_t *src, uint8_t *cw)
{
    if (nblocks <= 0) return LDPC_AMD_OK;
    const int grid = (int)std::min<int64_t>(nblocks, (int64_t)ctx->sm_count * 8);
    hipLaunchKernelGGL(rs_encode_kernel, dim3(grid), dim3(256), 0, ctx->stream, rs.n, rs.k, S, nblocks,
                       (const uint8_t *)rs.d_pt, src, cw);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_fpga_stats(ldpc_amd_ctx *ctx, const DevCode &code, int rs_n, int rs_k, int64_t nframes,
                      const uint8_t *erased0, const int32_t *residual_sys, unsigned long long *stats)
{
    if (nframes <= 0) return LDPC_AMD_OK;
    const int grid = (int)std::min<int64_t>((nframes + 3) / 4, 4096);
    hipLaunchKernelGGL(fpga_stats_kernel, dim3(grid), dim3(256), 0, ctx->stream, code.n, rs_n, rs_k, nframes, erased0,
                       residual_sys, stats);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_fpga_halves(ldpc_amd_ctx *ctx, const DevCode &code, int64_t nframes, const uint8_t *erased, int num_iter,
                       int32_t *residual_sys, int32_t *iterations)
{
    if (nframes <= 0) return LDPC_AMD_OK;
    const int wpb = 4;
    const int wave0 = align_up(code.degpad * code.mpad * 2, 16);
    const int wstride = 2 * align_up(code.n, 16);
    const size_t lds = (size_t)wave0 + (size_t)wpb * wstride;
    if (lds > (size_t)kLdsMax) return set_error(ctx, LDPC_AMD_EUNSUP, "FPGA perf decoder: LDS need %zu bytes", lds);
    const dim3 grid((unsigned)((nframes + wpb - 1) / wpb));
#define LDPC_FPGA_HALVES(D)                                                                                     \
    {                                                                                                          \
        auto kfn = fpga_halves_kernel<D>;                                                                      \
        LDPC_HIP_TRY(ctx, allow_max_lds(reinterpret_cast<const void *>(kfn))); \
        hipLaunchKernelGGL(kfn, grid, dim3(64 * wpb), lds, ctx->stream, code, nframes, erased, num_iter, residual_sys, iterations, wave0, wstride); \
    }
    if (code.degpad <= 8) LDPC_FPGA_HALVES(8)
    else if (code.degpad <= 14) LDPC_FPGA_HALVES(14)
    else if (code.degpad <= 16) LDPC_FPGA_HALVES(16)
    else LDPC_FPGA_HALVES(24)
#undef LDPC_FPGA_HALVES
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

#ifdef LDPC_AMD_STAMPS
extern "C" int ldpc_amd_debug_peel_stamps(ldpc_amd_ctx *ctx, unsigned long long *out32, int reset)
{
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    LDPC_HIP_TRY(ctx, hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_peel_stamps), 56 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[56] = {0};
        LDPC_HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_peel_stamps), z, sizeof(z)));
    }
    return LDPC_AMD_OK;
}
#endif

int launch_selftest(ldpc_amd_ctx *ctx)
{
    int rc;
    if ((rc = scratch_reserve(ctx, ctx->stage_i32, 64))) return rc;
    int *d = (int *)ctx->stage_i32.p;
    LDPC_HIP_TRY(ctx, hipMemsetAsync(d, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(selftest_kernel, dim3(256), dim3(256), 0, ctx->stream, d);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    int bad = -1;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(&bad, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != 0) return set_error(ctx, LDPC_AMD_EHIP, "GF(256) self-test: %d mismatches", bad);
    return LDPC_AMD_OK;
}

int launch_copy_probe(ldpc_amd_ctx *ctx, const uint8_t *src, uint8_t *dst, uint64_t bytes, int reps, double *ms)
{
    // Best single launch over `reps` repetitions of each of a few launch shapes (persistent grids of 1024- and 256-thread
    // workgroups): the figure is quoted as "what a linear read + write copy reaches on this box", so it has to be the best
    // the probe can do, not the average of one shape.
    hipEvent_t e0, e1;
    LDPC_HIP_TRY(ctx, hipEventCreate(&e0));
    LDPC_HIP_TRY(ctx, hipEventCreate(&e1));
    const uint64_t chunks = bytes / 16;
    static const int shapes[][2] = {{16384, 1024}, {2048, 1024}, {512, 1024}, {8192, 256}, {2048, 256}};
    float best = 0.f;
    for (const auto &sh : shapes) {
        const int grid = (int)std::min<uint64_t>((chunks + sh[1] - 1) / sh[1], (uint64_t)sh[0]);
        hipLaunchKernelGGL(copy_probe_kernel, dim3(grid), dim3(sh[1]), 0, ctx->stream, src, dst, chunks);  // warm-up
        for (int i = 0; i < reps; i++) {
            LDPC_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
            hipLaunchKernelGGL(copy_probe_kernel, dim3(grid), dim3(sh[1]), 0, ctx->stream, src, dst, chunks);
            LDPC_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
            LDPC_HIP_TRY(ctx, hipGetLastError());
            LDPC_HIP_TRY(ctx, hipEventSynchronize(e1));
            float t = 0.f;
            LDPC_HIP_TRY(ctx, hipEventElapsedTime(&t, e0, e1));
            if (best == 0.f || t < best) best = t;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = (double)best;
    return LDPC_AMD_OK;
}

int launch_synth_source(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int k, int S, uint8_t *d)
{
    const uint64_t base = (uint64_t)frame0 * (uint64_t)k * (uint64_t)S;
    const uint64_t count = (uint64_t)nframes * (uint64_t)k * (uint64_t)S;
    if (count == 0) return LDPC_AMD_OK;
    const uint64_t words = (count + 3) / 4;
    const int grid = (int)std::min<uint64_t>((words + 255) / 256, 16384);
    hipLaunchKernelGGL(synth_bytes_kernel, dim3(grid), dim3(256), 0, ctx->stream, seed, (uint32_t)LDPC_SYNTH_STREAM_SOURCE, base, count, d);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_synth_bursty(ldpc_amd_ctx *ctx, uint64_t seed, int64_t first, int64_t count, double alpha, double beta,
                        double bias, uint8_t *d)
{
    if (count <= first) return LDPC_AMD_OK;
    GeParams p{};
    p.seed = seed; p.count = (uint64_t)count; p.first = (uint64_t)first;
    const double transition = 0.1;  // Bursty_Error_Channel_Model_Generator.m:16
    p.ta = ldpc_synth_threshold(alpha); p.tb = ldpc_synth_threshold(beta);
    p.t10 = ldpc_synth_threshold(transition / bias); p.t01 = ldpc_synth_threshold(transition);
    const uint32_t nblocks = (uint32_t)((count + 256 * kGeSeg - 1) / (256 * kGeSeg));
    int rc = scratch_reserve(ctx, ctx->rsws, (size_t)nblocks * 5 + 64);
    if (rc) return rc;
    uint32_t *blockfn = (uint32_t *)ctx->rsws.p;
    uint8_t *entry = (uint8_t *)(blockfn + nblocks);
    hipLaunchKernelGGL(ge_block_fn_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, p, blockfn);
    hipLaunchKernelGGL(ge_scan_kernel, dim3(1), dim3(1), 0, ctx->stream, nblocks, (const uint32_t *)blockfn, entry);
    hipLaunchKernelGGL(ge_write_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, p, (const uint8_t *)entry, d);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_synth_fpga(ldpc_amd_ctx *ctx, uint32_t seed, uint64_t first, int64_t count, int per64, uint8_t *d)
{
    if (count <= 0) return LDPC_AMD_OK;
    const int grid = (int)std::min<int64_t>((count + 255) / 256, 16384);
    hipLaunchKernelGGL(synth_fpga_kernel, dim3(grid), dim3(256), 0, ctx->stream, seed, first, (uint64_t)count, per64, d);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

int launch_synth_erasures(ldpc_amd_ctx *ctx, uint64_t seed, uint32_t stream_id, int64_t first, int64_t count,
                          uint64_t thresh, uint8_t *d)
{
    if (count <= 0) return LDPC_AMD_OK;
    const int grid = (int)std::min<int64_t>((count + 255) / 256, 16384);
    hipLaunchKernelGGL(synth_bernoulli_kernel, dim3(grid), dim3(256), 0, ctx->stream, seed, stream_id, (uint64_t)first,
                       (uint64_t)count, thresh, d);
    LDPC_HIP_TRY(ctx, hipGetLastError());
    return LDPC_AMD_OK;
}

}  // namespace ldpc_amd
