// panel_probe.hip -- measurement + reference check of the PANEL step of a blocked GF(256) factorisation (DESIGN.md section 9,
// "plan for the next round").  NOT part of the library: a stand-alone probe of what one wavefront needs per column when the
// 16-byte panel chunk of every row lives in its registers.
//
// The step it times is the reference's forward elimination (Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:85-115)
// restricted to one 16-column chunk: lane l of slot q holds the row at LOGICAL position 64 q + l, so "first non-zero row at or
// below the diagonal" (:87) is the first set bit of a ballot; the swap (:92-97) moves the displaced row into the pivot's lane;
// the rows below are updated with f / pivot (:107-114), and that multiplier is left in the byte the update zeroes (flip trick:
// the pivot byte of the masked pivot chunk is XORed with 1, so f + (f / piv)(piv + 1) = f / piv).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/panel_probe tools/panel_probe.hip
//   tools/bin/panel_probe [rows=400] [density_percent=5] [reps=200]
// prints: check against a plain host elimination, then ns per column with one single-wavefront workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct alignas(16) U4 { uint32_t x, y, z, w; };
struct MulTab { uint32_t t0, t1, t2, t3, t4; };

__device__ __forceinline__ uint32_t gfmul4(const MulTab &t, uint32_t x)
{
    const uint32_t s0 = x & 0x07070707u, s1 = (x >> 3) & 0x07070707u, s2 = (x >> 6) & 0x03030303u;
    return __builtin_amdgcn_perm(t.t1, t.t0, s0) ^ __builtin_amdgcn_perm(t.t3, t.t2, s1) ^ __builtin_amdgcn_perm(t.t4, t.t4, s2);
}
__device__ __forceinline__ uint32_t rdl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
// "write lane": one select per register (v_writelane_b32 would do, but this compiler exposes no builtin for it and its two
// scalar operands do not fit gfx9's one-SGPR rule in inline asm); the lane compare is shared by the compiler
__device__ __forceinline__ uint32_t wrl(uint32_t old, uint32_t v, int lane) { return ((int)threadIdx.x == lane) ? v : old; }

constexpr int NR = 8;   // 64-row slots: up to 512 logical rows

struct Args {
    const uint8_t *rows;   // [nrows][16] the panel chunk of every row, logical order = row index
    const uint8_t *lg;     // [256]
    const uint8_t *ex;     // [512]
    const uint32_t *mt;    // [256][8] multiply tables indexed by LOG of the coefficient
    uint8_t *out;          // [gridDim][nrows][16] chunks after the panel, by ORIGINAL row index
    int *piv;              // [gridDim][16] original row index of every pivot, -1 past a break
    int nrows, reps;
};

__global__ __launch_bounds__(64) void panel_kernel(Args a)
{
    __shared__ uint8_t lg[256];
    __shared__ uint8_t ex[512];
    __shared__ __attribute__((aligned(16))) uint32_t mt[256 * 8];
    const int lane = (int)threadIdx.x;
    for (int i = lane; i < 256; i += 64) lg[i] = a.lg[i];
    for (int i = lane; i < 512; i += 64) ex[i] = a.ex[i];
    for (int i = lane; i < 2048; i += 64) mt[i] = a.mt[i];
    __syncthreads();
    const int n = a.nrows;
    U4 P[NR];
    uint32_t sr[NR];
    uint32_t rec_pp = 0xFFFFFFFFu;
    U4 recP = {0, 0, 0, 0};
    int done = 16;                                      // columns that found a pivot
    for (int rep = 0; rep < a.reps; rep++) {
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const int l = q * 64 + lane;
            sr[q] = l < n ? (uint32_t)l : 0xFFFFu;
            P[q] = l < n ? reinterpret_cast<const U4 *>(a.rows)[l] : U4{0, 0, 0, 0};
        }
        rec_pp = 0xFFFFFFFFu;
        done = 16;
        // one column; D = the dword of the chunk that holds it (compile time: no register selects), sh = its shift in the dword
        auto column = [&](auto dtag, const int j, const int sh) -> bool {
            constexpr int D = decltype(dtag)::value;
            // ---- a. the column's byte of every row, who has a non-zero (per-lane flags; ballots only for the decisions)
            uint32_t lgb[NR];
            bool nzl[NR];
#pragma unroll
            for (int q = 0; q < NR; q++) {
                const uint32_t w = D == 0 ? P[q].x : (D == 1 ? P[q].y : (D == 2 ? P[q].z : P[q].w));
                const uint32_t bq = (w >> sh) & 0xFFu;
                nzl[q] = bq != 0u;
                lgb[q] = lg[bq];
            }
            const bool d0 = nzl[0];                       // (the row at position j may be the one the swap displaces)
            nzl[0] = nzl[0] && lane >= j;                 // logical positions < j hold the pivots of this panel
            // ---- b. first non-zero row at or below the diagonal (:87)
            int pq = -1;
            uint64_t nzp = 0;
#pragma unroll
            for (int q = NR - 1; q >= 0; q--) {
                const uint64_t m = __ballot(nzl[q]);
                if (m) { pq = q; nzp = m; }
            }
            if (pq < 0) return false;                     // :88-90 (the caller would stop the whole elimination here)
            const int pl = __ffsll((long long)nzp) - 1;
            const int p = pq * 64 + pl;
            const bool at_pl = lane == pl, at_j = lane == j;
            // ---- c. the pivot row's chunk, its identity and the log of the pivot; d. the swap (:92-97): the row at logical
            //      position j moves to the pivot's position (registers, identity, log and flag travel with it)
            U4 pv = {0, 0, 0, 0};
            uint32_t ppw = 0, lgp = 0;
            const U4 dv = {rdl(P[0].x, j), rdl(P[0].y, j), rdl(P[0].z, j), rdl(P[0].w, j)};
            const uint32_t dsr = rdl(sr[0], j), dlg = rdl(lgb[0], j);
            const bool db = (__ballot(d0) >> j) & 1ull;
#pragma unroll
            for (int q = 0; q < NR; q++)
                if (q == pq) {
                    pv.x = rdl(P[q].x, pl); pv.y = rdl(P[q].y, pl); pv.z = rdl(P[q].z, pl); pv.w = rdl(P[q].w, pl);
                    ppw = rdl(sr[q], pl); lgp = rdl(lgb[q], pl);
                    if (p != j) {
                        P[q].x = at_pl ? dv.x : P[q].x; P[q].y = at_pl ? dv.y : P[q].y; P[q].z = at_pl ? dv.z : P[q].z; P[q].w = at_pl ? dv.w : P[q].w;
                        sr[q] = at_pl ? dsr : sr[q]; lgb[q] = at_pl ? dlg : lgb[q];
                        nzl[q] = at_pl ? db : nzl[q];
                    }
                }
            nzl[0] = nzl[0] && lane > j;                  // position j is the pivot's now
            // ---- e. the record of the column (lane j keeps it)
            const uint32_t lpi = lgp ? 255u - lgp : 0u;   // log(1 / pivot)
            rec_pp = at_j ? ppw : rec_pp;
            recP.x = at_j ? pv.x : recP.x; recP.y = at_j ? pv.y : recP.y; recP.z = at_j ? pv.z : recP.z; recP.w = at_j ? pv.w : recP.w;
            // ---- f. rows below with a non-zero: row += (f / piv) * pivot chunk, bytes >= j only, byte j takes f / piv
            const uint32_t keep = 0xFFFFFFFFu << sh, one = 1u << sh;
            const uint32_t pd = ((D == 0 ? pv.x : (D == 1 ? pv.y : (D == 2 ? pv.z : pv.w))) & keep) ^ one;
#pragma unroll
            for (int q = 0; q < NR; q++) {
                if (!__ballot(nzl[q])) continue;          // (wave-uniform)
                if (nzl[q]) {
                    uint32_t lf = lgb[q] + lpi;
                    lf = lf >= 255u ? lf - 255u : lf;
                    const U4 tq = *reinterpret_cast<const U4 *>(mt + lf * 8);
                    MulTab t;
                    t.t0 = tq.x; t.t1 = tq.y; t.t2 = tq.z; t.t3 = tq.w; t.t4 = mt[lf * 8 + 4];
                    if (D == 0) { P[q].x ^= gfmul4(t, pd); P[q].y ^= gfmul4(t, pv.y); P[q].z ^= gfmul4(t, pv.z); P[q].w ^= gfmul4(t, pv.w); }
                    if (D == 1) { P[q].y ^= gfmul4(t, pd); P[q].z ^= gfmul4(t, pv.z); P[q].w ^= gfmul4(t, pv.w); }
                    if (D == 2) { P[q].z ^= gfmul4(t, pd); P[q].w ^= gfmul4(t, pv.w); }
                    if (D == 3) { P[q].w ^= gfmul4(t, pd); }
                }
            }
            return true;
        };
        for (int j = 0; j < 16; j++) {
            const int sh = (j & 3) * 8;
            bool ok;
            switch (j >> 2) {
                case 0: ok = column(std::integral_constant<int, 0>{}, j, sh); break;
                case 1: ok = column(std::integral_constant<int, 1>{}, j, sh); break;
                case 2: ok = column(std::integral_constant<int, 2>{}, j, sh); break;
                default: ok = column(std::integral_constant<int, 3>{}, j, sh); break;
            }
            if (!ok) { done = j; break; }
        }
    }
    // results of the last repetition: live rows from their slots, pivot rows from the records
    uint8_t *out = a.out + (size_t)blockIdx.x * n * 16;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int l = q * 64 + lane;
        if (l < n && l >= done && sr[q] != 0xFFFFu) reinterpret_cast<U4 *>(out)[sr[q]] = P[q];
    }
    if (lane < 16) {
        a.piv[blockIdx.x * 16 + lane] = (int)rec_pp;
        if (rec_pp != 0xFFFFFFFFu) reinterpret_cast<U4 *>(out)[rec_pp] = recP;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static uint8_t h_lg[256], h_ex[512];
static uint8_t gmul(uint8_t x, uint8_t y) { return (x && y) ? h_ex[h_lg[x] + h_lg[y]] : 0; }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 400, dens = argc > 2 ? atoi(argv[2]) : 5, reps = argc > 3 ? atoi(argv[3]) : 200;
    if (n < 16 || n > 64 * NR) { fprintf(stderr, "rows: 16 ... %d\n", 64 * NR); return 2; }
    {   // GF(2^8), primitive polynomial 0x171, alpha = 2
        uint32_t v = 1;
        for (int i = 0; i < 255; i++) { h_ex[i] = (uint8_t)v; h_lg[v] = (uint8_t)i; v <<= 1; if (v & 0x100) v ^= 0x171; }
        for (int i = 255; i < 512; i++) h_ex[i] = h_ex[i - 255];
        h_lg[0] = 0;
    }
    std::vector<uint32_t> h_mt(256 * 8, 0);
    for (int l = 0; l < 255; l++) {
        const uint8_t c = h_ex[l];
        for (int i = 0; i < 8; i++) {
            h_mt[l * 8 + (i >> 2)] |= (uint32_t)gmul(c, (uint8_t)i) << (8 * (i & 3));
            h_mt[l * 8 + 2 + (i >> 2)] |= (uint32_t)gmul(c, (uint8_t)(i << 3)) << (8 * (i & 3));
        }
        for (int i = 0; i < 4; i++) h_mt[l * 8 + 4] |= (uint32_t)gmul(c, (uint8_t)(i << 6)) << (8 * i);
    }
    std::vector<uint8_t> rows((size_t)n * 16);
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(n * 131 + dens);
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); };
    for (auto &b : rows) b = (rnd() % 100 < (uint32_t)dens) ? (uint8_t)(1 + rnd() % 255) : 0;
    // host reference: the same elimination on bytes, explicit logical order
    std::vector<uint8_t> R = rows;
    std::vector<int> perm(n), piv_ref(16, -1);
    for (int l = 0; l < n; l++) perm[l] = l;
    for (int j = 0; j < 16; j++) {
        int p = -1;
        for (int l = j; l < n && p < 0; l++) if (R[(size_t)perm[l] * 16 + j]) p = l;
        if (p < 0) break;
        std::swap(perm[j], perm[p]);
        const int pp = perm[j];
        piv_ref[j] = pp;
        const uint8_t pinv = h_ex[255 - h_lg[R[(size_t)pp * 16 + j]]];
        for (int l = j + 1; l < n; l++) {
            uint8_t *x = &R[(size_t)perm[l] * 16];
            if (!x[j]) continue;
            const uint8_t mult = gmul(x[j], pinv);
            for (int c = j; c < 16; c++) x[c] ^= gmul(mult, R[(size_t)pp * 16 + c]);
            x[j] = mult;
        }
    }
    int dev_cus = 0;
    CHECK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = dev_cus > 0 ? dev_cus : 256;
    Args a{};
    uint8_t *d_rows, *d_lg, *d_ex, *d_out;
    uint32_t *d_mt;
    int *d_piv;
    CHECK(hipMalloc(&d_rows, rows.size())); CHECK(hipMalloc(&d_lg, 256)); CHECK(hipMalloc(&d_ex, 512));
    CHECK(hipMalloc(&d_mt, h_mt.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)grid * n * 16)); CHECK(hipMalloc(&d_piv, (size_t)grid * 16 * 4));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lg, h_lg, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ex, h_ex, 512, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_mt, h_mt.data(), h_mt.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_out, rows.data(), rows.size(), hipMemcpyHostToDevice));   // rows the panel never touches keep their bytes
    for (int g = 1; g < grid; g++) CHECK(hipMemcpy(d_out + (size_t)g * n * 16, rows.data(), rows.size(), hipMemcpyHostToDevice));
    a.rows = d_rows; a.lg = d_lg; a.ex = d_ex; a.mt = d_mt; a.out = d_out; a.piv = d_piv; a.nrows = n; a.reps = 1;
    hipLaunchKernelGGL(panel_kernel, dim3(grid), dim3(64), 0, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<uint8_t> out((size_t)n * 16);
    std::vector<int> piv(16);
    CHECK(hipMemcpy(out.data(), d_out + (size_t)(grid - 1) * n * 16, out.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(piv.data(), d_piv + (size_t)(grid - 1) * 16, 64, hipMemcpyDeviceToHost));
    int bad = 0, npiv = 0;
    for (int j = 0; j < 16; j++) { bad += piv[j] != piv_ref[j]; npiv += piv_ref[j] >= 0; }
    for (size_t i = 0; i < out.size(); i++) bad += out[i] != R[i];
    printf("rows %d, density %d %%: %d pivots, %s (%d differences)\n", n, dens, npiv, bad ? "MISMATCH against the host elimination" : "equal to the host elimination", bad);
    // timing: one single-wavefront workgroup per CU, `reps` panels each
    a.reps = reps;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(panel_kernel, dim3(grid), dim3(64), 0, 0, a);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(panel_kernel, dim3(grid), dim3(64), 0, 0, a);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %.1f ns per column (one wavefront per CU, %d CUs, %d panels of %d columns each)\n", ms * 1e6 / ((double)reps * npiv), grid, reps, npiv);
    return bad ? 1 : 0;
}
