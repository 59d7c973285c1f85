// peel_lab.hip -- stand-alone harness for csrc/peel_relax.inc (the S = 1 decoder by time-stamp relaxation): compiles in seconds,
// checks the kernel frame by frame against a plain C restatement of the sequential sweeps (Matlab/My_LDPC_HybridML_NonBinary_
// Erasure_Decoder.m:21-59: which check solves which symbol in which sweep, `iterations`, residual count, decoded bytes) and times it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/peel_lab tools/peel_lab.hip
//   tools/bin/peel_lab <code.csr.bin> <coef_seed> <frames> <per|bursty> <max_sweeps> [wpb]
// Diagnostic tool, not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define LDPC_AMD_ST_MP_DONE 0
#define LDPC_AMD_ST_ML_SKIPPED 3
#define RELAX_LAB 1
#define RELAX_LOG(i) a.glog[i]
#define RELAX_EXP(i) a.gexp[i]
__constant__ unsigned char c_inv[256];   // (only the packet-schedule mode of the kernel file reads it; the lab runs MODE 0)

namespace {
struct alignas(16) U4 { uint32_t x, y, z, w; };
constexpr int kWave = 64;
constexpr int kMlClasses = 16, kMlHdr = 32;
constexpr int kDevErrRelaxCap = 2;
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
#include "../ldpc_erasure_codes_amd/csrc/peel_relax.inc"
}  // namespace

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(2); } } while (0)

static uint64_t rng_state = 1;
static uint64_t next_u64()
{
    uint64_t x = (rng_state += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static double uni() { return ((double)(next_u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: peel_lab code.csr.bin coef_seed frames per|bursty max_sweeps [wpb]\n"); return 1; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 1; }
    char magic[8];
    uint32_t hdr[4];
    if (fread(magic, 1, 8, fp) != 8 || fread(hdr, 4, 4, fp) != 4) return 1;
    const int n = (int)hdr[0], k = (int)hdr[1], m = (int)hdr[2], nnz = (int)hdr[3];
    std::vector<uint32_t> row_ptr(m + 1);
    std::vector<uint16_t> cols(nnz);
    if (fread(row_ptr.data(), 4, m + 1, fp) != (size_t)m + 1 || fread(cols.data(), 2, nnz, fp) != (size_t)nnz) return 1;
    fclose(fp);
    rng_state = strtoull(argv[2], nullptr, 10);
    const int F = atoi(argv[3]);
    const bool bursty = !strcmp(argv[4], "bursty"), parity = !strcmp(argv[4], "parity");   // parity: every parity symbol erased (one long chain)
    const double per = (bursty || parity) ? 0.0 : atof(argv[4]);
    const int max_sweeps = atoi(argv[5]);
    int wpb_arg = argc > 6 ? atoi(argv[6]) : 0;
    const int gt = argc > 7 ? atoi(argv[7]) : 0, U = argc > 8 ? atoi(argv[8]) : 1, inner_max = argc > 9 ? atoi(argv[9]) : 0, lab_stop = argc > 10 ? atoi(argv[10]) : 0;

    // GF(256), poly 0x171
    uint8_t lg[256] = {0}, ex[512];
    {
        int x = 1;
        for (int i = 0; i < 255; i++) { ex[i] = (uint8_t)x; lg[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x171; }
        for (int i = 255; i < 512; i++) ex[i] = ex[i - 255];
    }
    auto mul = [&](uint8_t a_, uint8_t b_) -> uint8_t { return (a_ && b_) ? ex[lg[a_] + lg[b_]] : 0; };
    auto inv = [&](uint8_t a_) -> uint8_t { return ex[255 - lg[a_]]; };
    std::vector<uint8_t> coefs(nnz);
    for (int e = 0; e < nnz; e++) coefs[e] = (uint8_t)(1 + next_u64() % 255);
    int maxdeg = 0;
    for (int r = 0; r < m; r++) maxdeg = std::max(maxdeg, (int)(row_ptr[r + 1] - row_ptr[r]));
    const int degpad = maxdeg <= 8 ? 8 : (maxdeg <= 14 ? 14 : 16);
    if (maxdeg > 16) { fprintf(stderr, "row degree %d > 16\n", maxdeg); return 1; }
    const int mpad = (m + 63) / 64 * 64;
    int logM = 0;
    while ((1 << logM) < mpad) logM++;
    if ((long)(max_sweeps + 1) << logM > 65535 || max_sweeps > 62) { fprintf(stderr, "keys do not fit 16 bits\n"); return 1; }
    std::vector<uint16_t> ell_col((size_t)degpad * mpad, (uint16_t)(n * 2));   // (byte offsets of the key words: DevCode::rx_off)
    std::vector<uint8_t> ell_logc((size_t)degpad * mpad, 0);
    for (int r = 0; r < m; r++)
        for (uint32_t e = row_ptr[r], t = 0; e < row_ptr[r + 1]; e++, t++) {
            ell_col[((size_t)(r >> 6) * degpad + t) * 64 + (r & 63)] = (uint16_t)(cols[e] * 2);     // chunk-major: [chunk][slot][check in chunk]
            ell_logc[((size_t)(r >> 6) * degpad + t) * 64 + (r & 63)] = lg[coefs[e]];
        }

    // frames: codewords (systematic encode, ErasureCodes_NonBinaryLDPCSim.m:173-182) with erasures; every 7th frame gets a corrupted
    // received symbol (not a codeword): the (check, symbol) pairs decide the bytes there
    std::vector<uint8_t> sym((size_t)F * n), era((size_t)F * n);
    int state = 0;
    for (int f = 0; f < F; f++) {
        uint8_t *y = &sym[(size_t)f * n];
        for (int j = 0; j < k; j++) y[j] = (uint8_t)next_u64();
        for (int r = 0; r < m; r++) {
            uint8_t s = 0;
            for (uint32_t e = row_ptr[r]; e + 1 < row_ptr[r + 1]; e++) s ^= mul(coefs[e], y[cols[e]]);
            y[k + r] = mul(s, inv(coefs[row_ptr[r + 1] - 1]));
        }
        for (int j = 0; j < n; j++) {
            bool e;
            if (bursty) {   // Gilbert-Elliott like bench.py's cfg 3 (alpha 0.13, beta 0.8, bias 10)
                e = uni() < (state ? 0.8 : 0.13);
                state = state ? (uni() < 0.1 ? 0 : 1) : (uni() < 0.01 ? 1 : 0);
            } else if (parity) {
                e = j >= k || uni() < 0.01;
            } else {
                e = uni() < per;
            }
            era[(size_t)f * n + j] = e ? 1 : 0;
            if (e) y[j] = 0x5A;
        }
        if (f % 7 == 3) y[next_u64() % n] ^= 0x3C;
    }

    // ---- plain C sequential sweeps
    std::vector<uint8_t> want((size_t)F * n);
    std::vector<int32_t> wsw(F), wres(F);
    std::vector<uint32_t> wfire((size_t)F * mpad, 0);
    const auto c0 = std::chrono::steady_clock::now();
    for (int f = 0; f < F; f++) {
        std::vector<int> yv(n);
        for (int j = 0; j < n; j++) yv[j] = era[(size_t)f * n + j] ? -1 : sym[(size_t)f * n + j];
        int it = 0, left = 0;
        for (int j = 0; j < n; j++) left += yv[j] < 0;
        bool stop = false;
        while (!stop && it < max_sweeps) {
            it++;
            for (int r = 0; r < m; r++) {
                int cnt = 0, last = -1;
                for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++)
                    if (yv[cols[e]] < 0) { cnt++; last = (int)e; }
                if (cnt == 1) {
                    uint8_t s = 0;
                    for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++)
                        if ((int)e != last) s ^= mul(coefs[e], (uint8_t)yv[cols[e]]);
                    yv[cols[last]] = mul(s, inv(coefs[last]));
                    wfire[(size_t)f * mpad + r] = (uint32_t)(((uint32_t)it << logM | (uint32_t)r) << 16) | (uint32_t)(cols[last] * 2);
                    left--;
                }
            }
            if (left == 0) stop = true;
        }
        wsw[f] = it; wres[f] = left;
        for (int j = 0; j < n; j++) want[(size_t)f * n + j] = yv[j] < 0 ? 0 : (uint8_t)yv[j];
    }
    const double cpu_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();

    // ---- device
    RelaxArgs a{};
    a.n = n; a.k = k; a.m = m; a.mpad = mpad; a.logM = logM;
    uint16_t *d_col; uint8_t *d_logc, *d_sym, *d_era, *d_out, *d_lg, *d_ex;
    int32_t *d_sw, *d_res, *d_st;
    uint32_t *d_fire;
    unsigned long long *d_ev;
    CK(hipMalloc(&d_col, ell_col.size() * 2)); CK(hipMalloc(&d_logc, ell_logc.size()));
    CK(hipMalloc(&d_sym, sym.size())); CK(hipMalloc(&d_era, era.size())); CK(hipMalloc(&d_out, sym.size()));
    CK(hipMalloc(&d_lg, 256)); CK(hipMalloc(&d_ex, 512));
    CK(hipMalloc(&d_sw, 4 * F)); CK(hipMalloc(&d_res, 4 * F)); CK(hipMalloc(&d_st, 4 * F));
    CK(hipMalloc(&d_fire, (size_t)F * mpad * 4)); CK(hipMalloc(&d_ev, 128));
    CK(hipMemcpy(d_col, ell_col.data(), ell_col.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_logc, ell_logc.data(), ell_logc.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sym, sym.data(), sym.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_era, era.data(), era.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lg, lg, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ex, ex, 512, hipMemcpyHostToDevice));
    CK(hipMemset(d_ev, 0, 128));
    a.rx_off = d_col; a.ell_logc = d_logc; a.nframes = F; a.sym = d_sym; a.erased = d_era; a.max_sweeps = max_sweeps; a.do_ml = 0;
    a.out = d_out; a.sweeps = d_sw; a.residual = d_res; a.status = d_st; a.glog = d_lg; a.gexp = d_ex; a.dbg_fire = d_fire; a.dbg_evals = d_ev; a.inner_max = inner_max;
    auto al = [](int v, int q) { return (v + q - 1) / q * q; };
    RelaxLds L{};
    int off = 0;
    L.off16 = off; if (!gt) off += al(2 * degpad * mpad, 16);
    L.logc8 = off; if (!gt) off += al(degpad * mpad, 16);
    L.lg = off; off += 256;
    L.ex = off; off += 512;
    L.wave0 = off;
    int w = 0;
    L.key = w; w += al(2 * (n + 1), 16);
    L.fire = w; w += al(2 * mpad, 16);
    L.order = w; w += al(2 * mpad, 16);
    L.cnt = w; w += 256;
    L.wave_stride = w;
    int wpb = std::min(16, (160 * 1024 - off) / w);
    if (wpb_arg > 0) wpb = std::min(wpb, wpb_arg);
    L.total = off + wpb * w;
    a.lds = L;
    printf("variant: tables %s, %d chunk(s) per step\n", gt ? "in global memory" : "in LDS", U);
    printf("code n=%d k=%d m=%d maxdeg=%d (bucket %d) logM=%d; LDS tables %d B + %d B per frame, %d frames per workgroup (%d B)\n", n, k, m, maxdeg,
           degpad, logM, off, w, wpb, L.total);
    auto launch = [&]() {
        const int grid = (F + wpb - 1) / wpb;
#define LAB_CASE(D, G, UU)                                                                                                   \
    if (degpad == D && gt == G && U == UU) {                                                                                 \
        CK(hipFuncSetAttribute((const void *)ldpc_peel_relax_kernel<D, G != 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((ldpc_peel_relax_kernel<D, G != 0, 0>), dim3(grid), dim3(wpb * 64), (size_t)L.total, 0, a);      \
    }
        LAB_CASE(8, 0, 1) LAB_CASE(8, 1, 1) LAB_CASE(14, 0, 1) LAB_CASE(14, 1, 1) LAB_CASE(16, 0, 1) LAB_CASE(16, 1, 1)
#undef LAB_CASE
        CK(hipGetLastError());
    };
    launch();
    CK(hipDeviceSynchronize());
    std::vector<uint8_t> got(sym.size());
    std::vector<int32_t> gsw(F), gres(F);
    std::vector<uint32_t> gfire((size_t)F * mpad);
    unsigned long long ev[16];
    CK(hipMemcpy(got.data(), d_out, got.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(gsw.data(), d_sw, 4 * F, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gres.data(), d_res, 4 * F, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gfire.data(), d_fire, gfire.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ev, d_ev, 128, hipMemcpyDeviceToHost));
    long bad_bytes = 0, bad_sw = 0, bad_res = 0, bad_fire = 0, steps = 0, resid_frames = 0;
    double sw_sum = 0;
    for (int f = 0; f < F; f++) {
        bad_sw += gsw[f] != wsw[f];
        bad_res += gres[f] != wres[f];
        resid_frames += wres[f] > 0;
        sw_sum += wsw[f];
        for (int j = 0; j < n; j++) bad_bytes += got[(size_t)f * n + j] != want[(size_t)f * n + j];
        for (int r = 0; r < mpad; r++) {
            bad_fire += gfire[(size_t)f * mpad + r] != wfire[(size_t)f * mpad + r];
            steps += wfire[(size_t)f * mpad + r] != 0;
        }
    }
    printf("%d frames, %.1f steps and %.2f sweeps per frame, %ld frames with a residual; plain C: %.1f us per frame\n", F, (double)steps / F,
           sw_sum / F, resid_frames, cpu_s / F * 1e6);
    printf("mismatches: bytes %ld, sweeps %ld, residual %ld, (check, symbol, visit) words %ld  => %s\n", bad_bytes, bad_sw, bad_res, bad_fire,
           (bad_bytes || bad_sw || bad_res || bad_fire) ? "FAIL" : "OK");
    printf("chunk evaluations per frame %.1f (chunks %d), apply iterations per frame %.1f (batches %.1f)\n", (double)ev[0] / F, mpad / 64,
           (double)ev[1] / F, (double)((steps / F + 63) / 64));
    {
        const char *names[7] = {"flags -> keys", "relaxation", "status words", "time-order sort", "symbols -> values", "apply", "output store"};
        unsigned long long tot = 0;
        for (int i = 0; i < 7; i++) tot += ev[2 + i];
        printf("phases (memtime ticks per frame, wavefront view):");
        for (int i = 0; i < 7; i++) printf("  %s %.0f (%.0f %%)", names[i], (double)ev[2 + i] / F, 100.0 * ev[2 + i] / (double)tot);
        printf("\n");
    }
    // timing
    a.dbg_fire = nullptr; a.dbg_evals = nullptr; a.lab_stop = lab_stop;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel: %.4f ms per %d frames = %.2f M frames/s\n", ms / reps, F, F / (ms / reps * 1e-3) / 1e6);
    return (bad_bytes || bad_sw || bad_res || bad_fire) ? 3 : 0;
}
