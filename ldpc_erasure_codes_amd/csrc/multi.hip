// multi.hip -- the multi-device layer (include/ldpc_erasure_amd_multi.h): one context and one host thread per rank, frames
// sharded in contiguous blocks, one gather to rank 0's device at the end.  Everything below the group is the single-device
// C ABI of api.cpp; this file adds threads, the shard arithmetic, peer copies and a device-side compare for the C bench.
//
// Reference: frames are independent (Matlab/ErasureCodes_NonBinaryLDPCSim.m:218; OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:52);
// the reference host drives ONE device through init_opencl() / run() / cleanup() (OpenCL/host/src/main.cpp:266-310).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"
#include "../../include/ldpc_erasure_amd_multi.h"

struct ldpc_amd_group {
    std::vector<ldpc_amd_ctx *> ctx;
    std::vector<int> dev;
    std::vector<std::vector<int>> codes;   // group code handle -> the handle on every rank (a rank's context may hold other codes too)
    std::string err;
};

namespace {

int group_error(ldpc_amd_group *g, int code, const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    return code;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// fn(rank) on one host thread per rank (rank 0 on the calling thread); the first negative return code wins.  Exceptions do not
// cross the C ABI: a thread that cannot be started makes its rank run on the calling thread instead.
int for_each_rank(ldpc_amd_group *g, const std::function<int(int)> &fn)
{
    const int N = (int)g->ctx.size();
    std::vector<int> rc(N, 0);
    std::vector<std::thread> th;
    std::vector<int> inline_ranks;
    for (int r = 1; r < N; r++) {
        try {
            th.emplace_back([&, r] { rc[r] = fn(r); });
        } catch (...) {
            inline_ranks.push_back(r);
        }
    }
    rc[0] = fn(0);
    for (int r : inline_ranks) rc[r] = fn(r);
    for (std::thread &t : th) t.join();
    for (int r = 0; r < N; r++)
        if (rc[r] < 0) return group_error(g, rc[r], "rank %d (device %d): %s", r, g->dev[r], ldpc_amd_last_error(g->ctx[r]));
    return LDPC_AMD_OK;
}

__global__ void count_mismatch_kernel(const uint4 *a, const uint4 *b, uint64_t chunks, unsigned long long *bad)
{
    unsigned long long mine = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < chunks; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        mine += (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) ? 1ull : 0ull;
    }
    if (mine) atomicAdd(bad, mine);
}

}  // namespace

extern "C" {

void ldpc_amd_shard_frames(int64_t nframes, int nranks, int rank, int64_t *first, int64_t *count)
{
    if (nranks < 1) nranks = 1;
    if (nframes < 0) nframes = 0;
    const int64_t base = nframes / nranks, rem = nframes % nranks;
    if (count) *count = base + (rank < rem ? 1 : 0);
    if (first) *first = (int64_t)rank * base + std::min<int64_t>(rank, rem);
}

int ldpc_amd_group_create(int nranks, const int *devices, ldpc_amd_group **out)
{
    if (!out) return LDPC_AMD_EINVAL;
    *out = nullptr;
    if (nranks < 1 || nranks > 64) return LDPC_AMD_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LDPC_AMD_EHIP;
    ldpc_amd_group *g = new (std::nothrow) ldpc_amd_group();
    if (!g) return LDPC_AMD_ENOMEM;
    for (int r = 0; r < nranks; r++) {
        const int d = devices ? devices[r] : r % ndev;
        ldpc_amd_ctx *c = nullptr;
        const int rc = ldpc_amd_init(d, &c);
        if (rc != LDPC_AMD_OK) {
            for (ldpc_amd_ctx *p : g->ctx) ldpc_amd_cleanup(p);
            delete g;
            return rc;   // (the text is in ldpc_amd_last_error(NULL) of this thread)
        }
        g->ctx.push_back(c);
        g->dev.push_back(d);
    }
    // peer access towards rank 0's device for the final gather (a no-op between ranks that share a device)
    for (int r = 1; r < nranks; r++) {
        if (g->dev[r] == g->dev[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->dev[r], g->dev[0]) == hipSuccess && can) {
            (void)hipSetDevice(g->dev[r]);
            const hipError_t e = hipDeviceEnablePeerAccess(g->dev[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // peer copies still work, staged
        }
    }
    *out = g;
    return LDPC_AMD_OK;
}

void ldpc_amd_group_destroy(ldpc_amd_group *g)
{
    if (!g) return;
    for (ldpc_amd_ctx *c : g->ctx) ldpc_amd_cleanup(c);
    delete g;
}

int ldpc_amd_group_size(const ldpc_amd_group *g) { return g ? (int)g->ctx.size() : 0; }
int ldpc_amd_group_device(const ldpc_amd_group *g, int rank) { return (g && rank >= 0 && rank < (int)g->dev.size()) ? g->dev[rank] : -1; }
ldpc_amd_ctx *ldpc_amd_group_ctx(ldpc_amd_group *g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr; }
const char *ldpc_amd_group_last_error(const ldpc_amd_group *g) { return g ? g->err.c_str() : "null group"; }

int ldpc_amd_group_load_builtin_code(ldpc_amd_group *g, int code_ind, uint64_t coef_seed)
{
    if (!g) return LDPC_AMD_EINVAL;
    std::vector<int> hs;
    for (size_t r = 0; r < g->ctx.size(); r++) {
        const int h = ldpc_amd_load_builtin_code(g->ctx[r], code_ind, coef_seed);
        if (h < 0) return group_error(g, h, "rank %zu: %s", r, ldpc_amd_last_error(g->ctx[r]));
        hs.push_back(h);
    }
    g->codes.push_back(hs);
    return (int)g->codes.size() - 1;
}

int ldpc_amd_group_register_code(ldpc_amd_group *g, int n, int k, const uint32_t *row_ptr, const uint16_t *cols, const uint8_t *coefs)
{
    if (!g) return LDPC_AMD_EINVAL;
    std::vector<int> hs;
    for (size_t r = 0; r < g->ctx.size(); r++) {
        const int h = ldpc_amd_register_code(g->ctx[r], n, k, row_ptr, cols, coefs);
        if (h < 0) return group_error(g, h, "rank %zu: %s", r, ldpc_amd_last_error(g->ctx[r]));
        hs.push_back(h);
    }
    g->codes.push_back(hs);
    return (int)g->codes.size() - 1;
}

int ldpc_amd_group_decode_batch(ldpc_amd_group *g, int code, int S, int64_t nframes, const uint8_t *sym, const uint8_t *erased,
                                int max_sweeps, int do_ml, uint8_t *out, int32_t *sweeps, int32_t *residual, int32_t *status)
{
    if (!g) return LDPC_AMD_EINVAL;
    if (nframes < 0 || S < 1) return group_error(g, LDPC_AMD_EINVAL, "bad nframes/S");
    if (nframes == 0) return LDPC_AMD_OK;
    int n = 0, k = 0, nnz = 0;
    if (code < 0 || code >= (int)g->codes.size() || ldpc_amd_code_info(g->ctx[0], g->codes[code][0], &n, &k, &nnz) != LDPC_AMD_OK)
        return group_error(g, LDPC_AMD_ENOCODE, "unknown group code handle %d", code);
    if (!sym || !erased || !out) return group_error(g, LDPC_AMD_EINVAL, "sym/erased/out must not be null");
    const int N = (int)g->ctx.size();
    return for_each_rank(g, [&](int r) -> int {
        int64_t f0, cnt;
        ldpc_amd_shard_frames(nframes, N, r, &f0, &cnt);
        if (cnt == 0) return LDPC_AMD_OK;
        return ldpc_amd_decode_batch(g->ctx[r], g->codes[code][r], S, cnt, sym + (size_t)f0 * n * S, erased + (size_t)f0 * n, max_sweeps, do_ml,
                                     out + (size_t)f0 * n * S, sweeps ? sweeps + f0 : nullptr, residual ? residual + f0 : nullptr,
                                     status ? status + f0 : nullptr, 0);
    });
}

int ldpc_amd_group_decode_resident(ldpc_amd_group *g, int code, int S, int64_t nframes, const uint8_t *const *sym,
                                   const uint8_t *const *erased, int max_sweeps, int do_ml, uint8_t *const *out, int32_t *const *words,
                                   int32_t *gathered_words, uint8_t *gathered_out, double *decode_ms, double *gather_ms)
{
    if (!g) return LDPC_AMD_EINVAL;
    if (nframes < 0 || S < 1 || !sym || !erased || !out || !words) return group_error(g, LDPC_AMD_EINVAL, "bad arguments");
    int n = 0, k = 0, nnz = 0;
    if (code < 0 || code >= (int)g->codes.size() || ldpc_amd_code_info(g->ctx[0], g->codes[code][0], &n, &k, &nnz) != LDPC_AMD_OK)
        return group_error(g, LDPC_AMD_ENOCODE, "unknown group code handle %d", code);
    const int N = (int)g->ctx.size();
    const double t0 = now_ms();
    int rc = for_each_rank(g, [&](int r) -> int {
        int64_t f0, cnt;
        ldpc_amd_shard_frames(nframes, N, r, &f0, &cnt);
        if (cnt == 0) return LDPC_AMD_OK;
        if (!sym[r] || !erased[r] || !out[r] || !words[r]) return ldpc_amd::set_error(g->ctx[r], LDPC_AMD_EINVAL, "null device pointer for rank %d", r);
        int rcr = ldpc_amd_decode_batch(g->ctx[r], g->codes[code][r], S, cnt, sym[r], erased[r], max_sweeps, do_ml, out[r], words[r], words[r] + cnt,
                                        words[r] + 2 * cnt, LDPC_AMD_DEVICE_PTRS);
        if (rcr) return rcr;
        return ldpc_amd_synchronize(g->ctx[r]);
    });
    const double t1 = now_ms();
    if (decode_ms) *decode_ms = t1 - t0;
    if (rc) return rc;
    // ---- the one gather of the job: every rank pushes its shard to rank 0's device, on its own stream
    if (gathered_words || gathered_out) {
        rc = for_each_rank(g, [&](int r) -> int {
            int64_t f0, cnt;
            ldpc_amd_shard_frames(nframes, N, r, &f0, &cnt);
            if (cnt == 0) return LDPC_AMD_OK;
            ldpc_amd_ctx *c = g->ctx[r];
            LDPC_HIP_TRY(c, hipSetDevice(g->dev[r]));
            if (gathered_words)
                for (int w = 0; w < 3; w++)
                    LDPC_HIP_TRY(c, hipMemcpyPeerAsync(gathered_words + (size_t)w * nframes + f0, g->dev[0], words[r] + (size_t)w * cnt, g->dev[r],
                                                       sizeof(int32_t) * (size_t)cnt, c->stream));
            if (gathered_out)
                LDPC_HIP_TRY(c, hipMemcpyPeerAsync(gathered_out + (size_t)f0 * n * S, g->dev[0], out[r], g->dev[r], (size_t)cnt * n * S, c->stream));
            return ldpc_amd_synchronize(c);
        });
    }
    if (gather_ms) *gather_ms = now_ms() - t1;
    return rc;
}

int ldpc_amd_group_fpga_run(ldpc_amd_group *g, unsigned short nldpc, int seed, int PER_numerator_div_64, int code_ind, long numFrames,
                            short num_iter, int perf_tests_body, ldpc_amd_error_type *total)
{
    if (!g || !total) return LDPC_AMD_EINVAL;
    if (numFrames < 0) return group_error(g, LDPC_AMD_EINVAL, "numFrames < 0");
    const int N = (int)g->ctx.size();
    std::vector<ldpc_amd_error_type> part(N, ldpc_amd_error_type{0, 0});
    const int rc = for_each_rank(g, [&](int r) -> int {
        int64_t f0, cnt;
        ldpc_amd_shard_frames(numFrames, N, r, &f0, &cnt);
        ldpc_amd_ctx *c = g->ctx[r];
        int rcr = ldpc_amd_data_in_at(c, nullptr, nldpc, seed, PER_numerator_div_64, code_ind, (long)cnt, (long)f0);
        if (rcr) return rcr;
        rcr = perf_tests_body ? ldpc_amd_ldpc_erasure_decoder_perf_tests(c, num_iter, code_ind) : ldpc_amd_ldpc_erasure_decoder(c, num_iter, code_ind);
        if (rcr) return rcr;
        return ldpc_amd_data_out(c, nullptr, code_ind, (long)cnt, &part[r]);
    });
    if (rc) return rc;
    long long e0 = 0, e1 = 0;
    for (const ldpc_amd_error_type &p : part) { e0 += p.num_LDPC_errors; e1 += p.num_RS_errors; }
    total->num_LDPC_errors = (int)e0;   // (the reference's counters are ints too: ldpc_erasure_decoder_top.cl:46-49)
    total->num_RS_errors = (int)e1;
    return LDPC_AMD_OK;
}

int ldpc_amd_group_bench_resident(ldpc_amd_group *g, int code_ind, uint64_t coef_seed, int S, int64_t F, double per, int max_sweeps,
                                  int steps, double result[4])
{
    if (!g || !result) return LDPC_AMD_EINVAL;
    if (F < 1 || steps < 1 || S < 1) return group_error(g, LDPC_AMD_EINVAL, "bad frames / steps / S");
    const int N = (int)g->ctx.size();
    const int code = ldpc_amd_group_load_builtin_code(g, code_ind, coef_seed);
    if (code < 0) return code;
    int n = 0, k = 0, nnz = 0;
    ldpc_amd_code_info(g->ctx[0], g->codes[code][0], &n, &k, &nnz);
    struct Bufs { uint8_t *src = nullptr, *cw = nullptr, *era = nullptr, *out = nullptr; int32_t *words = nullptr; unsigned long long *bad = nullptr; };
    std::vector<Bufs> B(N);
    int32_t *gathered = nullptr;
    const int64_t total = F * N;
    auto free_all = [&]() {
        for (int r = 0; r < N; r++) {
            (void)hipSetDevice(g->dev[r]);
            for (void *p : {(void *)B[r].src, (void *)B[r].cw, (void *)B[r].era, (void *)B[r].out, (void *)B[r].words, (void *)B[r].bad})
                if (p) (void)hipFree(p);
        }
        if (gathered) { (void)hipSetDevice(g->dev[0]); (void)hipFree(gathered); }
    };
    int rc = for_each_rank(g, [&](int r) -> int {
        ldpc_amd_ctx *c = g->ctx[r];
        LDPC_HIP_TRY(c, hipSetDevice(g->dev[r]));
        LDPC_HIP_TRY(c, hipMalloc((void **)&B[r].src, (size_t)F * k * S));
        LDPC_HIP_TRY(c, hipMalloc((void **)&B[r].cw, (size_t)F * n * S));
        LDPC_HIP_TRY(c, hipMalloc((void **)&B[r].era, (size_t)F * n));
        LDPC_HIP_TRY(c, hipMalloc((void **)&B[r].out, (size_t)F * n * S));
        LDPC_HIP_TRY(c, hipMalloc((void **)&B[r].words, sizeof(int32_t) * 3 * (size_t)F));
        LDPC_HIP_TRY(c, hipMalloc((void **)&B[r].bad, 16));
        int rcr;
        // frame indices continue across the ranks: the group decodes frames [0, N F) of ONE synthetic stream
        if ((rcr = ldpc_amd_synth_source(c, 20261004ull + (uint64_t)code_ind, (int64_t)r * F, F, k, S, B[r].src))) return rcr;
        if ((rcr = ldpc_amd_synth_erasures_uniform(c, 20261005ull + (uint64_t)code_ind, (int64_t)r * F, F, n, per, B[r].era))) return rcr;
        if ((rcr = ldpc_amd_encode_batch(c, g->codes[code][r], S, F, B[r].src, B[r].cw, LDPC_AMD_DEVICE_PTRS))) return rcr;
        return ldpc_amd_synchronize(c);
    });
    if (!rc) {
        (void)hipSetDevice(g->dev[0]);
        if (hipMalloc((void **)&gathered, sizeof(int32_t) * 3 * (size_t)total) != hipSuccess) rc = group_error(g, LDPC_AMD_ENOMEM, "hipMalloc of the gather buffer failed");
    }
    std::vector<const uint8_t *> sym(N), era(N);
    std::vector<uint8_t *> out(N);
    std::vector<int32_t *> words(N);
    for (int r = 0; r < N; r++) { sym[r] = B[r].cw; era[r] = B[r].era; out[r] = B[r].out; words[r] = B[r].words; }
    // (the payload of an erased symbol is ignored by the decoder, ...Decoder.m:30-35: the codeword itself serves as the received frame)
    double dec = 0, gat = 0, dsum = 0;
    if (!rc) rc = ldpc_amd_group_decode_resident(g, code, S, total, sym.data(), era.data(), max_sweeps, 1, out.data(), words.data(), nullptr, nullptr, &dec, &gat);   // warm-up
    for (int s = 0; s < steps && !rc; s++) {
        const bool last = s == steps - 1;
        rc = ldpc_amd_group_decode_resident(g, code, S, total, sym.data(), era.data(), max_sweeps, 1, out.data(), words.data(), last ? gathered : nullptr,
                                            nullptr, &dec, &gat);
        dsum += dec;
    }
    // ---- verification on the devices: out == cw on every rank, status words all zero, gathered words == the ranks' own
    double ok = 1.0;
    if (!rc) {
        std::vector<unsigned long long> bad(N, 0);
        rc = for_each_rank(g, [&](int r) -> int {
            ldpc_amd_ctx *c = g->ctx[r];
            LDPC_HIP_TRY(c, hipSetDevice(g->dev[r]));
            LDPC_HIP_TRY(c, hipMemsetAsync(B[r].bad, 0, 16, c->stream));
            const uint64_t chunks = (uint64_t)F * n * S / 16;
            hipLaunchKernelGGL(count_mismatch_kernel, dim3(2048), dim3(256), 0, c->stream, (const uint4 *)B[r].out, (const uint4 *)B[r].cw, chunks, B[r].bad);
            LDPC_HIP_TRY(c, hipGetLastError());
            LDPC_HIP_TRY(c, hipMemcpyAsync(&bad[r], B[r].bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            std::vector<int32_t> mine(3 * (size_t)F), got(3 * (size_t)F);
            LDPC_HIP_TRY(c, hipMemcpyAsync(mine.data(), B[r].words, sizeof(int32_t) * 3 * (size_t)F, hipMemcpyDeviceToHost, c->stream));
            LDPC_HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int w = 0; w < 3; w++)
                LDPC_HIP_TRY(c, hipMemcpy(got.data() + (size_t)w * F, gathered + (size_t)w * total + (size_t)r * F, sizeof(int32_t) * (size_t)F, hipMemcpyDeviceToHost));
            if (mine != got) bad[r] += 1;
            for (int64_t f = 0; f < F; f++)
                if (mine[2 * (size_t)F + f] != 0) bad[r] += 1;   // status: every frame of this workload decodes by message passing
            return LDPC_AMD_OK;
        });
        for (unsigned long long b : bad)
            if (b) ok = 0.0;
    }
    free_all();
    if (rc) return rc;
    result[0] = (double)total * steps / (dsum * 1e-3);
    result[1] = dsum / steps;
    result[2] = gat;
    result[3] = ok;
    return LDPC_AMD_OK;
}

}  // extern "C"
