// api.cpp -- the C ABI of include/ldpc_erasure_amd.h (host side only; kernels live in kernels.hip).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

#include "internal.h"
#include "../../include/ldpc_erasure_amd_synth.h"

namespace ldpc_amd {

// ------------------------------------------------------------------------------------------------
// GF(256) host tables (poly 0x171)
// ------------------------------------------------------------------------------------------------
static GfHost build_gf_host()
{
    GfHost g;
    memset(&g, 0, sizeof(g));
    int x = 1;
    for (int i = 0; i < 255; i++) {
        g.exp[i] = (uint8_t)x;
        g.log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= kPrimPoly;
    }
    for (int i = 255; i < 512; i++) g.exp[i] = g.exp[i - 255];
    g.inv[0] = 0;
    for (int a = 1; a < 256; a++) g.inv[a] = g.exp[(255 - g.log[a]) % 255];
    return g;
}

const GfHost &gf_host()
{
    static const GfHost g = build_gf_host();   // C++11 magic static: built once, safe under concurrent first use
    return g;
}

void build_mul3_tables(uint32_t *tab)
{
    const GfHost &g = gf_host();
    for (int c = 0; c < 256; c++) {
        uint8_t b[32] = {0};
        for (int i = 0; i < 8; i++) b[i] = g.mul((uint8_t)c, (uint8_t)i);               // bits 0-2
        for (int i = 0; i < 8; i++) b[8 + i] = g.mul((uint8_t)c, (uint8_t)(i << 3));    // bits 3-5
        for (int i = 0; i < 4; i++) b[16 + i] = g.mul((uint8_t)c, (uint8_t)(i << 6));   // bits 6-7
        for (int w = 0; w < 8; w++)
            tab[c * 8 + w] = (uint32_t)b[4 * w] | ((uint32_t)b[4 * w + 1] << 8) | ((uint32_t)b[4 * w + 2] << 16) |
                             ((uint32_t)b[4 * w + 3] << 24);
    }
}

// ------------------------------------------------------------------------------------------------
// knobs: one table, read from the environment once per context (ldpc_amd_init) or set by ldpc_amd_configure
// ------------------------------------------------------------------------------------------------
namespace {
struct KnobDesc {
    const char *name;                                    // without the LDPC_AMD_ prefix
    bool (*set)(Knobs &, const char *value);             // value != nullptr
    void (*reset)(Knobs &);                              // back to the shipped default
    long long (*get)(const Knobs &);                     // current value (the two word-valued knobs: 0 / 1)
};
static bool parse_ll(const char *v, long long *out)
{
    char *end = nullptr;
    const long long x = strtoll(v, &end, 10);
    if (end == v || *end != '\0') return false;
    *out = x;
    return true;
}
#define LDPC_KNOB_INT(NAME, FIELD, COND)                                                                                          \
    {NAME, [](Knobs &k, const char *v) { long long x; if (!parse_ll(v, &x) || !(COND)) return false; k.FIELD = (decltype(k.FIELD))x; return true; }, \
     [](Knobs &k) { k.FIELD = Knobs{}.FIELD; }, [](const Knobs &k) { return (long long)k.FIELD; }}
static const KnobDesc kKnobs[] = {
    {"APPLY", [](Knobs &k, const char *v) { if (!strcmp(v, "gather")) k.apply_gather = 1; else if (!strcmp(v, "scatter")) k.apply_gather = 0; else return false; return true; },
     [](Knobs &k) { k.apply_gather = 0; }, [](const Knobs &k) { return (long long)k.apply_gather; }},
    LDPC_KNOB_INT("SCATTER_B", scatter_b, x == 256 || x == 128 || x == 64),
    LDPC_KNOB_INT("SCATTER_TIERS", scatter_tiers, x == 1 || x == 2),
    LDPC_KNOB_INT("SCATTER_NT", scatter_nt, x == 0 || x == 1),
    LDPC_KNOB_INT("SCATTER_XCD", scatter_xcd, x == 0 || x == 1),
    LDPC_KNOB_INT("SCATTER_DYN", scatter_dyn, x >= 0 && x <= 4),
    LDPC_KNOB_INT("SCATTER_R", scatter_r, x == 1 || x == 2 || x == 4),
    LDPC_KNOB_INT("SCATTER_R2", scatter_r2, x >= 2 && x <= 4),
    LDPC_KNOB_INT("SCATTER_T2B", scatter_t2b, x == 256 || x == 128),
    LDPC_KNOB_INT("SCATTER_XL", scatter_xl, x == 0 || x == 1),
    LDPC_KNOB_INT("SCATTER_PAIRS", scatter_pairs, x == 0 || x == 1),
    LDPC_KNOB_INT("SCATTER_LISTS", scatter_lists, x == 0 || x == 1),
    LDPC_KNOB_INT("PEEL_WPB", peel_wpb, x >= 0 && x <= 16),
    LDPC_KNOB_INT("PEEL_GT", peel_gt, x >= -1 && x <= 1),
    LDPC_KNOB_INT("PEEL_RELAX", peel_relax, x == 0 || x == 1),
    LDPC_KNOB_INT("SCATTER_T2P", scatter_t2p, x == 1 || x == 2 || x == 4 || x == 8),
    LDPC_KNOB_INT("SCATTER_T2P_FORCE", scatter_t2p_force, x == 0 || x == 1),
    LDPC_KNOB_INT("ENC_PERSIST", enc_persist, x == 0 || x == 1),
    LDPC_KNOB_INT("ML_SOLVE", ml_solve, x >= 0 && x <= 2),
    LDPC_KNOB_INT("ML_DBG", ml_dbg, x >= 0),
    LDPC_KNOB_INT("ML_SOLVE_B", ml_solve_b, x == 16 || x == 32 || x == 64 || x == 128),
    LDPC_KNOB_INT("ML_ARENA_WORDS", ml_arena_words, x == 0 || (x >= 1024 && x <= (1ll << 27))),
    LDPC_KNOB_INT("ML_THREADS", ml_threads, x == 0 || (x >= 256 && x <= 1024 && (x % 64) == 0)),
    LDPC_KNOB_INT("ML_PACK", ml_pack, x >= 1 && x <= 4),
    LDPC_KNOB_INT("ML_PI", ml_pi, x >= 0 && x <= 2),
    LDPC_KNOB_INT("ML_OVERLAP_PRIO", ml_overlap_prio, x >= 0 && x <= 1),
    LDPC_KNOB_INT("ML_OVERLAP", ml_overlap, x >= 0 && x <= 2),
    LDPC_KNOB_INT("ML_PI_ADAPTIVE", ml_pi_adaptive, x >= 0 && x <= 1),
    LDPC_KNOB_INT("ML_PI_IMAX", ml_pi_imax, x >= 0 && x <= 256),
    LDPC_KNOB_INT("ML_PI_WGS", ml_pi_wgs, x >= 0 && x <= 4096),
    LDPC_KNOB_INT("ML_PI_WAVES", ml_pi_waves, x >= 1 && x <= 4),
    LDPC_KNOB_INT("ML_PI_LDS", ml_pi_lds, x >= 32 && x <= 160),
    LDPC_KNOB_INT("ENC_CLIST", enc_clist, x == 0 || x == 1),
    LDPC_KNOB_INT("ENC_B", enc_b, x == 128 || x == 256),
    LDPC_KNOB_INT("ENC_LIST", enc_list, x == 0 || x == 1),
    LDPC_KNOB_INT("ENC_GROUP", enc_group, x == 0 || x == 1),
    LDPC_KNOB_INT("ENC_CAP", enc_cap, x >= 0 && x <= 64),
    {"RS", [](Knobs &k, const char *v) { if (!strcmp(v, "generic")) k.rs_generic = 1; else if (!strcmp(v, "fast")) k.rs_generic = 0; else return false; return true; },
     [](Knobs &k) { k.rs_generic = 0; }, [](const Knobs &k) { return (long long)k.rs_generic; }},
    LDPC_KNOB_INT("RS_VW", rs_vw, x == 0 || x == 1 || x == 2 || x == 4),
    LDPC_KNOB_INT("HOST_PIPELINE", host_pipeline, x == 0 || x == 1),
    LDPC_KNOB_INT("FPGA_CHUNK", fpga_chunk, x >= 1),
    LDPC_KNOB_INT("CHUNK_S1", chunk_s1, x >= 1024 && x <= (1ll << 20)),
};
#undef LDPC_KNOB_INT
}  // namespace

int knob_set(Knobs &k, const char *key, const char *value)
{
    if (!key) return -1;
    if (!strncasecmp(key, "LDPC_AMD_", 9)) key += 9;
    for (const KnobDesc &d : kKnobs) {
        if (strcasecmp(key, d.name)) continue;
        if (!value || !*value) { d.reset(k); return 0; }
        return d.set(k, value) ? 0 : -1;
    }
    return -1;
}

bool knobs_from_env(Knobs &k, std::string *rejected)
{
    char name[64];
    for (const KnobDesc &d : kKnobs) {
        snprintf(name, sizeof(name), "LDPC_AMD_%s", d.name);
        const char *v = getenv(name);   // the ONLY getenv of the library: once per context, inside ldpc_amd_init
        if (v && *v && !d.set(k, v)) {
            // the same rule as ldpc_amd_configure: a value the knob does not take is an error, never a silently kept default
            // (an A/B script with a typo would otherwise measure the default under the wrong label)
            if (rejected) *rejected = std::string(name) + "=" + v;
            return false;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// errors, scratch
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_init_error;

int set_error(ldpc_amd_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_init_error = buf;
    return code;
}

int check_device_error(ldpc_amd_ctx *ctx)
{
    if (!ctx->dev_err_host) return LDPC_AMD_OK;
    const int bits = __atomic_exchange_n(ctx->dev_err_host, 0, __ATOMIC_ACQ_REL);
    if (!bits) return LDPC_AMD_OK;
    return set_error(ctx, LDPC_AMD_EHIP, "a kernel reported a violated internal assumption (bits 0x%x: 1 = dynamic LDS not at LDS address 0, 2 = a loop of the relaxation kernel hit its safety cap); "
                     "the frames it handled were left undecoded", bits);
}

int scratch_reserve(ldpc_amd_ctx *ctx, Scratch &s, size_t bytes)
{
    if (bytes <= s.cap) return LDPC_AMD_OK;
    if (s.p) {
        // pending work may still use the old block
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && ctx->aux_in) e = hipStreamSynchronize(ctx->aux_in);
        if (e == hipSuccess && ctx->aux_out) e = hipStreamSynchronize(ctx->aux_out);
        if (e == hipSuccess && ctx->aux_ml) e = hipStreamSynchronize(ctx->aux_ml);
        if (e != hipSuccess) return set_error(ctx, LDPC_AMD_EHIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
        (void)hipFree(s.p);
        s.p = nullptr;
        s.cap = 0;
    }
    size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipMalloc(&s.p, want);
    if (e != hipSuccess) {
        s.p = nullptr;
        return set_error(ctx, LDPC_AMD_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    s.cap = want;
    return LDPC_AMD_OK;
}

static hipEvent_t prof_take(ldpc_amd_ctx *ctx)
{
    hipEvent_t e = nullptr;
    if (!ctx->prof_pool.empty()) {
        e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

hipEvent_t prof_begin(ldpc_amd_ctx *ctx, int level)
{
    if (ctx->profiling < level) return nullptr;
    hipEvent_t e = prof_take(ctx);
    if (e) (void)hipEventRecord(e, ctx->stream);
    return e;
}

void prof_end(ldpc_amd_ctx *ctx, int kind, hipEvent_t start)
{
    if (!start) return;
    hipEvent_t e = prof_take(ctx);
    if (!e) { ctx->prof_pool.push_back(start); return; }
    (void)hipEventRecord(e, ctx->stream);
    ctx->prof_events[kind].emplace_back(start, e);
}

static void scratch_free(Scratch &s)
{
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr;
    s.cap = 0;
}

// ------------------------------------------------------------------------------------------------
// built-in code ROM
// ------------------------------------------------------------------------------------------------
struct BuiltinCode {
    int code_ind, n, k, first_row, last_row, rs_n, rs_k, nnz;
    const uint32_t *row_ptr;
    const uint16_t *cols;
};
#include "builtin_codes_gen.inc"

static const BuiltinCode *find_builtin(int code_ind)
{
    for (const BuiltinCode &b : kBuiltinCodes)
        if (b.code_ind == code_ind) return &b;
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// code registration: CSR -> device tables
// ------------------------------------------------------------------------------------------------
template <typename T>
static int upload(ldpc_amd_ctx *ctx, HostCode *hc, const std::vector<T> &v, const T **dst)
{
    void *p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return set_error(ctx, LDPC_AMD_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    hc->allocs.push_back(p);
    if (!v.empty()) {
        e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return set_error(ctx, LDPC_AMD_EHIP, "hipMemcpy: %s", hipGetErrorString(e));
    }
    *dst = (const T *)p;
    return LDPC_AMD_OK;
}

static void free_code(HostCode *hc)
{
    if (!hc) return;
    for (void *p : hc->allocs) (void)hipFree(p);
    delete hc;
}

static int register_code(ldpc_amd_ctx *ctx, int n, int k, const uint32_t *row_ptr, const uint16_t *cols,
                         const uint8_t *coefs)
{
    if (!row_ptr || !cols) return set_error(ctx, LDPC_AMD_EINVAL, "null code arrays");
    if (n <= 1 || k < 1 || k >= n || n > 65534) return set_error(ctx, LDPC_AMD_EINVAL, "bad (n,k) = (%d,%d)", n, k);
    const int m = n - k;
    if (row_ptr[0] != 0) return set_error(ctx, LDPC_AMD_EINVAL, "row_ptr[0] != 0");
    const GfHost &gf = gf_host();
    HostCode *hc = new (std::nothrow) HostCode();
    if (!hc) return set_error(ctx, LDPC_AMD_ENOMEM, "out of host memory");
    hc->n = n; hc->k = k; hc->m = m; hc->nnz = (int)row_ptr[m];
    hc->row_ptr.assign(row_ptr, row_ptr + m + 1);
    hc->cols.assign(cols, cols + hc->nnz);
    if (coefs) hc->coefs.assign(coefs, coefs + hc->nnz);
    else hc->coefs.assign(hc->nnz, 1);
    int maxdeg = 0;
    for (int r = 0; r < m; r++) {
        if (row_ptr[r + 1] < row_ptr[r]) { delete hc; return set_error(ctx, LDPC_AMD_EINVAL, "row_ptr not monotone at row %d", r); }
        const int d = (int)(row_ptr[r + 1] - row_ptr[r]);
        maxdeg = std::max(maxdeg, d);
        int prev = -1;
        for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
            if ((int)cols[e] <= prev || cols[e] >= n) { delete hc; return set_error(ctx, LDPC_AMD_EINVAL, "row %d: columns must be ascending and < n", r); }
            if (hc->coefs[e] == 0) { delete hc; return set_error(ctx, LDPC_AMD_EINVAL, "row %d: zero coefficient", r); }
            prev = cols[e];
        }
    }
    if (maxdeg > 24) { delete hc; return set_error(ctx, LDPC_AMD_EUNSUP, "row degree %d > 24 not supported", maxdeg); }
    hc->maxdeg = maxdeg;
    const int degpad = maxdeg <= 8 ? 8 : (maxdeg <= 14 ? 14 : (maxdeg <= 16 ? 16 : 24));  // 14: the (2040,1530) rows (13/14 entries)
    const int mpad = (m + 63) / 64 * 64;

    std::vector<uint32_t> edges(hc->nnz);
    for (int e = 0; e < hc->nnz; e++)
        edges[e] = (uint32_t)cols[e] | ((uint32_t)hc->coefs[e] << 16) | ((uint32_t)gf.log[hc->coefs[e]] << 24);
    std::vector<uint16_t> ell_col((size_t)degpad * mpad, 0xFFFF);
    std::vector<uint8_t> ell_logc((size_t)degpad * mpad, 0), ell_coef((size_t)degpad * mpad, 0);
    std::vector<uint32_t> ell_pk((size_t)degpad * mpad, 0xFFFFu);
    std::vector<uint16_t> rx_off((size_t)degpad * mpad, (uint16_t)(2 * n));   // (n <= 32767 for the relaxation kernel: checked at launch)
    std::vector<uint8_t> rx_logc((size_t)degpad * mpad, 0);
    for (int r = 0; r < m; r++)
        for (uint32_t e = row_ptr[r], t = 0; e < row_ptr[r + 1]; e++, t++) {
            ell_col[(size_t)t * mpad + r] = cols[e];
            ell_logc[(size_t)t * mpad + r] = gf.log[hc->coefs[e]];
            ell_coef[(size_t)t * mpad + r] = hc->coefs[e];
            ell_pk[(size_t)t * mpad + r] = (uint32_t)cols[e] | ((uint32_t)gf.log[hc->coefs[e]] << 16);
            rx_off[((size_t)(r >> 6) * degpad + t) * 64 + (r & 63)] = (uint16_t)(2 * cols[e]);
            rx_logc[((size_t)(r >> 6) * degpad + t) * 64 + (r & 63)] = gf.log[hc->coefs[e]];
        }

    // static encode schedule: row i solves column k+i (triangle form) once the parity symbols among its other
    // neighbours are known.  Only valid when the code is in triangle form; otherwise encode is refused.
    std::vector<uint32_t> enc_steps;
    std::vector<uint16_t> enc_lvlend;
    std::vector<uint8_t> enc_invc(1, 1);
    int enc_nlevels = 0;
    {
        bool triangle = true;
        std::vector<int> lvl(m, 0);
        for (int r = 0; r < m && triangle; r++) {
            if (row_ptr[r + 1] == row_ptr[r] || cols[row_ptr[r + 1] - 1] != k + r) { triangle = false; break; }
            int L = 0;
            for (uint32_t e = row_ptr[r]; e + 1 < row_ptr[r + 1]; e++)
                if (cols[e] >= k) L = std::max(L, lvl[cols[e] - k]);
            lvl[r] = L + 1;
            enc_nlevels = std::max(enc_nlevels, L + 1);
        }
        if (triangle) {
            enc_lvlend.assign(enc_nlevels + 1, 0);
            for (int r = 0; r < m; r++) enc_lvlend[lvl[r]]++;
            std::vector<uint32_t> start(enc_nlevels + 2, 0);
            for (int L = 1; L <= enc_nlevels; L++) start[L + 1] = start[L] + enc_lvlend[L];
            enc_steps.assign(m, 0);
            std::vector<uint32_t> fill(start);
            for (int r = 0; r < m; r++) enc_steps[fill[lvl[r]]++] = (uint32_t)r | ((uint32_t)(k + r) << 16);
            enc_lvlend[0] = 0;
            for (int L = 1; L <= enc_nlevels; L++) enc_lvlend[L] = (uint16_t)start[L + 1];
            enc_invc.assign(m, 1);
            for (int s_ = 0; s_ < m; s_++) {
                const int r = (int)(enc_steps[s_] & 0xFFFFu);
                enc_invc[s_] = gf.inv[hc->coefs[row_ptr[r + 1] - 1]];   // the diagonal entry is the row's last
            }
        } else {
            enc_nlevels = 0;
            enc_steps.assign(1, 0);
            enc_lvlend.assign(1, 0);
        }
    }

    int maxcoldeg = 0, cdw_shift = 0;
    std::vector<uint32_t> cell;  // column lists: [n][maxcoldeg] check | coef << 16, 0xFFFFFFFF = none
    {
        std::vector<int> cdeg(n, 0);
        for (int e = 0; e < hc->nnz; e++) maxcoldeg = std::max(maxcoldeg, ++cdeg[cols[e]]);
        while ((1 << cdw_shift) < maxcoldeg) cdw_shift++;
        cell.assign((size_t)n << cdw_shift, 0xFFFFFFFFu);
        std::fill(cdeg.begin(), cdeg.end(), 0);
        for (int r = 0; r < m; r++)
            for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++)
                cell[((size_t)cols[e] << cdw_shift) + cdeg[cols[e]]++] = (uint32_t)r | ((uint32_t)hc->coefs[e] << 16);
    }
    // static per-symbol lists of the encoder (every check is used; slot = position in the level-sorted schedule;
    // the check whose target is the symbol itself is left out), same layout as the decoder's per-frame lists
    std::vector<uint32_t> enc_src((size_t)n << cdw_shift, 0xFFFFFFFFu);
    if (enc_nlevels > 0) {
        std::vector<uint32_t> slot_of_row(m, 0);
        for (int s_ = 0; s_ < m; s_++) slot_of_row[enc_steps[s_] & 0xFFFFu] = (uint32_t)s_;
        std::vector<int> fillc(n, 0);
        for (int r = 0; r < m; r++)
            for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
                const int j = cols[e];
                if (j == k + r) continue;
                enc_src[((size_t)j << cdw_shift) + fillc[j]++] = ((uint32_t)slot_of_row[r] * 128u) | ((uint32_t)hc->coefs[e] << 24);
            }
    }
    // ... and the lists of the PARITY symbols once more, compact and in schedule order (step s solves symbol k + row(s)): the
    // encoder's level phase reads them from LDS -- 1023 words + 511 offsets for the (2040,1530) code -- instead of one padded
    // list per level from global memory behind its own row stores
    std::vector<uint32_t> enc_lst;
    std::vector<uint16_t> enc_lst_off(m + 1, 0);
    if (enc_nlevels > 0) {
        for (int s_ = 0; s_ < m; s_++) {
            const int t = (int)(enc_steps[s_] >> 16);
            enc_lst_off[s_] = (uint16_t)enc_lst.size();
            for (int i = 0; i < (1 << cdw_shift); i++) {
                const uint32_t w = enc_src[((size_t)t << cdw_shift) + i];
                if (w == 0xFFFFFFFFu) break;
                enc_lst.push_back(w);
            }
        }
        enc_lst_off[m] = (uint16_t)enc_lst.size();
        if (enc_lst.size() >= 0xFFFFu) { enc_lst.clear(); }   // (offsets are 16-bit: such a code keeps the global lists)
    }
    const int enc_lst_n = (int)enc_lst.size();
    if (enc_lst.empty()) enc_lst.assign(1, 0xFFFFFFFFu);
    // ---- grouped static schedule (DevCode::encg_*): levels collapsed offline.  Rows are in topological order (triangle form: the
    // parity inputs of row r are columns k + j with j < r).  Greedy: row r joins the group of its latest parity input unless it
    // would then have to pull more than `cap` accumulators (its in-group inputs and theirs, transitively); else it opens the next.
    std::vector<uint32_t> encg_steps(1, 0), encg_src(1, 0xFFFFFFFFu), encg_ent(1, 0xFFFFFFFFu);
    std::vector<uint16_t> encg_lvlend(1, 0), encg_ent_off(1, 0);
    std::vector<uint8_t> encg_invc(1, 1), encg_npull(1, 0);
    int encg_nlevels = 0, encg_ent_n = 0, encg_pulls = 0, encg_maxpull = 0;
    const int cap = ctx->knobs.enc_cap;
    if (enc_nlevels > 1 && cap > 0) {
        std::vector<int> grp(m, 0);
        std::vector<std::vector<std::pair<int, uint8_t>>> anc(m);   // (ancestor row, composite coefficient c_ra), rows of r's group
        std::vector<uint8_t> acc_c(m, 0);
        std::vector<int> touched;
        for (int r = 0; r < m; r++) {
            int gm = -1;
            for (uint32_t e = row_ptr[r]; e + 1 < row_ptr[r + 1]; e++)
                if (cols[e] >= k) gm = std::max(gm, grp[cols[e] - k]);
            if (gm < 0) { grp[r] = 0; continue; }
            touched.clear();
            for (uint32_t e = row_ptr[r]; e + 1 < row_ptr[r + 1]; e++) {
                if (cols[e] < k) continue;
                const int j = cols[e] - k;
                if (grp[j] != gm) continue;
                // val_j = inv_j * (acc_j ^ sum_a c_ja acc_a) enters row r with h_rj
                const uint8_t f = gf.mul(hc->coefs[e], gf.inv[hc->coefs[row_ptr[j + 1] - 1]]);
                auto add = [&](int a_, uint8_t c_) {
                    if (!acc_c[a_]) touched.push_back(a_);   // (a coefficient that cancels to 0 and comes back is listed twice: deduplicated below)
                    acc_c[a_] ^= c_;
                };
                add(j, f);
                for (const auto &pa : anc[j]) add(pa.first, gf.mul(f, pa.second));
            }
            std::sort(touched.begin(), touched.end());
            touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
            if ((int)touched.size() > cap) {
                grp[r] = gm + 1;
            } else {
                grp[r] = gm;
                for (int a_ : touched)
                    if (acc_c[a_]) anc[r].emplace_back(a_, acc_c[a_]);
            }
            for (int a_ : touched) acc_c[a_] = 0;
        }
        encg_nlevels = 1 + *std::max_element(grp.begin(), grp.end());
        // steps by group; inside a group the steps with the longest pull lists first (the rounds of a group even out)
        std::vector<int> order(m);
        for (int r = 0; r < m; r++) order[r] = r;
        std::stable_sort(order.begin(), order.end(), [&](int a_, int b_) {
            if (grp[a_] != grp[b_]) return grp[a_] < grp[b_];
            return anc[a_].size() > anc[b_].size();
        });
        encg_steps.assign(m, 0); encg_invc.assign(m, 1); encg_lvlend.assign(encg_nlevels + 1, 0);
        std::vector<uint32_t> slot_of_row(m, 0);
        for (int s_ = 0; s_ < m; s_++) {
            const int r = order[s_];
            slot_of_row[r] = (uint32_t)s_;
            encg_steps[s_] = (uint32_t)r | ((uint32_t)(k + r) << 16);
            encg_invc[s_] = gf.inv[hc->coefs[row_ptr[r + 1] - 1]];
            encg_lvlend[grp[r] + 1] = (uint16_t)(s_ + 1);
        }
        for (int L = 1; L <= encg_nlevels; L++) encg_lvlend[L] = std::max(encg_lvlend[L], encg_lvlend[L - 1]);
        encg_src.assign((size_t)n << cdw_shift, 0xFFFFFFFFu);
        std::vector<int> fillc(n, 0);
        for (int r = 0; r < m; r++)
            for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
                const int j = cols[e];
                if (j == k + r) continue;
                if (j >= k && grp[j - k] == grp[r]) continue;   // pulled, not scattered
                encg_src[((size_t)j << cdw_shift) + fillc[j]++] = (slot_of_row[r] * 128u) | ((uint32_t)hc->coefs[e] << 24);
            }
        encg_ent.clear(); encg_ent_off.assign(m + 1, 0); encg_npull.assign(m, 0);
        for (int s_ = 0; s_ < m; s_++) {
            const int r = order[s_];
            encg_ent_off[s_] = (uint16_t)encg_ent.size();
            encg_npull[s_] = (uint8_t)anc[r].size();
            encg_pulls += (int)anc[r].size();
            encg_maxpull = std::max(encg_maxpull, (int)anc[r].size());
            for (const auto &pa : anc[r]) encg_ent.push_back((slot_of_row[pa.first] * 128u) | ((uint32_t)pa.second << 24));
            for (int i = 0; i < (1 << cdw_shift); i++) {
                const uint32_t w = encg_src[((size_t)(k + r) << cdw_shift) + i];
                if (w == 0xFFFFFFFFu) break;
                encg_ent.push_back(w);
            }
            if (encg_ent.size() >= 0xFFFFu) break;
        }
        if (encg_ent.size() >= 0xFFFFu) {   // 16-bit offsets: such a code keeps the level-by-level schedule
            encg_nlevels = 0;
        } else {
            encg_ent_off[m] = (uint16_t)encg_ent.size();
            encg_ent_n = (int)encg_ent.size();
            // Self-check on bytes (one pseudo-random source word): the grouped schedule, executed the way the kernel executes it,
            // must give the parity symbols of the row-by-row encoder (ErasureCodes_NonBinaryLDPCSim.m:173-182).
            std::vector<uint8_t> x(n, 0), accv(m, 0), y(n, 0);
            uint32_t lcg = 0x9E3779B9u ^ (uint32_t)n;
            for (int j = 0; j < k; j++) { lcg = lcg * 1664525u + 1013904223u; x[j] = y[j] = (uint8_t)(lcg >> 24); }
            for (int r = 0; r < m; r++) {
                uint8_t sacc = 0;
                for (uint32_t e = row_ptr[r]; e + 1 < row_ptr[r + 1]; e++) sacc ^= gf.mul(hc->coefs[e], x[cols[e]]);
                x[k + r] = gf.mul(sacc, gf.inv[hc->coefs[row_ptr[r + 1] - 1]]);
            }
            for (int j = 0; j < k; j++)
                for (int i = 0; i < (1 << cdw_shift); i++) {
                    const uint32_t w = encg_src[((size_t)j << cdw_shift) + i];
                    if (w == 0xFFFFFFFFu) break;
                    accv[(w & 0x00FFFFFFu) / 128u] ^= gf.mul((uint8_t)(w >> 24), y[j]);
                }
            bool ok = true;
            for (int L = 1; L <= encg_nlevels && ok; L++) {
                std::vector<std::pair<int, uint8_t>> vals;
                for (int s_ = encg_lvlend[L - 1]; s_ < encg_lvlend[L]; s_++) {   // every step of the group reads the accumulators as the group found them
                    uint8_t a_ = accv[s_];
                    for (int i = 0; i < encg_npull[s_]; i++) {
                        const uint32_t w = encg_ent[encg_ent_off[s_] + i];
                        a_ ^= gf.mul((uint8_t)(w >> 24), accv[(w & 0x00FFFFFFu) / 128u]);
                    }
                    vals.emplace_back(s_, gf.mul(a_, encg_invc[s_]));
                }
                for (const auto &sv : vals) {
                    const int s_ = sv.first;
                    y[encg_steps[s_] >> 16] = sv.second;
                    for (int i = encg_ent_off[s_] + encg_npull[s_]; i < encg_ent_off[s_ + 1]; i++) {
                        const uint32_t w = encg_ent[i];
                        const int tgt_slot = (int)((w & 0x00FFFFFFu) / 128u);
                        if (tgt_slot >= encg_lvlend[L - 1] && tgt_slot < encg_lvlend[L]) ok = false;   // a scatter inside the group would race with the pulls
                        accv[tgt_slot] ^= gf.mul((uint8_t)(w >> 24), sv.second);
                    }
                }
            }
            ok = ok && (x == y);
            if (!ok) {
                delete hc;
                return set_error(ctx, LDPC_AMD_EHIP, "internal: the grouped encode schedule failed its self-check (n=%d, k=%d, cap=%d)", n, k, cap);
            }
        }
        if (encg_nlevels == 0) {
            encg_steps.assign(1, 0); encg_src.assign(1, 0xFFFFFFFFu); encg_ent.assign(1, 0xFFFFFFFFu);
            encg_lvlend.assign(1, 0); encg_ent_off.assign(1, 0); encg_invc.assign(1, 1); encg_npull.assign(1, 0);
            encg_ent_n = 0;
        }
    }
    hc->enc_info[0] = enc_nlevels; hc->enc_info[1] = encg_nlevels; hc->enc_info[2] = encg_pulls;
    hc->enc_info[3] = encg_ent_n - encg_pulls; hc->enc_info[4] = encg_maxpull;
    // encoder: source rows in the order of their column degree (every check is a step of the static schedule, so a row's
    // degree IS the number of accumulators it feeds): the row pieces a wavefront handles at once then take the same number
    // of edge turns
    std::vector<uint16_t> enc_order(std::max(k, 1), 0);
    {
        std::vector<int> cdeg(n, 0);
        for (int e = 0; e < hc->nnz; e++) cdeg[cols[e]]++;
        std::vector<int> idx(k);
        for (int j = 0; j < k; j++) idx[j] = j;
        std::stable_sort(idx.begin(), idx.end(), [&](int a_, int b_) { return cdeg[a_] > cdeg[b_]; });
        for (int j = 0; j < k; j++) enc_order[j] = (uint16_t)idx[j];
    }
    DevCode &d = hc->dev;
    d.n = n; d.k = k; d.m = m; d.nnz = hc->nnz; d.maxdeg = maxdeg; d.degpad = degpad; d.mpad = mpad;
    d.maxcoldeg = maxcoldeg; d.cdw_shift = cdw_shift;
    d.enc_nlevels = enc_nlevels;
    d.enc_lst_n = enc_lst_n;
    d.encg_nlevels = encg_nlevels; d.encg_ent_n = encg_ent_n;
    int rc;
    if ((rc = upload(ctx, hc, hc->row_ptr, &d.row_ptr)) || (rc = upload(ctx, hc, edges, &d.edges)) ||
        (rc = upload(ctx, hc, ell_col, &d.ell_col)) || (rc = upload(ctx, hc, ell_logc, &d.ell_logc)) ||
        (rc = upload(ctx, hc, ell_coef, &d.ell_coef)) || (rc = upload(ctx, hc, ell_pk, &d.ell_pk)) || (rc = upload(ctx, hc, rx_off, &d.rx_off)) || (rc = upload(ctx, hc, rx_logc, &d.rx_logc)) || (rc = upload(ctx, hc, cell, &d.cell)) ||
        (rc = upload(ctx, hc, enc_src, &d.enc_src)) || (rc = upload(ctx, hc, enc_order, &d.enc_order)) || (rc = upload(ctx, hc, enc_invc, &d.enc_invc)) ||
        (rc = upload(ctx, hc, enc_steps, &d.enc_steps)) || (rc = upload(ctx, hc, enc_lvlend, &d.enc_lvlend)) ||
        (rc = upload(ctx, hc, enc_lst, &d.enc_lst)) || (rc = upload(ctx, hc, enc_lst_off, &d.enc_lst_off)) ||
        (rc = upload(ctx, hc, encg_steps, &d.encg_steps)) || (rc = upload(ctx, hc, encg_lvlend, &d.encg_lvlend)) ||
        (rc = upload(ctx, hc, encg_invc, &d.encg_invc)) || (rc = upload(ctx, hc, encg_src, &d.encg_src)) ||
        (rc = upload(ctx, hc, encg_ent, &d.encg_ent)) || (rc = upload(ctx, hc, encg_ent_off, &d.encg_ent_off)) ||
        (rc = upload(ctx, hc, encg_npull, &d.encg_npull))) {
        free_code(hc);
        return rc;
    }
    ctx->codes.push_back(hc);
    return (int)ctx->codes.size() - 1;
}

static HostCode *get_code(ldpc_amd_ctx *ctx, int code)
{
    if (!ctx || code < 0 || code >= (int)ctx->codes.size()) return nullptr;
    return ctx->codes[code];
}

// ------------------------------------------------------------------------------------------------
// host-pointer staging
// ------------------------------------------------------------------------------------------------
struct Staged {
    const uint8_t *sym = nullptr;
    const uint8_t *erased = nullptr;
    uint8_t *out = nullptr;
    int32_t *sweeps = nullptr, *residual = nullptr, *status = nullptr;
};

// Host buffers, large batch: the reference's three FPGA kernels (data_in / decoder / data_out, coupled by channels,
// OpenCL/host/src/main.cpp:513-517,617-625) run concurrently; the equivalent here is a chunked pipeline in which the
// upload of chunk c+1, the kernels of chunk c and the download of chunk c-1 overlap.  Caller memory is pageable, so a
// copy call occupies its calling thread: uploads and kernel launches are issued from this thread, downloads from a
// helper thread, each on its own stream; device staging is double-buffered.  One helper for every host-pointer entry
// point (decode, encode, RS encode, RS decode): up to two input arrays and up to four output arrays, each `item_bytes`
// per item (frame or block); launch(count, in0, in1, out0, out1, out2, out3) enqueues the kernels of one chunk on the
// context's stream.
struct PipeIn {
    const void *host = nullptr;
    size_t item_bytes = 0;
};
struct PipeOut {
    void *host = nullptr;      // nullptr: the array is produced on the device but not downloaded
    size_t item_bytes = 0;     // 0: unused slot
};
static const size_t kPipeChunkBytes = (size_t)96 << 20;    // device staging per chunk and array
static const size_t kPipeThreshold = (size_t)192 << 20;    // batches below this go in one shot

template <class Launch>
static int host_pipeline(ldpc_amd_ctx *ctx, int64_t nitems, const PipeIn (&ins)[2], const PipeOut (&outs)[4], Launch launch)
{
    size_t big = 1;
    for (const PipeIn &i : ins) big = std::max(big, i.item_bytes);
    for (const PipeOut &o : outs) big = std::max(big, o.item_bytes);
    const int64_t C = std::max<int64_t>(1, std::min<int64_t>(nitems, (int64_t)(kPipeChunkBytes / big)));
    const int64_t nc = (nitems + C - 1) / C;
    // small output arrays (the three status words of the decoder) share one staging block
    size_t small_stride[4] = {0, 0, 0, 0}, small_total = 0;
    for (int i = 1; i < 4; i++) { small_stride[i] = small_total; small_total += ((outs[i].item_bytes * (size_t)C + 255) & ~(size_t)255); }
    int rc;
    if ((rc = scratch_reserve(ctx, ctx->stage_in, 2 * std::max<size_t>(ins[0].item_bytes, 1) * C)) ||
        (rc = scratch_reserve(ctx, ctx->stage_er, 2 * std::max<size_t>(ins[1].item_bytes, 1) * C)) ||
        (rc = scratch_reserve(ctx, ctx->stage_out, 2 * std::max<size_t>(outs[0].item_bytes, 1) * C)) ||
        (rc = scratch_reserve(ctx, ctx->stage_i32, 2 * std::max<size_t>(small_total, 256))))
        return rc;
    if (!ctx->aux_in) LDPC_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_in, hipStreamNonBlocking));
    if (!ctx->aux_out) LDPC_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_out, hipStreamNonBlocking));
    // the five events live in the context (created once, destroyed by ldpc_amd_cleanup): no error path leaks them
    for (int i = 0; i < 5; i++)
        if (!ctx->pipe_events[i]) LDPC_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_events[i], hipEventDisableTiming));
    hipEvent_t e_in[2] = {ctx->pipe_events[0], ctx->pipe_events[1]}, e_k[2] = {ctx->pipe_events[2], ctx->pipe_events[3]};
    hipEvent_t e_free = ctx->pipe_events[4];
    // the staging buffers may still be in use by earlier work on the context's stream
    LDPC_HIP_TRY(ctx, hipEventRecord(e_free, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_in, e_free, 0));

    std::mutex mu;
    std::condition_variable cv;
    int64_t launched = 0, downloaded = 0;   // chunks whose kernels have been enqueued / whose results are on the host
    bool failed = false;
    std::string helper_err;
    const int device = ctx->device;
    hipStream_t s_out = ctx->aux_out;
    uint8_t *dev_out0 = (uint8_t *)ctx->stage_out.p;
    uint8_t *dev_small = (uint8_t *)ctx->stage_i32.p;
    const size_t small_buf = std::max<size_t>(small_total, 256);
    auto dev_out = [&](int i, int b) -> uint8_t * {
        return i == 0 ? dev_out0 + (size_t)b * outs[0].item_bytes * C : dev_small + (size_t)b * small_buf + small_stride[i];
    };

    std::thread helper;
    auto helper_body = [&]() {
        (void)hipSetDevice(device);
        for (int64_t c = 0; c < nc; c++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return launched > c || failed; });
                if (failed) return;
            }
            const int b = (int)(c & 1);
            const int64_t f0 = c * C, cnt = std::min(C, nitems - f0);
            hipError_t e = hipStreamWaitEvent(s_out, e_k[b], 0);
            for (int i = 0; i < 4 && e == hipSuccess; i++)
                if (outs[i].host && outs[i].item_bytes)
                    e = hipMemcpyAsync((uint8_t *)outs[i].host + (size_t)f0 * outs[i].item_bytes, dev_out(i, b), outs[i].item_bytes * cnt, hipMemcpyDeviceToHost, s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(s_out);
            std::lock_guard<std::mutex> lk(mu);
            if (e != hipSuccess) { failed = true; helper_err = hipGetErrorString(e); cv.notify_all(); return; }
            downloaded = c + 1;
            cv.notify_all();
        }
    };
    try {
        helper = std::thread(helper_body);
    } catch (...) {   // std::system_error must not cross the C ABI
        return set_error(ctx, LDPC_AMD_ENOMEM, "host pipeline: could not start the download thread");
    }

    int result = LDPC_AMD_OK;
    for (int64_t c = 0; c < nc && result == LDPC_AMD_OK; c++) {
        const int b = (int)(c & 1);
        const int64_t f0 = c * C, cnt = std::min(C, nitems - f0);
        uint8_t *din0 = (uint8_t *)ctx->stage_in.p + (size_t)b * ins[0].item_bytes * C, *din1 = (uint8_t *)ctx->stage_er.p + (size_t)b * ins[1].item_bytes * C;
        hipError_t e = hipSuccess;
        if (c >= 2) e = hipStreamWaitEvent(ctx->aux_in, e_k[b], 0);          // in[b] was read by the kernels of chunk c-2
        if (e == hipSuccess && ins[0].item_bytes) e = hipMemcpyAsync(din0, (const uint8_t *)ins[0].host + (size_t)f0 * ins[0].item_bytes, ins[0].item_bytes * cnt, hipMemcpyHostToDevice, ctx->aux_in);
        if (e == hipSuccess && ins[1].item_bytes) e = hipMemcpyAsync(din1, (const uint8_t *)ins[1].host + (size_t)f0 * ins[1].item_bytes, ins[1].item_bytes * cnt, hipMemcpyHostToDevice, ctx->aux_in);
        if (e == hipSuccess) e = hipEventRecord(e_in[b], ctx->aux_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, e_in[b], 0);
        if (e != hipSuccess) { result = set_error(ctx, LDPC_AMD_EHIP, "host pipeline: %s", hipGetErrorString(e)); break; }
        {   // out[b] and e_k[b] belong to chunk c-2 until its results are on the host
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return downloaded >= c - 1 || failed; });
            if (failed) break;
        }
        if ((rc = launch(cnt, din0, din1, dev_out(0, b), dev_out(1, b), dev_out(2, b), dev_out(3, b)))) { result = rc; break; }
        e = hipEventRecord(e_k[b], ctx->stream);
        if (e != hipSuccess) { result = set_error(ctx, LDPC_AMD_EHIP, "host pipeline: %s", hipGetErrorString(e)); break; }
        std::lock_guard<std::mutex> lk(mu);
        launched = c + 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (result != LDPC_AMD_OK) failed = true;
        cv.notify_all();
    }
    helper.join();
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->aux_in);
    (void)hipStreamSynchronize(ctx->aux_out);
    if (result == LDPC_AMD_OK && failed) result = set_error(ctx, LDPC_AMD_EHIP, "host pipeline (download): %s", helper_err.c_str());
    return result;
}

// Small host-pointer calls (a single Matlab-style frame is the extreme): the call is bound by the NUMBER of runtime calls, not
// by bytes -- two copies in, four out, each a separate pageable transfer.  Inputs are packed into one pinned block (one
// upload), outputs come back as one download and are unpacked on the host.
static const size_t kSmallCall = (size_t)1 << 20;
static int pinned_reserve(ldpc_amd_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pin_cap) return LDPC_AMD_OK;
    if (ctx->pin) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->pin); ctx->pin = nullptr; ctx->pin_cap = 0; }
    const size_t want = std::max<size_t>(bytes, 2 * kSmallCall);
    hipError_t e = hipHostMalloc(&ctx->pin, want, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->pin = nullptr; return set_error(ctx, LDPC_AMD_ENOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); }
    ctx->pin_cap = want;
    return LDPC_AMD_OK;
}

}  // namespace ldpc_amd

using namespace ldpc_amd;

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *ldpc_amd_version(void) { return "ldpc_erasure_amd 0.1 (gfx950)"; }

int ldpc_amd_init(int device_ordinal, ldpc_amd_ctx **out)
{
    if (!out) return set_error(nullptr, LDPC_AMD_EINVAL, "ctx output pointer is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return set_error(nullptr, LDPC_AMD_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return set_error(nullptr, LDPC_AMD_EINVAL, "device ordinal %d out of range (0..%d)", device_ordinal, ndev - 1);
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess)
        return set_error(nullptr, LDPC_AMD_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess)
        return set_error(nullptr, LDPC_AMD_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(nullptr, LDPC_AMD_EUNSUP, "device %d is %s; this library is built for gfx950 only",
                         device_ordinal, prop.gcnArchName);
    ldpc_amd_ctx *ctx = new (std::nothrow) ldpc_amd_ctx();
    if (!ctx) return set_error(nullptr, LDPC_AMD_ENOMEM, "out of host memory");
    ctx->device = device_ordinal;
    ctx->sm_count = prop.multiProcessorCount;
    std::string bad_knob;
    if (!knobs_from_env(ctx->knobs, &bad_knob)) {   // the environment is looked at here and nowhere else
        delete ctx;
        return set_error(nullptr, LDPC_AMD_EINVAL, "ldpc_amd_init: environment knob rejected (unparsable or out of range): %s", bad_knob.c_str());
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return set_error(nullptr, LDPC_AMD_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
    // device-visible pinned word the kernels report violated assumptions through (checked after every synchronisation)
    if (hipHostMalloc((void **)&ctx->dev_err_host, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) *ctx->dev_err_host = 0;
    else ctx->dev_err_host = nullptr;
    if ((e = upload_constants(ctx->stream)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return set_error(nullptr, LDPC_AMD_EHIP, "constant upload: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return LDPC_AMD_OK;
}

void ldpc_amd_cleanup(ldpc_amd_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (HostCode *c : ctx->codes) free_code(c);
    for (HostRs *r : ctx->rs) {
        if (r->d_g) (void)hipFree(r->d_g);
        if (r->d_pt) (void)hipFree(r->d_pt);
        delete r;
    }
    Scratch *all[] = {&ctx->sched, &ctx->mlws, &ctx->mlstate, &ctx->mlops, &ctx->mlrec, &ctx->mllist, &ctx->biglist, &ctx->encctr, &ctx->stage_in, &ctx->stage_er,
                      &ctx->stage_out, &ctx->stage_i32, &ctx->schedpull, &ctx->schedlists, &ctx->rsws, &ctx->rsbad, &ctx->fpga_erased, &ctx->fpga_stats};
    for (Scratch *s : all) scratch_free(*s);
    for (auto &v : ctx->prof_events)
        for (auto &pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->pipe_events)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ml_events)
        if (e) (void)hipEventDestroy(e);
    if (ctx->aux_in) (void)hipStreamDestroy(ctx->aux_in);
    if (ctx->aux_out) (void)hipStreamDestroy(ctx->aux_out);
    if (ctx->aux_ml) (void)hipStreamDestroy(ctx->aux_ml);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->ml_head_host) (void)hipHostFree(ctx->ml_head_host);
    if (ctx->dev_err_host) (void)hipHostFree(ctx->dev_err_host);
    delete ctx;
}

const char *ldpc_amd_last_error(const ldpc_amd_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int ldpc_amd_set_stream(ldpc_amd_ctx *ctx, void *hip_stream)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_in) (void)hipStreamSynchronize(ctx->aux_in);
    if (ctx->aux_out) (void)hipStreamSynchronize(ctx->aux_out);
    if (ctx->aux_ml) (void)hipStreamSynchronize(ctx->aux_ml);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return LDPC_AMD_OK;
}

int ldpc_amd_configure(ldpc_amd_ctx *ctx, const char *key, const char *value)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    if (knob_set(ctx->knobs, key, value))
        return set_error(ctx, LDPC_AMD_EINVAL, "ldpc_amd_configure: unknown key or bad value: %s = %s", key ? key : "(null)", value ? value : "(default)");
    return LDPC_AMD_OK;
}

int ldpc_amd_knobs(ldpc_amd_ctx *ctx, char *buf, int cap)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    const Knobs def{};
    std::string out;
    for (const KnobDesc &d : kKnobs) {
        const long long v = d.get(ctx->knobs);
        if (v == d.get(def)) continue;
        if (!out.empty()) out += ' ';
        out += d.name;
        out += '=';
        out += std::to_string(v);
    }
    if (buf && cap > 0) {
        const size_t nb = std::min(out.size(), (size_t)cap - 1);
        memcpy(buf, out.data(), nb);
        buf[nb] = '\0';
    }
    return (int)out.size();
}

int ldpc_amd_synchronize(ldpc_amd_ctx *ctx)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

int ldpc_amd_code_params(int code_ind, int params[6])
{
    const BuiltinCode *b = find_builtin(code_ind);
    if (!b || !params) return LDPC_AMD_ENOCODE;
    params[0] = b->n; params[1] = b->k; params[2] = b->first_row; params[3] = b->last_row;
    params[4] = b->rs_n; params[5] = b->rs_k;
    return LDPC_AMD_OK;
}

int ldpc_amd_load_builtin_code(ldpc_amd_ctx *ctx, int code_ind, uint64_t coef_seed)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    const BuiltinCode *b = find_builtin(code_ind);
    if (!b) return set_error(ctx, LDPC_AMD_ENOCODE, "no built-in code with index %d", code_ind);
    std::vector<uint8_t> coefs(b->nnz, 1);
    if (coef_seed)
        for (int i = 0; i < b->nnz; i++) coefs[i] = ldpc_synth_nonzero(coef_seed, LDPC_SYNTH_STREAM_COEF, (uint64_t)i);
    return register_code(ctx, b->n, b->k, b->row_ptr, b->cols, coefs.data());
}

int ldpc_amd_register_code(ldpc_amd_ctx *ctx, int n, int k, const uint32_t *row_ptr, const uint16_t *cols,
                           const uint8_t *coefs)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    return register_code(ctx, n, k, row_ptr, cols, coefs);
}

int ldpc_amd_code_info(ldpc_amd_ctx *ctx, int code, int *n, int *k, int *nnz)
{
    HostCode *hc = get_code(ctx, code);
    if (!hc) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown code handle %d", code) : LDPC_AMD_EINVAL;
    if (n) *n = hc->n;
    if (k) *k = hc->k;
    if (nnz) *nnz = hc->nnz;
    return LDPC_AMD_OK;
}

int ldpc_amd_encode_info(ldpc_amd_ctx *ctx, int code, int info[6])
{
    HostCode *hc = get_code(ctx, code);
    if (!hc) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown code handle %d", code) : LDPC_AMD_EINVAL;
    if (!info) return set_error(ctx, LDPC_AMD_EINVAL, "info must not be null");
    for (int i = 0; i < 5; i++) info[i] = hc->enc_info[i];
    info[5] = ctx->last_enc_grouped;
    return LDPC_AMD_OK;
}

int ldpc_amd_code_csr(ldpc_amd_ctx *ctx, int code, uint32_t *row_ptr, uint16_t *cols, uint8_t *coefs)
{
    HostCode *hc = get_code(ctx, code);
    if (!hc) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown code handle %d", code) : LDPC_AMD_EINVAL;
    if (row_ptr) memcpy(row_ptr, hc->row_ptr.data(), hc->row_ptr.size() * sizeof(uint32_t));
    if (cols) memcpy(cols, hc->cols.data(), hc->cols.size() * sizeof(uint16_t));
    if (coefs) memcpy(coefs, hc->coefs.data(), hc->coefs.size());
    return LDPC_AMD_OK;
}

int ldpc_amd_decode_batch(ldpc_amd_ctx *ctx, int code, int S, int64_t nframes, const uint8_t *sym,
                          const uint8_t *erased, int max_sweeps, int do_ml, uint8_t *out, int32_t *sweeps,
                          int32_t *residual, int32_t *status, unsigned flags)
{
    HostCode *hc = get_code(ctx, code);
    if (!hc) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown code handle %d", code) : LDPC_AMD_EINVAL;
    if (nframes < 0 || S < 1) return set_error(ctx, LDPC_AMD_EINVAL, "bad nframes/S");
    if (nframes == 0) return LDPC_AMD_OK;
    if (!sym || !erased || !out) return set_error(ctx, LDPC_AMD_EINVAL, "sym/erased/out must not be null");
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t fbytes = (size_t)hc->n * S;
    DecodeArgs d{};
    d.code = hc->dev; d.S = S; d.nframes = nframes; d.in_rows = hc->n; d.max_sweeps = max_sweeps; d.do_ml = do_ml ? 1 : 0;
    if (flags & LDPC_AMD_DEVICE_PTRS) {
        d.sym = sym; d.erased = erased; d.out = out; d.sweeps = sweeps; d.residual = residual; d.status = status;
        if (flags & LDPC_AMD_INPLACE) {
            if (out != sym || S < 16) return set_error(ctx, LDPC_AMD_EINVAL, "LDPC_AMD_INPLACE needs out == sym and S >= 16");
            d.inplace = 1;
        }
        return launch_decode(ctx, d);
    }
    if (flags & LDPC_AMD_INPLACE) return set_error(ctx, LDPC_AMD_EINVAL, "LDPC_AMD_INPLACE needs LDPC_AMD_DEVICE_PTRS");
    int rc;
    const size_t ebytes = (size_t)hc->n;
    if (fbytes * (size_t)nframes >= kPipeThreshold && ctx->knobs.host_pipeline) {
        const PipeIn ins[2] = {{sym, fbytes}, {erased, ebytes}};
        const PipeOut outs[4] = {{out, fbytes}, {sweeps, sizeof(int32_t)}, {residual, sizeof(int32_t)}, {status, sizeof(int32_t)}};
        return host_pipeline(ctx, nframes, ins, outs, [&](int64_t cnt, uint8_t *i0, uint8_t *i1, uint8_t *o0, uint8_t *o1, uint8_t *o2, uint8_t *o3) {
            DecodeArgs dc = d;
            dc.nframes = cnt; dc.sym = i0; dc.erased = i1; dc.out = o0;
            dc.sweeps = (int32_t *)o1; dc.residual = (int32_t *)o2; dc.status = (int32_t *)o3;
            return launch_decode(ctx, dc);
        });
    }
    const size_t in_b = (fbytes + ebytes) * nframes, out_b = (fbytes + 3 * sizeof(int32_t)) * nframes;
    if (in_b + out_b + 64 <= kSmallCall) {
        // one upload, one download (pinned bounce block): sym | erased -> device ; out | sweeps | residual | status <- device
        const size_t o_er = (fbytes * nframes + 15) & ~(size_t)15, o_out = (o_er + ebytes * nframes + 15) & ~(size_t)15;
        const size_t o_i32 = (o_out + fbytes * nframes + 15) & ~(size_t)15, total = o_i32 + 3 * sizeof(int32_t) * nframes;
        if ((rc = pinned_reserve(ctx, total)) || (rc = scratch_reserve(ctx, ctx->stage_in, std::max(total, 2 * kSmallCall)))) return rc;
        uint8_t *hp = (uint8_t *)ctx->pin, *dp = (uint8_t *)ctx->stage_in.p;
        memcpy(hp, sym, fbytes * nframes);
        memcpy(hp + o_er, erased, ebytes * nframes);
        LDPC_HIP_TRY(ctx, hipMemcpyAsync(dp, hp, o_out, hipMemcpyHostToDevice, ctx->stream));
        int32_t *i32 = (int32_t *)(dp + o_i32);
        d.sym = dp; d.erased = dp + o_er; d.out = dp + o_out;
        d.sweeps = i32; d.residual = i32 + nframes; d.status = i32 + 2 * nframes;
        if ((rc = launch_decode(ctx, d))) return rc;
        LDPC_HIP_TRY(ctx, hipMemcpyAsync(hp + o_out, dp + o_out, total - o_out, hipMemcpyDeviceToHost, ctx->stream));
        LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = check_device_error(ctx))) return rc;
        memcpy(out, hp + o_out, fbytes * nframes);
        const int32_t *h32 = (const int32_t *)(hp + o_i32);
        if (sweeps) memcpy(sweeps, h32, sizeof(int32_t) * nframes);
        if (residual) memcpy(residual, h32 + nframes, sizeof(int32_t) * nframes);
        if (status) memcpy(status, h32 + 2 * nframes, sizeof(int32_t) * nframes);
        return LDPC_AMD_OK;
    }
    if ((rc = scratch_reserve(ctx, ctx->stage_in, fbytes * nframes)) || (rc = scratch_reserve(ctx, ctx->stage_er, (size_t)hc->n * nframes)) ||
        (rc = scratch_reserve(ctx, ctx->stage_out, fbytes * nframes)) || (rc = scratch_reserve(ctx, ctx->stage_i32, 3 * sizeof(int32_t) * (size_t)nframes)))
        return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in.p, sym, fbytes * nframes, hipMemcpyHostToDevice, ctx->stream));
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_er.p, erased, (size_t)hc->n * nframes, hipMemcpyHostToDevice, ctx->stream));
    int32_t *i32 = (int32_t *)ctx->stage_i32.p;
    d.sym = (const uint8_t *)ctx->stage_in.p; d.erased = (const uint8_t *)ctx->stage_er.p; d.out = (uint8_t *)ctx->stage_out.p;
    d.sweeps = i32; d.residual = i32 + nframes; d.status = i32 + 2 * nframes;
    if ((rc = launch_decode(ctx, d))) return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(out, d.out, fbytes * nframes, hipMemcpyDeviceToHost, ctx->stream));
    if (sweeps) LDPC_HIP_TRY(ctx, hipMemcpyAsync(sweeps, d.sweeps, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, ctx->stream));
    if (residual) LDPC_HIP_TRY(ctx, hipMemcpyAsync(residual, d.residual, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, ctx->stream));
    if (status) LDPC_HIP_TRY(ctx, hipMemcpyAsync(status, d.status, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

int ldpc_amd_encode_batch(ldpc_amd_ctx *ctx, int code, int S, int64_t nframes, const uint8_t *source,
                          uint8_t *codeword, unsigned flags)
{
    HostCode *hc = get_code(ctx, code);
    if (!hc) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown code handle %d", code) : LDPC_AMD_EINVAL;
    if (nframes < 0 || S < 1) return set_error(ctx, LDPC_AMD_EINVAL, "bad nframes/S");
    if (nframes == 0) return LDPC_AMD_OK;
    if (!source || !codeword) return set_error(ctx, LDPC_AMD_EINVAL, "source/codeword must not be null");
    if (hc->dev.enc_nlevels == 0) return set_error(ctx, LDPC_AMD_EUNSUP, "code is not in triangle form: no systematic encoder");
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (flags & LDPC_AMD_DEVICE_PTRS) return launch_encode(ctx, hc->dev, S, nframes, source, codeword);
    const size_t ib = (size_t)hc->k * S * nframes, ob = (size_t)hc->n * S * nframes;
    int rc;
    if (ob >= kPipeThreshold && ctx->knobs.host_pipeline) {   // large batch: upload / encode / download overlap, chunk by chunk
        const PipeIn ins[2] = {{source, (size_t)hc->k * S}, {nullptr, 0}};
        const PipeOut outs[4] = {{codeword, (size_t)hc->n * S}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}};
        return host_pipeline(ctx, nframes, ins, outs, [&](int64_t cnt, uint8_t *i0, uint8_t *, uint8_t *o0, uint8_t *, uint8_t *, uint8_t *) {
            return launch_encode(ctx, hc->dev, S, cnt, i0, o0);
        });
    }
    if ((rc = scratch_reserve(ctx, ctx->stage_in, ib)) || (rc = scratch_reserve(ctx, ctx->stage_out, ob))) return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in.p, source, ib, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_encode(ctx, hc->dev, S, nframes, (const uint8_t *)ctx->stage_in.p, (uint8_t *)ctx->stage_out.p))) return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(codeword, ctx->stage_out.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

// ---- Reed-Solomon ----------------------------------------------------------------------------------
static int gf_invert(const GfHost &gf, std::vector<uint8_t> &A, int k, std::vector<uint8_t> &I)
{
    I.assign((size_t)k * k, 0);
    for (int i = 0; i < k; i++) I[(size_t)i * k + i] = 1;
    for (int col = 0; col < k; col++) {
        int p = -1;
        for (int r = col; r < k; r++)
            if (A[(size_t)r * k + col]) { p = r; break; }
        if (p < 0) return -1;
        if (p != col)
            for (int t = 0; t < k; t++) {
                std::swap(A[(size_t)col * k + t], A[(size_t)p * k + t]);
                std::swap(I[(size_t)col * k + t], I[(size_t)p * k + t]);
            }
        const uint8_t iv = gf.inv[A[(size_t)col * k + col]];
        for (int t = 0; t < k; t++) {
            A[(size_t)col * k + t] = gf.mul(A[(size_t)col * k + t], iv);
            I[(size_t)col * k + t] = gf.mul(I[(size_t)col * k + t], iv);
        }
        for (int r = 0; r < k; r++) {
            const uint8_t f = A[(size_t)r * k + col];
            if (r == col || !f) continue;
            for (int t = 0; t < k; t++) {
                A[(size_t)r * k + t] ^= gf.mul(f, A[(size_t)col * k + t]);
                I[(size_t)r * k + t] ^= gf.mul(f, I[(size_t)col * k + t]);
            }
        }
    }
    return 0;
}

int ldpc_amd_rs_create(ldpc_amd_ctx *ctx, int n, int k)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    if (n < 2 || n > 255 || k < 1 || k >= n) return set_error(ctx, LDPC_AMD_EINVAL, "bad RS (n,k) = (%d,%d)", n, k);
    const GfHost &gf = gf_host();
    // Matlab/Test_My_RS_Decode.m:30-37: G(row,col) = alpha^(row*col) (1-based), G <- inv(G(1:k,1:k)) * G
    std::vector<uint8_t> V((size_t)k * n), A((size_t)k * k), I;
    for (int row = 1; row <= k; row++)
        for (int col = 1; col <= n; col++) V[(size_t)(row - 1) * n + col - 1] = gf.exp[(row * col) % 255];
    for (int r = 0; r < k; r++) memcpy(&A[(size_t)r * k], &V[(size_t)r * n], k);
    if (gf_invert(gf, A, k, I)) return set_error(ctx, LDPC_AMD_EINVAL, "singular Vandermonde block");
    HostRs *rs = new (std::nothrow) HostRs();
    if (!rs) return set_error(ctx, LDPC_AMD_ENOMEM, "out of host memory");
    rs->n = n; rs->k = k;
    rs->g.assign((size_t)k * n, 0);
    for (int r = 0; r < k; r++)
        for (int c = 0; c < n; c++) {
            uint8_t s = 0;
            for (int t = 0; t < k; t++) s ^= gf.mul(I[(size_t)r * k + t], V[(size_t)t * n + c]);
            rs->g[(size_t)r * n + c] = s;
        }
    std::vector<uint8_t> pt((size_t)(n - k) * k);
    for (int j = 0; j < n - k; j++)
        for (int i = 0; i < k; i++) pt[(size_t)j * k + i] = rs->g[(size_t)i * n + k + j];
    hipError_t e;
    if ((e = hipMalloc((void **)&rs->d_g, rs->g.size())) != hipSuccess || (e = hipMalloc((void **)&rs->d_pt, pt.size())) != hipSuccess ||
        (e = hipMemcpy(rs->d_g, rs->g.data(), rs->g.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(rs->d_pt, pt.data(), pt.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        if (rs->d_g) (void)hipFree(rs->d_g);
        if (rs->d_pt) (void)hipFree(rs->d_pt);
        delete rs;
        return set_error(ctx, LDPC_AMD_EHIP, "RS table upload: %s", hipGetErrorString(e));
    }
    ctx->rs.push_back(rs);
    return (int)ctx->rs.size() - 1;
}

static HostRs *get_rs(ldpc_amd_ctx *ctx, int rs)
{
    if (!ctx || rs < 0 || rs >= (int)ctx->rs.size()) return nullptr;
    return ctx->rs[rs];
}

int ldpc_amd_rs_generator(ldpc_amd_ctx *ctx, int rs, uint8_t *g)
{
    HostRs *r = get_rs(ctx, rs);
    if (!r || !g) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown RS handle %d", rs) : LDPC_AMD_EINVAL;
    memcpy(g, r->g.data(), r->g.size());
    return LDPC_AMD_OK;
}

int ldpc_amd_rs_encode_batch(ldpc_amd_ctx *ctx, int rs, int S, int64_t nblocks, const uint8_t *source,
                             uint8_t *codeword, unsigned flags)
{
    HostRs *r = get_rs(ctx, rs);
    if (!r) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown RS handle %d", rs) : LDPC_AMD_EINVAL;
    if (nblocks < 0 || (S != 1 && S % 16)) return set_error(ctx, LDPC_AMD_EUNSUP, "S must be 1 or a multiple of 16");
    if (nblocks == 0) return LDPC_AMD_OK;
    if (!source || !codeword) return set_error(ctx, LDPC_AMD_EINVAL, "null data pointer");
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (flags & LDPC_AMD_DEVICE_PTRS) return launch_rs_encode(ctx, *r, S, nblocks, source, codeword);
    const size_t ib = (size_t)r->k * S * nblocks, ob = (size_t)r->n * S * nblocks;
    int rc;
    if (ob >= kPipeThreshold && ctx->knobs.host_pipeline) {
        const PipeIn ins[2] = {{source, (size_t)r->k * S}, {nullptr, 0}};
        const PipeOut outs[4] = {{codeword, (size_t)r->n * S}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}};
        return host_pipeline(ctx, nblocks, ins, outs, [&](int64_t cnt, uint8_t *i0, uint8_t *, uint8_t *o0, uint8_t *, uint8_t *, uint8_t *) {
            return launch_rs_encode(ctx, *r, S, cnt, i0, o0);
        });
    }
    if ((rc = scratch_reserve(ctx, ctx->stage_in, ib)) || (rc = scratch_reserve(ctx, ctx->stage_out, ob))) return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in.p, source, ib, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_rs_encode(ctx, *r, S, nblocks, (const uint8_t *)ctx->stage_in.p, (uint8_t *)ctx->stage_out.p))) return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(codeword, ctx->stage_out.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

int ldpc_amd_rs_decode_batch(ldpc_amd_ctx *ctx, int rs, int S, int64_t nblocks, const uint16_t *recv_idx,
                             const uint8_t *recv_val, uint8_t *msg, unsigned flags)
{
    HostRs *r = get_rs(ctx, rs);
    if (!r) return ctx ? set_error(ctx, LDPC_AMD_ENOCODE, "unknown RS handle %d", rs) : LDPC_AMD_EINVAL;
    if (nblocks < 0 || (S != 1 && S % 16)) return set_error(ctx, LDPC_AMD_EUNSUP, "S must be 1 or a multiple of 16");
    if (nblocks == 0) return LDPC_AMD_OK;
    if (!recv_idx || !recv_val || !msg) return set_error(ctx, LDPC_AMD_EINVAL, "null data pointer");
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (flags & LDPC_AMD_DEVICE_PTRS) return launch_rs_decode(ctx, *r, S, nblocks, recv_idx, recv_val, msg);
    // host pointers: the positions can be checked here (with device pointers the kernels check per block and decode a
    // malformed block to zeros)
    for (int64_t b = 0; b < nblocks; b++) {
        const uint16_t *ix = recv_idx + b * r->k;
        for (int t = 0; t < r->k; t++)
            if (ix[t] >= r->n || (t > 0 && ix[t - 1] >= ix[t]))
                return set_error(ctx, LDPC_AMD_EINVAL, "rs_decode: block %lld: recv_idx must be < n and strictly ascending (entry %d)", (long long)b, t);
    }
    const size_t vb = (size_t)r->k * S * nblocks, xb = (size_t)r->k * 2 * nblocks;
    int rc;
    if (vb >= kPipeThreshold && ctx->knobs.host_pipeline) {
        const PipeIn ins[2] = {{recv_val, (size_t)r->k * S}, {recv_idx, (size_t)r->k * 2}};
        const PipeOut outs[4] = {{msg, (size_t)r->k * S}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}};
        return host_pipeline(ctx, nblocks, ins, outs, [&](int64_t cnt, uint8_t *i0, uint8_t *i1, uint8_t *o0, uint8_t *, uint8_t *, uint8_t *) {
            return launch_rs_decode(ctx, *r, S, cnt, (const uint16_t *)i1, i0, o0);
        });
    }
    if ((rc = scratch_reserve(ctx, ctx->stage_in, vb)) || (rc = scratch_reserve(ctx, ctx->stage_er, xb)) ||
        (rc = scratch_reserve(ctx, ctx->stage_out, vb)))
        return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in.p, recv_val, vb, hipMemcpyHostToDevice, ctx->stream));
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_er.p, recv_idx, xb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_rs_decode(ctx, *r, S, nblocks, (const uint16_t *)ctx->stage_er.p, (const uint8_t *)ctx->stage_in.p,
                               (uint8_t *)ctx->stage_out.p)))
        return rc;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(msg, ctx->stage_out.p, vb, hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LDPC_AMD_OK;
}

int ldpc_amd_rs_bad_blocks(ldpc_amd_ctx *ctx, long long *count)
{
    if (!ctx || !count) return ctx ? set_error(ctx, LDPC_AMD_EINVAL, "null count pointer") : LDPC_AMD_EINVAL;
    *count = 0;
    if (!ctx->rsbad.p) return LDPC_AMD_OK;   // no RS decode ran on this context yet
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int v = 0;
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(&v, ctx->rsbad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = v;
    return LDPC_AMD_OK;
}

// ---- synthetic inputs --------------------------------------------------------------------------------
int ldpc_amd_synth_source(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int k, int S,
                          uint8_t *d_source)
{
    if (!ctx || !d_source || nframes < 0 || k < 1 || S < 1) return ctx ? set_error(ctx, LDPC_AMD_EINVAL, "bad argument") : LDPC_AMD_EINVAL;
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_synth_source(ctx, seed, frame0, nframes, k, S, d_source);
}

int ldpc_amd_synth_erasures_uniform(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int n,
                                    double per, uint8_t *d_erased)
{
    if (!ctx || !d_erased || nframes < 0 || n < 1) return ctx ? set_error(ctx, LDPC_AMD_EINVAL, "bad argument") : LDPC_AMD_EINVAL;
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_synth_erasures(ctx, seed, LDPC_SYNTH_STREAM_ERASE, frame0 * n, nframes * n, ldpc_synth_threshold(per), d_erased);
}

int ldpc_amd_synth_erasures_bursty(ldpc_amd_ctx *ctx, uint64_t seed, int64_t frame0, int64_t nframes, int n,
                                   double alpha, double beta, double good_transition_bias, uint8_t *d_erased)
{
    if (!ctx || !d_erased || nframes < 0 || frame0 < 0 || n < 1 || good_transition_bias <= 0)
        return ctx ? set_error(ctx, LDPC_AMD_EINVAL, "bad argument") : LDPC_AMD_EINVAL;
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_synth_bursty(ctx, seed, frame0 * n, (frame0 + nframes) * n, alpha, beta, good_transition_bias, d_erased);
}

// ---- FPGA harness drop-in (three kernels of OpenCL/host/src/main.cpp:578-626) ---------------------------
static int fpga_code(ldpc_amd_ctx *ctx, int code_ind)
{
    if (code_ind < 0 || code_ind >= 4) return set_error(ctx, LDPC_AMD_ENOCODE, "code_ind %d out of range", code_ind);
    if (ctx->fpga_binary_code[code_ind] < 0) {
        int h = ldpc_amd_load_builtin_code(ctx, code_ind, 0);  // binary H: the FPGA decoder XORs packets
        if (h < 0) return h;
        ctx->fpga_binary_code[code_ind] = h;
    }
    return ctx->fpga_binary_code[code_ind];
}

// Frames per chunk of the streamed run (LDPC_AMD_FPGA_CHUNK overrides it: tests use small chunks to cross chunk borders).
static long fpga_chunk_frames(const ldpc_amd_ctx *ctx) { return ctx->knobs.fpga_chunk > 0 ? ctx->knobs.fpga_chunk : 65536; }
// Runs up to this many frames keep their per-frame results for ldpc_amd_fpga_frame_stats (8 bytes per frame).
static const long kFpgaKeepFrames = 1l << 22;

int ldpc_amd_data_in(ldpc_amd_ctx *ctx, const ldpc_amd_symbol_type *data_in, unsigned short nldpc, int seed,
                     int PER_numerator_div_64, int code_ind, long numFrames)
{
    return ldpc_amd_data_in_at(ctx, data_in, nldpc, seed, PER_numerator_div_64, code_ind, numFrames, 0);
}

int ldpc_amd_data_in_at(ldpc_amd_ctx *ctx, const ldpc_amd_symbol_type *data_in, unsigned short nldpc, int seed,
                        int PER_numerator_div_64, int code_ind, long numFrames, long firstFrame)
{
    (void)data_in;  // the FPGA kernel never reads its buffer either (ldpc_erasure_decoder_top.cl:84-117)
    if (!ctx) return LDPC_AMD_EINVAL;
    const BuiltinCode *b = find_builtin(code_ind);
    if (!b) return set_error(ctx, LDPC_AMD_ENOCODE, "no built-in code with index %d", code_ind);
    if (numFrames < 0 || firstFrame < 0) return set_error(ctx, LDPC_AMD_EINVAL, "numFrames / firstFrame < 0");
    (void)nldpc;  // the kernel takes n from ldpc_params[code_ind] (ldpc_erasure_decoder_top.cl:70-71), nldpc is only printed
    // The FPGA source is a frame loop feeding a channel (:84-117): nothing is materialised.  The erasure stream is a pure
    // function of (seed, symbol index), so the source is only armed here and its symbols are drawn chunk by chunk inside
    // the decoder call, the way the decoder kernel pulls them from LDPC_DEC_DIN.
    ctx->fpga_frames = numFrames; ctx->fpga_code_ind = code_ind; ctx->fpga_per64 = PER_numerator_div_64; ctx->fpga_seed = seed;
    ctx->fpga_first = firstFrame;
    ctx->fpga_decoded = -1; ctx->fpga_kept = false;
    return LDPC_AMD_OK;
}

// The streamed run: for every chunk of frames  draw the flags (data_in) -> decode -> add to the two running counters
// (num_frame_errors / num_RS_frame_errors of ldpc_erasure_decoder_perf_tests.cl:46-47,70-80,229-236).  O(chunk) memory,
// everything enqueued on the context's stream; ldpc_amd_data_out synchronises.
static int fpga_run(ldpc_amd_ctx *ctx, short num_iter, int code_ind, bool halves)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    if (ctx->fpga_frames < 0 || ctx->fpga_code_ind != code_ind)
        return set_error(ctx, LDPC_AMD_EINVAL, "ldpc_amd_data_in was not called for code_ind %d", code_ind);
    if (num_iter < 1) return set_error(ctx, LDPC_AMD_EINVAL, "num_iter must be >= 1");
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int h = fpga_code(ctx, code_ind);
    if (h < 0) return h;
    HostCode *hc = ctx->codes[h];
    const BuiltinCode *b = find_builtin(code_ind);
    const long nf = ctx->fpga_frames;
    const long C = std::min<long>(std::max<long>(nf, 1), fpga_chunk_frames(ctx));
    const bool keep = nf <= kFpgaKeepFrames;
    const long slots = keep ? std::max<long>(nf, 1) : C;
    ctx->fpga_decoded = -1;
    int rc;
    if ((rc = scratch_reserve(ctx, ctx->fpga_erased, (size_t)C * hc->n))) return rc;
    if ((rc = scratch_reserve(ctx, ctx->fpga_stats, 64 + sizeof(int32_t) * 2 * (size_t)slots))) return rc;
    unsigned long long *counters = (unsigned long long *)ctx->fpga_stats.p;
    int32_t *res_base = (int32_t *)((unsigned char *)ctx->fpga_stats.p + 64), *it_base = res_base + slots;
    LDPC_HIP_TRY(ctx, hipMemsetAsync(counters, 0, 64, ctx->stream));
    uint8_t *flags = (uint8_t *)ctx->fpga_erased.p;
    for (long f0 = 0; f0 < nf; f0 += C) {
        const long cnt = std::min(C, nf - f0);
        int32_t *res = keep ? res_base + f0 : res_base, *its = keep ? it_base + f0 : it_base;
        // erased iff (rv & 0x3F) < PER_numerator_div_64 (ldpc_erasure_decoder_top.cl:105), rv from threefry4x32 with key
        // {1, seed} and the running symbol counter (:74-75,96-98): the FPGA's own erasure stream for this seed
        if ((rc = launch_synth_fpga(ctx, (uint32_t)ctx->fpga_seed, (uint64_t)(ctx->fpga_first + f0) * (uint64_t)hc->n, (int64_t)cnt * hc->n, ctx->fpga_per64, flags)))
            return rc;
        if (halves) {
            if ((rc = launch_fpga_halves(ctx, hc->dev, cnt, flags, num_iter, res, its))) return rc;
        } else {
            // flags-only decode: the payload is the all-zero codeword, only the erasure pattern matters
            DecodeArgs d{};
            d.code = hc->dev; d.S = 16; d.nframes = cnt; d.in_rows = hc->n; d.max_sweeps = num_iter; d.do_ml = 0;
            d.erased = flags; d.flags_only = 1; d.residual_sys = res; d.sweeps = its;
            if ((rc = launch_decode(ctx, d))) return rc;
        }
        if ((rc = launch_fpga_stats(ctx, hc->dev, b->rs_n, b->rs_k, cnt, flags, res, counters))) return rc;
    }
    ctx->fpga_decoded = nf;
    ctx->fpga_kept = keep;
    return LDPC_AMD_OK;
}

int ldpc_amd_ldpc_erasure_decoder(ldpc_amd_ctx *ctx, short num_iter, int code_ind) { return fpga_run(ctx, num_iter, code_ind, false); }

int ldpc_amd_ldpc_erasure_decoder_perf_tests(ldpc_amd_ctx *ctx, short num_iter, int code_ind) { return fpga_run(ctx, num_iter, code_ind, true); }

int ldpc_amd_fpga_frame_stats(ldpc_amd_ctx *ctx, long numFrames, int32_t *residual_sys, int32_t *iterations)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    if (numFrames < 0 || numFrames != ctx->fpga_frames || ctx->fpga_decoded != numFrames)
        return set_error(ctx, LDPC_AMD_EINVAL, "fpga_frame_stats: no decoded run of %ld frames (last data_in: %ld, decoded: %ld)",
                         numFrames, ctx->fpga_frames, ctx->fpga_decoded);
    if (numFrames == 0) return LDPC_AMD_OK;
    if (!ctx->fpga_kept)
        return set_error(ctx, LDPC_AMD_EUNSUP, "fpga_frame_stats: per-frame results are kept for runs of up to %ld frames", kFpgaKeepFrames);
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int32_t *res = (const int32_t *)((const unsigned char *)ctx->fpga_stats.p + 64);
    if (residual_sys) LDPC_HIP_TRY(ctx, hipMemcpyAsync(residual_sys, res, sizeof(int32_t) * (size_t)numFrames, hipMemcpyDeviceToHost, ctx->stream));
    if (iterations) LDPC_HIP_TRY(ctx, hipMemcpyAsync(iterations, res + numFrames, sizeof(int32_t) * (size_t)numFrames, hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LDPC_AMD_OK;
}

int ldpc_amd_data_out(ldpc_amd_ctx *ctx, ldpc_amd_symbol_type *data_out, int code_ind, long numFrames,
                      ldpc_amd_error_type *stats)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    if (ctx->fpga_code_ind != code_ind || numFrames < 0 || numFrames != ctx->fpga_frames)
        return set_error(ctx, LDPC_AMD_EINVAL, "data_out arguments do not match the last data_in call");
    if (ctx->fpga_decoded != numFrames)
        return set_error(ctx, LDPC_AMD_EINVAL, "data_out: no decoder call since the last data_in (the FPGA's data_out would block on ERROR_STAT)");
    const BuiltinCode *b = find_builtin(code_ind);
    if (!b) return set_error(ctx, LDPC_AMD_ENOCODE, "no built-in code with index %d", code_ind);
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long host[2] = {0, 0};
    if (numFrames > 0) {
        LDPC_HIP_TRY(ctx, hipMemcpyAsync(host, ctx->fpga_stats.p, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
        LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    // error_type holds two ints (ldpc_erasure_decoder_top.cl:46-49); N_T = 2e8 frames x 8 RS blocks still fits
    if (stats) { stats->num_LDPC_errors = (int)host[0]; stats->num_RS_errors = (int)host[1]; }
    if (data_out) {
        // the FPGA data_out never writes its buffer (ldpc_erasure_decoder_top.cl:140-150 is commented out);
        // the all-zero codeword is what a correct decode returns.
        memset(data_out, 0, sizeof(ldpc_amd_symbol_type) * (size_t)b->k);
    }
    return LDPC_AMD_OK;
}

// ---- measurement -----------------------------------------------------------------------------------------
int ldpc_amd_set_profiling(ldpc_amd_ctx *ctx, int enable)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    ctx->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return LDPC_AMD_OK;
}

int ldpc_amd_get_profile(ldpc_amd_ctx *ctx, double ms[LDPC_AMD_PROF_KINDS], int64_t launches[LDPC_AMD_PROF_KINDS])
{
    if (!ctx) return LDPC_AMD_EINVAL;
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int kd = 0; kd < LDPC_AMD_PROF_KINDS; kd++) {
        double total = 0;
        for (auto &pr : ctx->prof_events[kd]) {
            float t = 0;
            if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) total += t;
            ctx->prof_pool.push_back(pr.first);
            ctx->prof_pool.push_back(pr.second);
        }
        if (ms) ms[kd] = total;
        if (launches) launches[kd] = (int64_t)ctx->prof_events[kd].size();
        ctx->prof_events[kd].clear();
    }
    return LDPC_AMD_OK;
}

const char *ldpc_amd_profile_kernel_name(ldpc_amd_ctx *ctx, int kind)
{
    if (!ctx || kind < 0 || kind >= LDPC_AMD_PROF_KINDS) return "";
    return ctx->prof_names[kind].c_str();
}

int ldpc_amd_last_plan(ldpc_amd_ctx *ctx, int info[8])
{
    if (!ctx || !info) return LDPC_AMD_EINVAL;
    memcpy(info, ctx->last_plan, sizeof(ctx->last_plan));
    return LDPC_AMD_OK;
}

int ldpc_amd_ml_stats(ldpc_amd_ctx *ctx, long long stats[4])
{
    if (!ctx || !stats) return LDPC_AMD_EINVAL;
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (!ctx->mllist.p) return LDPC_AMD_OK;   // no decode yet
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t hdr[32];
    LDPC_HIP_TRY(ctx, hipMemcpyAsync(hdr, ctx->mllist.p, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
    LDPC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    stats[0] = hdr[0]; stats[1] = hdr[27]; stats[2] = hdr[24]; stats[3] = hdr[19];   // layout: launch_decode (kernels.hip)
    return LDPC_AMD_OK;
}

// ---- diagnostics ---------------------------------------------------------------------------------------
int ldpc_amd_selftest(ldpc_amd_ctx *ctx)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_selftest(ctx);
}

int ldpc_amd_copy_probe(ldpc_amd_ctx *ctx, const void *src, void *dst, uint64_t bytes, int reps, double *ms_per_copy)
{
    if (!ctx) return LDPC_AMD_EINVAL;
    if (!src || !dst || !ms_per_copy || reps < 1 || bytes < 16 || (bytes & 15u) || ((uintptr_t)src & 15u) || ((uintptr_t)dst & 15u))
        return set_error(ctx, LDPC_AMD_EINVAL, "copy_probe: device pointers and size must be 16-byte multiples, reps >= 1");
    LDPC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_copy_probe(ctx, (const uint8_t *)src, (uint8_t *)dst, bytes, reps, ms_per_copy);
}

int ldpc_amd_gf_tables(uint8_t *mult, uint8_t *inv)
{
    const GfHost &g = gf_host();
    if (mult)
        for (int a = 0; a < 256; a++)
            for (int b = 0; b < 256; b++) mult[a * 256 + b] = g.mul((uint8_t)a, (uint8_t)b);
    if (inv) memcpy(inv, g.inv, 256);
    return LDPC_AMD_OK;
}

}  // extern "C"
