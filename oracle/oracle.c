/*
 * oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; pinning status is
 * stated there).
 *
 * Conventions: Matlab indices are 1-based; C arrays here are 0-based.  Vlist keeps the reference's
 * 1-based column numbers (as OpenCL/device/LDPC_Vlist_data.h does) and every use subtracts 1, like
 * OpenCL/device/ldpc_erasure_decoder.cl:70,74.  An erasure is the value -1 (...Decoder.m:9).
 * Line numbers in comments refer to the reference file named at the top of each function.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#include "../include/ldpc_erasure_amd_synth.h"

/* ============================ GF(256) tables ==================================================== */
/* Matlab/Build_GF256_Lookup_Tables.m */
void oracle_gf_build(oracle_gf *t, int prim_poly)
{
    const int GF_SIZE = 256; /* :9 */
    memset(t, 0, sizeof(*t));
    /* :21-29  gf_log_inv(1)=0, (2)=1, then successive powers of alpha = x_gf(3) = 2 */
    t->antilog[0] = 0;
    t->antilog[1] = 1;
    int gf_temp = 2; /* alpha */
    for (int ii = 3; ii <= GF_SIZE; ii++) {
        t->antilog[ii - 1] = (uint8_t)gf_temp;
        /* gf_temp = alpha*gf_temp : multiply by x and reduce by the primitive polynomial */
        gf_temp <<= 1;
        if (gf_temp & 0x100) gf_temp ^= prim_poly;
    }
    /* :31-32  Log_array = [-inf 0:GF_SIZE-2]; log_lookup(gf_log_inv+1) = Log_array */
    for (int i = 0; i < 256; i++) t->log[i] = -1;
    for (int ii = 2; ii <= GF_SIZE; ii++) t->log[t->antilog[ii - 1]] = (int16_t)(ii - 2);
    t->log[0] = -1; /* -inf */
    /* :35-41  gf_inv(2)=1; gf_inv(ii) = gf_log_inv(GF_SIZE - log_lookup(ii) + 1); gf_inv = gf_inv(2:end) */
    t->inv[0] = 1;
    for (int ii = 3; ii <= GF_SIZE; ii++) {
        int x = ii - 1;
        t->inv[x - 1] = t->antilog[(GF_SIZE - t->log[x] + 1) - 1];
    }
    /* :44-54  mult(row,col) = gf_log_inv(mod(log(row)+log(col),255)+2), rows/cols of 0 stay zero */
    for (int row = 2; row <= GF_SIZE; row++)
        for (int col = 2; col <= GF_SIZE; col++) {
            int a = row - 1, b = col - 1;
            t->mult[a][b] = t->antilog[((t->log[a] + t->log[b]) % 255 + 2) - 1];
        }
    /* :57-67  add = bitxor */
    for (int a = 0; a < 256; a++)
        for (int b = 0; b < 256; b++) t->add[a][b] = (uint8_t)(a ^ b);
}

const oracle_gf *oracle_gf_default(void)
{
    static oracle_gf tab;
    static int built = 0;
    if (!built) {
        oracle_gf_build(&tab, 369); /* [1 0 1 1 1 0 0 0 1], ErasureCodes_NonBinaryLDPCSim.m:70 */
        built = 1;
    }
    return &tab;
}

#define GF_ADD(a, b) (gf->add[(a)][(b)])   /* GF_add_lookup(a+1, b+1)  */
#define GF_MUL(a, b) (gf->mult[(a)][(b)])  /* GF_mult_lookup(a+1, b+1) */
#define GF_INV(x) (gf->inv[(x) - 1])       /* GF_inv_lookup(x)         */

/* ============================ code container ==================================================== */
/* Matlab/ErasureCodes_NonBinaryLDPCSim.m:91-107 (Vlist, Vlist_val; ascending columns) */
oracle_code *oracle_code_create(int n, int k, const uint32_t *row_ptr, const uint16_t *cols,
                                const uint8_t *coefs)
{
    if (n <= 0 || k < 0 || k >= n) return NULL;
    int m = n - k;
    int maxdeg = 0;
    for (int r = 0; r < m; r++) {
        int d = (int)(row_ptr[r + 1] - row_ptr[r]);
        if (d < 0) return NULL;
        if (d > maxdeg) maxdeg = d;
    }
    oracle_code *c = (oracle_code *)calloc(1, sizeof(*c));
    c->n = n; c->k = k; c->m = m; c->nnz = (int)row_ptr[m]; c->width = maxdeg + 1;
    c->vlist = (int *)calloc((size_t)m * c->width, sizeof(int));
    c->vlist_val = (int *)calloc((size_t)m * c->width, sizeof(int));
    for (int jj = 0; jj < m; jj++) {
        int d = (int)(row_ptr[jj + 1] - row_ptr[jj]);
        c->vlist[jj * c->width] = d;      /* :97 */
        c->vlist_val[jj * c->width] = d;  /* :98 */
        int prev = -1;
        for (int t = 0; t < d; t++) {
            int col = cols[row_ptr[jj] + t];
            if (col <= prev || col >= n || coefs[row_ptr[jj] + t] == 0) { oracle_code_destroy(c); return NULL; }
            prev = col;
            c->vlist[jj * c->width + t + 1] = col + 1;                     /* :103 (1-based) */
            c->vlist_val[jj * c->width + t + 1] = coefs[row_ptr[jj] + t]; /* :104 */
        }
    }
    return c;
}

void oracle_code_destroy(oracle_code *c)
{
    if (!c) return;
    free(c->vlist);
    free(c->vlist_val);
    free(c);
}

#define VL(ii, j) (c->vlist[(ii) * c->width + (j)])        /* Vlist(ii+1, j+1)     */
#define VV(ii, j) (c->vlist_val[(ii) * c->width + (j)])    /* Vlist_val(ii+1, j+1) */

/* H_sparse(ii, col): coefficient of 1-based column `col` in 0-based row ii, 0 if absent */
static int h_at(const oracle_code *c, int ii, int col)
{
    int d = VL(ii, 0);
    for (int j = 1; j <= d; j++)
        if (VL(ii, j) == col) return VV(ii, j);
    return 0;
}

/* ============================ encoder ============================================================ */
/* Matlab/ErasureCodes_NonBinaryLDPCSim.m:174-182 */
void oracle_ldpc_encode(const oracle_code *c, const uint8_t *source, uint8_t *codeword)
{
    const oracle_gf *gf = oracle_gf_default();
    memset(codeword, 0, (size_t)c->n);          /* :174 */
    memcpy(codeword, source, (size_t)c->k);     /* :175 */
    for (int pp = 0; pp < c->m; pp++) {         /* :176 */
        int gf_sum = 0;                         /* :177 */
        for (int ll = 1; ll <= VV(pp, 0) - 1; ll++) /* :178 all but the last (diagonal) entry */
            gf_sum = GF_ADD(gf_sum, GF_MUL(codeword[VL(pp, ll) - 1], VV(pp, ll))); /* :179 */
        codeword[c->k + pp] = GF_MUL(gf_sum, GF_INV(VV(pp, VV(pp, 0)))); /* :181 */
    }
}

void oracle_ldpc_encode_packets(const oracle_code *c, int S, const uint8_t *source, uint8_t *codeword)
{
    const oracle_gf *gf = oracle_gf_default();
    memset(codeword, 0, (size_t)c->n * S);      /* :174 */
    memcpy(codeword, source, (size_t)c->k * S); /* :175 */
    uint8_t *gf_sum = (uint8_t *)malloc((size_t)S);
    for (int pp = 0; pp < c->m; pp++) {         /* :176 */
        memset(gf_sum, 0, (size_t)S);           /* :177 */
        for (int ll = 1; ll <= VV(pp, 0) - 1; ll++) { /* :178-179 */
            const uint8_t *row = gf->mult[VV(pp, ll)];
            const uint8_t *src = &codeword[(size_t)(VL(pp, ll) - 1) * S];
            for (int l = 0; l < S; l++) gf_sum[l] = GF_ADD(gf_sum[l], row[src[l]]);
        }
        const uint8_t *irow = gf->mult[GF_INV(VV(pp, VV(pp, 0)))]; /* :181 */
        uint8_t *dst = &codeword[(size_t)(c->k + pp) * S];
        for (int l = 0; l < S; l++) dst[l] = irow[gf_sum[l]];
    }
    free(gf_sum);
}

/* ============================ hybrid MP + ML decoder, lane-vectorised ============================ */
/* Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m.  y[n*S] with flag era[n] standing for the
 * value -1; S = 1 is the reference exactly, S > 1 runs the same statements on S independent byte
 * lanes that share the erasure pattern.  binary = 1 selects the GF(2) siblings' arithmetic
 * (mod(sum,2) == XOR on 0/1 values, coefficient 1) -- see the wrappers below. */
typedef struct {
    int num_cur_erasures, ml_ran, dont_do_jordan;
} dec_info;

static void lane_axpy(const oracle_gf *gf, uint8_t *dst, int mult, const uint8_t *src, int S)
{
    /* dst(l) = GF_add(dst(l), GF_mult(mult, src(l))) for every lane */
    const uint8_t *row = gf->mult[mult];
    for (int l = 0; l < S; l++) dst[l] = GF_ADD(dst[l], row[src[l]]);
}

static void lane_scale(const oracle_gf *gf, uint8_t *dst, int mult, int S)
{
    const uint8_t *row = gf->mult[mult];
    for (int l = 0; l < S; l++) dst[l] = row[dst[l]];
}

static int hybrid_decode_core(const oracle_code *c, int S, uint8_t *y, uint8_t *era, int itenum,
                              int do_ML_decode, int *iterations, dec_info *info)
{
    const oracle_gf *gf = oracle_gf_default();
    const int n = c->n, k = c->k, m = c->m;
    int stopsig = 0;  /* :16 */
    int itestep = 0;  /* :17 */
    int num_cur_erasures = 0;
    uint8_t *gf_sum = (uint8_t *)malloc((size_t)S);
    info->ml_ran = 0;
    info->dont_do_jordan = 0;

    while (stopsig == 0 && itestep < itenum) { /* :21 */
        itestep = itestep + 1;                 /* :23 */
        for (int ii = 0; ii < m; ii++) {       /* :27 */
            int num_erasures = 0;              /* :28 */
            int erasure_ind = 0;               /* :29 */
            for (int jj = 1; jj <= VL(ii, 0); jj++) { /* :30 */
                if (era[VL(ii, jj) - 1]) {            /* :31 y_current(...) == -1 */
                    num_erasures = num_erasures + 1;  /* :32 */
                    erasure_ind = VL(ii, jj);         /* :33 last erased neighbour */
                }
            }
            if (num_erasures == 1) { /* :37 */
                /* :39 check_indices = setxor(neighbours, erasure_ind): ascending, erasure_ind removed */
                memset(gf_sum, 0, (size_t)S); /* :40 */
                for (int kk = 1; kk <= VL(ii, 0); kk++) { /* :41 */
                    int ci = VL(ii, kk);
                    if (ci == erasure_ind) continue;
                    lane_axpy(gf, gf_sum, h_at(c, ii, ci), &y[(size_t)(ci - 1) * S], S); /* :45 */
                }
                int invh = GF_INV(h_at(c, ii, erasure_ind)); /* :47 */
                uint8_t *dst = &y[(size_t)(erasure_ind - 1) * S];
                for (int l = 0; l < S; l++) dst[l] = GF_MUL(gf_sum[l], invh);
                era[erasure_ind - 1] = 0;
            }
        }
        num_cur_erasures = 0; /* :51 over ALL n symbols */
        for (int j = 0; j < n; j++) num_cur_erasures += era[j] ? 1 : 0;
        if (num_cur_erasures == 0) stopsig = 1; /* :52-54 */
    }
    info->num_cur_erasures = num_cur_erasures;
    *iterations = itestep; /* :130 */

    int rc = 0;
    if (num_cur_erasures > 0 && do_ML_decode == 1) { /* :61 */
        info->ml_ran = 1;
        /* :63-64 erasure_ind = find(y_current == -1), ascending */
        int num_erasures = num_cur_erasures;
        if (num_erasures > m) { /* Matlab would fail at :127 (rhs has n-k entries) */
            free(gf_sum);
            return -2;
        }
        int *erasure_ind = (int *)malloc(sizeof(int) * (size_t)num_erasures);
        for (int j = 0, t = 0; j < n; j++)
            if (era[j]) erasure_ind[t++] = j + 1;
        /* :65 find_inv = H_sparse(:, erasure_ind)  (m x num_erasures, dense here) */
        const int E = num_erasures;
        uint8_t *find_inv = (uint8_t *)calloc((size_t)m * E, 1);
#define FI(r, cc) find_inv[(size_t)(r) * E + (cc)]
        for (int r = 0; r < m; r++)
            for (int t = 0; t < E; t++) FI(r, t) = (uint8_t)h_at(c, r, erasure_ind[t]);
        /* :72-82 rhs(kk) = sum over the known neighbours of row kk */
        uint8_t *rhs = (uint8_t *)calloc((size_t)m * S, 1); /* :74 zeros(n-k,1) */
        for (int kk = 0; kk < n - k; kk++) { /* :75 */
            memset(gf_sum, 0, (size_t)S);    /* :77 */
            for (int ll = 1; ll <= VL(kk, 0); ll++) { /* :76,78 intersect(neighbours, non_erasure_ind) */
                int ci = VL(kk, ll);
                if (era[ci - 1]) continue;
                lane_axpy(gf, gf_sum, h_at(c, kk, ci), &y[(size_t)(ci - 1) * S], S); /* :79 */
            }
            memcpy(&rhs[(size_t)kk * S], gf_sum, (size_t)S); /* :81 */
        }
        int dont_do_jordan = 0; /* :83 */
        int *non_zero_ind = (int *)malloc(sizeof(int) * (size_t)m);
        int *non_zero_indices = (int *)malloc(sizeof(int) * (size_t)E);
        uint8_t *temp_row = (uint8_t *)malloc((size_t)(E > S ? E : S));
        for (int col = 0; col < E; col++) { /* :85 */
            /* :86 non_zero_ind = find(find_inv(col:end, col)) + col - 1 */
            int nnzr = 0;
            for (int r = col; r < m; r++)
                if (FI(r, col)) non_zero_ind[nnzr++] = r;
            if (nnzr == 0) { /* :87-90 */
                dont_do_jordan = 1;
                break;
            }
            int p = non_zero_ind[0];
            /* :92-94 swap rhs(col) <-> rhs(p) */
            memcpy(temp_row, &rhs[(size_t)col * S], (size_t)S);
            memcpy(&rhs[(size_t)col * S], &rhs[(size_t)p * S], (size_t)S);
            memcpy(&rhs[(size_t)p * S], temp_row, (size_t)S);
            /* :95-97 swap rows */
            memcpy(temp_row, &FI(col, 0), (size_t)E);
            memcpy(&FI(col, 0), &FI(p, 0), (size_t)E);
            memcpy(&FI(p, 0), temp_row, (size_t)E);
            /* :99 non_zero_indices = find(find_inv(col,:)) */
            int nzi = 0;
            for (int t = 0; t < E; t++)
                if (FI(col, t)) non_zero_indices[nzi++] = t;
            int multiplier = GF_INV(FI(col, non_zero_indices[0])); /* :100 */
            FI(col, col) = GF_MUL(FI(col, col), multiplier);       /* :101 */
            lane_scale(gf, &rhs[(size_t)col * S], multiplier, S);  /* :102 */
            for (int kk = 1; kk < nzi; kk++)                       /* :103-105 */
                FI(col, non_zero_indices[kk]) = GF_MUL(FI(col, non_zero_indices[kk]), multiplier);
            /* :107-114 zero out the other non-zero rows below the diagonal */
            for (int ii = 1; ii < nnzr; ii++) {
                int r = non_zero_ind[ii];
                multiplier = FI(r, col); /* :109 */
                /* :108,110-112 over union(find(row r), find(row col)) == every column where either is non-zero */
                for (int ll = 0; ll < E; ll++)
                    if (FI(r, ll) || FI(col, ll)) FI(r, ll) = GF_ADD(FI(r, ll), GF_MUL(multiplier, FI(col, ll)));
                lane_axpy(gf, &rhs[(size_t)r * S], multiplier, &rhs[(size_t)col * S], S); /* :113 */
            }
        }
        if (!dont_do_jordan) { /* :117 */
            for (int col = E - 1; col >= 1; col--) { /* :118 col = num_erasures:-1:2 */
                for (int r = 0; r <= col - 1; r++) { /* :119 find(find_inv(1:col-1, col)) */
                    if (!FI(r, col)) continue;
                    lane_axpy(gf, &rhs[(size_t)r * S], FI(r, col), &rhs[(size_t)col * S], S); /* :122 */
                    FI(r, col) = 0; /* :123 */
                }
            }
        }
        /* :127 y_current(erasure_ind) = rhs(1:num_erasures)  -- unconditional */
        for (int t = 0; t < E; t++) {
            memcpy(&y[(size_t)(erasure_ind[t] - 1) * S], &rhs[(size_t)t * S], (size_t)S);
            era[erasure_ind[t] - 1] = 0;
        }
        info->dont_do_jordan = dont_do_jordan;
#undef FI
        free(temp_row); free(non_zero_indices); free(non_zero_ind);
        free(rhs); free(find_inv); free(erasure_ind);
    }
    free(gf_sum);
    return rc;
}

static void fill_info(int *info, const dec_info *di)
{
    if (!info) return;
    info[0] = di->num_cur_erasures;
    info[1] = di->ml_ran;
    info[2] = di->dont_do_jordan;
}

int oracle_ldpc_hybridml_nonbinary_decode(const oracle_code *c, const int16_t *recv, int itenum,
                                          int do_ml, int16_t *msg, int *iterations, int *info)
{
    const int n = c->n;
    uint8_t *y = (uint8_t *)malloc((size_t)n);
    uint8_t *era = (uint8_t *)malloc((size_t)n);
    for (int j = 0; j < n; j++) { /* :10 y_current = recv_vec_val */
        era[j] = recv[j] == -1;
        y[j] = era[j] ? 0 : (uint8_t)recv[j];
    }
    dec_info di;
    int rc = hybrid_decode_core(c, 1, y, era, itenum, do_ml, iterations, &di);
    for (int j = 0; j < n; j++) msg[j] = era[j] ? -1 : (int16_t)y[j]; /* :129 */
    fill_info(info, &di);
    free(y); free(era);
    return rc;
}

int oracle_ldpc_hybridml_nonbinary_decode_packets(const oracle_code *c, int S, const uint8_t *sym,
                                                  const uint8_t *erased, int itenum, int do_ml,
                                                  uint8_t *out, uint8_t *out_erased, int *iterations,
                                                  int *info)
{
    const int n = c->n;
    uint8_t *era = (uint8_t *)malloc((size_t)n);
    for (int j = 0; j < n; j++) {
        era[j] = erased[j] != 0;
        if (era[j]) memset(&out[(size_t)j * S], 0, (size_t)S);
        else memcpy(&out[(size_t)j * S], &sym[(size_t)j * S], (size_t)S);
    }
    dec_info di;
    int rc = hybrid_decode_core(c, S, out, era, itenum, do_ml, iterations, &di);
    for (int j = 0; j < n; j++)
        if (era[j]) memset(&out[(size_t)j * S], 0, (size_t)S);
    if (out_erased) memcpy(out_erased, era, (size_t)n);
    fill_info(info, &di);
    free(era);
    return rc;
}

int oracle_ldpc_decode_batch_s1(const oracle_code *c, int nframes, const uint8_t *sym,
                                const uint8_t *erased, int itenum, int do_ml, uint8_t *out,
                                int32_t *sweeps, int32_t *residual, int32_t *status)
{
    const int n = c->n;
    for (int f = 0; f < nframes; f++) {
        int it = 0, info[3];
        int rc = oracle_ldpc_hybridml_nonbinary_decode_packets(
            c, 1, sym + (size_t)f * n, erased + (size_t)f * n, itenum, do_ml, out + (size_t)f * n, NULL,
            &it, info);
        sweeps[f] = it;
        residual[f] = info[0];
        /* status: 0 MP finished, 1 ML solved, 2 ML rank deficient, 3 ML not run */
        if (info[0] == 0) status[f] = 0;
        else if (rc == -2 || !info[1]) status[f] = 3;
        else status[f] = info[2] ? 2 : 1;
    }
    return 0;
}

/* ============================ binary siblings ==================================================== */
/* Matlab/My_LDPC_Erasure_Decoder.m */
int oracle_ldpc_binary_mp_decode(const oracle_code *c, const int16_t *recv, int itenum, int16_t *msg,
                                 int *iterations)
{
    const int n = c->n, m = c->m;
    int16_t *y_current = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    memcpy(y_current, recv, sizeof(int16_t) * (size_t)n); /* :7 */
    int stopsig = 0, itestep = 0;                         /* :14-15 */
    while (stopsig == 0 && itestep < itenum) {            /* :19 */
        itestep = itestep + 1;
        for (int ii = 0; ii < m; ii++) { /* :25 */
            int num_erasures = 0, erasure_ind = 0;
            for (int jj = 1; jj <= VL(ii, 0); jj++)
                if (y_current[VL(ii, jj) - 1] == -1) { /* :29 */
                    num_erasures = num_erasures + 1;
                    erasure_ind = VL(ii, jj);
                }
            if (num_erasures == 1) { /* :35-36 mod(sum(others), 2) */
                int s = 0;
                for (int jj = 1; jj <= VL(ii, 0); jj++)
                    if (VL(ii, jj) != erasure_ind) s += y_current[VL(ii, jj) - 1];
                y_current[erasure_ind - 1] = (int16_t)(s % 2);
            }
        }
        int num_cur_erasures = 0; /* :40 */
        for (int j = 0; j < n; j++) num_cur_erasures += y_current[j] == -1;
        if (num_cur_erasures == 0) stopsig = 1;
    }
    memcpy(msg, y_current, sizeof(int16_t) * (size_t)n); /* :49 */
    *iterations = itestep;                               /* :50 */
    free(y_current);
    return 0;
}

/* Matlab/My_LDPC_HybridML_Erasure_Decoder.m (GF(2): H_sparse is the 0/1 structure) */
int oracle_ldpc_binary_hybridml_decode(const oracle_code *c, const int16_t *recv, int itenum,
                                       int16_t *msg, int *iterations, int *info)
{
    const int n = c->n, m = c->m;
    int16_t *y_current = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    int num_cur_erasures = 0;
    /* :6-46 identical to My_LDPC_Erasure_Decoder with itenum = 10 */
    oracle_ldpc_binary_mp_decode(c, recv, itenum, y_current, iterations);
    for (int j = 0; j < n; j++) num_cur_erasures += y_current[j] == -1;
    if (info) { info[0] = num_cur_erasures; info[1] = 0; info[2] = 0; }
    if (num_cur_erasures > 0) { /* :48 */
        if (num_cur_erasures > m) { memcpy(msg, y_current, sizeof(int16_t) * (size_t)n); free(y_current); return -2; }
        const int E = num_cur_erasures;
        int *erasure_ind = (int *)malloc(sizeof(int) * (size_t)E); /* :50 */
        for (int j = 0, t = 0; j < n; j++)
            if (y_current[j] == -1) erasure_ind[t++] = j + 1;
        uint8_t *find_inv = (uint8_t *)calloc((size_t)m * E, 1); /* :52 */
#define FI(r, cc) find_inv[(size_t)(r) * E + (cc)]
        for (int r = 0; r < m; r++)
            for (int t = 0; t < E; t++) FI(r, t) = h_at(c, r, erasure_ind[t]) ? 1 : 0;
        uint8_t *rhs = (uint8_t *)calloc((size_t)m, 1); /* :54 mod(H(:,known)*y(known)', 2) */
        for (int r = 0; r < m; r++) {
            int s = 0;
            for (int jj = 1; jj <= VL(r, 0); jj++)
                if (y_current[VL(r, jj) - 1] != -1) s += y_current[VL(r, jj) - 1];
            rhs[r] = (uint8_t)(s % 2);
        }
        int dont_do_jordan = 0; /* :55 */
        for (int col = 0; col < E; col++) { /* :57 */
            int p = -1; /* :58 first non-zero row >= col */
            for (int r = col; r < m; r++)
                if (FI(r, col)) { p = r; break; }
            if (p < 0) { dont_do_jordan = 1; break; } /* :59-62 */
            uint8_t tv = rhs[col]; rhs[col] = rhs[p]; rhs[p] = tv; /* :64-66 */
            for (int t = 0; t < E; t++) { uint8_t tr = FI(col, t); FI(col, t) = FI(p, t); FI(p, t) = tr; } /* :67-69 */
            for (int r = p + 1; r < m; r++) { /* :71-74 the other non-zero rows (found before the swap) */
                if (!FI(r, col)) continue;
                for (int t = 0; t < E; t++) FI(r, t) = (uint8_t)((FI(r, t) + FI(col, t)) % 2);
                rhs[r] = (uint8_t)((rhs[r] + rhs[col]) % 2);
            }
        }
        if (!dont_do_jordan) { /* :77 */
            for (int col = E - 1; col >= 1; col--)      /* :78 */
                for (int r = 0; r <= col - 1; r++) {    /* :79 */
                    if (!FI(r, col)) continue;
                    for (int t = 0; t < E; t++) FI(r, t) = (uint8_t)((FI(r, t) + FI(col, t)) % 2); /* :82 */
                    rhs[r] = (uint8_t)((rhs[r] + rhs[col]) % 2);                                   /* :83 */
                }
        }
        for (int t = 0; t < E; t++) y_current[erasure_ind[t] - 1] = rhs[t]; /* :87 */
        if (info) { info[1] = 1; info[2] = dont_do_jordan; }
#undef FI
        free(rhs); free(find_inv); free(erasure_ind);
    }
    memcpy(msg, y_current, sizeof(int16_t) * (size_t)n); /* :89 */
    free(y_current);
    return 0;
}

/* ============================ Reed-Solomon ======================================================= */
/* Matlab/Test_My_RS_Decode.m:22,30-37 */
int oracle_rs_generator(int n, int k, uint8_t *g)
{
    const oracle_gf *gf = oracle_gf_default();
    /* :30-34 G(row,col) = alpha^(col*row), alpha = x_gf(3) = 2: alpha^e = antilog[(e mod 255) + 1] */
    uint8_t *G = (uint8_t *)malloc((size_t)k * n);
    for (int row = 1; row <= k; row++)
        for (int col = 1; col <= n; col++) G[(size_t)(row - 1) * n + (col - 1)] = gf->antilog[((row * col) % 255) + 1];
    /* :36 G_k_inv = inv(G(1:k,1:k)) by Gauss-Jordan on [A | I] (the inverse is unique) */
    uint8_t *A = (uint8_t *)malloc((size_t)k * k), *I = (uint8_t *)calloc((size_t)k * k, 1);
    for (int r = 0; r < k; r++) {
        memcpy(&A[(size_t)r * k], &G[(size_t)r * n], (size_t)k);
        I[(size_t)r * k + r] = 1;
    }
    for (int col = 0; col < k; col++) {
        int p = -1;
        for (int r = col; r < k; r++)
            if (A[(size_t)r * k + col]) { p = r; break; }
        if (p < 0) { free(G); free(A); free(I); return -1; }
        if (p != col)
            for (int t = 0; t < k; t++) {
                uint8_t x = A[(size_t)col * k + t]; A[(size_t)col * k + t] = A[(size_t)p * k + t]; A[(size_t)p * k + t] = x;
                x = I[(size_t)col * k + t]; I[(size_t)col * k + t] = I[(size_t)p * k + t]; I[(size_t)p * k + t] = x;
            }
        int iv = GF_INV(A[(size_t)col * k + col]);
        for (int t = 0; t < k; t++) {
            A[(size_t)col * k + t] = GF_MUL(A[(size_t)col * k + t], iv);
            I[(size_t)col * k + t] = GF_MUL(I[(size_t)col * k + t], iv);
        }
        for (int r = 0; r < k; r++) {
            if (r == col) continue;
            int f = A[(size_t)r * k + col];
            if (!f) continue;
            for (int t = 0; t < k; t++) {
                A[(size_t)r * k + t] = GF_ADD(A[(size_t)r * k + t], GF_MUL(f, A[(size_t)col * k + t]));
                I[(size_t)r * k + t] = GF_ADD(I[(size_t)r * k + t], GF_MUL(f, I[(size_t)col * k + t]));
            }
        }
    }
    /* :37 G = G_k_inv * G */
    for (int r = 0; r < k; r++)
        for (int col = 0; col < n; col++) {
            int s = 0;
            for (int t = 0; t < k; t++) s = GF_ADD(s, GF_MUL(I[(size_t)r * k + t], G[(size_t)t * n + col]));
            g[(size_t)r * n + col] = (uint8_t)s;
        }
    free(G); free(A); free(I);
    return 0;
}

/* Matlab/Test_My_RS_Decode.m:48  source_encode_vec = source_vec*G */
void oracle_rs_encode(int n, int k, const uint8_t *g, const uint8_t *source, uint8_t *codeword)
{
    const oracle_gf *gf = oracle_gf_default();
    for (int col = 0; col < n; col++) {
        int s = 0;
        for (int r = 0; r < k; r++) s = GF_ADD(s, GF_MUL(source[r], g[(size_t)r * n + col]));
        codeword[col] = (uint8_t)s;
    }
}

/* Matlab/My_RS_Decode_Optimize_With_GFTables.m */
int oracle_rs_decode(int n, int k, const uint8_t *g, const uint16_t *recv_ind0, const uint8_t *recv_val,
                     uint8_t *msg)
{
    const oracle_gf *gf = oracle_gf_default();
    int rank_deficient = 0;
    /* 1-based working copies so that the statements below read like the reference */
    int *recv_vec_ind = (int *)malloc(sizeof(int) * (size_t)(k + 1));
    int *acc = (int *)malloc(sizeof(int) * (size_t)(k + 1)); /* repair_multiply_accumulator */
    int *bit_order_vec = (int *)malloc(sizeof(int) * (size_t)(k + 1));
    uint8_t *GJ = (uint8_t *)calloc((size_t)(k + 1) * (k + 1), 1);
#define GJ_mat(r, cc) GJ[(size_t)(r) * (k + 1) + (cc)]
#define Gm(r, cc) g[(size_t)((r) - 1) * n + ((cc) - 1)]
    for (int ii = 1; ii <= k; ii++) recv_vec_ind[ii] = recv_ind0[ii - 1] + 1;
    /* :19-23 GJ_mat(ii,:) = G(:, recv_vec_ind(ii)) */
    for (int ii = 1; ii <= k; ii++)
        for (int t = 1; t <= k; t++) GJ_mat(ii, t) = Gm(t, recv_vec_ind[ii]);
    /* :29 */
    int num_sys_symbols = 0;
    for (int ii = 1; ii <= k; ii++) num_sys_symbols += recv_vec_ind[ii] <= k;
    for (int ii = 1; ii <= k; ii++) bit_order_vec[ii] = ii; /* :30 */
    for (int ii = 1; ii <= num_sys_symbols; ii++) {          /* :33 */
        int col_ind = 0, ind = 1;                            /* :34-35 */
        while (col_ind == 0) {                               /* :36-41 */
            if (GJ_mat(ii, ind) != 0) col_ind = ind;
            ind = ind + 1;
        }
        for (int r = 1; r <= k; r++) { /* :42-44 swap columns ii and col_ind */
            uint8_t temp = GJ_mat(r, ii);
            GJ_mat(r, ii) = GJ_mat(r, col_ind);
            GJ_mat(r, col_ind) = temp;
        }
        int temp_col_ind = bit_order_vec[ii]; /* :45-47 */
        bit_order_vec[ii] = col_ind;
        bit_order_vec[col_ind] = temp_col_ind;
    }
    for (int ii = 1; ii <= k; ii++) acc[ii] = recv_val[ii - 1]; /* :51 */
    int row_index = num_sys_symbols + 1;                         /* :52 */
    int swap_ind = row_index + 1;                                /* :53 */
    int NotDone = 1;                                             /* :54 */
    uint8_t *row_temp = (uint8_t *)malloc((size_t)(k + 1));
    while (row_index <= k && NotDone == 1) { /* :55 */
        for (int jj = 1; jj <= num_sys_symbols; jj++) { /* :57-60 */
            acc[row_index] = GF_ADD(acc[row_index], GF_MUL(GJ_mat(row_index, jj), acc[jj]));
            GJ_mat(row_index, jj) = 0;
        }
        for (int jj = num_sys_symbols + 1; jj <= row_index - 1; jj++) { /* :61-67 */
            acc[row_index] = GF_ADD(acc[row_index], GF_MUL(GJ_mat(row_index, jj), acc[jj]));
            int row_multiplier = GJ_mat(row_index, jj); /* :63 */
            for (int ll = jj; ll <= k; ll++)            /* :64-66 */
                GJ_mat(row_index, ll) = GF_ADD(GJ_mat(row_index, ll), GF_MUL(row_multiplier, GJ_mat(jj, ll)));
        }
        if (GJ_mat(row_index, row_index) != 0) { /* :70 */
            int GF_mult = GF_INV(GJ_mat(row_index, row_index)); /* :71 */
            for (int ll = row_index; ll <= k; ll++)             /* :72-74 */
                GJ_mat(row_index, ll) = GF_MUL(GF_mult, GJ_mat(row_index, ll));
            acc[row_index] = GF_MUL(GF_mult, acc[row_index]); /* :75 */
            row_index = row_index + 1;                        /* :76 */
            swap_ind = row_index + 1;                         /* :77 */
        } else {                                              /* :78 */
            if (swap_ind > k) {
                NotDone = 0; /* :80 */
            } else {
                memcpy(row_temp, &GJ_mat(row_index, 0), (size_t)(k + 1)); /* :82-84 */
                memcpy(&GJ_mat(row_index, 0), &GJ_mat(swap_ind, 0), (size_t)(k + 1));
                memcpy(&GJ_mat(swap_ind, 0), row_temp, (size_t)(k + 1));
                int t = acc[row_index]; /* :85-87 */
                acc[row_index] = acc[swap_ind];
                acc[swap_ind] = t;
                swap_ind = swap_ind + 1; /* :88 */
            }
        }
    }
    if (row_index <= k) rank_deficient = 1; /* :95-97 (reference: empty body) */
    for (int ii = k - 1; ii >= num_sys_symbols + 1; ii--) /* :100 */
        for (int jj = ii + 1; jj <= k; jj++) {            /* :101-104 */
            acc[ii] = GF_ADD(acc[ii], GF_MUL(acc[jj], GJ_mat(ii, jj)));
            GJ_mat(ii, jj) = 0;
        }
    /* :110-116 */
    memset(msg, 0, (size_t)k);
    for (int ii = 1; ii <= num_sys_symbols; ii++) msg[bit_order_vec[ii] - 1] = recv_val[ii - 1];
    for (int ii = num_sys_symbols + 1; ii <= k; ii++) msg[bit_order_vec[ii] - 1] = (uint8_t)acc[ii];
#undef GJ_mat
#undef Gm
    free(row_temp); free(GJ); free(bit_order_vec); free(acc); free(recv_vec_ind);
    return rank_deficient;
}

/* ============================ channel models ===================================================== */
/* Matlab/Bursty_Error_Channel_Model_Generator.m */
int oracle_bursty_channel_step(int current_state, double alpha, double beta, double good_transition_bias,
                               double rand_num, double state_rand_num, int *next_state)
{
    double transition = 0.1;                                   /* :16 */
    double Prob_1_given_0 = transition / good_transition_bias; /* :19 */
    double Prob_0_given_1 = transition;                        /* :20 */
    int error_out = 0;                                         /* :24 */
    if (current_state == 0) {                                  /* :27 */
        if (rand_num <= alpha) error_out = 1;                  /* :28-30 */
        if (state_rand_num <= Prob_1_given_0) *next_state = 1; /* :32-33 */
        else *next_state = current_state;                      /* :35 */
    } else {
        if (rand_num <= beta) error_out = 1;                   /* :38-40 */
        if (state_rand_num <= Prob_0_given_1) *next_state = 0; /* :42-43 */
        else *next_state = current_state;                      /* :45 */
    }
    return error_out;
}

/* ============================ synthetic inputs =================================================== */
void oracle_synth_coefs(uint64_t seed, int nnz, uint8_t *coefs)
{
    for (int i = 0; i < nnz; i++) coefs[i] = ldpc_synth_nonzero(seed, LDPC_SYNTH_STREAM_COEF, (uint64_t)i);
}

void oracle_synth_source(uint64_t seed, int64_t frame0, int nframes, int k, int S, uint8_t *src)
{
    uint64_t base = (uint64_t)frame0 * (uint64_t)k * (uint64_t)S;
    uint64_t total = (uint64_t)nframes * (uint64_t)k * (uint64_t)S;
    for (uint64_t i = 0; i < total; i++) src[i] = ldpc_synth_byte(seed, LDPC_SYNTH_STREAM_SOURCE, base + i);
}

void oracle_synth_erasures_uniform(uint64_t seed, int64_t frame0, int nframes, int n, double per,
                                   uint8_t *erased)
{
    /* LDPCErasureCodes_MessagePassingAlgSim.m:183-188: erased iff rand(1) <= PER */
    uint64_t base = (uint64_t)frame0 * (uint64_t)n;
    uint64_t total = (uint64_t)nframes * (uint64_t)n;
    for (uint64_t i = 0; i < total; i++)
        erased[i] = ldpc_synth_uniform(seed, LDPC_SYNTH_STREAM_ERASE, base + i) <= per;
}

void oracle_synth_erasures_bursty(uint64_t seed, int64_t frame0, int nframes, int n, double alpha,
                                  double beta, double good_transition_bias, uint8_t *erased)
{
    /* ErasureCodes_NonBinaryLDPCSim.m:163 next_state = 0 once, then :191-198 per symbol */
    int next_state = 0;
    uint64_t first = (uint64_t)frame0 * (uint64_t)n;
    uint64_t last = first + (uint64_t)nframes * (uint64_t)n;
    for (uint64_t i = 0; i < last; i++) {
        double r1 = ldpc_synth_uniform(seed, LDPC_SYNTH_STREAM_BURST_E, i);
        double r2 = ldpc_synth_uniform(seed, LDPC_SYNTH_STREAM_BURST_S, i);
        int e = oracle_bursty_channel_step(next_state, alpha, beta, good_transition_bias, r1, r2, &next_state);
        if (i >= first) erased[i - first] = (uint8_t)e;
    }
}

/* ============================ FPGA source kernel ================================================== */
/* OpenCL/device/ldpc_erasure_decoder_top.cl:57-120 (data_in), erasure flags only */
void oracle_threefry4x32_20(const uint32_t ctr[4], const uint32_t key[4], uint32_t out[4])
{
    ldpc_threefry4x32_20(ctr, key, out);
}

/* OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:56-236, one pass of the while(1) frame loop */
int oracle_fpga_perf_decoder_frame(const oracle_code *c, int num_iter, uint8_t *is_erasure, uint64_t *payload,
                                   int *iterations)
{
    const int n_ldpc = c->n, k_ldpc = c->k;
    const int num_parity_checks = n_ldpc - k_ldpc; /* :62 */
    /* :60-61,68-69 codeword / codeword2 start as the same received frame */
    uint8_t *er1 = (uint8_t *)malloc((size_t)n_ldpc), *er2 = (uint8_t *)malloc((size_t)n_ldpc);
    uint64_t *sy1 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n_ldpc), *sy2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n_ldpc);
    for (int ii = 0; ii < n_ldpc; ii++) {
        er1[ii] = er2[ii] = is_erasure[ii];
        sy1[ii] = sy2[ii] = payload ? payload[ii] : 0;
    }
    int iter_ind = 0, stop_sig = 0; /* :86-87 */
    while (iter_ind < num_iter && stop_sig == 0) { /* :88 */
        for (int half = 0; half < 2; half++) {     /* :98-137 on codeword, :139-176 on codeword2 */
            uint8_t *er = half ? er2 : er1;
            uint64_t *sy = half ? sy2 : sy1;
            const int k0 = half ? num_parity_checks / 2 : 0, k1 = half ? num_parity_checks : num_parity_checks / 2;
            for (int kk = k0; kk < k1; kk++) {
                uint64_t parity_accumulator = 0;   /* :102-108 */
                unsigned char num_erasures = 0;
                unsigned int erasure_ind = 0;
                for (int ii = 0; ii < VL(kk, 0); ii++) { /* :112 */
                    const int col = VL(kk, ii + 1) - 1;  /* 1-based list (:114) */
                    parity_accumulator ^= sy[col];       /* :116-119: erased symbols are XORed too (stored as zeros) */
                    if (er[col] == 1) {                  /* :120-124 */
                        num_erasures = (unsigned char)(num_erasures + er[col]);
                        erasure_ind = (unsigned int)col;
                    }
                }
                if (num_erasures == 1) { /* :126-135 */
                    er[erasure_ind] = 0;
                    sy[erasure_ind] = parity_accumulator;
                }
            }
        }
        int num_current_correct = 0; /* :178 */
        for (int ii = 0; ii < n_ldpc; ii++) {
            if (er1[ii] + er2[ii] == 1) { /* :180 one copy (not both) has it */
                if (er1[ii]) { sy1[ii] = sy2[ii]; er1[ii] = 0; }
                else { sy2[ii] = sy1[ii]; er2[ii] = 0; }
                num_current_correct += 1; /* :197 */
            } else if ((er1[ii] + er2[ii] == 0) && ii < k_ldpc) { /* :199 */
                num_current_correct += 1;
            }
        }
        if (num_current_correct == k_ldpc) stop_sig = 1; /* :205-207 */
        iter_ind += 1;
    }
    int num_final_erasures = 0; /* :213-220 */
    for (int ii = 0; ii < k_ldpc; ii++)
        if (er1[ii] == 1) num_final_erasures += 1;
    for (int ii = 0; ii < n_ldpc; ii++) {
        is_erasure[ii] = er1[ii];
        if (payload) payload[ii] = sy1[ii];
    }
    if (iterations) *iterations = iter_ind;
    free(er1); free(er2); free(sy1); free(sy2);
    return num_final_erasures;
}

void oracle_fpga_data_in_erasures(int seed, int per_numerator_div_64, int64_t count, uint8_t *erased)
{
    const uint32_t key[4] = {1u /* tid, :69 */, (uint32_t)seed /* useed, :68 */, 0u, 0u}; /* :74 */
    uint32_t c[4] = {0u, 0u, 0u, 0u};                                                    /* :75 */
    for (int64_t i = 0; i < count; i++) { /* :84,89 itr over frames, k over symbols */
        uint32_t u[4];
        c[0]++;                           /* :96 */
        ldpc_threefry4x32_20(c, key, u);  /* :97 */
        long rv = (long)(int32_t)u[0];    /* :98 long rv = u.i.x */
        erased[i] = (rv & 0x3F) < per_numerator_div_64 ? 1 : 0; /* :105-110 */
    }
}
