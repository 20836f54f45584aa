/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C CPU restatement of the reference's recode -> k-mer window -> count -> cosine path
 * for sizes the pure-Python restatement (oracle/ref_path.py) cannot reach in seconds.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library, and
 * only as the checker.  Nothing under snekmer_amd/ links or loads it.
 *
 * Parity status: PINNED via tests/test_oracle_golden.py, which checks every function here
 * against fixtures produced by the imported reference (tests/golden/make_golden.py).
 *
 * Integer conventions shared with the device path: class letters are
 * ranked in ASCII order, code = sum rank_i * nsym^(k-1-i), so numeric code order equals
 * lexicographic k-mer string order.  Codes are carried as uint64 here regardless of width.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INVALID 0xFF
#define ORC_SENTINEL UINT64_MAX

/* snekmer/vectorize.py:193-195: rstrip("*") then translate; length shrinks by the number of
 * trailing '*'.  Output keeps the input offsets; outlen[i] is the stripped length. */
void orc_recode(const uint8_t *table, const uint8_t *seq, const int64_t *off, int64_t n,
                uint8_t *out, int32_t *outlen)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t b = off[i], e = off[i + 1];
        while (e > b && seq[e - 1] == '*')
            --e;
        for (int64_t p = b; p < e; ++p)
            out[p] = table[seq[p]];
        outlen[i] = (int32_t)(e - b);
    }
}

/* snekmer/vectorize.py:239-249 + :292-328: one slot per window start p in [0, len-k]; slot holds
 * the window's code when all k translated characters are class letters, else the sentinel. */
void orc_kmer_codes(const uint8_t *rank, int nsym, int k, const uint8_t *seq, const int64_t *off,
                    int64_t n, uint64_t *codes, int32_t *nwin)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t b = off[i], e = off[i + 1];
        while (e > b && seq[e - 1] == '*')
            --e;
        int64_t len = e - b, w = len - k + 1;
        nwin[i] = (int32_t)(w > 0 ? w : 0);
        for (int64_t p = 0; p < w; ++p) {
            uint64_t c = 0;
            int ok = 1;
            for (int j = 0; j < k; ++j) {
                uint8_t r = rank[seq[b + p + j]];
                if (r == ORC_INVALID) {
                    ok = 0;
                    break;
                }
                c = c * (uint64_t)nsym + r;
            }
            codes[b + p] = ok ? c : ORC_SENTINEL;
        }
    }
}

typedef struct {
    uint64_t code;
    uint32_t pos;
} orc_win;

static int cmp_win(const void *a, const void *b)
{
    const orc_win *x = (const orc_win *)a, *y = (const orc_win *)b;
    if (x->code != y->code)
        return x->code < y->code ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

/* rules/learn.smk:359-383 / rules/apply.smk:188-206 restricted to valid windows (invalid ones can
 * never match a basis k-mer built by rules/kmerize.smk:89-104): per sequence, distinct codes in
 * ascending order with their multiplicity and the index of their first window.
 * Returns nnz; rowptr has n+1 entries; outputs need capacity >= total windows. */
int64_t orc_count_csr(const uint8_t *rank, int nsym, int k, const uint8_t *seq, const int64_t *off,
                      int64_t n, int64_t *rowptr, uint64_t *codes, uint32_t *counts,
                      uint32_t *firstpos)
{
    int64_t nnz = 0, maxlen = 0;
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] - off[i] > maxlen)
            maxlen = off[i + 1] - off[i];
    orc_win *buf = (orc_win *)malloc(sizeof(orc_win) * (size_t)(maxlen + 1));
    for (int64_t i = 0; i < n; ++i) {
        int64_t b = off[i], e = off[i + 1];
        while (e > b && seq[e - 1] == '*')
            --e;
        int64_t w = (e - b) - k + 1, m = 0;
        rowptr[i] = nnz;
        for (int64_t p = 0; p < w; ++p) {
            uint64_t c = 0;
            int ok = 1;
            for (int j = 0; j < k; ++j) {
                uint8_t r = rank[seq[b + p + j]];
                if (r == ORC_INVALID) {
                    ok = 0;
                    break;
                }
                c = c * (uint64_t)nsym + r;
            }
            if (ok) {
                buf[m].code = c;
                buf[m].pos = (uint32_t)p;
                ++m;
            }
        }
        qsort(buf, (size_t)m, sizeof(orc_win), cmp_win);
        for (int64_t t = 0; t < m;) {
            int64_t u = t;
            while (u < m && buf[u].code == buf[t].code)
                ++u;
            codes[nnz] = buf[t].code;
            counts[nnz] = (uint32_t)(u - t);
            if (firstpos)
                firstpos[nnz] = buf[t].pos;
            ++nnz;
            t = u;
        }
    }
    rowptr[n] = nnz;
    free(buf);
    return nnz;
}

typedef struct {
    uint64_t code;
    int64_t idx;
} orc_ent;

static int cmp_ent(const void *a, const void *b)
{
    const orc_ent *x = (const orc_ent *)a, *y = (const orc_ent *)b;
    if (x->code != y->code)
        return x->code < y->code ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* rules/kmerize.smk:89-104: the observed basis.  Emitted in ascending code order together with,
 * per basis k-mer, document frequency, total occurrences (what min_filter tests) and the
 * first-seen key (row << 32 | first window) whose ascending order is the reference's dict
 * insertion order.  colidx[e] maps CSR entry e to its basis column.  Returns B. */
int64_t orc_basis(const uint64_t *codes, const uint32_t *counts, const uint32_t *firstpos,
                  const int64_t *rowptr, int64_t n, int64_t nnz, uint64_t *basis, uint32_t *df,
                  uint64_t *total, uint64_t *firstkey, uint32_t *colidx)
{
    orc_ent *ent = (orc_ent *)malloc(sizeof(orc_ent) * (size_t)(nnz + 1));
    int64_t *rowof = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
            rowof[e] = i;
    for (int64_t e = 0; e < nnz; ++e) {
        ent[e].code = codes[e];
        ent[e].idx = e;
    }
    qsort(ent, (size_t)nnz, sizeof(orc_ent), cmp_ent);
    int64_t B = 0;
    for (int64_t t = 0; t < nnz;) {
        int64_t u = t;
        uint64_t tot = 0;
        while (u < nnz && ent[u].code == ent[t].code) {
            tot += counts[ent[u].idx];
            colidx[ent[u].idx] = (uint32_t)B;
            ++u;
        }
        basis[B] = ent[t].code;
        df[B] = (uint32_t)(u - t);
        total[B] = tot;
        firstkey[B] = ((uint64_t)rowof[ent[t].idx] << 32) | (firstpos ? firstpos[ent[t].idx] : 0);
        ++B;
        t = u;
    }
    free(ent);
    free(rowof);
    return B;
}

/* sklearn cosine_similarity(X, Y) for count rows held as CSR over a shared column space
 * (call sites rules/apply.smk:282-284, rules/learn.smk:821-823): float64, zero-norm rows give 0.
 * Computes only the requested rows of X against every row of Y: out[r*m + j].
 * The integer Gram entry is exact; g / (|x||y|) differs from sklearn's normalise-then-dot by
 * O(1e-16) (checked against it in tests). */
void orc_cosine_rows(int64_t n, const int64_t *xrowptr, const uint32_t *xcol, const uint32_t *xval,
                     int64_t m, const int64_t *yrowptr, const uint32_t *ycol, const uint32_t *yval,
                     int64_t ncols, const int64_t *rows, int64_t nrows, double *out)
{
    (void)n;
    int64_t ynnz = yrowptr[m];
    int64_t *colptr = (int64_t *)calloc((size_t)(ncols + 2), sizeof(int64_t));
    uint32_t *prow = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(ynnz + 1));
    uint32_t *pval = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(ynnz + 1));
    double *ynorm = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    for (int64_t e = 0; e < ynnz; ++e)
        colptr[ycol[e] + 2]++;
    for (int64_t c = 0; c < ncols; ++c)
        colptr[c + 2] += colptr[c + 1];
    for (int64_t j = 0; j < m; ++j) {
        double s = 0;
        for (int64_t e = yrowptr[j]; e < yrowptr[j + 1]; ++e) {
            int64_t slot = colptr[ycol[e] + 1]++;
            prow[slot] = (uint32_t)j;
            pval[slot] = yval[e];
            s += (double)yval[e] * (double)yval[e];
        }
        ynorm[j] = s > 0 ? sqrt(s) : 1.0;
    }
    int64_t *acc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    for (int64_t r = 0; r < nrows; ++r) {
        int64_t i = rows[r];
        memset(acc, 0, sizeof(int64_t) * (size_t)m);
        double s = 0;
        for (int64_t e = xrowptr[i]; e < xrowptr[i + 1]; ++e) {
            uint32_t c = xcol[e];
            int64_t v = xval[e];
            s += (double)v * (double)v;
            for (int64_t t = colptr[c]; t < colptr[c + 1]; ++t)
                acc[prow[t]] += v * (int64_t)pval[t];
        }
        double xn = s > 0 ? sqrt(s) : 1.0;
        for (int64_t j = 0; j < m; ++j)
            out[r * m + j] = (double)acc[j] / (xn * ynorm[j]);
    }
    free(acc);
    free(ynorm);
    free(pval);
    free(prow);
    free(colptr);
}
