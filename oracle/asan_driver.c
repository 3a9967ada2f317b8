/* TEST INFRASTRUCTURE ONLY (never linked into the product).
 * Drives every entry point of the C oracle on seeded random inputs under AddressSanitizer + UBSan
 * (oracle/Makefile target `asan_driver`; run by tests/test_oracle_asan.py on the CPU).  Besides memory errors it
 * cross-checks the bit-vector search against the plain DP, so a sanitizer-clean run is also a consistency run. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_hw_bruteforce(const uint8_t *q, int m, const uint8_t *t, int n, int k, int *ed, int *n_loc, int *starts, int *ends,
                         int cap, int *first_ins, uint8_t *ops, int ops_cap, int *ops_len);
int oracle_hw_bitvector(const uint8_t *q, int m, const uint8_t *t, int n, int k, int *ed, int *n_loc, int *starts, int *ends,
                        int cap, int *first_ins, uint8_t *ops, int ops_cap, int *ops_len, int want_path);
int oracle_hw_forward_only(const uint8_t *q, int m, const uint8_t *t, int n, int k, int *ed, int *n_loc);
int oracle_chain_quadratic(int n, const int32_t *c, const int32_t *d, const double *val, int thr, int32_t *out_idx, double *value);
int oracle_chain_nlogn(int n, const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d, const double *val, int thr,
                       int32_t *out_idx, double *value);
int oracle_nam_quadratic(int n, const int32_t *y, const int32_t *c, const int32_t *d, int64_t max_intron, int64_t *C, int32_t *trace);
int oracle_nam_nlogn(int n, const int32_t *c, const int32_t *d, int64_t *C, int32_t *trace);

static uint64_t g_state = 0x243F6A8885A308D3ull;
static uint32_t rnd(void)
{
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(g_state >> 33);
}
static int rint_(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

static int cmp_y(const void *a, const void *b)
{
    const int32_t *p = (const int32_t *)a, *q = (const int32_t *)b;
    return p[1] < q[1] ? -1 : p[1] > q[1];
}

int main(void)
{
    int fails = 0;
    /* ---- Myers: exact-size buffers so that any overrun trips the sanitizer ---- */
    for (int it = 0; it < 1500; it++) {
        const int n = rint_(1, 300), m = rint_(1, it % 7 == 0 ? 200 : 70), sigma = it % 5 == 0 ? 2 : 4;
        uint8_t *t = (uint8_t *)malloc((size_t)n), *q = (uint8_t *)malloc((size_t)m);
        for (int i = 0; i < n; i++) t[i] = (uint8_t)("ACGT"[rnd() % (uint32_t)sigma]);
        if (m < n && it % 2 == 0) {
            const int a = rint_(0, n - m);
            for (int i = 0; i < m; i++) q[i] = rnd() % 10 == 0 ? (uint8_t)("ACGT"[rnd() % 4]) : t[a + i];
        } else
            for (int i = 0; i < m; i++) q[i] = (uint8_t)("ACGT"[rnd() % (uint32_t)sigma]);
        const int k = it % 3 == 0 ? -1 : it % 3 == 1 ? (2 * (m < n ? m : n)) / 5 : 0;
        int *s1 = (int *)malloc(sizeof(int) * (size_t)n), *e1 = (int *)malloc(sizeof(int) * (size_t)n);
        int *s2 = (int *)malloc(sizeof(int) * (size_t)n), *e2 = (int *)malloc(sizeof(int) * (size_t)n);
        uint8_t *o1 = (uint8_t *)malloc((size_t)(m + n)), *o2 = (uint8_t *)malloc((size_t)(m + n));
        int ed1, nl1, fi1, ol1, ed2, nl2, fi2, ol2, ed3, nl3;
        const int r1 = oracle_hw_bruteforce(q, m, t, n, k, &ed1, &nl1, s1, e1, n, &fi1, o1, m + n, &ol1);
        const int r2 = oracle_hw_bitvector(q, m, t, n, k, &ed2, &nl2, s2, e2, n, &fi2, o2, m + n, &ol2, 1);
        oracle_hw_forward_only(q, m, t, n, k, &ed3, &nl3);
        int ok = r1 == 0 && r2 == 0 && ed1 == ed2 && nl1 == nl2 && fi1 == fi2 && ol1 == ol2 && ed3 == ed1 && (ed1 < 0 || nl3 == nl1);
        for (int a = 0; ok && ed1 >= 0 && a < nl1; a++) ok = s1[a] == s2[a] && e1[a] == e2[a];
        if (ok && ed1 >= 0) ok = memcmp(o1, o2, (size_t)ol1) == 0;
        if (!ok) { fails++; fprintf(stderr, "myers case %d differs (m=%d n=%d k=%d)\n", it, m, n, k); }
        free(t); free(q); free(s1); free(e1); free(s2); free(e2); free(o1); free(o2);
    }
    /* ---- MAM / NAM chaining ---- */
    for (int it = 0; it < 400; it++) {
        const int n = it % 40 == 0 ? rint_(201, 700) : rint_(1, 220);
        int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)n);
        for (int i = 0; i < n; i++) {
            const int L = rint_(5, 60), c = rint_(0, 900), x = rint_(0, 5000);
            rows[4 * i] = x; rows[4 * i + 1] = x + L + rint_(0, 3); rows[4 * i + 2] = c; rows[4 * i + 3] = c + L - 1;
        }
        qsort(rows, (size_t)n, sizeof(int32_t) * 4, cmp_y);
        int32_t *x = (int32_t *)malloc(4 * (size_t)n), *y = (int32_t *)malloc(4 * (size_t)n), *c = (int32_t *)malloc(4 * (size_t)n),
                *d = (int32_t *)malloc(4 * (size_t)n), *out = (int32_t *)malloc(4 * (size_t)n), *tr = (int32_t *)malloc(4 * ((size_t)n + 1));
        double *val = (double *)malloc(8 * (size_t)n), v1, v2;
        int64_t *C = (int64_t *)malloc(8 * ((size_t)n + 1));
        for (int i = 0; i < n; i++) {
            x[i] = rows[4 * i]; y[i] = rows[4 * i + 1]; c[i] = rows[4 * i + 2]; d[i] = rows[4 * i + 3];
            val[i] = (double)(d[i] - c[i] + 1) * (rnd() % 2 ? 1.0 : 0.85);
        }
        const int l1 = oracle_chain_quadratic(n, c, d, val, 20, out, &v1);
        const int l2 = oracle_chain_nlogn(n, x, y, c, d, val, 5, out, &v2);
        oracle_nam_quadratic(n, y, c, d, 200000, C, tr);
        const int r4 = oracle_nam_nlogn(n, c, d, C, tr);
        if (l1 < 0 || l1 > n || l2 > n || r4 != 0) { fails++; fprintf(stderr, "chain case %d: bad return\n", it); }
        free(rows); free(x); free(y); free(c); free(d); free(out); free(tr); free(val); free(C);
    }
    if (fails) { fprintf(stderr, "%d failures\n", fails); return 1; }
    printf("asan_driver ok\n");
    return 0;
}
