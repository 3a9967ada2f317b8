/* TEST INFRASTRUCTURE ONLY (oracle; never linked into or called by the product).
 *
 * CPU restatement of the base-level alignment behind the MAM stage:
 *     help_functions.parasail_alignment(s1, s2, match_score=2, mismatch_penalty=-2, opening_penalty=3, gap_ext=1)
 *         -> parasail.sg_trace_scan_16 / _32            /root/reference/modules/help_functions.py:204-226
 *     called from align.get_exact_alignment              /root/reference/modules/align.py:338-354
 * parasail (C library, PyPI `parasail`, version unpinned by setup.py) is absent from /root/reference and from the
 * image, so this follows its published algorithm: semi-global alignment (end gaps free in BOTH sequences, at both
 * ends), affine gaps with the convention "a gap of length L costs open + (L - 1) * extend", substitution matrix
 * parasail.matrix_create("ACGT", match, mismatch).
 *
 * PARITY UNPINNED for everything that depends on parasail's tie-breaking (which of several optimal end cells and
 * which of several optimal paths its scan kernels report); the SCORE is mathematically defined and is pinned against a
 * plain O(mn) three-matrix DP in tests/test_oracle_affine.py.
 * Choices made here (and mirrored bit for bit by the HIP kernel):
 *   - A, C, G, T match case-insensitively (score `match`), two different ones score `mismatch`; a pair with any other
 *     character (N, IUPAC codes) scores 0: matrix_create builds a (4 + 1) x (4 + 1) matrix whose extra "wildcard" row
 *     and column are 0 and maps every byte outside the alphabet to it (parasail's published matrix_create; recalled, not
 *     checkable here).  Such a pair is still written as 'X' in the edit script (cigar_to_seq treats '=' and 'X' alike);
 *   - end cell: the last row is scanned left to right, then the last column top to bottom; a later cell replaces
 *     the current best only when strictly better;
 *   - H: diagonal, then F (gap in the reference, consumes a query base, cigar 'I'), then E (consumes a reference
 *     base, cigar 'D'); E and F: extension wins a tie against opening;
 *   - the edit script covers BOTH sequences completely (leading / trailing free gaps appear as I / D), as
 *     help_functions.cigar_to_seq applies it to the whole of s1 and s2 (:217).
 * Edit script codes as in edlib_port.c: 0 '=', 1 'I', 2 'D', 3 'X'; written in forward order. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG (-(1 << 29))

static int sub_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* returns 0; -3 when ops_cap is too small (then *ops_len = needed). ops may be NULL (score and end cell only). */
int oracle_sg_affine(const uint8_t *q, int m, const uint8_t *t, int n, int match, int mismatch, int gap_open, int gap_extend,
                     int *score, int *end_q, int *end_t, uint8_t *ops, int ops_cap, int *ops_len)
{
    *score = 0; *end_q = -1; *end_t = -1;
    if (ops_len) *ops_len = 0;
    if (m <= 0 || n <= 0) return 0;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1) * 2);
    int32_t *F = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t *lastcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    uint8_t *tr = ops ? (uint8_t *)malloc((size_t)m * (size_t)n) : NULL; /* bits 0-1 H source, bit 2 E extended, bit 3 F extended */
    int32_t *prev = H, *cur = H + (n + 1);
    for (int j = 0; j <= n; j++) { prev[j] = 0; F[j] = NEG; }
    for (int i = 0; i < m; i++) {
        const int qc = sub_code(q[i]);
        cur[0] = 0;
        int32_t E = NEG;
        for (int j = 1; j <= n; j++) {
            const int32_t e_ext = E - gap_extend, e_opn = cur[j - 1] - gap_open;
            const int e_is_ext = e_ext >= e_opn;
            E = e_is_ext ? e_ext : e_opn;
            const int32_t f_ext = F[j] - gap_extend, f_opn = prev[j] - gap_open;
            const int f_is_ext = f_ext >= f_opn;
            F[j] = f_is_ext ? f_ext : f_opn;
            const int tc = sub_code(t[j - 1]);
            const int32_t d = prev[j - 1] + ((qc >= 0 && tc >= 0) ? (qc == tc ? match : mismatch) : 0);
            int src = 0;
            int32_t h = d;
            if (F[j] > h) { h = F[j]; src = 1; }
            if (E > h) { h = E; src = 2; }
            cur[j] = h;
            if (tr) tr[(size_t)i * n + (j - 1)] = (uint8_t)(src | (e_is_ext << 2) | (f_is_ext << 3));
        }
        lastcol[i] = cur[n];
        int32_t *tmp = prev; prev = cur; cur = tmp;
    }
    /* prev = last row */
    int best = NEG, bi = -1, bj = -1;
    for (int j = 1; j <= n; j++) if (prev[j] > best) { best = prev[j]; bi = m - 1; bj = j - 1; }
    for (int i = 0; i < m; i++) if (lastcol[i] > best) { best = lastcol[i]; bi = i; bj = n - 1; }
    *score = best; *end_q = bi; *end_t = bj;
    int rc = 0;
    if (ops) {
        /* backwards: trailing free gaps, the path, leading free gaps */
        size_t cap = (size_t)m + (size_t)n, cnt = 0;
        uint8_t *rev = (uint8_t *)malloc(cap);
        for (int i = m - 1; i > bi; i--) rev[cnt++] = 1;
        for (int j = n - 1; j > bj; j--) rev[cnt++] = 2;
        int i = bi, j = bj, state = 0; /* 0 = H, 1 = F, 2 = E */
        while (i >= 0 && j >= 0) {
            const uint8_t b = tr[(size_t)i * n + j];
            if (state == 0) state = b & 3;
            if (state == 0) { rev[cnt++] = (sub_code(q[i]) >= 0 && sub_code(q[i]) == sub_code(t[j])) ? 0 : 3; i--; j--; }
            else if (state == 1) { rev[cnt++] = 1; if (!((b >> 3) & 1)) state = 0; i--; }
            else { rev[cnt++] = 2; if (!((b >> 2) & 1)) state = 0; j--; }
        }
        for (; i >= 0; i--) rev[cnt++] = 1;
        for (; j >= 0; j--) rev[cnt++] = 2;
        if (ops_len) *ops_len = (int)cnt;
        if ((int)cnt > ops_cap) rc = -3;
        else for (size_t a = 0; a < cnt; a++) ops[a] = rev[cnt - 1 - a];
        free(rev);
    }
    free(H); free(F); free(lastcol); free(tr);
    return rc;
}
