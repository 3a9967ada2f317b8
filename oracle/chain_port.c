/*
 * oracle/chain_port.c — TEST INFRASTRUCTURE ONLY (CPU oracle), never linked or imported by the
 * product path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement, on flat arrays, of the two MAM colinear-chaining solvers of the reference:
 *   read_coverage_mam_score      /root/reference/modules/colinear_solver.py:276-389  (n <= 200)
 *   n_logn_read_coverage_mams    /root/reference/modules/colinear_solver.py:394-493  (n  > 200)
 * and of the range-max search tree the second one uses:
 *   make_leafs_power_of_2        colinear_solver.py:64-83
 *   construct_tree/range_query/update  /root/reference/modules/range_query_max_search_tree.py:19-157
 *
 * Pinned (tests/test_oracle_chain.py) against golden vectors generated in the build container by
 * importing the reference's own pure-Python modules (script: tests/golden/make_chain_golden.py).
 *
 * All float arithmetic is IEEE binary64 in the reference's association order; compile with
 * -ffp-contract=off.  MAMs arrive already stable-sorted by y, index j = position.
 */
#include <stdint.h>
#include <stdlib.h>

#define NEG_INIT (-4294967296.0) /* the reference's -2**32 "not yet processed" marker */

/* ---- quadratic solver (colinear_solver.py:276-389) ---------------------------------------- */
/* c,d: inclusive read coordinates; val: MAM score.  Writes the chain (ascending indices) into
 * out_idx, returns its length; *value = C[argmax]. */
int oracle_chain_quadratic(int n, const int32_t *c, const int32_t *d, const double *val, int thr,
                           int32_t *out_idx, double *value)
{
    if (n <= 0) { *value = -1.0; return 0; }
    double *C = (double *)malloc(sizeof(double) * n);
    int *ptr = (int *)malloc(sizeof(int) * n);
    for (int j = 0; j < n; j++) {
        /* T set: strictly before on the read; gap penalty 0.1 per skipped read base (:312) */
        int ta = -1; double tmax = 0.0;
        /* I set: overlap of at most thr bases (:322,:327) */
        int ia = -1; double imax = 0.0;
        for (int jp = j - 1; jp >= 0; jp--) { /* descending: first maximum met = largest j' (:316,:329) */
            if (d[jp] < c[j]) {
                double v = C[jp] - 0.1 * (double)(c[j] - d[jp] - 1);
                if (ta < 0 || v > tmax) { ta = jp; tmax = v; }
            }
            if (c[j] <= d[jp] && d[jp] <= c[j] - 1 + thr) {
                int o = d[jp] - c[j] + 1;
                double v = C[jp] + ((val[j] - (double)o) - 0.1 * (double)o);
                if (ia < 0 || v > imax) { ia = jp; imax = v; }
            }
        }
        double Ca = (ta < 0 ? 0.0 : tmax) + val[j];
        double Cb = (ia < 0 ? 0.0 : imax);
        if (Ca >= Cb) { ptr[j] = ta; C[j] = Ca; } else { ptr[j] = ia; C[j] = Cb; }
        /* C[j] = max(Ca, Cb): same number either way */
    }
    int best = 0;
    for (int j = 1; j < n; j++) if (C[j] > C[best]) best = j; /* first arg-max (:356) */
    *value = C[best];
    int len = 0;
    for (int j = best; j >= 0; j = ptr[j]) len++;
    int w = len;
    for (int j = best; j >= 0; j = ptr[j]) out_idx[--w] = j;
    free(C); free(ptr);
    return len;
}

/* ---- range-max search tree (range_query_max_search_tree.py) ------------------------------- */
typedef struct { int nleaf; int32_t *d; double *Cj; int32_t *jmax; } rtree;

static int lex_better(double ca, int ja, double cb, int jb) /* (ca,ja) strictly beats (cb,jb) */
{
    return ca > cb || (ca == cb && ja > jb);
}

/* leaves: start node, MAMs in index order, pad nodes; stable sort by d (:64-83) */
static int build_leaves(int n, const int32_t *dm, int32_t **ld, int32_t **lj)
{
    int cnt = n + 1, nleaf = cnt;
    for (int i = 0; i < 20; i++) {
        int lo = 1 << i, hi = 1 << (i + 1);
        if (cnt == lo || cnt == hi) break;
        if (lo < cnt && cnt < hi) { nleaf = hi; break; }
    }
    int32_t *d0 = (int32_t *)malloc(sizeof(int32_t) * nleaf), *j0 = (int32_t *)malloc(sizeof(int32_t) * nleaf);
    d0[0] = -1; j0[0] = -1;
    for (int i = 0; i < n; i++) { d0[i + 1] = dm[i]; j0[i + 1] = i; }
    for (int i = 0; i < nleaf - cnt; i++) { d0[cnt + i] = -1; j0[cnt + i] = -i - 2; }
    /* stable merge sort by d */
    int32_t *d1 = (int32_t *)malloc(sizeof(int32_t) * nleaf), *j1 = (int32_t *)malloc(sizeof(int32_t) * nleaf);
    for (int w = 1; w < nleaf; w *= 2) {
        for (int s = 0; s < nleaf; s += 2 * w) {
            int a = s, ae = s + w < nleaf ? s + w : nleaf, b = ae, be = s + 2 * w < nleaf ? s + 2 * w : nleaf, o = s;
            while (a < ae && b < be) {
                if (d0[b] < d0[a]) { d1[o] = d0[b]; j1[o++] = j0[b++]; }
                else { d1[o] = d0[a]; j1[o++] = j0[a++]; }
            }
            while (a < ae) { d1[o] = d0[a]; j1[o++] = j0[a++]; }
            while (b < be) { d1[o] = d0[b]; j1[o++] = j0[b++]; }
        }
        int32_t *t = d0; d0 = d1; d1 = t; t = j0; j0 = j1; j1 = t;
    }
    free(d1); free(j1);
    *ld = d0; *lj = j0;
    return nleaf;
}

static void tree_construct(rtree *T, int nleaf, const int32_t *ld, const int32_t *lj)
{
    T->nleaf = nleaf;
    T->d = (int32_t *)malloc(sizeof(int32_t) * 2 * nleaf);
    T->Cj = (double *)malloc(sizeof(double) * 2 * nleaf);
    T->jmax = (int32_t *)malloc(sizeof(int32_t) * 2 * nleaf);
    for (int i = 0; i < nleaf; i++) { T->d[nleaf + i] = ld[i]; T->Cj[nleaf + i] = NEG_INIT; T->jmax[nleaf + i] = lj[i]; }
    for (int i = nleaf - 1; i >= 1; i--) { /* copy of the child with the larger d, left on ties (:27-32) */
        int ch = (T->d[2 * i + 1] > T->d[2 * i]) ? 2 * i + 1 : 2 * i;
        T->d[i] = T->d[ch]; T->Cj[i] = T->Cj[ch]; T->jmax[i] = T->jmax[ch];
    }
}

static void tree_update(rtree *T, int leaf, double value) /* :137-157 */
{
    int pos = leaf + T->nleaf;
    T->Cj[pos] = value;
    while (pos > 1) {
        pos >>= 1;
        int L = 2 * pos, R = 2 * pos + 1;
        int ch = lex_better(T->Cj[R], T->jmax[R], T->Cj[L], T->jmax[L]) ? R : L;
        T->Cj[pos] = T->Cj[ch]; T->jmax[pos] = T->jmax[ch];
    }
}

/* :35-133.  Returns 0 and (Cmax, jmax), or -1 for the reference's "BUG" (empty node set). */
static int tree_query(const rtree *T, int l, int r, double *Cmax, int *jm)
{
    int n = T->nleaf, lp = 1, rp = 1, v = 1, have = 0;
    double bc = 0; int bj = 0;
#define TAKE(node) do { if (!have || lex_better(T->Cj[node], T->jmax[node], bc, bj)) { bc = T->Cj[node]; bj = T->jmax[node]; } have = 1; } while (0)
    for (;;) {
        int ll = 2 * lp, lr = ll + 1, rl = 2 * rp, rr = rl + 1;
        rp = (r >= T->d[rl]) ? rr : rl;
        lp = (l > T->d[ll]) ? lr : ll;
        if (lp == rp) v = lp;
        if (lp != rp && v != (lp >> 1)) {
            if (rp == rr) TAKE(rl);
            if (lp == ll) TAKE(lr);
        }
        if (lp >= n && rp >= n) break;
    }
    if (T->d[rp] == r) TAKE(rp);
    if (T->d[lp] >= l) TAKE(lp);
#undef TAKE
    if (!have) return -1;
    *Cmax = bc; *jm = bj;
    return 0;
}

static void tree_free(rtree *T) { free(T->d); free(T->Cj); free(T->jmax); }

/* exposed for the RMQ golden vectors (G2): one tree over d[], a list of (leaf j, value) updates,
 * then queries.  upd_j = MAM index (or -1 for the start leaf). */
int oracle_rmq_run(int n, const int32_t *dm, int n_upd, const int32_t *upd_j, const double *upd_v,
                   int n_q, const int32_t *ql, const int32_t *qr, double *out_c, int32_t *out_j, int32_t *out_ok)
{
    int32_t *ld, *lj;
    int nleaf = build_leaves(n, dm, &ld, &lj);
    rtree T; tree_construct(&T, nleaf, ld, lj);
    for (int u = 0; u < n_upd; u++) {
        int leaf = -1;
        for (int i = 0; i < nleaf; i++) if (lj[i] == upd_j[u]) { leaf = i; break; }
        if (leaf < 0) return -1;
        tree_update(&T, leaf, upd_v[u]);
    }
    for (int a = 0; a < n_q; a++) {
        double cm = 0; int jm = 0;
        int rc = tree_query(&T, ql[a], qr[a], &cm, &jm);
        out_ok[a] = (rc == 0); out_c[a] = cm; out_j[a] = jm;
    }
    tree_free(&T); free(ld); free(lj);
    return 0;
}

/* ---- n log n solver (colinear_solver.py:394-493) ------------------------------------------ */
/* returns chain length, or -1 if the reference would have hit its "BUG"/exit path. */
int oracle_chain_nlogn(int n, const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d,
                       const double *val, int thr, int32_t *out_idx, double *value)
{
    if (n <= 0) { *value = -1.0; return 0; }
    int32_t *ld, *lj;
    int nleaf = build_leaves(n, d, &ld, &lj);
    rtree T, I; tree_construct(&T, nleaf, ld, lj); tree_construct(&I, nleaf, ld, lj);
    int *leaf_of = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < nleaf; i++) if (lj[i] >= 0) leaf_of[lj[i]] = i;
    double *C = (double *)calloc((size_t)n + 1, sizeof(double));
    int *trace = (int *)calloc((size_t)n + 1, sizeof(int));
    tree_update(&T, 0, 0.0); tree_update(&I, 0, 0.0); /* leaf 0 is the start node after the sort */
    int rc = 0;
    for (int j = 0; j < n && rc == 0; j++) {
        double Tmax, Imax; int ja, jb;
        if (tree_query(&T, -1, c[j] - 1, &Tmax, &ja) != 0 || Tmax < 0) { rc = -1; break; }
        double Ca = Tmax + val[j];
        if (tree_query(&I, c[j], c[j] - 1 + thr, &Imax, &jb) != 0) { rc = -1; break; }
        int jbw = jb < 0 ? jb + n : jb; /* Python negative indexing of mams[j_prime_b] (:448-451) */
        if (jbw < 0 || jbw >= n) { rc = -1; break; }
        double pen = (x[j] != y[jbw]) ? ((double)(d[jbw] - (c[j] - 1)) + 0.0001) : 0.0001;
        double Cb = (Imax + val[j]) - pen;
        double v; int jp;
        if (Cb > Ca) { v = Cb; jp = jb; } else { v = Ca; jp = ja; } /* max_both: first on ties (:458) */
        C[j + 1] = v;
        trace[j + 1] = (jp < 0 || v == 0.0) ? 0 : jp + 1;
        tree_update(&T, leaf_of[j], v); tree_update(&I, leaf_of[j], v);
    }
    int len = -1;
    if (rc == 0) {
        int best = 0;
        for (int i = 1; i <= n; i++) if (C[i] > C[best]) best = i;
        *value = C[best];
        len = 0;
        for (int i = best; i > 0; i = trace[i]) len++;
        int w = len;
        for (int i = best; i > 0; i = trace[i]) out_idx[--w] = i - 1;
    }
    tree_free(&T); tree_free(&I); free(ld); free(lj); free(leaf_of); free(C); free(trace);
    return len;
}

/* Batched driver: instances in CSR form; switch rule of classify_read_with_mams.py:560-575. */
int oracle_chain_batch(int n_inst, const int64_t *inst_off,
                       const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d, const double *val,
                       int switch_n, int thr_small, int thr_large,
                       int32_t *chain_idx, int32_t *chain_len, double *value)
{
    for (int i = 0; i < n_inst; i++) {
        int64_t o = inst_off[i]; int n = (int)(inst_off[i + 1] - o);
        int len;
        if (n > switch_n) len = oracle_chain_nlogn(n, x + o, y + o, c + o, d + o, val + o, thr_large, chain_idx + o, &value[i]);
        else len = oracle_chain_quadratic(n, c + o, d + o, val + o, thr_small, chain_idx + o, &value[i]);
        if (len < 0) return -1;
        chain_len[i] = len;
    }
    return 0;
}

/* ============================================================================================== */
/* NAM (seed) chaining - SURVEY.md section 8 row f1.  The step right before the MAM stage:           */
/*   read_coverage(mems, max_intron)   /root/reference/modules/colinear_solver.py:169-272  (< 90 NAMs) */
/*   n_logn_read_coverage(mems)        /root/reference/modules/colinear_solver.py:85-167                 */
/* called at /root/reference/modules/align.py:448-453.  All-integer objective (read coverage).      */
/* Both return EVERY co-optimal chain (all_solutions_c_max_indicies :30, reconstruct_all_solutions  */
/* :49-62); here the DP vectors C[0..n] and trace[0..n] are the output and the (cheap) enumeration  */
/* of chains is done by the caller.  Pinned against tests/golden/nam_golden.json.                   */
/* ============================================================================================== */

/* NAMs sorted by y, j = position.  C[0] = 0, trace[0] = -1 (the reference's None). */
int oracle_nam_quadratic(int n, const int32_t *y, const int32_t *c, const int32_t *d, int64_t max_intron,
                         int64_t *C, int32_t *trace)
{
    C[0] = 0; trace[0] = -1;
    for (int j = 0; j < n; j++) {
        int ta = -1, ia = -1;
        int64_t tmax = 0, imax = 0;
        for (int jp = j - 1; jp >= 0; jp--) { /* descending: first maximum met = largest j' (:199,:209) */
            if ((int64_t)y[j] - (int64_t)y[jp] >= max_intron) continue;
            if (d[jp] < c[j]) {
                int64_t v = C[jp + 1];
                if (ta < 0 || v > tmax) { ta = jp; tmax = v; }
            }
            if (c[j] <= d[jp] && d[jp] <= d[j]) {
                int64_t v = C[jp + 1] + (int64_t)(d[j] - d[jp]);
                if (ia < 0 || v > imax) { ia = jp; imax = v; }
            }
        }
        int64_t Ca = (int64_t)(d[j] - c[j] + 1) + (ta < 0 ? 0 : tmax);
        int64_t Cb = ia < 0 ? 0 : imax;
        int jp = Ca >= Cb ? ta : ia; /* max_both: first maximum -> C_a on ties (:232) */
        C[j + 1] = Ca >= Cb ? Ca : Cb;
        trace[j + 1] = jp < 0 ? 0 : jp + 1;
    }
    return 0;
}

/* returns 0, or -1 where the reference would print "BUG" and exit */
int oracle_nam_nlogn(int n, const int32_t *c, const int32_t *d, int64_t *C, int32_t *trace)
{
    C[0] = 0; trace[0] = -1;
    if (n <= 0) return 0;
    int32_t *ld, *lj;
    int nleaf = build_leaves(n, d, &ld, &lj);
    rtree T, I; tree_construct(&T, nleaf, ld, lj); tree_construct(&I, nleaf, ld, lj);
    int *leaf_of = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < nleaf; i++) if (lj[i] >= 0) leaf_of[lj[i]] = i;
    tree_update(&T, 0, 0.0); tree_update(&I, 0, 0.0);
    int rc = 0;
    for (int j = 0; j < n; j++) {
        double Tmax, Imax; int ja, jb;
        if (tree_query(&T, -1, c[j] - 1, &Tmax, &ja) != 0 || Tmax < 0) { rc = -1; break; }
        double Ca = Tmax + (double)(d[j] - c[j] + 1);
        if (tree_query(&I, c[j], d[j], &Imax, &jb) != 0) { rc = -1; break; }
        double Cb = Imax + (double)d[j];
        double v; int jp;
        if (Cb > Ca) { v = Cb; jp = jb; } else { v = Ca; jp = ja; }
        C[j + 1] = (int64_t)v;
        trace[j + 1] = (jp < 0 || v == 0.0) ? 0 : jp + 1;
        tree_update(&T, leaf_of[j], v); tree_update(&I, leaf_of[j], v - (double)d[j]);
    }
    tree_free(&T); tree_free(&I); free(ld); free(lj); free(leaf_of);
    return rc;
}
