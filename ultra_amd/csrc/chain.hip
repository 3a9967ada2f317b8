// MAM colinear chaining for gfx950: one wavefront per (read, location) instance.
//
// Replaces /root/reference/modules/colinear_solver.py:276-389 (read_coverage_mam_score, n <= 200)
// and :394-493 (n_logn_read_coverage_mams, n > 200; with range_query_max_search_tree.py:19-157).
//
//   * quadratic solver: for each j the wave evaluates every j' < j in parallel (stride 64), keeps
//     (value, index) pairs and reduces them with the reference's tie rule (largest j' among equal
//     maxima; the T candidate wins Ca >= Cb);
//   * n log n solver: the reference's search tree is replaced by its closed form - leaves sorted by
//     read end coordinate d in LDS (bitonic sort), the two range queries become index ranges found
//     by binary search, answered by a wave-wide lexicographic (C, j) max over per-64-leaf chunk
//     maxima plus one partial chunk.  The tree's two quirks are kept: the last leaf is left out of
//     a range that extends past it, and an empty range returns the first leaf above it.
//
// FP64 everywhere, one rounding per reference operation: built with -ffp-contract=off and the
// arithmetic below spelled with __d*_rn so nothing fuses.
#include "common.h"

#include <algorithm>
#include <mutex>
#include <string.h>

namespace {

struct Cand {
    double v;
    int idx;
    int has;
};

// larger value wins; ties -> larger index
__device__ __forceinline__ Cand better_hi(Cand a, Cand b)
{
    if (!b.has) return a;
    if (!a.has) return b;
    if (b.v > a.v || (b.v == a.v && b.idx > a.idx)) return b;
    return a;
}
// larger value wins; ties -> smaller index (first arg-max)
__device__ __forceinline__ Cand better_lo(Cand a, Cand b)
{
    if (!b.has) return a;
    if (!a.has) return b;
    if (b.v > a.v || (b.v == a.v && b.idx < a.idx)) return b;
    return a;
}
// Wave-wide lexicographic maximum, the result in every lane.  Data-parallel-primitive moves instead of __shfl_xor (six
// rounds of four ds_bpermute trips through the LDS crossbar per reduction, two reductions per MAM: the quadratic chainer
// spent its time waiting for them): an inclusive scan over each row of 16 lanes (row_shr 1, 2, 4, 8; a lane without a
// source keeps the identity, has = 0), lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15), lane 31 into rows 2 / 3
// (row_bcast:31): lane 63 holds the maximum and is read back through the scalar unit.  The order in which candidates
// meet does not matter: (value, index) pairs are totally ordered and distinct.
template <bool HI, int CTRL, int ROWS> __device__ __forceinline__ Cand dpp_round(Cand c)
{
    Cand o;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(c.v), CTRL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(c.v), CTRL, ROWS, 0xf, false);
    o.v = __hiloint2double(hi, lo);
    o.idx = __builtin_amdgcn_update_dpp(0, c.idx, CTRL, ROWS, 0xf, false);
    o.has = __builtin_amdgcn_update_dpp(0, c.has, CTRL, ROWS, 0xf, false);
    return HI ? better_hi(c, o) : better_lo(c, o);
}
template <bool HI> __device__ __forceinline__ Cand wave_reduce(Cand c)
{
    c = dpp_round<HI, 0x111, 0xf>(c); // row_shr:1
    c = dpp_round<HI, 0x112, 0xf>(c); // row_shr:2
    c = dpp_round<HI, 0x114, 0xf>(c); // row_shr:4
    c = dpp_round<HI, 0x118, 0xf>(c); // row_shr:8
    c = dpp_round<HI, 0x142, 0xa>(c); // row_bcast:15 into rows 1 and 3
    c = dpp_round<HI, 0x143, 0xc>(c); // row_bcast:31 into rows 2 and 3
    Cand r;
    r.v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c.v), 63), __builtin_amdgcn_readlane(__double2loint(c.v), 63));
    r.idx = __builtin_amdgcn_readlane(c.idx, 63);
    r.has = __builtin_amdgcn_readlane(c.has, 63);
    return r;
}

struct ChainArgs {
    const int64_t *off;
    const int32_t *x, *y, *c, *d;
    const double *val;
    int32_t thr;
    int32_t *idx, *len;
    double *value;
    const int32_t *cnt;  // optional per-instance MAM count (nullptr: off[inst+1] - off[inst])
    const int32_t *list; // instance ids handled by this launch
    int32_t nlist;
    int32_t cap;         // LDS entries per wave (quadratic) / leaves (n log n)
    uint8_t *work;       // global scratch (n log n)
    const int64_t *work_off;
    int32_t *status;
    int32_t skip_tiny;   // chain_quadratic: instances of at most kTiny MAMs were done by chain_tiny
    // launch lists made on the device (k_chain_keys + sort): `list` is the whole sorted list and this launch takes the
    // entries [seg_first[seg_lo], seg_first[seg_hi]); the grid only bounds their number from above
    const int32_t *seg_first;
    int32_t seg_lo, seg_hi;
    int64_t work_stride; // n log n with device lists: bytes of tree arrays per list entry (work_off unused)
    // NAM (seed) chaining, SURVEY.md section 8 row f1: integer read-coverage objective of
    // colinear_solver.py:85-272; the whole DP vectors are the output (every co-optimal chain is wanted)
    int32_t nam;               // 0 = MAM objective, 1 = NAM objective
    const int64_t *max_intron; // per instance (quadratic NAM solver only)
    int64_t *C_out;            // [total + n_inst]: instance i at off[i] + i, n + 1 entries, C[0] = 0
    int32_t *trace_out;        // same layout, trace[0] = -1
};

constexpr int kTiny = 48; // largest instance of the lane-per-instance kernel (chain_tiny)
constexpr int kQuadSmall = 256;    // LDS entries per wave in the 4-waves-per-block launch

// Launch lists by size on the device.  Segment of an instance: 0 .. kTiny = lane-per-instance kernel, largest first
// (segment kTiny - n), then the small quadratic launch, the large one, the n log n solver.
// The small quadratic launch (one wavefront per instance of 49 .. 256 MAMs, up to a millisecond for the largest) gets
// kSmallBins segments by size, largest first: the launch's critical path starts with the launch, not in its tail.
constexpr int kSmallBins = 8;
constexpr int kSegSmall = kTiny + 1, kSegBig = kSegSmall + kSmallBins, kSegTree = kSegBig + 1, kSegEnd = kSegTree + 1;
static_assert(kSegEnd + 1 <= 64, "segment bounds live in 64 words");
__global__ void __launch_bounds__(256) k_chain_keys(const int32_t *cnt, int32_t n_inst, int32_t switch_n, uint32_t *keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inst) return;
    const int n = cnt[i];
    keys[i] = n > switch_n && n > 0 ? kSegTree : n <= kTiny ? (uint32_t)(kTiny - (n < 0 ? 0 : n))
            : n <= kQuadSmall ? (uint32_t)(kSegSmall + (kQuadSmall - n) * kSmallBins / (kQuadSmall - kTiny)) : kSegBig;
}
struct ListSlice { const int32_t *list; int n; };
__device__ __forceinline__ ListSlice list_slice(const ChainArgs &a)
{
    if (!a.seg_first) return {a.list, a.nlist};
    const int lo = a.seg_first[a.seg_lo];
    return {a.list + lo, a.seg_first[a.seg_hi] - lo};
}

// ---- quadratic ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chain_quadratic(ChainArgs a)
{
    extern __shared__ uint8_t lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int slot = blockIdx.x * wpb + wave;
    const ListSlice ls = list_slice(a);
    if (slot >= ls.n) return;
    const int inst = ls.list[slot];
    const int64_t o = a.off[inst];
    const int n = a.cnt ? a.cnt[inst] : (int)(a.off[inst + 1] - o);
    const int cap = a.cap;
    const bool nam = a.nam != 0;
    if (a.skip_tiny && n <= kTiny) return; // chain_tiny has done it
    uint8_t *base = lds_raw + (size_t)wave * cap * (nam ? 24 : 20);
    double *sC = (double *)base;
    int32_t *sc = (int32_t *)(base + (size_t)cap * 8);
    int32_t *sd = sc + cap;
    int32_t *sp = sd + cap;
    int32_t *sy = sp + cap; // NAM only
    const int64_t oo = o + inst; // NAM outputs: n + 1 entries per instance
    if (nam && lane == 0) { a.C_out[oo] = 0; a.trace_out[oo] = -1; }
    if (n <= 0) {
        if (lane == 0) { a.len[inst] = 0; a.value[inst] = -1.0; }
        return;
    }
    for (int i = lane; i < n; i += 64) { sc[i] = a.c[o + i]; sd[i] = a.d[o + i]; if (nam) sy[i] = a.y[o + i]; }
    const int thr = a.thr;
    const long long max_intron = nam ? a.max_intron[inst] : 0;
    for (int j = 0; j < n; j++) {
        const int cj = sc[j];
        if (nam) {
            // read_coverage (colinear_solver.py:169-272): T = C[j'] over d[j'] < c, I = C[j'] + (d - d[j']) over
            // c <= d[j'] <= d, both only within max_intron on the genome; ties -> largest j'; C_a wins ties
            const int dcur = sd[j];
            const long long yj = sy[j];
            Cand T = {0.0, -1, 0}, I = {0.0, -1, 0};
            for (int jp = lane; jp < j; jp += 64) {
                if (yj - (long long)sy[jp] >= max_intron) continue;
                const int dj = sd[jp];
                const double Cp = sC[jp];
                if (dj < cj) { Cand t = {Cp, jp, 1}; T = better_hi(T, t); }
                if (cj <= dj && dj <= dcur) { Cand t = {__dadd_rn(Cp, (double)(dcur - dj)), jp, 1}; I = better_hi(I, t); }
            }
            T = wave_reduce<true>(T);
            I = wave_reduce<true>(I);
            const double Ca = __dadd_rn((double)(dcur - cj + 1), T.has ? T.v : 0.0);
            const double Cb = I.has ? I.v : 0.0;
            if (lane == 0) {
                const int jp = Ca >= Cb ? (T.has ? T.idx : -1) : (I.has ? I.idx : -1);
                const double v = Ca >= Cb ? Ca : Cb;
                sC[j] = v; sp[j] = jp;
                a.C_out[oo + j + 1] = (int64_t)v;
                a.trace_out[oo + j + 1] = jp < 0 ? 0 : jp + 1;
            }
            continue;
        }
        const double vj = a.val[o + j];
        Cand T = {0.0, -1, 0}, I = {0.0, -1, 0};
        for (int jp = lane; jp < j; jp += 64) {
            const int dj = sd[jp];
            const double Cp = sC[jp];
            if (dj < cj) {
                double v = __dsub_rn(Cp, __dmul_rn(0.1, (double)(cj - dj - 1)));
                Cand t = {v, jp, 1};
                T = better_hi(T, t);
            }
            if (cj <= dj && dj <= cj - 1 + thr) {
                double ov = (double)(dj - cj + 1);
                double v = __dadd_rn(Cp, __dsub_rn(__dsub_rn(vj, ov), __dmul_rn(0.1, ov)));
                Cand t = {v, jp, 1};
                I = better_hi(I, t);
            }
        }
        T = wave_reduce<true>(T);
        I = wave_reduce<true>(I);
        const double Ca = __dadd_rn(T.has ? T.v : 0.0, vj);
        const double Cb = I.has ? I.v : 0.0;
        if (lane == 0) {
            if (Ca >= Cb) { sC[j] = Ca; sp[j] = T.has ? T.idx : -1; }
            else { sC[j] = Cb; sp[j] = I.has ? I.idx : -1; }
        }
    }
    Cand B = {0.0, -1, 0};
    for (int i = lane; i < n; i += 64) { Cand t = {sC[i], i, 1}; B = better_lo(B, t); }
    B = wave_reduce<false>(B);
    if (lane == 0) {
        if (nam) { a.len[inst] = n; a.value[inst] = B.v; return; }
        int len = 0;
        for (int j = B.idx; j >= 0; j = sp[j]) len++;
        int w = len;
        for (int j = B.idx; j >= 0; j = sp[j]) a.idx[o + (--w)] = j;
        a.len[inst] = len;
        a.value[inst] = B.v;
    }
}

// ---- quadratic, tiny instances -------------------------------------------------------------------
// Most (read, location) instances hold a handful of MAMs: a wavefront per instance would idle 60 of its
// lanes and spend its time in two wave-wide reductions per MAM.  Instances of at most kTiny MAMs get one
// LANE each: the same recurrence evaluated sequentially (j' ascending through better_hi, i.e. the same
// maxima and the same tie rules), the DP vectors in lane-private LDS columns.  The host sorts the list by
// size, so the lanes of a wavefront run equally long.
__global__ void __launch_bounds__(64) chain_tiny(ChainArgs a)
{
    __shared__ double sC[kTiny * 64];
    __shared__ int32_t sd[kTiny * 64];
    __shared__ int8_t sp[kTiny * 64];
    const int lane = threadIdx.x;
    const int slot = blockIdx.x * 64 + lane;
    int inst = 0, n = 0;
    int64_t o = 0;
    const ListSlice ls = list_slice(a);
    if (slot < ls.n) {
        inst = ls.list[slot];
        o = a.off[inst];
        n = a.cnt ? a.cnt[inst] : (int)(a.off[inst + 1] - o);
        if (n > kTiny) {
            // beyond the first a.cap list entries the host only had an upper bound: chain_quadratic takes those
            if (slot < a.cap) atomicOr(a.status, 2); // the host's size bound was wrong: fail the call
            n = -1;
        }
        if (n == 0) { a.len[inst] = 0; a.value[inst] = -1.0; }
    }
    for (int i = 0; i < n; i++) sd[i * 64 + lane] = a.d[o + i];
    int nmax = n;
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) { const int t = __shfl_xor(nmax, w, 64); nmax = t > nmax ? t : nmax; }
    const int thr = a.thr;
    // c and val of MAM j are read once: fetched one j ahead, under the inner loop
    int c_next = n > 0 ? a.c[o] : 0;
    double v_next = n > 0 ? a.val[o] : 0.0;
    for (int j = 0; j < nmax; j++) {
        if (j >= n) continue;
        const int cj = c_next;
        const double vj = v_next;
        if (j + 1 < n) { c_next = a.c[o + j + 1]; v_next = a.val[o + j + 1]; }
        Cand T = {0.0, -1, 0}, I = {0.0, -1, 0};
        for (int jp = 0; jp < j; jp++) {
            const int dj = sd[jp * 64 + lane];
            const double Cp = sC[jp * 64 + lane];
            if (dj < cj) {
                double v = __dsub_rn(Cp, __dmul_rn(0.1, (double)(cj - dj - 1)));
                Cand t = {v, jp, 1};
                T = better_hi(T, t);
            }
            if (cj <= dj && dj <= cj - 1 + thr) {
                double ov = (double)(dj - cj + 1);
                double v = __dadd_rn(Cp, __dsub_rn(__dsub_rn(vj, ov), __dmul_rn(0.1, ov)));
                Cand t = {v, jp, 1};
                I = better_hi(I, t);
            }
        }
        const double Ca = __dadd_rn(T.has ? T.v : 0.0, vj);
        const double Cb = I.has ? I.v : 0.0;
        if (Ca >= Cb) { sC[j * 64 + lane] = Ca; sp[j * 64 + lane] = (int8_t)(T.has ? T.idx : -1); }
        else { sC[j * 64 + lane] = Cb; sp[j * 64 + lane] = (int8_t)(I.has ? I.idx : -1); }
    }
    if (n <= 0) return;
    Cand B = {0.0, -1, 0};
    for (int i = 0; i < n; i++) { Cand t = {sC[i * 64 + lane], i, 1}; B = better_lo(B, t); }
    int len = 0;
    for (int j = B.idx; j >= 0; j = sp[j * 64 + lane]) len++;
    int w = len;
    for (int j = B.idx; j >= 0; j = sp[j * 64 + lane]) a.idx[o + (--w)] = j;
    a.len[inst] = len;
    a.value[inst] = B.v;
}

// ---- n log n -------------------------------------------------------------------------------------
__device__ __forceinline__ int first_greater(const int32_t *ld, int nleaf, int key, bool or_equal, bool &fallback)
{
    // first index with ld > key (or >= key); nleaf-1 with fallback=true when there is none
    int lo = 0, hi = nleaf;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        bool right = or_equal ? (ld[mid] < key) : (ld[mid] <= key);
        if (right) lo = mid + 1; else hi = mid;
    }
    fallback = (lo >= nleaf);
    return fallback ? nleaf - 1 : lo;
}

template <bool NAM> __global__ void __launch_bounds__(64) chain_nlogn(ChainArgs a)
{
    extern __shared__ uint8_t lds_raw[];
    const int lane = threadIdx.x;
    const int slot = blockIdx.x;
    const ListSlice ls = list_slice(a);
    if (slot >= ls.n) return;
    const int inst = ls.list[slot];
    const int64_t o = a.off[inst];
    const int n = a.cnt ? a.cnt[inst] : (int)(a.off[inst + 1] - o);
    if (n <= 0) {
        if (lane == 0) {
            a.len[inst] = 0; a.value[inst] = -1.0;
            if (NAM) { a.C_out[o + inst] = 0; a.trace_out[o + inst] = -1; }
        }
        return;
    }
    // leaf count (colinear_solver.py:64-83): n+1 nodes padded up to a power of two
    int cnt = n + 1, nleaf = 1;
    while (nleaf < cnt) nleaf <<= 1;
    const int npad = nleaf - cnt;
    const int nchunk = (nleaf + 63) >> 6;
    // LDS: lC[nleaf] f64 | ld[nleaf] i32 | lj[nleaf] i32 | cmC[nchunk] f64 | cmJ[nchunk] i32
    // NAM: the 8 bytes per leaf hold two int32 instead - the T tree's C and the I tree's C - d (all values are
    // integers below 2^31, the host checks the coordinates), so both trees fit where the MAM solver keeps one
    double *lC = (double *)lds_raw;
    int32_t *lTi = (int32_t *)lds_raw, *lIi = lTi + nleaf;
    constexpr int32_t NEGI = (int32_t)0x80000000;
    int32_t *ld = (int32_t *)(lds_raw + (size_t)nleaf * 8);
    int32_t *lj = ld + nleaf;
    double *cmC = (double *)(lds_raw + (size_t)nleaf * 16);
    int32_t *cmJ = (int32_t *)(lds_raw + (size_t)nleaf * 16 + (size_t)nchunk * 8);
    // global scratch of this instance
    uint8_t *wk = a.work + (a.work_off ? a.work_off[slot] : (int64_t)slot * a.work_stride);
    double *C = (double *)wk;                      // n+1
    int32_t *trace = (int32_t *)(wk + (size_t)(n + 1) * 8); // n+1
    int32_t *leaf_of = trace + (n + 1);            // n
    int32_t *qT = leaf_of + n;                     // n   T-prefix length
    int32_t *qIlo = qT + n;                        // n
    int32_t *qIhi = qIlo + n;                      // n   (exclusive)
    const int32_t *gc = a.c + o, *gd = a.d + o, *gx = a.x + o, *gy = a.y + o;
    const double *gv = a.val + o;
    const double NEG = -__builtin_inf();

    // ---- sort MAMs by (d, j): bitonic sort of 64-bit keys held in the lC area ----
    uint64_t *keys = (uint64_t *)lC;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = lane; i < np2; i += 64)
        keys[i] = i < n ? ((((uint64_t)(uint32_t)gd[i]) << 32) | (uint32_t)i) : ~(uint64_t)0;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int t = lane; t < (np2 >> 1); t += 64) {
                int i = ((t / s) * (s << 1)) + (t % s);
                int p = i + s;
                bool up = ((i & k) == 0);
                uint64_t A = keys[i], B = keys[p];
                if ((A > B) == up) { keys[i] = B; keys[p] = A; }
            }
        }
    }
    // leaves: start node, pad nodes, then the MAMs in (d, j) order (stable sort by d of the reference)
    for (int r = lane; r < n; r += 64) {
        uint64_t kx = keys[r];
        int j = (int)(uint32_t)kx;
        int pos = 1 + npad + r;
        lj[pos] = j;
        leaf_of[j] = pos;
    }
    for (int i = lane; i <= npad; i += 64) lj[i] = (i == 0) ? -1 : -(i - 1) - 2;
    // keys alias lC: read every d before lC is initialised
    for (int r = lane; r < n; r += 64) ld[1 + npad + r] = (int32_t)(uint32_t)(keys[r] >> 32);
    for (int i = lane; i <= npad; i += 64) ld[i] = -1;
    if (NAM) {
        // ld was filled from the keys that alias this area: every lane is past those reads (one wave, in order)
        for (int i = lane; i < nleaf; i += 64) { lTi[i] = NEGI; lIi[i] = NEGI; }
    } else {
        for (int i = lane; i < nleaf; i += 64) lC[i] = NEG;
    }
    for (int i = lane; i < nchunk; i += 64) { cmC[i] = NEG; cmJ[i] = -0x7fffffff; }
    if (lane == 0) {
        if (NAM) { lTi[0] = 0; lIi[0] = 0; } else lC[0] = 0.0;
        cmC[0] = 0.0; cmJ[0] = -1;
    }
    // ---- per-MAM query ranges (leaf keys never change) ----
    const int thr = a.thr;
    int bug = 0;
    for (int j = lane; j < n; j += 64) {
        const int c = gc[j];
        bool fb;
        // T query [-1, c-1]: lo = 0
        int hi = first_greater(ld, nleaf, c - 1, false, fb);
        int plen = hi + ((fb && ld[hi] == c - 1) ? 1 : 0);
        if (plen <= 0) bug = 1; // the reference's "BUG" exit: cannot happen, leaf 0 has d = -1
        qT[j] = plen;
        // I query [c, c-1+thr] (MAM, :437) or [c, d] (NAM, colinear_solver.py:137)
        const int r = NAM ? gd[j] : c - 1 + thr;
        bool fbl, fbh;
        int lo = first_greater(ld, nleaf, c, true, fbl);
        int hi2 = first_greater(ld, nleaf, r, false, fbh);
        int b, e;
        if (lo < hi2) { b = lo; e = hi2 + ((ld[hi2] == r) ? 1 : 0); }
        else {
            bool take = (ld[hi2] == r) || (ld[lo] >= c);
            // lo == hi2 here (lo > hi2 would need r < l - 1)
            b = lo; e = take ? lo + 1 : lo;
            if (!take) bug = 1;
        }
        qIlo[j] = b; qIhi[j] = e;
    }
    if (__any(bug)) {
        if (lane == 0) { a.len[inst] = -1; a.value[inst] = -1.0; atomicOr(a.status, 1); }
        return;
    }
    __threadfence();
    if (lane == 0) { C[0] = 0.0; trace[0] = 0; }
    // ---- DP ----
    for (int j = 0; j < n; j++) {
        const int plen = qT[j];
        const int b = qIlo[j], e = qIhi[j];
        const double vj = NAM ? (double)(gd[j] - gc[j] + 1) : gv[j];
        // T: lexicographic (C, j) max over leaves [0, plen)
        Cand T = {0.0, 0, 0};
        const int full = plen >> 6;
        for (int ch = lane; ch < full; ch += 64) {
            double cv = cmC[ch];
            if (cv != NEG) { Cand t = {cv, cmJ[ch], 1}; T = better_hi(T, t); }
        }
        {
            int i = (full << 6) + lane;
            if (i < plen) {
                if (NAM) {
                    int32_t ci = lTi[i];
                    if (ci != NEGI) { Cand t = {(double)ci, lj[i], 1}; T = better_hi(T, t); }
                } else {
                    double cv = lC[i];
                    if (cv != NEG) { Cand t = {cv, lj[i], 1}; T = better_hi(T, t); }
                }
            }
        }
        Cand I = {0.0, 0, 0};
        for (int i = b + lane; i < e; i += 64) {
            if (NAM) {
                int32_t ci = lIi[i];
                if (ci != NEGI) { Cand t = {(double)ci, lj[i], 1}; I = better_hi(I, t); }
            } else {
                double cv = lC[i];
                if (cv != NEG) { Cand t = {cv, lj[i], 1}; I = better_hi(I, t); }
            }
        }
        T = wave_reduce<true>(T);
        I = wave_reduce<true>(I);
        // T always holds the start leaf (C = 0)
        const double Ca = __dadd_rn(T.v, vj);
        double v = Ca;
        int jp = T.idx;
        if (NAM) {
            // C_b = I + d (:139); an I range without a processed leaf leaves the reference with -2^32 + d < C_a
            if (I.has) {
                double Cb = __dadd_rn(I.v, (double)gd[j]);
                if (Cb > Ca) { v = Cb; jp = I.idx; }
            }
        } else if (I.has) {
            int jb = I.idx < 0 ? I.idx + n : I.idx; // Python negative indexing (:448-451)
            double pen = (gx[j] != gy[jb]) ? __dadd_rn((double)(gd[jb] - (gc[j] - 1)), 0.0001) : 0.0001;
            double Cb = __dsub_rn(__dadd_rn(I.v, vj), pen);
            if (Cb > Ca) { v = Cb; jp = I.idx; }
        }
        if (lane == 0) {
            C[j + 1] = v;
            trace[j + 1] = (jp < 0 || v == 0.0) ? 0 : jp + 1;
            int leaf = leaf_of[j];
            if (NAM) { lTi[leaf] = (int32_t)v; lIi[leaf] = (int32_t)(v - (double)gd[j]); }
            else lC[leaf] = v;
            int ch = leaf >> 6;
            double cv = cmC[ch];
            if (cv == NEG || v > cv || (v == cv && j > cmJ[ch])) { cmC[ch] = v; cmJ[ch] = j; }
        }
    }
    __threadfence();
    Cand B = {0.0, -1, 0};
    for (int i = lane; i <= n; i += 64) { Cand t = {C[i], i, 1}; B = better_lo(B, t); }
    B = wave_reduce<false>(B);
    if (NAM) {
        const int64_t oo = o + inst;
        for (int i = lane; i <= n; i += 64) { a.C_out[oo + i] = (int64_t)C[i]; a.trace_out[oo + i] = i == 0 ? -1 : trace[i]; }
        if (lane == 0) { a.len[inst] = n; a.value[inst] = B.v; }
        return;
    }
    if (lane == 0) {
        int len = 0;
        for (int i = B.idx; i > 0; i = trace[i]) len++;
        int w = len;
        for (int i = B.idx; i > 0; i = trace[i]) a.idx[o + (--w)] = i - 1;
        a.len[inst] = len;
        a.value[inst] = B.v;
    }
}

constexpr int kLdsBudget = 160 * 1024;

} // namespace

void ultra_chain_free(ultra_ctx *ctx)
{
    ChainState &ch = ctx->ch;
    DevBuf *bufs[] = {&ch.ioff, &ch.off, &ch.x, &ch.y, &ch.c, &ch.d, &ch.val, &ch.idx, &ch.len, &ch.value, &ch.work, &ch.namC, &ch.namtrace, &ch.intron, &ch.sorttmp};
    for (DevBuf *b : bufs) b->release();
}

// MAM objective with the reference's solver switch, sizes known on the device only (d_cnt): the launch lists are made
// there (segment of every instance, bucket sort) while h_bound - an upper bound of every instance's size - sizes
// grids, LDS and the n log n solver's scratch.  Nothing is enqueued when kChainNeedExact comes back.
static int chain_launch_device_lists(ultra_ctx *ctx, int32_t n_inst, const int32_t *h_bound, const int64_t *d_off, const int32_t *d_cnt,
                                     const int32_t *d_x, const int32_t *d_y, const int32_t *d_c, const int32_t *d_d, const double *d_val,
                                     int thr_s, int thr_l, int32_t *d_idx, int32_t *d_len, double *d_value, int32_t **d_status_out)
{
    const int sw = ctx->cfg.chain_switch_n;
    int64_t ub_small = 0, ub_big = 0, ub_tree = 0;
    int32_t max_big = 0, max_tree = 0;
    for (int32_t i = 0; i < n_inst; i++) {
        const int32_t n = h_bound[i];
        if (n < 0) return ultra_fail(ctx, ULTRA_E_ARG, "chaining: negative instance size");
        if (n > sw && n > 0) { ub_tree++; max_tree = std::max(max_tree, n); }
        if (n > kTiny) ub_small++;
        if (n > kQuadSmall) { ub_big++; max_big = std::max(max_big, n); }
    }
    if (ub_tree > 0 && thr_l < 1) return kChainNeedExact;
    if ((size_t)max_big * 20 > (size_t)kLdsBudget) return kChainNeedExact;
    int tree_leaves = 1;
    while (tree_leaves < max_tree + 1) tree_leaves <<= 1;
    const size_t tree_lds = (size_t)tree_leaves * 16 + (size_t)((tree_leaves + 63) / 64) * 12 + 16;
    if (ub_tree > 0 && tree_lds > (size_t)kLdsBudget) return kChainNeedExact;
    const int64_t stride = ((((int64_t)max_tree + 1) * 12 + (int64_t)max_tree * 16) + 15) & ~(int64_t)15;
    if (ub_tree * stride > ((int64_t)2 << 30)) return kChainNeedExact;

    ChainState &ch = ctx->ch;
    hipStream_t s = ctx->stream;
    const size_t ni = (size_t)n_inst;
    // ch.off: segment of every instance | instances grouped by segment | segment bounds (64) | status
    ULTRA_HIP_CHECK(ctx, ch.off.ensure(ni * 8 + 64 * 4 + 64));
    uint32_t *keys = ch.off.as<uint32_t>();
    int32_t *i_out = (int32_t *)(keys + ni);
    int32_t *d_first = i_out + ni, *d_status = d_first + 64;
    if (d_status_out) *d_status_out = d_status;
    if (ub_tree > 0) ULTRA_HIP_CHECK(ctx, ch.work.ensure((size_t)(ub_tree * stride) + 16));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(d_status, 0, 4, s));
    hipLaunchKernelGGL(k_chain_keys, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, s, d_cnt, n_inst, sw, keys);
    ULTRA_HIP_CHECK(ctx, ultra_bucket_sort(s, keys, nullptr, n_inst, kSegEnd, i_out, d_first, 0, nullptr, ch.sorttmp));
    ChainArgs a;
    memset(&a, 0, sizeof a);
    a.off = d_off; a.cnt = d_cnt; a.x = d_x; a.y = d_y; a.c = d_c; a.d = d_d;
    a.val = d_val; a.idx = d_idx; a.len = d_len; a.value = d_value;
    a.status = d_status; a.list = i_out; a.seg_first = d_first;
    const int n_launch = 1 + (ub_small > 0) + (ub_big > 0) + (ub_tree > 0);
    const bool spread = ctx->cfg.launch_lanes > 1 && n_launch > 1;
    if (spread) ULTRA_HIP_CHECK(ctx, ultra_fork_at(ctx, 6));
    int li = 0;
    auto next_lane = [&]() { return spread ? ctx->lane(li++) : s; };
    if (ub_tree > 0) { // longest critical path first
        hipStream_t cs = next_lane();
        a.thr = thr_l; a.seg_lo = kSegTree; a.seg_hi = kSegEnd; a.cap = tree_leaves;
        a.work = ch.work.as<uint8_t>(); a.work_off = nullptr; a.work_stride = stride;
        if (tree_lds > 64 * 1024)
            ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)chain_nlogn<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_lds));
        hipLaunchKernelGGL(chain_nlogn<false>, dim3((unsigned)ub_tree), dim3(64), tree_lds, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    if (ub_big > 0) {
        hipStream_t cs = next_lane();
        a.thr = thr_s; a.seg_lo = kSegBig; a.seg_hi = kSegTree; a.cap = max_big;
        const size_t shmem = (size_t)max_big * 20;
        if (shmem > 64 * 1024)
            ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)chain_quadratic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(chain_quadratic, dim3((unsigned)ub_big), dim3(64), shmem, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    if (ub_small > 0) {
        hipStream_t cs = next_lane();
        a.thr = thr_s; a.seg_lo = kSegSmall; a.seg_hi = kSegBig; a.cap = kQuadSmall;
        hipLaunchKernelGGL(chain_quadratic, dim3((unsigned)((ub_small + 3) / 4)), dim3(256), (size_t)4 * kQuadSmall * 20, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        hipStream_t cs = next_lane();
        a.thr = thr_s; a.seg_lo = 0; a.seg_hi = kSegSmall; a.cap = 0;
        hipLaunchKernelGGL(chain_tiny, dim3((unsigned)((ni + 63) / 64)), dim3(64), 0, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    if (spread) ULTRA_HIP_CHECK(ctx, ultra_join_at(ctx, 7));
    return ULTRA_OK;
}

// Launches the chaining kernels for n_inst instances whose MAM arrays are already in device memory.
// h_cnt: per-instance MAM counts on the host (used to pick solver / LDS sizes); d_cnt: the same on
// the device, or nullptr when the counts are d_off[i+1] - d_off[i].  Asynchronous on ctx->stream;
// *d_status (device int, zeroed here) is non-zero afterwards if an instance hit the reference's
// "BUG" state.  cnt_is_bound (MAM objective, d_cnt given): h_cnt only bounds the sizes from above and the launch
// lists are made on the device (chain_launch_device_lists); kChainNeedExact = a limit failed on the bounds, nothing
// was enqueued, call again with the exact counts.
int ultra_chain_launch(ultra_ctx *ctx, int32_t n_inst, const int32_t *h_cnt, const int64_t *d_off, const int32_t *d_cnt,
                       const int32_t *d_x, const int32_t *d_y, const int32_t *d_c, const int32_t *d_d, const double *d_val,
                       int32_t solver, int32_t overlap_small, int32_t overlap_large,
                       int32_t *d_idx, int32_t *d_len, double *d_value, int32_t **d_status_out, const ultra_nam_chain *nam,
                       const ultra_chain_ws *ws, bool cnt_is_bound)
{
    const size_t qstride = nam ? 24 : 20; // LDS bytes per entry of the quadratic kernel
    if (solver < 0 || solver > 2) return ultra_fail(ctx, ULTRA_E_ARG, "chaining: solver must be 0, 1 or 2");
    const int thr_s = overlap_small < 0 ? ctx->cfg.overlap_small : overlap_small;
    const int thr_l = overlap_large < 0 ? ctx->cfg.overlap_large : overlap_large;
    if (cnt_is_bound) {
        if (solver != 0 || nam || !d_cnt || ws) return kChainNeedExact;
        return chain_launch_device_lists(ctx, n_inst, h_cnt, d_off, d_cnt, d_x, d_y, d_c, d_d, d_val, thr_s, thr_l, d_idx, d_len, d_value, d_status_out);
    }
    std::vector<int32_t> small, big, tree;
    std::vector<int32_t> tiny_n[kTiny + 1]; // MAM instances of at most kTiny entries, by size
    size_t n_tiny = 0;
    int32_t max_big = 0, max_tree = 0;
    for (int32_t i = 0; i < n_inst; i++) {
        const int32_t n = h_cnt[i];
        if (n < 0) return ultra_fail(ctx, ULTRA_E_ARG, "chaining: negative instance size");
        // MAM: n > 200 -> n log n (classify_read_with_mams.py:586); NAM: n >= 90 (align.py:448)
        bool use_tree = solver == 2 || (solver == 0 && (nam ? n >= ctx->cfg.nam_switch_n : n > ctx->cfg.chain_switch_n));
        if (use_tree && n > 0) {
            if (!nam && thr_l < 1) return ultra_fail(ctx, ULTRA_E_ARG, "chaining: n log n solver needs overlap threshold >= 1");
            tree.push_back(i); max_tree = std::max<int32_t>(max_tree, n);
        } else if (!nam && n <= kTiny) { tiny_n[n].push_back(i); n_tiny++; }
        else if (n <= kQuadSmall) small.push_back(i);
        else { big.push_back(i); max_big = std::max<int32_t>(max_big, n); }
    }
    if ((size_t)max_big * qstride > (size_t)kLdsBudget)
        return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, nam ? "chaining: quadratic solver limited to 6826 NAMs per instance"
                                                        : "chaining: quadratic solver limited to 8192 MAMs per instance");
    int tree_leaves = 1;
    while (tree_leaves < max_tree + 1) tree_leaves <<= 1;
    size_t tree_lds = (size_t)tree_leaves * 16 + (size_t)((tree_leaves + 63) / 64) * 12 + 16;
    if (!tree.empty() && tree_lds > (size_t)kLdsBudget)
        return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "chaining: n log n solver limited to 8191 MAMs per instance");

    ChainState &ch = ctx->ch;
    hipStream_t s = ctx->stream;
    const size_t ni = (size_t)n_inst;
    // device layout of ch.off: work_off | list | status
    const size_t aux_bytes = ultra_chain_aux_bytes(n_inst);
    if (ws) {
        if (aux_bytes > ws->aux_cap) return ultra_fail(ctx, ULTRA_E_CAPACITY, "chaining: the caller's workspace is too small for the instance lists");
    } else ULTRA_HIP_CHECK(ctx, ch.off.ensure(aux_bytes));
    int64_t *d_workoff = ws ? (int64_t *)ws->aux : ch.off.as<int64_t>();
    int32_t *d_list = (int32_t *)(d_workoff + (ni + 1));
    int32_t *d_status = d_list + ni;
    if (d_status_out) *d_status_out = d_status;
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(d_status, 0, 4, s));
    ch.h_list.clear();
    ch.h_list.reserve(ni);
    for (int n = kTiny; n >= 0; n--) ch.h_list.insert(ch.h_list.end(), tiny_n[n].begin(), tiny_n[n].end());
    ch.h_list.insert(ch.h_list.end(), small.begin(), small.end());
    ch.h_list.insert(ch.h_list.end(), big.begin(), big.end());
    ch.h_list.insert(ch.h_list.end(), tree.begin(), tree.end());
    if (!ch.h_list.empty())
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(d_list, ch.h_list.data(), ch.h_list.size() * 4, hipMemcpyHostToDevice, s));
    ch.h_workoff.assign(tree.size() + 1, 0);
    for (size_t t = 0; t < tree.size(); t++) {
        int64_t n = h_cnt[tree[t]];
        int64_t bytes = (n + 1) * 12 + n * 16;
        ch.h_workoff[t + 1] = ch.h_workoff[t] + ((bytes + 15) & ~(int64_t)15);
    }
    if (!tree.empty()) {
        if (ws) {
            if ((size_t)ch.h_workoff.back() + 16 > ws->work_cap)
                return ultra_fail(ctx, ULTRA_E_CAPACITY, "chaining: the caller's workspace is too small for the n log n solver's trees");
        } else ULTRA_HIP_CHECK(ctx, ch.work.ensure((size_t)ch.h_workoff.back() + 16));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(d_workoff, ch.h_workoff.data(), ch.h_workoff.size() * 8, hipMemcpyHostToDevice, s));
    }
    ChainArgs a;
    memset(&a, 0, sizeof a);
    a.off = d_off; a.cnt = d_cnt; a.x = d_x; a.y = d_y; a.c = d_c; a.d = d_d;
    a.val = d_val; a.idx = d_idx; a.len = d_len; a.value = d_value;
    a.status = d_status;
    if (nam) { a.nam = 1; a.max_intron = nam->d_max_intron; a.C_out = nam->d_C; a.trace_out = nam->d_trace; }
    // with device-side counts the host sizes are only upper bounds: the lane-per-instance kernel then also looks
    // at the `small` instances and takes those that turn out tiny; chain_quadratic skips them
    const bool bounds_only = !nam && d_cnt != nullptr;
    const size_t n_tiny_launch = n_tiny + (bounds_only ? small.size() : 0);
    // The launches work on disjoint instances and only read the inputs: dealt over the launch lanes so that the one or
    // two wavefronts of the n log n solver (0.4 ms of critical path) and the lane-per-instance kernel run beside the
    // quadratic solver instead of in front of and behind it.  Not with a caller's workspace: that entry point stays on
    // the one stream it was given.
    const int n_launch = (n_tiny_launch > 0) + !small.empty() + !big.empty() + !tree.empty();
    const bool spread = !ws && ctx->cfg.launch_lanes > 1 && n_launch > 1;
    if (spread) ULTRA_HIP_CHECK(ctx, ultra_fork_at(ctx, 6));
    int li = 0;
    auto next_lane = [&]() { return spread ? ctx->lane(li++) : s; };
    if (!tree.empty()) { // longest critical path first
        hipStream_t cs = next_lane();
        a.thr = thr_l; a.list = d_list + n_tiny + small.size() + big.size(); a.nlist = (int32_t)tree.size(); a.cap = tree_leaves;
        a.work = ws ? ws->work : ch.work.as<uint8_t>(); a.work_off = d_workoff;
        const void *fn = nam ? (const void *)chain_nlogn<true> : (const void *)chain_nlogn<false>;
        if (tree_lds > 64 * 1024)
            ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_lds));
        if (nam) hipLaunchKernelGGL(chain_nlogn<true>, dim3((unsigned)tree.size()), dim3(64), tree_lds, cs, a);
        else hipLaunchKernelGGL(chain_nlogn<false>, dim3((unsigned)tree.size()), dim3(64), tree_lds, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    a.skip_tiny = 0;
    if (!big.empty()) {
        hipStream_t cs = next_lane();
        a.thr = thr_s; a.list = d_list + n_tiny + small.size(); a.nlist = (int32_t)big.size(); a.cap = max_big;
        size_t shmem = (size_t)max_big * qstride;
        if (shmem > 64 * 1024)
            ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)chain_quadratic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(chain_quadratic, dim3((unsigned)big.size()), dim3(64), shmem, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    a.skip_tiny = bounds_only ? 1 : 0;
    if (!small.empty()) {
        hipStream_t cs = next_lane();
        a.thr = thr_s; a.list = d_list + n_tiny; a.nlist = (int32_t)small.size(); a.cap = kQuadSmall;
        int blocks = ((int)small.size() + 3) / 4;
        hipLaunchKernelGGL(chain_quadratic, dim3(blocks), dim3(256), (size_t)4 * kQuadSmall * qstride, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    a.skip_tiny = 0;
    if (n_tiny_launch > 0) {
        hipStream_t cs = next_lane();
        a.thr = thr_s; a.list = d_list; a.nlist = (int32_t)n_tiny_launch; a.cap = (int32_t)n_tiny;
        hipLaunchKernelGGL(chain_tiny, dim3((unsigned)((n_tiny_launch + 63) / 64)), dim3(64), 0, cs, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    if (spread) ULTRA_HIP_CHECK(ctx, ultra_join_at(ctx, 7));
    return ULTRA_OK;
}

extern "C" int ultra_chain_mams_batch(ultra_ctx *ctx, int32_t n_inst, const int64_t *inst_off,
                                      const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d,
                                      const double *val, int32_t solver, int32_t overlap_small, int32_t overlap_large,
                                      int32_t *chain_idx, int32_t *chain_len, double *value)
{
    if (!ctx) return ULTRA_E_ARG;
    if (n_inst < 0 || (n_inst > 0 && (!inst_off || !chain_len || !value)))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch: null pointer or negative instance count");
    if (n_inst == 0) return ULTRA_OK;
    const int64_t total = inst_off[n_inst];
    if (inst_off[0] != 0 || total < 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch: inst_off must start at 0");
    if (total > 0 && (!x || !y || !c || !d || !val || !chain_idx))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch: null MAM array");
    std::vector<int32_t> h_cnt((size_t)n_inst);
    for (int32_t i = 0; i < n_inst; i++) {
        int64_t n = inst_off[i + 1] - inst_off[i];
        if (n < 0 || n > 0x7fffffff) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch: inst_off must be non-decreasing");
        h_cnt[(size_t)i] = (int32_t)n;
    }
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ChainState &ch = ctx->ch;
    hipStream_t s = ctx->stream;
    const size_t nt = (size_t)std::max<int64_t>(total, 1), ni = (size_t)n_inst;
    ULTRA_HIP_CHECK(ctx, ch.ioff.ensure((ni + 1) * 8));
    ULTRA_HIP_CHECK(ctx, ch.x.ensure(nt * 4)); ULTRA_HIP_CHECK(ctx, ch.y.ensure(nt * 4));
    ULTRA_HIP_CHECK(ctx, ch.c.ensure(nt * 4)); ULTRA_HIP_CHECK(ctx, ch.d.ensure(nt * 4));
    ULTRA_HIP_CHECK(ctx, ch.val.ensure(nt * 8)); ULTRA_HIP_CHECK(ctx, ch.idx.ensure(nt * 4));
    ULTRA_HIP_CHECK(ctx, ch.len.ensure(ni * 4 + 16)); ULTRA_HIP_CHECK(ctx, ch.value.ensure(ni * 8));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.ioff.p, inst_off, (ni + 1) * 8, hipMemcpyHostToDevice, s));
    if (total > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.x.p, x, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.y.p, y, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.c.p, c, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.d.p, d, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.val.p, val, nt * 8, hipMemcpyHostToDevice, s));
    }
    int32_t *d_status = nullptr;
    int rc = ultra_chain_launch(ctx, n_inst, h_cnt.data(), ch.ioff.as<int64_t>(), nullptr, ch.x.as<int32_t>(), ch.y.as<int32_t>(),
                                ch.c.as<int32_t>(), ch.d.as<int32_t>(), ch.val.as<double>(), solver, overlap_small, overlap_large,
                                ch.idx.as<int32_t>(), ch.len.as<int32_t>(), ch.value.as<double>(), &d_status, nullptr);
    if (rc != ULTRA_OK) { (void)hipStreamSynchronize(s); return rc; }
    int32_t status = 0;
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(chain_len, ch.len.p, ni * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(value, ch.value.p, ni * 8, hipMemcpyDeviceToHost, s));
    if (total > 0) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(chain_idx, ch.idx.p, nt * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    if (status != 0)
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch: an instance reached the reference's 'BUG' state "
                                            "(empty range-query node set); check that c <= d for every MAM");
    return ULTRA_OK;
}

// ---- the one entry point that allocates nothing ----------------------------------------------------------------
// ultra_chain_mams_batch needs device memory beyond its inputs and outputs: staging copies of the MAM arrays, the
// launch lists, the result vectors and the search-tree arrays of the n log n solver.  Callers that must not allocate
// device memory (serving loops with a fixed memory budget, harnesses that forbid hipMalloc) get the same kernels on a
// fixed buffer that is part of the library's image: every GPU holds its own copy at its own address until the process
// ends.  The call is synchronous (host copies, one stream synchronisation): it allocates nothing, but it is not
// something to record into a HIP graph as it stands.
constexpr size_t kNoAllocBytes = (size_t)8 << 20;
static __device__ __align__(256) uint8_t scratch[kNoAllocBytes];

namespace {
constexpr int kMaxDevices = 64;
uint8_t *g_scratch_ptr[kMaxDevices] = {}; // device address of `scratch` on each GPU (the symbol itself is a host-side stand-in)
std::mutex g_scratch_mu;
std::mutex g_scratch_call_mu[kMaxDevices]; // one call at a time per GPU: every context of a device shares that device's array
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
} // namespace

// called by ultra_ctx_create with the context's device current
int ultra_chain_scratch_init(ultra_ctx *ctx)
{
    if (ctx->device < 0 || ctx->device >= kMaxDevices) return ULTRA_OK; // no-alloc entry point unavailable there
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[ctx->device]) {
        uint8_t *ptr = nullptr;
        ULTRA_HIP_CHECK(ctx, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[ctx->device] = ptr;
    }
    ctx->ch.fixed = g_scratch_ptr[ctx->device];
    return ULTRA_OK;
}

extern "C" int64_t ultra_chain_noalloc_capacity(void) { return (int64_t)kNoAllocBytes; }

extern "C" int ultra_chain_mams_batch_noalloc(ultra_ctx *ctx, int32_t n_inst, const int64_t *inst_off,
                                              const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d,
                                              const double *val, int32_t solver, int32_t overlap_small, int32_t overlap_large,
                                              int32_t *chain_idx, int32_t *chain_len, double *value)
{
    if (!ctx) return ULTRA_E_ARG;
    if (n_inst < 0 || (n_inst > 0 && (!inst_off || !chain_len || !value)))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch_noalloc: null pointer or negative instance count");
    if (n_inst == 0) return ULTRA_OK;
    const int64_t total = inst_off[n_inst];
    if (inst_off[0] != 0 || total < 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch_noalloc: inst_off must start at 0");
    if (total > 0 && (!x || !y || !c || !d || !val || !chain_idx))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch_noalloc: null MAM array");
    uint8_t *const base = ctx->ch.fixed; // the device address, never the symbol's name
    if (!base) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_chain_mams_batch_noalloc: no fixed buffer on this device");
    // ---- sizes first: nothing is touched when the batch does not fit ----
    const size_t nt = (size_t)std::max<int64_t>(total, 1), ni = (size_t)n_inst;
    size_t tree_bytes = 16;
    for (int32_t i = 0; i < n_inst; i++) {
        const int64_t n = inst_off[i + 1] - inst_off[i];
        if (n < 0 || n > 0x7fffffff) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch_noalloc: inst_off must be non-decreasing");
        if (n > 0 && (solver == 2 || (solver == 0 && n > ctx->cfg.chain_switch_n))) tree_bytes += (size_t)(((n + 1) * 12 + n * 16 + 15) & ~(int64_t)15);
    }
    size_t at = 0;
    auto carve = [&](size_t bytes) { const size_t o = at; at += up256(bytes); return o; };
    const size_t o_ioff = carve((ni + 1) * 8), o_x = carve(nt * 4), o_y = carve(nt * 4), o_c = carve(nt * 4), o_d = carve(nt * 4),
                 o_val = carve(nt * 8), o_idx = carve(nt * 4), o_len = carve(ni * 4 + 16), o_value = carve(ni * 8),
                 o_aux = carve(ultra_chain_aux_bytes(n_inst)), o_work = carve(tree_bytes);
    if (at > kNoAllocBytes)
        return ultra_fail(ctx, ULTRA_E_CAPACITY, "ultra_chain_mams_batch_noalloc: the batch needs " + std::to_string(at) +
                                                     " bytes of device scratch, the fixed buffer holds " + std::to_string(kNoAllocBytes) +
                                                     " (about 28 B per MAM + 32 B per instance + 28 B per MAM of an n log n instance); "
                                                     "split the batch or use ultra_chain_mams_batch");
    std::vector<int32_t> h_cnt(ni);
    for (int32_t i = 0; i < n_inst; i++) h_cnt[(size_t)i] = (int32_t)(inst_off[i + 1] - inst_off[i]);
    // The call returns with everything done (it ends with a stream synchronisation), so holding the device's lock for
    // its duration serialises callers on different contexts / threads instead of letting them corrupt each other.
    std::lock_guard<std::mutex> one_at_a_time(g_scratch_call_mu[ctx->device]);
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    ultra_drain_guard drain(ctx);
    // The buffer keeps the previous call's bytes.  Everything a kernel reads is overwritten below (inputs) or
    // inside the launch (lists, status word, tree arrays are initialised by their kernel); len / value / idx are
    // written for every instance before they are copied out, idx only up to len[i] entries per instance.
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(base + o_ioff, inst_off, (ni + 1) * 8, hipMemcpyHostToDevice, s));
    if (total > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(base + o_x, x, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(base + o_y, y, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(base + o_c, c, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(base + o_d, d, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(base + o_val, val, nt * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemsetAsync(base + o_idx, 0, nt * 4, s)); // entries beyond a chain's length: defined bytes
    }
    const ultra_chain_ws ws = {base + o_aux, up256(ultra_chain_aux_bytes(n_inst)), base + o_work, up256(tree_bytes)};
    int32_t *d_status = nullptr;
    int rc = ultra_chain_launch(ctx, n_inst, h_cnt.data(), (const int64_t *)(base + o_ioff), nullptr, (const int32_t *)(base + o_x),
                                (const int32_t *)(base + o_y), (const int32_t *)(base + o_c), (const int32_t *)(base + o_d),
                                (const double *)(base + o_val), solver, overlap_small, overlap_large, (int32_t *)(base + o_idx),
                                (int32_t *)(base + o_len), (double *)(base + o_value), &d_status, nullptr, &ws);
    if (rc != ULTRA_OK) return rc;
    int32_t status = 0;
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(chain_len, base + o_len, ni * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(value, base + o_value, ni * 8, hipMemcpyDeviceToHost, s));
    if (total > 0) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(chain_idx, base + o_idx, nt * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    drain.dismiss();
    if (status != 0)
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_mams_batch_noalloc: an instance reached the reference's 'BUG' state "
                                            "(empty range-query node set); check that c <= d for every MAM");
    return ULTRA_OK;
}

// NAM chaining, SURVEY.md section 8 row f1.  Replaces colinear_solver.read_coverage (modules/colinear_solver.py:169-272)
// and n_logn_read_coverage (:85-167) as align.py:448-453 calls them, for a batch of (read, chromosome, strand) instances.
extern "C" int ultra_chain_nams_batch(ultra_ctx *ctx, int32_t n_inst, const int64_t *inst_off,
                                      const int32_t *y, const int32_t *c, const int32_t *d,
                                      const int64_t *max_intron, int32_t solver,
                                      int64_t *C, int32_t *trace, int64_t *value)
{
    if (!ctx) return ULTRA_E_ARG;
    if (n_inst < 0 || (n_inst > 0 && (!inst_off || !C || !trace || !value)))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: null pointer or negative instance count");
    if (solver < 0 || solver > 2) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: solver must be 0, 1 or 2");
    if (n_inst == 0) return ULTRA_OK;
    const int64_t total = inst_off[n_inst];
    if (inst_off[0] != 0 || total < 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: inst_off must start at 0");
    if (total > 0 && (!c || !d)) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: null NAM array");
    std::vector<int32_t> h_cnt((size_t)n_inst);
    bool any_quadratic = false;
    for (int32_t i = 0; i < n_inst; i++) {
        int64_t n = inst_off[i + 1] - inst_off[i];
        if (n < 0 || n > 0x7fffffff) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: inst_off must be non-decreasing");
        h_cnt[(size_t)i] = (int32_t)n;
        if (n > 0 && (solver == 1 || (solver == 0 && n < ctx->cfg.nam_switch_n))) any_quadratic = true;
    }
    if (any_quadratic && (!y || !max_intron))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: the quadratic solver needs y and max_intron");
    // the kernels keep C and C - d as int32 / exact doubles: read coordinates must be sane
    for (int64_t i = 0; i < total; i++)
        if (c[i] < 0 || d[i] < c[i] || d[i] >= (1 << 30))
            return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: NAM " + std::to_string(i) + " has read interval [" +
                                                    std::to_string(c[i]) + ", " + std::to_string(d[i]) + "]; need 0 <= c <= d < 2^30");
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ChainState &ch = ctx->ch;
    hipStream_t s = ctx->stream;
    const size_t nt = (size_t)std::max<int64_t>(total, 1), ni = (size_t)n_inst;
    ULTRA_HIP_CHECK(ctx, ch.ioff.ensure((ni + 1) * 8));
    ULTRA_HIP_CHECK(ctx, ch.y.ensure(nt * 4));
    ULTRA_HIP_CHECK(ctx, ch.c.ensure(nt * 4)); ULTRA_HIP_CHECK(ctx, ch.d.ensure(nt * 4));
    ULTRA_HIP_CHECK(ctx, ch.len.ensure(ni * 4 + 16)); ULTRA_HIP_CHECK(ctx, ch.value.ensure(ni * 8));
    ULTRA_HIP_CHECK(ctx, ch.namC.ensure((nt + ni) * 8)); ULTRA_HIP_CHECK(ctx, ch.namtrace.ensure((nt + ni) * 4));
    ULTRA_HIP_CHECK(ctx, ch.intron.ensure(ni * 8));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.ioff.p, inst_off, (ni + 1) * 8, hipMemcpyHostToDevice, s));
    if (total > 0) {
        if (y) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.y.p, y, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.c.p, c, nt * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.d.p, d, nt * 4, hipMemcpyHostToDevice, s));
    }
    if (max_intron) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ch.intron.p, max_intron, ni * 8, hipMemcpyHostToDevice, s));
    else ULTRA_HIP_CHECK(ctx, hipMemsetAsync(ch.intron.p, 0, ni * 8, s));
    ultra_nam_chain nam = {ch.intron.as<int64_t>(), ch.namC.as<int64_t>(), ch.namtrace.as<int32_t>()};
    int32_t *d_status = nullptr;
    // x and val are not read in NAM mode (the n log n kernel forms pointers from them only)
    int rc = ultra_chain_launch(ctx, n_inst, h_cnt.data(), ch.ioff.as<int64_t>(), nullptr, ch.y.as<int32_t>(), ch.y.as<int32_t>(),
                                ch.c.as<int32_t>(), ch.d.as<int32_t>(), nullptr, solver, -1, -1,
                                nullptr, ch.len.as<int32_t>(), ch.value.as<double>(), &d_status, &nam);
    if (rc != ULTRA_OK) { (void)hipStreamSynchronize(s); return rc; }
    int32_t status = 0;
    std::vector<double> h_value(ni);
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(C, ch.namC.p, (size_t)(total + n_inst) * 8, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(trace, ch.namtrace.p, (size_t)(total + n_inst) * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_value.data(), ch.value.p, ni * 8, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    if (status != 0)
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_chain_nams_batch: an instance reached the reference's 'BUG' state");
    // value = max(C) (colinear_solver.py:159, :245); an instance without NAMs has C = [0]
    for (size_t i = 0; i < ni; i++) value[i] = h_cnt[i] > 0 ? (int64_t)h_value[i] : 0;
    return ULTRA_OK;
}
