// Candidate enumeration ON THE DEVICE: the packed index lives in HBM, one wavefront walks one (NAM chain, read) instance.
//
// Same work as plan.hip (the host planner), i.e. the dictionary-driven part of
//     classify_read_with_mams.main                      /root/reference/modules/classify_read_with_mams.py:406-523
//       get_unique_exon_and_flank_locations             :180-253
//       get_unique_segment_and_flank_choordinates       :256-300
//       the candidate loops and the edlib call shapes   :303-377, :441-523
// but the host never sees a candidate: it hands over the NAM chains and the reads of a batch, and the MAM stage's
// inputs (candidates, (query, target, k) pairs over the resident sequence pool + the batch's reads) are produced
// in device memory where ultra_myers_stage_pairs / ultra_mam_run pick them up.  The host planner costs ~11 us per
// instance and core, the device stage behind it ~0.13 us: this file is what lets a batch be bound by the GPU.
//
// Index layout (built once per GPU by ultra_dindex_create from the host index):
//   * every distinct segment gets an id = its rank in (start, stop, chromosome) order, i.e. the order in which the
//     reference's sorted(..., key=start) and this repository's host planner (loc_less) list segments; a record holds
//     its coordinates, its slice of the sequence pool and its slice of the segment -> gene list;
//   * one open-addressing hash table answers "is this exon_part_id a part (-> its segment ids) or a flank";
//   * gene -> small-segment ids in the stored order; the raw sequence pool.
// Per instance (template capacity CAP entries in LDS; instances that do not fit are re-run with a larger CAP):
//   raw hits (segment, NAM) in NAM order -> sort by (segment, arrival) -> unique hit segments with the first NAM's
//   start and the last NAM's end; genes of the hit segments -> sort/unique; entry list = small segments of every gene
//   in gene order, then the hit segments, de-duplicated by first arrival, sorted by (start, arrival) = the reference's
//   candidate order; hit segments with equal starts in the iteration order of the reference's Python set (pyset.h); flanks in first-seen order.  Two passes (count, scan, fill) give deterministic CSR outputs.
#include "common.h"

#include <algorithm>
#include <string.h>
#include <string>

#include "plan.h"
#include "pyset.h"

namespace {

struct SegRec { int32_t chr, s0, s1, len; int64_t off; int32_t gene_b, gene_e; };   // off < 0: no sequence in the index
struct FlankRec { int32_t chr, s0, s1, len; int64_t off; };
struct HtEntry { int32_t chr, p0, p1, val; };   // val: part id, or kFlankBit | flank id; -1 = empty slot
constexpr int32_t kFlankBit = 0x40000000;
constexpr int kMaxFlanks = 16;
constexpr int kMaxGenes = 64;

__host__ __device__ inline uint32_t ht_hash(int32_t chr, int32_t p0, int32_t p1)
{
    uint64_t h = (uint64_t)(uint32_t)chr * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)p0 + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    h ^= ((uint64_t)(uint32_t)p1 + 0x1CE4E5B9ull) * 0x94D049BB133111EBull;
    h ^= h >> 29;
    return (uint32_t)h;
}

enum { PERR_NONE = 0, PERR_NO_PART = 1, PERR_NO_GENE = 2, PERR_NO_SEQ = 3, PERR_RANGE = 4, PERR_TOO_BIG = 5 };

struct PlanArgs {
    const HtEntry *ht; uint32_t ht_mask;
    const int32_t *part_off, *part_seg;
    const SegRec *seg;
    const int32_t *gene_flat, *gene_small_off, *gene_small;
    const FlankRec *flank;
    const uint8_t *pool;
    int64_t pool_len;
    int32_t n_inst;
    const int64_t *nam_off, *nam_part, *nam_x, *nam_y, *read_off;
    const int32_t *list; int32_t n_list;
    const int32_t *n_list_dev; // when set: the list's length lies on the device (the grid only bounds it from above)
    int32_t *n_cand, *n_pair;
    int32_t *overflow_list, *overflow_cnt, *err, *max_cand;
    const int64_t *cand_off, *pair_off;
    uint8_t *kind;
    int32_t *e_start, *e_stop, *pair, *pairc, *seq_len, *seq_id, *chr;
    PairRec *rec;        // the pairs, unsorted (MyersState::u_rec)
    int32_t *read_len;
    int64_t *fps, *lps;
    // what the count pass found out about an instance, kept for the fill pass (which then starts at the candidate list
    // instead of repeating the look-ups and sorts): state_stride ints per instance / list entry, plan_state_ints<CAP>()
    int32_t *state; int64_t state_stride;
};

__device__ __forceinline__ void plan_error(const PlanArgs &a, int code, int inst)
{
    if (atomicCAS(&a.err[0], 0, code) == 0) a.err[1] = inst;
}

// Python's s[start:stop] on a sequence of length n (either bound may be absent); as plan.hip's py_slice
__device__ __forceinline__ void dev_py_slice(int64_t n, bool has_start, int64_t start, bool has_stop, int64_t stop, int64_t &b, int64_t &e)
{
    if (!has_start) start = 0;
    else { if (start < 0) start += n; if (start < 0) start = 0; if (start > n) start = n; }
    if (!has_stop) stop = n;
    else { if (stop < 0) stop += n; if (stop < 0) stop = 0; if (stop > n) stop = n; }
    b = start; e = stop < start ? start : stop;
}

__device__ __forceinline__ bool dev_overlaps(int64_t a0, int64_t a1, int64_t b0, int64_t b1) // classify_read_with_mams.py:175-178
{
    return (a0 <= b0 && b0 <= a1) || (a0 <= b1 && b1 <= a1) || (b0 <= a0 && a0 <= a1 && a1 <= b1);
}

// ascending bitonic sort of keys[0 .. n) in LDS by the 64 lanes of the wavefront (n <= cap, cap a power of two)
__device__ __forceinline__ void wave_sort(uint64_t *keys, int n, int lane)
{
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = n + lane; i < np2; i += 64) keys[i] = ~(uint64_t)0;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int t = lane; t < (np2 >> 1); t += 64) {
                const int i = ((t / s) * (s << 1)) + (t % s);
                const int p = i + s;
                const bool up = ((i & k) == 0);
                const uint64_t A = keys[i], B = keys[p];
                if ((A > B) == up) { keys[i] = B; keys[p] = A; }
            }
            __syncthreads();
        }
    }
}

// One candidate of the instance, in the reference's emission order.
struct CandView {
    int kind;
    int32_t chr, s0, s1;
    int64_t off;    // pool offset of the piece handed to add_segment_to_mam
    int64_t len;    // its length
    bool whole;     // the piece is the whole segment / flank
    int full_idx;   // partial pieces: index of the full candidate of the same segment / flank
    bool missing;   // no sequence in the index
};

template <int CAP, bool FILL>
__global__ void __launch_bounds__(64) k_plan(PlanArgs a)
{
    extern __shared__ uint64_t plan_lds[];
    uint64_t *keysA = plan_lds;                 // CAP
    uint64_t *keysB = keysA + CAP;              // CAP
    int32_t *A1 = (int32_t *)(keysB + CAP);     // CAP   raw hit x, later: segment id by arrival rank
    int32_t *A2 = A1 + CAP;                     // CAP   raw hit y, later: segment ids of the full candidates, sorted
    int32_t *uh_seg = A2 + CAP;                 // CAP   unique hit segments (ascending id)
    int32_t *uh_x = uh_seg + CAP;               // CAP
    int32_t *uh_y = uh_x + CAP;                 // CAP
    int32_t *uh_full = uh_y + CAP;              // CAP   index of the hit segment among the full candidates
    int32_t *genes = uh_full + CAP;             // kMaxGenes + 1 (then offsets)
    int32_t *goff = genes + kMaxGenes + 1;      // kMaxGenes + 1
    int32_t *fl_id = goff + kMaxGenes + 1;      // kMaxFlanks each:
    int32_t *fl_x = fl_id + kMaxFlanks, *fl_y = fl_x + kMaxFlanks, *fl_pos = fl_y + kMaxFlanks, *fl_part = fl_pos + kMaxFlanks;
    int32_t *pre = (int32_t *)keysA;            // pairs in front of every full candidate (keysA is dead by then)

    const int lane = threadIdx.x;
    const int inst = a.list ? a.list[blockIdx.x] : (int)blockIdx.x;
    if (a.list ? (int)blockIdx.x >= (a.n_list_dev ? *a.n_list_dev : a.n_list) : inst >= a.n_inst) return;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int64_t n0 = a.nam_off[inst], n1 = a.nam_off[inst + 1];
    const int64_t lr = a.read_off[inst + 1] - a.read_off[inst];
    int64_t fps = (int64_t)1 << 32, lps = 0;
    int nh = 0, nf = 0;
    bool too_big = false;
    int U = 0, C0 = 0, nfp = 0, npair = 0;
    int32_t *const stp = a.state + (int64_t)(a.list ? (int)blockIdx.x : inst) * a.state_stride;
    constexpr int kStateHdr = 16;
    if constexpr (FILL) {
        // ---- the count pass's findings: header, then the lists back to back ----
        // an overflow instance of the small-capacity launch: the launch with the larger capacity takes it (its count pass has
        // overwritten n_cand by now, so the mark is in the state's header)
        if (a.n_cand[inst] < 0 || stp[0] < 0) return;
        C0 = stp[0]; U = stp[1]; nf = stp[2]; nfp = stp[3]; npair = stp[4];
        fps = (int64_t)(((uint64_t)(uint32_t)stp[6] << 32) | (uint32_t)stp[5]);
        lps = (int64_t)(((uint64_t)(uint32_t)stp[8] << 32) | (uint32_t)stp[7]);
        const int32_t *q = stp + kStateHdr;
        for (int i = lane; i < C0; i += 64) A2[i] = q[i];
        q += C0;
        for (int i = lane; i < U; i += 64) { uh_seg[i] = q[i]; uh_x[i] = q[U + i]; uh_y[i] = q[2 * U + i]; uh_full[i] = q[3 * U + i]; }
        q += 4 * U;
        for (int i = lane; i < C0 + nf; i += 64) pre[i] = q[i];
        q += C0 + nf;
        for (int i = lane; i < nf; i += 64) { fl_id[i] = q[i]; fl_x[i] = q[nf + i]; fl_y[i] = q[2 * nf + i]; fl_pos[i] = q[3 * nf + i]; fl_part[i] = q[4 * nf + i]; }
        __syncthreads();
    } else {
    // ---- get_unique_exon_and_flank_locations (:180-253) ----
    for (int64_t base = n0; base < n1; base += 64) {
        const int64_t u = base + lane;
        const bool valid = u < n1;
        int64_t c64 = 0, p064 = 0, p164 = 0, x64 = 0, y64 = 0;
        if (valid) { c64 = a.nam_part[3 * u]; p064 = a.nam_part[3 * u + 1]; p164 = a.nam_part[3 * u + 2]; x64 = a.nam_x[u]; y64 = a.nam_y[u]; }
        const int64_t LIM = 0x7fffffff;
        if (valid && (c64 < 0 || c64 > LIM || p064 < 0 || p064 > LIM || p164 < 0 || p164 > LIM || x64 < 0 || x64 > LIM || y64 < 0 || y64 > LIM))
            plan_error(a, PERR_RANGE, inst);
        const int32_t chr = (int32_t)c64, p0 = (int32_t)p064, p1 = (int32_t)p164, x = (int32_t)x64, y = (int32_t)y64;
        int32_t val = -1;
        if (valid) {
            uint32_t h = ht_hash(chr, p0, p1) & a.ht_mask;
            for (;;) {
                const HtEntry e = a.ht[h];
                if (e.val == -1) break;
                if (e.chr == chr && e.p0 == p0 && e.p1 == p1) { val = e.val; break; }
                h = (h + 1) & a.ht_mask;
            }
        }
        if (valid && val == -1) plan_error(a, PERR_NO_PART, inst);
        const bool is_part = valid && val != -1 && !(val & kFlankBit);
        const bool is_flank = valid && val != -1 && (val & kFlankBit);
        // first / last part bounds (:221-226): a running minimum and maximum, whatever the order
        {
            int64_t mn = is_part ? (int64_t)p1 : (int64_t)1 << 40, mx = is_part ? (int64_t)p0 : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int64_t on = __shfl_xor(mn, o, 64), ox = __shfl_xor(mx, o, 64);
                mn = on < mn ? on : mn; mx = ox > mx ? ox : mx;
            }
            if (mn <= fps) fps = mn;
            if (mx >= lps) lps = mx;
        }
        // flanks: first-seen list with the first NAM's start and the last NAM's end; few per chain, taken one by one
        for (uint64_t fm = __ballot(is_flank); fm; fm &= fm - 1) {
            const int j = __ffsll((long long)fm) - 1;
            const int32_t v = __shfl(val, j, 64), xj = __shfl(x, j, 64), yj = __shfl(y, j, 64);
            const int f = v & ~kFlankBit;
            int found = -1;
            for (int t = 0; t < nf; t++) if (fl_id[t] == f) found = t;
            __syncthreads();
            if (found < 0) {
                if (nf < kMaxFlanks) { if (lane == 0) { fl_id[nf] = f; fl_x[nf] = xj; fl_y[nf] = yj; } nf++; }
                else too_big = true;
            } else if (lane == 0) fl_y[found] = yj;
            __syncthreads();
        }
        // parts: the (NAM, segment of its part) pairs of these 64 NAMs are laid out one after the other and tested 64 at
        // a time, one pair per lane - in (NAM, segment) order, so the position a hit is appended at is its arrival rank.
        // (One NAM at a time, with the lanes over that part's handful of segments, cost three dependent loads per NAM.)
        const int sb = is_part ? a.part_off[val] : 0;
        const int len = is_part ? a.part_off[val + 1] - sb : 0;
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        const int P = incl - len, T = __shfl(incl, 63, 64);
        for (int b0 = 0; b0 < T; b0 += 64) {
            const int idx = b0 + lane;
            int lo = 0, hi = 63; // largest lane j with P_j <= idx: the NAM this pair belongs to
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int mid = (lo + hi + 1) >> 1;
                const int pm = __shfl(P, mid, 64);
                if (pm <= idx) lo = mid; else hi = mid - 1;
            }
            const int sbj = __shfl(sb, lo, 64), pj = __shfl(P, lo, 64); // cross-lane reads stay outside the divergent part below
            const int32_t xj = __shfl(x, lo, 64), yj = __shfl(y, lo, 64);
            bool ov = false;
            int32_t sid = 0;
            if (idx < T) {
                sid = a.part_seg[sbj + (idx - pj)];
                const SegRec r = a.seg[sid];
                ov = dev_overlaps(r.s0, r.s1, xj, yj);
            }
            const uint64_t m = __ballot(ov);
            const int pos = nh + __popcll(m & lt_mask);
            if (ov && pos < CAP) { keysA[pos] = ((uint64_t)(uint32_t)sid << 32) | (uint32_t)pos; A1[pos] = xj; A2[pos] = yj; }
            nh += __popcll(m);
        }
    }
    if (nh > CAP) too_big = true;
    __syncthreads();

    // ---- unique hit segments: first NAM's start, last NAM's end ----
    if (!too_big && nh > 0) {
        wave_sort(keysA, nh, lane);
        for (int base = 0; base < nh; base += 64) {
            const int i = base + lane;
            bool first = false, last = false;
            uint64_t kx = 0;
            if (i < nh) {
                kx = keysA[i];
                const uint32_t sid = (uint32_t)(kx >> 32);
                first = i == 0 || (uint32_t)(keysA[i - 1] >> 32) != sid;
                last = i == nh - 1 || (uint32_t)(keysA[i + 1] >> 32) != sid;
            }
            const uint64_t m = __ballot(first);
            const int run = U + __popcll(m & lt_mask) + (first ? 1 : 0) - 1; // index of this entry's run
            if (i < nh) {
                const int raw = (int)(uint32_t)kx;
                if (first) { uh_seg[run] = (int32_t)(kx >> 32); uh_x[run] = A1[raw]; uh_full[run] = raw; }
                if (last) uh_y[run] = A2[raw];
            }
            U += __popcll(m);
        }
    }
    __syncthreads();

    // ---- hit segments that START at the same coordinate: the reference's sorted(set(...), key=start) leaves them in
    //      the set's iteration order, which decides between identical partial pieces later on (pyset.h) ----
    if (!too_big && U > 1) {
        static_assert(pyset::final_size(CAP) * sizeof(uint16_t) <= CAP * sizeof(uint64_t), "the set's table is laid over keysB");
        bool tie = false;
        for (int i = lane; i + 1 < U; i += 64) tie |= a.seg[uh_seg[i]].s0 == a.seg[uh_seg[i + 1]].s0;
        if (__ballot(tie)) {
            // insertion order = order of first arrival among the raw hits
            for (int i = lane; i < U; i += 64) keysB[i] = ((uint64_t)(uint32_t)uh_full[i] << 32) | (uint32_t)i;
            __syncthreads();
            wave_sort(keysB, U, lane);
            for (int k = lane; k < U; k += 64) {
                const int r = (int)(uint32_t)keysB[k];
                const SegRec sr = a.seg[uh_seg[r]];
                A1[k] = r;
                keysA[k] = pyset::tuple3_hash((uint64_t)sr.chr, (uint64_t)sr.s0, (uint64_t)sr.s1);
            }
            __syncthreads();
            uint16_t *table = (uint16_t *)keysB, *list = (uint16_t *)A2;
            if (lane == 0) genes[0] = (int32_t)pyset::build(keysA, U, table, list);
            __syncthreads();
            const int size = genes[0];
            for (int sl = lane; sl < size; sl += 64) {
                const int v = table[sl];
                if (v) uh_full[A1[v - 1]] = sl;
            }
            __syncthreads();
            for (int i = lane; i < U; i += 64)
                keysA[i] = ((uint64_t)(uint32_t)a.seg[uh_seg[i]].s0 << 32) | ((uint64_t)(uint32_t)uh_full[i] << 12) | (uint32_t)i;
            __syncthreads();
            wave_sort(keysA, U, lane);
            for (int q = lane; q < U; q += 64) { const int r = (int)(keysA[q] & 0xfff); A1[q] = uh_seg[r]; A2[q] = uh_x[r]; uh_full[q] = uh_y[r]; }
            __syncthreads();
            for (int q = lane; q < U; q += 64) { uh_seg[q] = A1[q]; uh_x[q] = A2[q]; uh_y[q] = uh_full[q]; }
            __syncthreads();
        }
    }

    // ---- get_unique_segment_and_flank_choordinates (:256-300): small segments of every hit gene, then the hit segments ----
    if (!too_big && U > 0) {
        // genes of the hit segments -> sorted, unique
        int ng_raw = 0;
        for (int base = 0; base < U; base += 64) {
            const int i = base + lane;
            int gb = 0, ge = 0;
            if (i < U) {
                const SegRec r = a.seg[uh_seg[i]];
                gb = r.gene_b; ge = r.gene_e;
                if (gb < 0) { plan_error(a, PERR_NO_GENE, inst); gb = ge = 0; }
            }
            // lanes append their gene lists one lane after the other (lists are 1-2 entries long)
            for (int j = 0; j < 64 && base + j < U; j++) {
                const int jb = __shfl(gb, j, 64), je = __shfl(ge, j, 64);
                for (int t = jb + lane; t < je; t += 64) {
                    const int pos = ng_raw + (t - jb);
                    if (pos < CAP) keysB[pos] = (uint64_t)(uint32_t)a.gene_flat[t];
                }
                ng_raw += je - jb;
            }
        }
        if (ng_raw > CAP) too_big = true;
        int ng = 0;
        if (!too_big) {
            __syncthreads();
            wave_sort(keysB, ng_raw, lane);
            for (int base = 0; base < ng_raw; base += 64) {
                const int i = base + lane;
                const bool first = i < ng_raw && (i == 0 || keysB[i - 1] != keysB[i]);
                const uint64_t m = __ballot(first);
                const int pos = ng + __popcll(m & lt_mask);
                if (first && pos < kMaxGenes) genes[pos] = (int32_t)keysB[i];
                ng += __popcll(m);
            }
            if (ng > kMaxGenes) too_big = true;
        }
        __syncthreads();
        int E = 0;
        if (!too_big) {
            // entry list in arrival order: key = (segment id, arrival rank)
            if (lane == 0) {
                int acc = 0;
                for (int g = 0; g < ng; g++) { goff[g] = acc; acc += a.gene_small_off[genes[g] + 1] - a.gene_small_off[genes[g]]; }
                goff[ng] = acc;
            }
            __syncthreads();
            const int Es = goff[ng];
            E = Es + U;
            if (E > CAP) too_big = true;
            else {
                for (int g = 0; g < ng; g++) {
                    const int sb = a.gene_small_off[genes[g]], cntg = goff[g + 1] - goff[g];
                    for (int t = lane; t < cntg; t += 64) {
                        const int rank = goff[g] + t;
                        keysA[rank] = ((uint64_t)(uint32_t)a.gene_small[sb + t] << 32) | (uint32_t)rank;
                    }
                }
                for (int i = lane; i < U; i += 64) keysA[Es + i] = ((uint64_t)(uint32_t)uh_seg[i] << 32) | (uint32_t)(Es + i);
                __syncthreads();
                wave_sort(keysA, E, lane);
                // first arrival of every segment -> key (start, arrival rank); A1[rank] = segment id
                for (int base = 0; base < E; base += 64) {
                    const int i = base + lane;
                    bool first = false;
                    uint64_t kx = 0;
                    if (i < E) { kx = keysA[i]; first = i == 0 || (uint32_t)(keysA[i - 1] >> 32) != (uint32_t)(kx >> 32); }
                    const uint64_t m = __ballot(first);
                    const int pos = C0 + __popcll(m & lt_mask);
                    if (first) {
                        const int32_t sid = (int32_t)(kx >> 32);
                        const uint32_t rank = (uint32_t)kx;
                        A1[rank] = sid;
                        keysB[pos] = ((uint64_t)(uint32_t)a.seg[sid].s0 << 32) | rank;
                    }
                    C0 += __popcll(m);
                }
                __syncthreads();
                wave_sort(keysB, C0, lane);
                for (int i = lane; i < C0; i += 64) A2[i] = A1[(uint32_t)keysB[i]];
                __syncthreads();
                // where every hit segment sits among the full candidates
                for (int i = lane; i < U; i += 64) {
                    const int32_t sid = uh_seg[i];
                    int at = 0;
                    for (int t = 0; t < C0; t++) if (A2[t] == sid) { at = t; break; }
                    uh_full[i] = at;
                }
            }
        }
    }
    __syncthreads();
    if (too_big) {
        if (CAP >= 2048) { plan_error(a, PERR_TOO_BIG, inst); if (lane == 0) { a.n_cand[inst] = 0; a.n_pair[inst] = 0; } }
        else if (lane == 0) { const int w = atomicAdd(a.overflow_cnt, 1); a.overflow_list[w] = inst; a.n_cand[inst] = -1; a.n_pair[inst] = 0; stp[0] = -1; }
        return;
    }

    // ---- flanks: candidate order by start (stable in first-seen order), 5 % rule for the partial pieces (:288-295) ----
    if (lane == 0) {
        for (int t = 0; t < nf; t++) {
            const FlankRec r = a.flank[fl_id[t]];
            int rank = 0;
            for (int o = 0; o < nf; o++) { const int32_t so = a.flank[fl_id[o]].s0; if (so < r.s0 || (so == r.s0 && o < t)) rank++; }
            fl_pos[t] = rank;
            const double width = (double)(r.s1 - r.s0);
            const bool part = (double)(fl_x[t] - r.s0) > 0.05 * width || (double)(r.s1 - fl_y[t]) > 0.05 * width;
            fl_part[t] = part ? nfp : -1;
            if (part) nfp++;
        }
        genes[0] = nfp;
    }
    __syncthreads();
    nfp = genes[0];
    } // !FILL
    const int B1 = C0, B2 = C0 + nf, B3 = B2 + 2 * U, C = B3 + 2 * nfp;

    auto view = [&](int ci) {
        CandView v;
        v.missing = false; v.full_idx = -1;
        int64_t wlen, woff;
        if (ci < B1) {
            const SegRec r = a.seg[A2[ci]];
            v.kind = 0; v.chr = r.chr; v.s0 = r.s0; v.s1 = r.s1; woff = r.off; wlen = r.len;
            v.off = woff; v.len = wlen; v.whole = true;
        } else if (ci < B2) {
            int t = 0;
            for (int o = 0; o < nf; o++) if (fl_pos[o] == ci - B1) t = o;
            const FlankRec r = a.flank[fl_id[t]];
            v.kind = 1; v.chr = r.chr; v.s0 = r.s0; v.s1 = r.s1; woff = r.off; wlen = r.len;
            v.off = woff; v.len = wlen; v.whole = true;
        } else {
            int j, g0, g1;
            if (ci < B3) {
                j = ci - B2;
                const int h = j >> 1;
                const SegRec r = a.seg[uh_seg[h]];
                v.kind = 2 + (j & 1); v.chr = r.chr; v.s0 = r.s0; v.s1 = r.s1; woff = r.off; wlen = r.len;
                g0 = uh_x[h]; g1 = uh_y[h];
                v.full_idx = uh_full[h];
            } else {
                j = ci - B3;
                int t = 0;
                for (int o = 0; o < nf; o++) if (fl_part[o] == (j >> 1)) t = o;
                const FlankRec r = a.flank[fl_id[t]];
                v.kind = 4 + (j & 1); v.chr = r.chr; v.s0 = r.s0; v.s1 = r.s1; woff = r.off; wlen = r.len;
                g0 = fl_x[t]; g1 = fl_y[t];
                v.full_idx = B1 + fl_pos[t];
            }
            int64_t b, e;
            if (!(j & 1)) dev_py_slice(wlen, true, (int64_t)g0 - v.s0, false, 0, b, e);                    // seq[g0 - start:]
            else dev_py_slice(wlen, false, 0, true, wlen - ((int64_t)v.s1 - ((int64_t)g1 + 1)), b, e);       // seq[:len - (stop - (g1 + 1))]
            v.off = woff + b; v.len = e - b; v.whole = b == 0 && e == wlen;
        }
        if (woff < 0) { v.missing = true; v.off = 0; v.len = 0; }
        return v;
    };
    // the edlib calls add_segment_to_mam would make for a candidate (:304-377): which of them are new pairs
    auto calls = [&](const CandView &v, bool &has_pr, bool &has_prc, bool &dup) {
        const int64_t width = (int64_t)v.s1 - v.s0;
        const bool skip = v.len == 0 || lr == 0 || (v.kind >= 2 && v.len <= 5); // pieces of at most 5 bases are never aligned (:488-520)
        has_pr = !skip && width >= 5;
        has_prc = !skip && (double)width >= 0.8 * (double)lr;
        dup = v.kind >= 2 && v.whole; // the same (query, target, k) as the full candidate's call: shared, as the host planner does
    };

    if constexpr (!FILL) {
    // ---- pass over the candidates: pair numbering (prefix sums in candidate order) ----
    for (int base = 0; base < C; base += 64) {
        const int ci = base + lane;
        int add = 0;
        if (ci < C) {
            const CandView v = view(ci);
            if (v.missing) plan_error(a, PERR_NO_SEQ, inst);
            bool hp, hpc, dup;
            calls(v, hp, hpc, dup);
            add = dup ? 0 : (hp ? 1 : 0) + (hpc ? 1 : 0);
        }
        int incl = add;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (ci < B2) pre[ci] = npair + incl - add;
        npair += __shfl(incl, 63, 64);
    }
    __syncthreads();
    if (lane == 0) {
        a.n_cand[inst] = C; a.n_pair[inst] = npair; if (C > *(volatile int32_t *)a.max_cand) atomicMax(a.max_cand, C); // one address: an unconditional atomic per wavefront serialised the launch (3 ms)
        stp[0] = C0; stp[1] = U; stp[2] = nf; stp[3] = nfp; stp[4] = npair;
        stp[5] = (int32_t)(uint32_t)(uint64_t)fps; stp[6] = (int32_t)(uint32_t)((uint64_t)fps >> 32);
        stp[7] = (int32_t)(uint32_t)(uint64_t)lps; stp[8] = (int32_t)(uint32_t)((uint64_t)lps >> 32);
    }
    int32_t *q = stp + kStateHdr;
    for (int i = lane; i < C0; i += 64) q[i] = A2[i];
    q += C0;
    for (int i = lane; i < U; i += 64) { q[i] = uh_seg[i]; q[U + i] = uh_x[i]; q[2 * U + i] = uh_y[i]; q[3 * U + i] = uh_full[i]; }
    q += 4 * U;
    for (int i = lane; i < B2; i += 64) q[i] = pre[i];
    q += B2;
    for (int i = lane; i < nf; i += 64) { q[i] = fl_id[i]; q[nf + i] = fl_x[i]; q[2 * nf + i] = fl_y[i]; q[3 * nf + i] = fl_pos[i]; q[4 * nf + i] = fl_part[i]; }
    return;
    } // !FILL

    // ---- fill ----
    const int64_t cbase = a.cand_off[inst], pbase = a.pair_off[inst];
    const int64_t read_at = a.pool_len + a.read_off[inst];
    if (lane == 0) { a.read_len[inst] = (int32_t)lr; a.fps[inst] = fps; a.lps[inst] = lps; }
    int run = 0;
    for (int base = 0; base < C; base += 64) {
        const int ci = base + lane;
        int add = 0;
        CandView v = {};
        bool hp = false, hpc = false, dup = false;
        if (ci < C) {
            v = view(ci);
            calls(v, hp, hpc, dup);
            add = dup ? 0 : (hp ? 1 : 0) + (hpc ? 1 : 0);
        }
        int incl = add;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (ci < C) {
            const int first_new = run + incl - add;
            int pr = -1, prc = -1;
            const int64_t width = (int64_t)v.s1 - v.s0, ls = v.len;
            const int32_t kk = (int32_t)(0.4 * (double)(lr < ls ? lr : ls)); // the binding truncates the float
            if (dup) {
                // the full candidate of the same segment made the same calls
                const CandView fv = view(v.full_idx);
                bool fhp, fhpc, fdup;
                calls(fv, fhp, fhpc, fdup);
                const int fbase = pre[v.full_idx];
                if (hp) pr = fbase;
                if (hpc) prc = fbase + (fhp ? 1 : 0);
            } else {
                int w = first_new;
                if (hp) {
                    pr = w++;
                    const int64_t g = pbase + pr;
                    a.rec[g] = PairRec{v.off, read_at, (int32_t)ls, (int32_t)lr, width >= 9 ? kk : 0, 0};
                }
                if (hpc) {
                    prc = w++;
                    const int64_t g = pbase + prc;
                    a.rec[g] = PairRec{read_at, v.off, (int32_t)lr, (int32_t)ls, kk, 0};
                }
            }
            // equal sequences share an id within the partial-segment loop and within the partial-flank loop (:479-523)
            int sid = ci;
            if (v.kind >= 2) {
                const int gb = v.kind < 4 ? B2 : B3;
                for (int cj = gb; cj < ci; cj++) {
                    const CandView o = view(cj);
                    if (o.len != v.len) continue;
                    bool eq = o.off == v.off;
                    if (!eq) {
                        eq = true;
                        for (int64_t t = 0; t < v.len; t++) if (a.pool[o.off + t] != a.pool[v.off + t]) { eq = false; break; }
                    }
                    if (eq) { sid = cj; break; }
                }
            }
            const int64_t g = cbase + ci;
            a.kind[g] = (uint8_t)v.kind; a.e_start[g] = v.s0; a.e_stop[g] = v.s1;
            a.pair[g] = pr < 0 ? -1 : (int32_t)(pbase + pr); a.pairc[g] = prc < 0 ? -1 : (int32_t)(pbase + prc);
            a.seq_len[g] = (int32_t)ls; a.seq_id[g] = sid; a.chr[g] = v.chr;
        }
        run += __shfl(incl, 63, 64);
    }
}

__global__ void k_widen_i32(const int32_t *in, uint64_t *out, int32_t n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] > 0 ? (uint64_t)in[i] : 0;
}
__global__ void k_close_scan(const uint64_t *in, uint64_t *out, int64_t n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[n] = n > 0 ? out[n - 1] + in[n - 1] : 0;
}

template <int CAP> constexpr int64_t plan_state_ints() { return 16 + 6 * (int64_t)CAP + 6 * kMaxFlanks; } // header, full candidates, 4 x hit segments, pair prefixes, 5 x flanks
template <int CAP> constexpr size_t plan_lds_bytes() { return (size_t)CAP * 16 + (size_t)CAP * 4 * 6 + (size_t)(2 * (kMaxGenes + 1) + 5 * kMaxFlanks) * 4 + 64; }

} // namespace

struct ultra_dindex {
    int device = 0;
    int64_t pool_len = 0;
    int64_t n_seg = 0, n_flank = 0, n_part = 0, n_gene = 0;
    uint32_t ht_mask = 0;
    uint32_t pool_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // bytes present in the pool
    uint64_t serial = 0;                                 // identity of this upload (contexts cache the encoded pool by it)
    DevBuf pool, ht, part_off, part_seg, seg, gene_flat, gene_small_off, gene_small, flank;
    std::string err;
};

namespace {
thread_local std::string g_dplan_err;
int dfail(int code, const std::string &m) { g_dplan_err = m; return code; }
uint64_t g_dindex_serial = 0;
#define DIX_CHECK(call)                                                                                              \
    do {                                                                                                             \
        hipError_t e__ = (call);                                                                                     \
        if (e__ != hipSuccess) { ultra_dindex_destroy(dx); return dfail(ULTRA_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } \
    } while (0)
} // namespace

extern "C" const char *ultra_dindex_last_error(void) { return g_dplan_err.c_str(); }

extern "C" void ultra_dindex_destroy(ultra_dindex *dx)
{
    if (!dx) return;
    (void)hipSetDevice(dx->device);
    for (DevBuf *b : {&dx->pool, &dx->ht, &dx->part_off, &dx->part_seg, &dx->seg, &dx->gene_flat, &dx->gene_small_off, &dx->gene_small, &dx->flank})
        b->release();
    delete dx;
}

extern "C" int ultra_dindex_create(int device, const ultra_index *ix, ultra_dindex **out)
{
    if (!ix || !out) return dfail(ULTRA_E_ARG, "ultra_dindex_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return dfail(ULTRA_E_NODEVICE, "ultra_dindex_create: no such device");
    if (hipSetDevice(device) != hipSuccess) return dfail(ULTRA_E_NODEVICE, "ultra_dindex_create: hipSetDevice failed");
    const int64_t LIM = 0x7fffffff;
    auto fits = [&](int64_t v) { return v >= 0 && v <= LIM; };
    // ---- segment ids in (start, stop, chromosome) order ----
    struct T3 { int64_t chr, s0, s1; };
    std::vector<T3> segs;
    segs.reserve(ix->seg_seq.size() + ix->small_flat.size() / 3);
    for (size_t i = 0; i + 2 < ix->seg_flat.size(); i += 3) segs.push_back({ix->seg_flat[i], ix->seg_flat[i + 1], ix->seg_flat[i + 2]});
    for (size_t i = 0; i + 2 < ix->small_flat.size(); i += 3) segs.push_back({ix->small_flat[i], ix->small_flat[i + 1], ix->small_flat[i + 2]});
    for (auto &kv : ix->seg_seq) segs.push_back({kv.first.a, kv.first.b, kv.first.c});
    for (auto &kv : ix->seg_genes) segs.push_back({kv.first.a, kv.first.b, kv.first.c});
    auto less3 = [](const T3 &x, const T3 &y) { return x.s0 != y.s0 ? x.s0 < y.s0 : x.s1 != y.s1 ? x.s1 < y.s1 : x.chr < y.chr; };
    std::sort(segs.begin(), segs.end(), less3);
    segs.erase(std::unique(segs.begin(), segs.end(), [](const T3 &x, const T3 &y) { return x.chr == y.chr && x.s0 == y.s0 && x.s1 == y.s1; }), segs.end());
    if ((int64_t)segs.size() > LIM) return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: too many segments");
    std::unordered_map<Key3, int32_t, Key3Hash> seg_id;
    seg_id.reserve(segs.size() * 2);
    std::vector<SegRec> h_seg(segs.size());
    for (size_t i = 0; i < segs.size(); i++) {
        const T3 &t = segs[i];
        if (!fits(t.chr) || !fits(t.s0) || !fits(t.s1)) return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: coordinates beyond 2^31 - 1");
        seg_id[Key3{t.chr, t.s0, t.s1}] = (int32_t)i;
        SegRec r; r.chr = (int32_t)t.chr; r.s0 = (int32_t)t.s0; r.s1 = (int32_t)t.s1; r.len = 0; r.off = -1; r.gene_b = r.gene_e = -1;
        auto sq = ix->seg_seq.find(Key3{t.chr, t.s0, t.s1});
        if (sq != ix->seg_seq.end()) {
            if (!fits(sq->second.len)) return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: sequence longer than 2^31 - 1");
            r.off = sq->second.off; r.len = (int32_t)sq->second.len;
        }
        auto gn = ix->seg_genes.find(Key3{t.chr, t.s0, t.s1});
        if (gn != ix->seg_genes.end()) {
            if (!fits(gn->second.first) || !fits(gn->second.second)) return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: gene list too long");
            r.gene_b = (int32_t)gn->second.first; r.gene_e = (int32_t)gn->second.second;
        }
        h_seg[i] = r;
    }
    // ---- parts and flanks behind one hash table ----
    std::vector<int32_t> h_part_off(1, 0), h_part_seg;
    std::vector<FlankRec> h_flank;
    size_t slots = 64;
    while (slots < 2 * (ix->part_segs.size() + ix->flank_seq.size()) + 2) slots <<= 1;
    std::vector<HtEntry> h_ht(slots, HtEntry{0, 0, 0, -1});
    auto ht_put = [&](int64_t chr, int64_t p0, int64_t p1, int32_t val) {
        uint32_t h = ht_hash((int32_t)chr, (int32_t)p0, (int32_t)p1) & (uint32_t)(slots - 1);
        while (h_ht[h].val != -1) {
            if (h_ht[h].chr == (int32_t)chr && h_ht[h].p0 == (int32_t)p0 && h_ht[h].p1 == (int32_t)p1) return false;
            h = (h + 1) & (uint32_t)(slots - 1);
        }
        h_ht[h] = HtEntry{(int32_t)chr, (int32_t)p0, (int32_t)p1, val};
        return true;
    };
    for (auto &kv : ix->part_segs) {
        if (!fits(kv.first.a) || !fits(kv.first.b) || !fits(kv.first.c)) return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: coordinates beyond 2^31 - 1");
        const int32_t pid = (int32_t)h_part_off.size() - 1;
        for (int64_t s = kv.second.first; s < kv.second.second; s++) {
            const int64_t t_chr = ix->seg_flat[3 * s], s0 = ix->seg_flat[3 * s + 1], s1 = ix->seg_flat[3 * s + 2];
            // the reference keys a hit segment's sequence by the NAM's chromosome and its gene by the segment's own
            // chromosome field; one id per segment is only the same thing when the two agree
            if (t_chr != kv.first.a)
                return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: a part lists a segment of another chromosome; use the host planner");
            h_part_seg.push_back(seg_id[Key3{t_chr, s0, s1}]);
        }
        h_part_off.push_back((int32_t)h_part_seg.size());
        ht_put(kv.first.a, kv.first.b, kv.first.c, pid);
    }
    for (auto &kv : ix->flank_seq) {
        if (!fits(kv.first.a) || !fits(kv.first.b) || !fits(kv.first.c) || !fits(kv.second.len))
            return dfail(ULTRA_E_UNSUPPORTED, "ultra_dindex_create: coordinates beyond 2^31 - 1");
        if (!ht_put(kv.first.a, kv.first.b, kv.first.c, kFlankBit | (int32_t)h_flank.size())) continue; // a part of the same name wins (:188)
        FlankRec r; r.chr = (int32_t)kv.first.a; r.s0 = (int32_t)kv.first.b; r.s1 = (int32_t)kv.first.c; r.len = (int32_t)kv.second.len; r.off = kv.second.off;
        h_flank.push_back(r);
    }
    std::vector<int32_t> h_gs_off(1, 0), h_gs;
    for (auto &rg : ix->gene_small) {
        for (int64_t v = rg.first; v < rg.second; v++)
            h_gs.push_back(seg_id[Key3{ix->small_flat[3 * v], ix->small_flat[3 * v + 1], ix->small_flat[3 * v + 2]}]);
        h_gs_off.push_back((int32_t)h_gs.size());
    }
    ultra_dindex *dx = new ultra_dindex();
    dx->device = device;
    dx->pool_len = (int64_t)ix->pool.size();
    dx->n_seg = (int64_t)h_seg.size(); dx->n_flank = (int64_t)h_flank.size(); dx->n_part = (int64_t)h_part_off.size() - 1; dx->n_gene = (int64_t)ix->gene_small.size();
    dx->ht_mask = (uint32_t)(slots - 1);
    dx->serial = ++g_dindex_serial;
    for (uint8_t b : ix->pool) dx->pool_mask[b >> 5] |= 1u << (b & 31);
    auto up = [&](DevBuf &buf, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = buf.ensure(bytes + 64);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice);
    };
    DIX_CHECK(up(dx->pool, ix->pool.data(), ix->pool.size()));
    DIX_CHECK(up(dx->ht, h_ht.data(), h_ht.size() * sizeof(HtEntry)));
    DIX_CHECK(up(dx->part_off, h_part_off.data(), h_part_off.size() * 4));
    DIX_CHECK(up(dx->part_seg, h_part_seg.data(), h_part_seg.size() * 4));
    DIX_CHECK(up(dx->seg, h_seg.data(), h_seg.size() * sizeof(SegRec)));
    DIX_CHECK(up(dx->gene_flat, ix->gene_flat.data(), ix->gene_flat.size() * 4));
    DIX_CHECK(up(dx->gene_small_off, h_gs_off.data(), h_gs_off.size() * 4));
    DIX_CHECK(up(dx->gene_small, h_gs.data(), h_gs.size() * 4));
    DIX_CHECK(up(dx->flank, h_flank.data(), h_flank.size() * sizeof(FlankRec)));
    *out = dx;
    return ULTRA_OK;
}

// The request of a batch (NAM chains + reads) goes up into the context's buffers and stays there: ultra_mam_plan_resident
// plans from it any number of times.
extern "C" int ultra_mam_upload_nams(ultra_ctx *ctx, const ultra_plan_request *rq)
{
    if (!ctx || !rq) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    MyersState &st = ctx->my;
    m.staged = false; m.ran = false; st.staged = false; st.ran = false;
    m.up_valid = false;
    const int32_t ni = rq->n_inst;
    if (ni < 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: negative instance count");
    if (ni > 0 && (!rq->inst_nam_off || !rq->read_off || !rq->reads)) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: null array");
    int64_t n_nam = 0, read_bytes = 0;
    if (ni > 0) {
        if (rq->inst_nam_off[0] != 0 || rq->read_off[0] != 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: offsets must start at 0");
        for (int32_t i = 0; i < ni; i++) {
            if (rq->inst_nam_off[i + 1] < rq->inst_nam_off[i]) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: inst_nam_off must be non-decreasing");
            const int64_t lr = rq->read_off[i + 1] - rq->read_off[i];
            if (lr < 1 || lr > 0x7fffffff) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: read length < 1");
        }
        n_nam = rq->inst_nam_off[ni]; read_bytes = rq->read_off[ni];
        if (n_nam > 0 && (!rq->nam_part || !rq->nam_x || !rq->nam_y)) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: null NAM array");
    }
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    ultra_drain_guard drain(ctx);
    const size_t NI = (size_t)std::max(ni, 1), NN = (size_t)std::max<int64_t>(n_nam, 1);
    ULTRA_HIP_CHECK(ctx, st.raw.ensure((size_t)read_bytes + 16));
    ULTRA_HIP_CHECK(ctx, m.d_namoff.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.d_readoff.ensure((NI + 1) * 8));
    ULTRA_HIP_CHECK(ctx, m.d_nampart.ensure(NN * 24)); ULTRA_HIP_CHECK(ctx, m.d_namx.ensure(NN * 8)); ULTRA_HIP_CHECK(ctx, m.d_namy.ensure(NN * 8));
    if (ni > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.raw.p, rq->reads, (size_t)read_bytes, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.d_namoff.p, rq->inst_nam_off, (NI + 1) * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.d_readoff.p, rq->read_off, (NI + 1) * 8, hipMemcpyHostToDevice, s));
    }
    if (n_nam > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.d_nampart.p, rq->nam_part, (size_t)n_nam * 24, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.d_namx.p, rq->nam_x, (size_t)n_nam * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.d_namy.p, rq->nam_y, (size_t)n_nam * 8, hipMemcpyHostToDevice, s));
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s)); // the caller's arrays may go away
    drain.dismiss();
    m.up_ni = ni; m.up_nnam = n_nam; m.up_read_bytes = read_bytes; m.up_valid = true;
    return ULTRA_OK;
}

// Plans and stages the uploaded batch on the device: afterwards the context is in the state ultra_mam_stage leaves it
// in.  Everything it reads lies in HBM (the request, the resident index); what comes back to the host are counts.
extern "C" int ultra_mam_plan_resident(ultra_ctx *ctx, const ultra_dindex *dx)
{
    if (!ctx || !dx) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    MyersState &st = ctx->my;
    m.staged = false; m.ran = false; st.staged = false; st.ran = false;
    if (!m.up_valid) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_plan_resident: no uploaded batch (ultra_mam_upload_nams)");
    if (dx->device != ctx->device) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: index and context are on different devices");
    const int32_t ni = m.up_ni;
    const int64_t n_nam = m.up_nnam, read_bytes = m.up_read_bytes;
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    ultra_drain_guard drain(ctx);
    // ---- arena = [pad | pool | reads (capacity) | pad]: the pool part is encoded once per (context, index) ----
    const int64_t P = dx->pool_len;
    bool fresh = st.pool_serial != dx->serial;
    if (fresh || read_bytes > st.reads_cap) {
        const int64_t cap = std::max<int64_t>(read_bytes + read_bytes / 4, 1 << 20);
        st.reads_cap = cap;
        st.N = (int64_t)kArenaPad + P + cap + (int64_t)kArenaPad;
        ULTRA_HIP_CHECK(ctx, st.seq.ensure((size_t)st.N + 16)); ULTRA_HIP_CHECK(ctx, st.rseq.ensure((size_t)st.N + 16));
        ULTRA_HIP_CHECK(ctx, st.lut.ensure(256 + 32));
        ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.seq.p, 0, (size_t)st.N + 16, s));
        ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.rseq.p, 0, (size_t)st.N + 16, s));
        if (fresh) {
            // codes are handed out in order of first appearance and never change: the pool's bytes first
            for (auto &v : st.h_lut) v = 0;
            for (auto &v : st.h_seen) v = false;
            st.S = 0;
            for (int b = 0; b < 256; b++)
                if (dx->pool_mask[b >> 5] >> (b & 31) & 1u) { st.h_seen[b] = true; st.h_lut[b] = (uint8_t)(st.S < 255 ? st.S : 255); st.S++; }
        }
        st.pool_serial = 0; // the pool part below is (re)written
        if (st.S > kMaxSymbols) return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_mam_stage_nams: more than 36 distinct symbols in the index");
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.lut.p, st.h_lut, 256, hipMemcpyHostToDevice, s));
        ultra_launch_encode(s, dx->pool.as<uint8_t>(), P, st.lut.as<uint8_t>(), st.seq.as<uint8_t>(), st.rseq.as<uint8_t>(), (int64_t)kArenaPad, st.N);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
        st.pool_serial = dx->serial;
    }
    if (ni == 0) {
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        drain.dismiss();
        m.n_cand = 0; m.n_inst = 0; m.max_cand = 0;
        for (auto &e : m.ev) if (!e) ULTRA_HIP_CHECK(ctx, hipEventCreate(&e));
        const int rc0 = ultra_myers_stage_pairs(ctx, 0);
        if (rc0 != ULTRA_OK) return rc0;
        m.staged = true;
        return ULTRA_OK;
    }
    // ---- new symbols of the reads get the next codes ----
    const size_t NI = (size_t)ni;
    (void)n_nam;
    uint32_t *d_mask = (uint32_t *)(st.lut.as<uint8_t>() + 256);
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(d_mask, 0, 32, s));
    ultra_launch_presence(s, st.raw.as<uint8_t>(), read_bytes, d_mask);
    // ---- count pass (runs while the presence mask travels back) ----
    ULTRA_HIP_CHECK(ctx, m.p_ncand.ensure(NI * 4)); ULTRA_HIP_CHECK(ctx, m.p_npair.ensure(NI * 4));
    ULTRA_HIP_CHECK(ctx, m.p_over.ensure(2 * NI * 4 + 64)); // overflow list of the small capacity, then of the middle one
    ULTRA_HIP_CHECK(ctx, m.p_misc.ensure(64));
    ULTRA_HIP_CHECK(ctx, m.p_wide.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.p_poff.ensure((NI + 1) * 8));
    ULTRA_HIP_CHECK(ctx, m.candoff.ensure((NI + 1) * 8));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(m.p_misc.p, 0, 64, s));
    PlanArgs a;
    memset(&a, 0, sizeof a);
    a.ht = dx->ht.as<HtEntry>(); a.ht_mask = dx->ht_mask; a.part_off = dx->part_off.as<int32_t>(); a.part_seg = dx->part_seg.as<int32_t>();
    a.seg = dx->seg.as<SegRec>(); a.gene_flat = dx->gene_flat.as<int32_t>(); a.gene_small_off = dx->gene_small_off.as<int32_t>();
    a.gene_small = dx->gene_small.as<int32_t>(); a.flank = dx->flank.as<FlankRec>(); a.pool = dx->pool.as<uint8_t>(); a.pool_len = P;
    a.n_inst = ni;
    a.nam_off = m.d_namoff.as<int64_t>(); a.nam_part = m.d_nampart.as<int64_t>(); a.nam_x = m.d_namx.as<int64_t>(); a.nam_y = m.d_namy.as<int64_t>();
    a.read_off = m.d_readoff.as<int64_t>();
    a.n_cand = m.p_ncand.as<int32_t>(); a.n_pair = m.p_npair.as<int32_t>();
    a.overflow_list = m.p_over.as<int32_t>();
    int32_t *misc = m.p_misc.as<int32_t>(); // [0] overflow count, [1..2] error code / instance, [3] largest candidate count
    a.overflow_cnt = misc; a.err = misc + 1; a.max_cand = misc + 3;
    ULTRA_HIP_CHECK(ctx, m.p_state.ensure(NI * (size_t)plan_state_ints<128>() * 4));
    a.state = m.p_state.as<int32_t>(); a.state_stride = plan_state_ints<128>();
    constexpr int CAP_S = 128, CAP_M = 512, CAP_L = 2048; // 128 entries hold 99 % of the instances and let a CU keep twice the wavefronts in flight (the kernel is a chain of dependent loads)
    hipLaunchKernelGGL((k_plan<CAP_S, false>), dim3((unsigned)ni), dim3(64), plan_lds_bytes<CAP_S>(), s, a);
    ULTRA_HIP_CHECK(ctx, hipGetLastError());
    uint32_t h_mask[8];
    int32_t h_misc[4];
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_mask, d_mask, 32, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_misc, misc, 16, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    bool lut_changed = false;
    for (int b = 0; b < 256; b++)
        if ((h_mask[b >> 5] >> (b & 31) & 1u) && !st.h_seen[b]) { st.h_seen[b] = true; st.h_lut[b] = (uint8_t)(st.S < 255 ? st.S : 255); st.S++; lut_changed = true; }
    if (st.S > kMaxSymbols) { st.pool_serial = 0; return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_mam_stage_nams: more than 36 distinct symbols in index + reads"); }
    if (st.S == 0) st.S = 1;
    if (lut_changed) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.lut.p, st.h_lut, 256, hipMemcpyHostToDevice, s));
    ultra_launch_encode(s, st.raw.as<uint8_t>(), read_bytes, st.lut.as<uint8_t>(), st.seq.as<uint8_t>(), st.rseq.as<uint8_t>(), (int64_t)kArenaPad + P, st.N);
    ULTRA_HIP_CHECK(ctx, hipGetLastError());
    // The instances that overflow the small capacity (1 % of a cDNA batch) run again at a middle one - seven wavefronts per
    // CU instead of the one the large capacity's 83 KB of LDS leave room for (0.36 -> 0.1 ms per 262144 reads) - and what
    // overflows that, at the large one.  The third list's length stays on the device: its launches take the second list's
    // length as their grid and the blocks beyond the list end at once.
    const int32_t n_over = h_misc[0];
    const void *big_count = (const void *)k_plan<CAP_L, false>, *big_fill = (const void *)k_plan<CAP_L, true>;
    PlanArgs mid = a, big = a;
    if (n_over > 0) {
        ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute(big_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_lds_bytes<CAP_L>()));
        ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute(big_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_lds_bytes<CAP_L>()));
        ULTRA_HIP_CHECK(ctx, m.p_state_mid.ensure((size_t)n_over * (size_t)plan_state_ints<CAP_M>() * 4));
        ULTRA_HIP_CHECK(ctx, m.p_state_big.ensure((size_t)n_over * (size_t)plan_state_ints<CAP_L>() * 4));
        mid.list = a.overflow_list; mid.n_list = n_over;
        mid.state = m.p_state_mid.as<int32_t>(); mid.state_stride = plan_state_ints<CAP_M>();
        mid.overflow_list = a.overflow_list + NI; mid.overflow_cnt = misc + 4;
        hipLaunchKernelGGL((k_plan<CAP_M, false>), dim3((unsigned)n_over), dim3(64), plan_lds_bytes<CAP_M>(), s, mid);
        big.list = mid.overflow_list; big.n_list = n_over; big.n_list_dev = misc + 4;
        big.state = m.p_state_big.as<int32_t>(); big.state_stride = plan_state_ints<CAP_L>();
        hipLaunchKernelGGL((k_plan<CAP_L, false>), dim3((unsigned)n_over), dim3(64), plan_lds_bytes<CAP_L>(), s, big);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    // ---- CSR offsets ----
    const unsigned gi = (unsigned)((ni + 255) / 256);
    hipLaunchKernelGGL(k_widen_i32, dim3(gi), dim3(256), 0, s, a.n_cand, m.p_wide.as<uint64_t>(), ni);
    ULTRA_HIP_CHECK(ctx, ultra_scan_u64(ctx, m.p_wide.as<uint64_t>(), m.candoff.as<uint64_t>(), ni, m.scantmp));
    hipLaunchKernelGGL(k_close_scan, dim3(1), dim3(64), 0, s, m.p_wide.as<uint64_t>(), m.candoff.as<uint64_t>(), (int64_t)ni);
    int64_t h_tot[2] = {0, 0};
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_tot[0], m.candoff.as<int64_t>() + ni, 8, hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_widen_i32, dim3(gi), dim3(256), 0, s, a.n_pair, m.p_wide.as<uint64_t>(), ni);
    ULTRA_HIP_CHECK(ctx, ultra_scan_u64(ctx, m.p_wide.as<uint64_t>(), m.p_poff.as<uint64_t>(), ni, m.scantmp));
    hipLaunchKernelGGL(k_close_scan, dim3(1), dim3(64), 0, s, m.p_wide.as<uint64_t>(), m.p_poff.as<uint64_t>(), (int64_t)ni);
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_tot[1], m.p_poff.as<int64_t>() + ni, 8, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_misc, misc, 16, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    auto plan_failed = [&](const int32_t *mi) -> int {
        static const char *what[] = {"", "a NAM's exon_part_id is neither a part nor a flank of the index (KeyError in ref_flank_sequences)",
                                     "hit segment without a gene (KeyError in segment_to_gene, classify_read_with_mams.py:264)",
                                     "candidate without a sequence in the index (KeyError in ref_segment_sequences / ref_flank_sequences)",
                                     "coordinate beyond 2^31 - 1", "more than 2048 hits / entries, 64 genes or 16 flanks in one instance: use the host planner"};
        const int code = mi[1] >= 1 && mi[1] <= 5 ? mi[1] : 1;
        return ultra_fail(ctx, code == PERR_TOO_BIG || code == PERR_RANGE ? ULTRA_E_UNSUPPORTED : ULTRA_E_ARG,
                          "ultra_mam_stage_nams: instance " + std::to_string(mi[2]) + ": " + what[code]);
    };
    if (h_misc[1] != 0) return plan_failed(h_misc);
    const int64_t nc = h_tot[0], npairs = h_tot[1];
    if (nc > 0x7fffffff || npairs > 0x7fffffff) return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_mam_stage_nams: more than 2^31 - 1 candidates or pairs in one batch");
    // ---- fill pass ----
    const size_t NC = (size_t)std::max<int64_t>(nc, 1), NP = (size_t)std::max<int64_t>(npairs, 1);
    ULTRA_HIP_CHECK(ctx, m.kind.ensure(NC)); ULTRA_HIP_CHECK(ctx, m.estart.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.estop.ensure(NC * 4));
    ULTRA_HIP_CHECK(ctx, m.pair.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.pairc.ensure(NC * 4));
    ULTRA_HIP_CHECK(ctx, m.seqlen.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.seqid.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.chr.ensure(NC * 4));
    ULTRA_HIP_CHECK(ctx, m.cap.ensure((NC + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.capoff.ensure((NC + 1) * 8));
    ULTRA_HIP_CHECK(ctx, m.readlen.ensure(NI * 4)); ULTRA_HIP_CHECK(ctx, m.fps.ensure(NI * 8)); ULTRA_HIP_CHECK(ctx, m.lps.ensure(NI * 8));
    ULTRA_HIP_CHECK(ctx, m.instbase.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.nmam.ensure(NI * 4));
    ULTRA_HIP_CHECK(ctx, m.chainlen.ensure(NI * 4 + 16)); ULTRA_HIP_CHECK(ctx, m.value.ensure(NI * 8));
    ULTRA_HIP_CHECK(ctx, m.lenw.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.chainoff.ensure((NI + 1) * 8));
    ULTRA_HIP_CHECK(ctx, st.u_rec.ensure(NP * sizeof(PairRec)));
    a.cand_off = m.candoff.as<int64_t>(); a.pair_off = m.p_poff.as<int64_t>();
    a.kind = m.kind.as<uint8_t>(); a.e_start = m.estart.as<int32_t>(); a.e_stop = m.estop.as<int32_t>(); a.pair = m.pair.as<int32_t>();
    a.pairc = m.pairc.as<int32_t>(); a.seq_len = m.seqlen.as<int32_t>(); a.seq_id = m.seqid.as<int32_t>(); a.chr = m.chr.as<int32_t>();
    a.rec = st.u_rec.as<PairRec>();
    a.read_len = m.readlen.as<int32_t>(); a.fps = m.fps.as<int64_t>(); a.lps = m.lps.as<int64_t>();
    hipLaunchKernelGGL((k_plan<CAP_S, true>), dim3((unsigned)ni), dim3(64), plan_lds_bytes<CAP_S>(), s, a);
    if (n_over > 0) {
        // the fill launches see the arrays the count launches did not have yet
        PlanArgs fm = a, fb = a;
        fm.list = mid.list; fm.n_list = mid.n_list; fm.state = mid.state; fm.state_stride = mid.state_stride;
        fb.list = big.list; fb.n_list = big.n_list; fb.n_list_dev = big.n_list_dev; fb.state = big.state; fb.state_stride = big.state_stride;
        hipLaunchKernelGGL((k_plan<CAP_M, true>), dim3((unsigned)n_over), dim3(64), plan_lds_bytes<CAP_M>(), s, fm);
        hipLaunchKernelGGL((k_plan<CAP_L, true>), dim3((unsigned)n_over), dim3(64), plan_lds_bytes<CAP_L>(), s, fb);
    }
    ULTRA_HIP_CHECK(ctx, hipGetLastError());
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_misc, misc, 16, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    if (h_misc[1] != 0) return plan_failed(h_misc);
    drain.dismiss();
    const int32_t max_c = h_misc[3];
    if ((size_t)max_c * 28 > (size_t)160 * 1024 - 64)
        return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_mam_stage_nams: more than 5849 candidates in one instance");
    for (auto &e : m.ev) if (!e) ULTRA_HIP_CHECK(ctx, hipEventCreate(&e));
    st.u_packed = true;
    int rc = ultra_myers_stage_pairs(ctx, (int32_t)npairs);
    if (rc != ULTRA_OK) return rc;
    if (nc > 0) {
        ultra_mam_remap_pairs(ctx, (int32_t)nc);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    }
    m.n_cand = (int32_t)nc; m.n_inst = ni; m.max_cand = max_c; m.has_chr = true;
    m.staged = true;
    return ULTRA_OK;
}

// The same, with the planner's kernels ordered behind the start pass of the run that `front` (another context of the same
// GPU) has in flight.  The forward and start passes fill the chip: a planner launched beside them crawls and slows them
// down; the rest of a run (traceback, MAM build, chaining) is bound by latency and leaves the ALUs to a neighbour.  If
// `front` is inside ultra_mam_run_begin on another thread, the call first waits until that run's start pass has been
// enqueued; with no run in flight on `front` it is ultra_mam_plan_resident.
extern "C" int ultra_mam_plan_resident_behind(ultra_ctx *ctx, const ultra_dindex *dx, ultra_ctx *front)
{
    if (!ctx || !dx) return ULTRA_E_ARG;
    if (front && front != ctx && front->device == ctx->device) {
        std::unique_lock<std::mutex> lk(front->tail_mu);
        front->tail_cv.wait(lk, [&] { return !front->run_active || front->tail_open; });
        if (front->run_active && front->tail_has_event) {
            ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
            ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, front->my.ev[2], 0));
        }
    }
    return ultra_mam_plan_resident(ctx, dx);
}

extern "C" int ultra_mam_stage_nams(ultra_ctx *ctx, const ultra_dindex *dx, const ultra_plan_request *rq)
{
    if (!ctx || !dx || !rq) return ULTRA_E_ARG;
    if (dx->device != ctx->device) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage_nams: index and context are on different devices");
    const int rc = ultra_mam_upload_nams(ctx, rq);
    if (rc != ULTRA_OK) return rc;
    return ultra_mam_plan_resident(ctx, dx);
}
