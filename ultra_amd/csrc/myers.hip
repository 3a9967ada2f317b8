// Myers / Hyyro bit-vector semi-global search for gfx950 (wave64), hand-written for CDNA4.
//
// Replaces the arithmetic behind edlib.align(query, target, task="path", mode="HW", k) as the
// reference calls it at /root/reference/modules/classify_read_with_mams.py:99 (:312, :353, :377).
//
// Layout choices (not a translation of edlib):
//   * the query is padded at the TOP of its first 64-row word, so the last query row is always
//     bit 63 of the last word and the bottom-row score is the running sum of the last word's
//     horizontal carry - no per-column bit extraction, no "last W columns" epilogue;
//   * lane-per-pair kernels (W = 1..8 words in registers) for queries up to 512 bases: 64 independent
//     pairs per wavefront, zero cross-lane traffic, Peq[sym][word][lane] in LDS (conflict-free
//     ds_read_b64: the bank depends on the lane only);
//   * lane-group kernels for longer queries: G lanes per pair, W words per lane, anti-diagonal
//     (skewed) schedule, the horizontal carry moves one lane per step;
//   * three passes: forward HW (distance + end columns), reverse anchored pass per end column
//     (start = smallest start), NW pass that stores every column followed by a traceback of the
//     first location (insertion count / edit script).
#include "common.h"

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

namespace {

enum { MODE_FWD = 0, MODE_FWD_EXACT = 1, MODE_REV = 2, MODE_NW = 3 };
#ifndef ULTRA_BAND32W
#define ULTRA_BAND32W 1
#endif
#if ULTRA_BAND32W
enum { kBandClasses = 9, kBandFallback = 9 }; // banded traceback: 32 rows, then WB = 1, 2, 4, 6, 8, 12, 16, 32 words; class 9 = not banded
#else
enum { kBandClasses = 8, kBandFallback = 8 }; // banded traceback: 32 rows, then WB = 1, 2, 4, 6, 8, 12, 16 words; class 8 = not banded
#endif
// A lane-per-hit band kernel is bound by its longest lane (words x columns, twice): up to 2 words a hit keeps a lane
// to itself (myers_band, and myers_band32 for the 86 % of hits with distance <= 22: half the arithmetic, checkpoint
// bytes and LDS of the 1-word band); from 4 words on the window is split over 2 - 8 neighbouring lanes (myers_bandg),
// which is also what lets the band reach 16 words (distance <= 1014) before the full-height kernels take over (a
// 24-word class over 8 lanes measured slower than those for the few hits beyond).
__host__ __device__ constexpr int band_words(int c) { return c < 3 ? c : c == 3 ? 4 : c == 4 ? 6 : c == 5 ? 8 : c == 6 ? 12 : c == 7 ? 16 : 32; } // classes 1..8
#ifndef ULTRA_BANDG_GEOM
#define ULTRA_BANDG_GEOM 2
#endif
#if ULTRA_BANDG_GEOM == 0
__host__ __device__ constexpr int band_group(int c) { return c < 3 ? 1 : c == 3 ? 2 : c == 4 ? 3 : c == 5 ? 4 : c == 6 ? 4 : c == 7 ? 8 : 16; }    // lanes per hit
#elif ULTRA_BANDG_GEOM == 1
// one word per lane up to 8 words, two beyond: the wide bands are few, long wavefronts whose length is the length of the
// traceback phase, and a column costs a lane about 30 instructions + 40 per word
__host__ __device__ constexpr int band_group(int c) { return c < 3 ? 1 : c == 3 ? 4 : c == 4 ? 6 : c == 5 ? 8 : c == 6 ? 6 : c == 7 ? 8 : 16; }    // lanes per hit
#else
// one word per lane throughout (12 lanes: five hits per wavefront; 16: four, a DPP row each; 32: two)
__host__ __device__ constexpr int band_group(int c) { return c < 3 ? 1 : c == 3 ? 4 : c == 4 ? 6 : c == 5 ? 8 : c == 6 ? 12 : c == 7 ? 16 : 32; }    // lanes per hit
#endif
__host__ __device__ constexpr int band_rows(int c) { return c == 0 ? 32 : 64 * band_words(c); }
#ifndef ULTRA_BAND_K0
#define ULTRA_BAND_K0 32
#endif
#ifndef ULTRA_BAND_K1
#define ULTRA_BAND_K1 16
#endif
__host__ __device__ constexpr int band_chunk(int c) { return c == 0 ? ULTRA_BAND_K0 : c == 1 ? ULTRA_BAND_K1 : 8; } // columns per checkpoint
// Hits whose insertion count follows from arithmetic and need no traceback.  A global alignment of m query bases with L
// target bases at distance ed has I - D = m - L and X + I + D = ed, so X = ed - |m - L| - 2 j for some j >= 0: when
// ed - |m - L| <= 1 there is no choice (j = 0) and I = max(m - L, 0) whatever path the tie rule takes.  A third of the
// cDNA hits (all with ed <= 1, 78 % of ed = 2, 12 % of ed 6 .. 22), 24 % of the 32-row band class's columns.
// k_band_keys writes their count and sorts them behind the last real hit of their class; the band kernels skip them.
#ifndef ULTRA_WALK_BY_COLUMN
#define ULTRA_WALK_BY_COLUMN 1
#endif
#ifndef ULTRA_TB_ARITH
#define ULTRA_TB_ARITH 1
#endif
__device__ __forceinline__ bool tb_by_arithmetic(int ed, int m, int L)
{
    const int d = m - L;
    return ULTRA_TB_ARITH && ed - (d < 0 ? -d : d) <= 1;
}

struct MyArgs {
    const uint8_t *seq, *rseq;
    int64_t N;
    int S;
    const int64_t *q_off; const int32_t *q_len; const int64_t *t_off; const int32_t *t_len; const int32_t *k;
    int32_t *ed; int32_t *nloc; int32_t *slots; int slot_cap;
    const int64_t *loc_off; int32_t *loc_end; int32_t *loc_start; const int32_t *loc_pair;
    const int32_t *hit_pairs; int32_t *first_ins;
    const int32_t *nw_items; // full-height traceback: dense list of pairs to do, or nullptr = the hit list itself
    ulonglong2 *tr; int64_t region; // trace scratch {Pv, Ph} per word and column; entries per wave (lane kernels) / per pair slot (group kernels)
    int32_t Lmax;
    uint8_t *ops; const int64_t *ops_off; int32_t *n_ops;
    int64_t begin, end;
    int32_t *wflag;  // forward launch: dense list of the work items in which some pair has more end columns than first-pass slots
    int32_t *wcount; // its length (one counter per register bucket)
    int32_t bstride, boff; // myers_mega: workgroup x of this launch takes work item x * bstride + boff (a bucket's list dealt over two launches)
    // start pass: the work items of a launch follow from the location range of each size class, so they are
    // computed from blockIdx (heaviest class first) instead of being listed by the host
    struct { int32_t cls, first_block; int64_t b, e; } rr[5];
    int32_t n_rr;
    // long class (queries beyond 32768 rows, strips of 32768): two byte-per-column buffers per item for the horizontal
    // deltas that leave one strip and enter the next; item `base` of a launch uses buffer pair base - hbase
    uint8_t *hbuf; int64_t hstride, hbase;
    int32_t nbb;    // lane-group traceback: 64-row blocks stored per column (the ones the path's band crosses); 0 = all
    const int32_t *rev_items; // start pass over a list of locations (what the banded start kernels left), or nullptr = the range itself
};

struct Item {
    const uint8_t *q, *t;
    int32_t m, n, k, pair, best;
    int32_t loc; // start pass: the location this item stands for
    bool valid;
};

template <int MODE> __device__ __forceinline__ Item setup_item(const MyArgs &a, int64_t item)
{
    Item it;
    it.valid = item < a.end;
    it.q = a.seq; it.t = a.seq; it.m = 1; it.n = 0; it.k = 0; it.pair = 0; it.best = 0; it.loc = 0;
    if (!it.valid) return it;
    if (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) {
        int32_t p = (int32_t)item;
        it.pair = p;
        it.m = a.q_len[p]; it.n = a.t_len[p];
        it.q = a.seq + a.q_off[p]; it.t = a.seq + a.t_off[p];
        int32_t k = a.k[p];
        it.k = (k < 0 || k > it.m) ? it.m : k;
        if (MODE == MODE_FWD_EXACT) it.valid = a.nloc[p] > a.slot_cap;
    } else if (MODE == MODE_REV) {
        const int32_t loc = a.rev_items ? a.rev_items[item] : (int32_t)item;
        int32_t p = a.loc_pair[loc];
        int32_t e = a.loc_end[loc];
        it.loc = loc;
        it.pair = p;
        it.m = a.q_len[p];
        it.best = a.ed[p];
        int32_t lim = it.m + it.best;
        if (lim > e + 1) lim = e + 1;
        it.n = lim;
        it.k = it.best;
        it.q = a.rseq + (a.N - a.q_off[p] - it.m);
        it.t = a.rseq + (a.N - a.t_off[p] - (int64_t)(e + 1));
    } else {
        int32_t p = a.nw_items ? a.nw_items[item] : a.hit_pairs[item];
        int64_t lo = a.loc_off[p];
        int32_t s = a.loc_start[lo], e = a.loc_end[lo];
        it.pair = p;
        it.m = a.q_len[p];
        it.best = a.ed[p];
        it.n = e - s + 1;
        it.k = it.best;
        it.q = a.seq + a.q_off[p];
        it.t = a.seq + a.t_off[p] + s;
    }
    return it;
}

__device__ __forceinline__ uint64_t load8(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8); // arena is padded: reading a few bytes past a slice is safe
    return v;
}

// One 64-row word, one column (Hyyro 2003 formulation of Myers 1999).  hp/hm = horizontal
// carry into the word's top row (+1 / -1), updated to the carry out of its bottom row.
// Written on 32-bit halves: gfx950 has a three-input bitwise op (v_bitop3_b32, truth table in the
// immediate: a = 0xF0, b = 0xCC, c = 0xAA), v_lshl_or_b32 and v_alignbit_b32, so a word costs
// 23 VALU instructions instead of the ~37 the 64-bit formulation compiles to.
template <int TT> __device__ __forceinline__ uint32_t bop3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
template <int TT> __device__ __forceinline__ uint64_t bop3_64(uint64_t a, uint64_t b, uint64_t c)
{
    return pack64(bop3<TT>((uint32_t)a, (uint32_t)b, (uint32_t)c), bop3<TT>((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)));
}
// (x >> 1) | (next << 63): two v_alignbit_b32
__device__ __forceinline__ uint64_t shr1_in(uint64_t x, uint32_t next_lo)
{
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    return pack64(__builtin_amdgcn_alignbit(xh, xl, 1), __builtin_amdgcn_alignbit(next_lo, xh, 1));
}
__device__ __forceinline__ void word_step(uint64_t &Pv, uint64_t &Mv, uint64_t Eq, uint32_t &hp, uint32_t &hm, uint64_t &Ph0)
{
    const uint32_t pvl = (uint32_t)Pv, pvh = (uint32_t)(Pv >> 32), mvl = (uint32_t)Mv, mvh = (uint32_t)(Mv >> 32);
    const uint32_t eqh = (uint32_t)(Eq >> 32);
    uint32_t eql = (uint32_t)Eq;
    const uint32_t xvl = eql | mvl, xvh = eqh | mvh;             // Xv = Eq | Mv
    eql |= hm;                                                   // Eq |= (hin < 0)
    const uint64_t sum = pack64(eql & pvl, eqh & pvh) + Pv;      // (Eq & Pv) + Pv: the one 64-bit carry chain
    const uint32_t xhl = bop3<0xBE>((uint32_t)sum, pvl, eql);    // Xh = (sum ^ Pv) | Eq
    const uint32_t xhh = bop3<0xBE>((uint32_t)(sum >> 32), pvh, eqh);
    const uint32_t phl = bop3<0xF1>(mvl, xhl, pvl);              // Ph = Mv | ~(Xh | Pv)
    const uint32_t phh = bop3<0xF1>(mvh, xhh, pvh);
    const uint32_t mhl = pvl & xhl, mhh = pvh & xhh;             // Mh = Pv & Xh
    Ph0 = pack64(phl, phh); // bit i: D[i][j] - D[i][j-1] == +1
    const uint32_t psl = (phl << 1) | hp;                        // Ph = (Ph << 1) | (hin > 0)
    const uint32_t psh = __builtin_amdgcn_alignbit(phh, phl, 31);
    const uint32_t msl = (mhl << 1) | hm;                        // Mh = (Mh << 1) | (hin < 0)
    const uint32_t msh = __builtin_amdgcn_alignbit(mhh, mhl, 31);
    hp = phh >> 31; hm = mhh >> 31;
    Pv = pack64(bop3<0xF1>(msl, xvl, psl), bop3<0xF1>(msh, xvh, psh)); // Pv' = Mh | ~(Xv | Ph)
    Mv = pack64(psl & xvl, psh & xvh);                                 // Mv' = Ph & Xv
}

// The same step for a FIRST word of the forward pass whose lower 32 rows are all top padding: those rows match
// everything, carry no vertical delta and, in the HW forward pass, no horizontal one (the top boundary is 0), so they
// stay as they are and only the upper half is computed: 12 instructions instead of 23.
__device__ __forceinline__ void word_step_hi(uint64_t &Pv, uint64_t &Mv, uint64_t Eq, uint32_t &hp, uint32_t &hm)
{
    const uint32_t pvh = (uint32_t)(Pv >> 32), mvh = (uint32_t)(Mv >> 32), eqh = (uint32_t)(Eq >> 32);
    const uint32_t xvh = eqh | mvh;
    const uint32_t sum = (eqh & pvh) + pvh;
    const uint32_t xhh = bop3<0xBE>(sum, pvh, eqh);
    const uint32_t phh = bop3<0xF1>(mvh, xhh, pvh);
    const uint32_t mhh = pvh & xhh;
    const uint32_t psh = phh << 1, msh = mhh << 1;
    hp = phh >> 31; hm = mhh >> 31;
    Pv = pack64((uint32_t)Pv, bop3<0xF1>(msh, xvh, psh));
    Mv = pack64((uint32_t)Mv, psh & xvh);
}

__device__ __forceinline__ uint64_t pad_mask(int pad, int blk) // bits of word blk whose global row index < pad
{
    int r = pad - 64 * blk;
    if (r <= 0) return 0;
    if (r >= 64) return ~(uint64_t)0;
    return (((uint64_t)1) << r) - 1;
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// Traceback of the first location over the stored NW columns.  Per word and column the pass keeps
// Pv (bit i: D[i][c] - D[i-1][c] == +1, "up" is an optimal predecessor) and Ph (bit i:
// D[i][c] - D[i][c-1] == +1, "left" is one); neither bit set means the diagonal.  Tie rule of the
// walk from the bottom-right cell: up (I), then left (D), then diagonal - so one 16-byte load per
// step decides it.  stride = distance between consecutive words of one column, colstride = between
// columns.  Rows above `pad` are top padding: once the query is used up only deletions remain; once
// the target is used up only insertions remain.
constexpr int kStage = 8; // columns (x 2 words) staged in LDS per refill of the walk

// Columns may hold only nbb of the nwords blocks: those from first(c) = clamp((c + lo_pad) >> 6, 0, nwords - nbb) on
// (lo_pad = lowest reachable diagonal - 1 + pad); nbb == nwords is the full-height layout.
__device__ __forceinline__ void traceback(const ulonglong2 *tr, int64_t stride, int64_t colstride, int pad, int rows, int L,
                                          const uint8_t *q, const uint8_t *t, uint8_t *ops, int ops_cap, int &n_ins, int &n_ops,
                                          ulonglong2 *stage, int nbb, int lo_pad)
{
    auto at = [&](int ck, int w) {
        int f = (ck + lo_pad) >> 6;
        const int fmax = (rows >> 6) - nbb;
        f = f < 0 ? 0 : f > fmax ? fmax : f;
        int rel = w - f;
        rel = rel < 0 ? 0 : rel >= nbb ? nbb - 1 : rel; // blocks a staged column does not hold are never walked
        return (int64_t)ck * colstride + (int64_t)rel * stride;
    };
    // The walk goes through the columns backwards: kStage columns x 2 adjacent words around the
    // current row are staged per refill (all loads in flight together) instead of one dependent
    // HBM load per step.  stage: this lane's slots at stage[i * 64].
    const int nwords = rows >> 6;
    int ip = rows - 1, c = L - 1;
    int ins = 0, cnt = 0;
    int win_hi = -1, win_lo = 0x7fffffff, wlo = 0;
    while (ip >= pad && c >= 0) {
        const int w = ip >> 6;
        if (c < win_lo || w < wlo || w > wlo + 1) {
            wlo = (ip - 32) >> 6;
            wlo = wlo > nwords - 2 ? nwords - 2 : wlo;
            wlo = wlo < 0 ? 0 : wlo;
            const int w1 = wlo + 1 < nwords ? wlo + 1 : wlo;
            ulonglong2 buf[kStage * 2];
#pragma unroll
            for (int k = 0; k < kStage; k++) {
                const int ck = c - k > 0 ? c - k : 0;
                buf[2 * k] = tr[at(ck, wlo)];
                buf[2 * k + 1] = tr[at(ck, w1)];
            }
#pragma unroll
            for (int k = 0; k < kStage * 2; k++) stage[k * 64] = buf[k];
            win_hi = c; win_lo = c - kStage + 1;
        }
        const ulonglong2 e = stage[((win_hi - c) * 2 + (w - wlo)) * 64];
        const uint64_t bit = ((uint64_t)1) << (ip & 63);
        int code;
        if (e.x & bit) { code = 1; ins++; ip--; }
        else if (e.y & bit) { code = 2; c--; }
        else { code = (ops && q[ip - pad] != t[c]) ? 3 : 0; ip--; c--; }
        if (ops && cnt < ops_cap) ops[cnt] = (uint8_t)code;
        cnt++;
    }
    if (ip >= pad) { // left boundary reached: the remaining query bases are insertions
        const int r = ip - pad + 1;
        if (ops) for (int u = 0; u < r; u++) if (cnt + u < ops_cap) ops[cnt + u] = 1;
        ins += r; cnt += r;
    } else if (c >= 0) { // query used up: the remaining target bases are deletions
        if (ops) for (int u = 0; u <= c; u++) if (cnt + u < ops_cap) ops[cnt + u] = 2;
        cnt += c + 1;
    }
    n_ins = ins; n_ops = cnt;
}

// The same walk for the lane-group kernels, where G lanes share a pair and a single walker would fetch every tile
// alone (measured: 2.7 - 4.5 ms of walk behind 1.1 ms of matrix fill for a 1742 x 1965 pair).  All lanes of the
// wavefront call this: per round the walker's position goes to its group, up to 8 lanes of the group fetch a tile
// of 16 entries each (8 G columns x the 2 blocks around the current row; every load in flight at once) into the
// pair's LDS region, and the walker (lane g == 0) walks until it leaves the tile.  One wavefront per workgroup.
template <int G>
__device__ __forceinline__ void traceback_group(const ulonglong2 *tr, int nbb, int lo_pad, int pad, int rows, int L, bool valid,
                                                const uint8_t *q, const uint8_t *t, uint8_t *ops, int ops_cap, int &n_ins, int &n_ops,
                                                ulonglong2 *tile, int lane, int g)
{
    constexpr int GL = G < 8 ? G : 8; // lanes of a group that fetch
    constexpr int E = 16 * GL;        // entries per tile
    constexpr int TC = E / 2;         // columns per tile
    const int nwords = rows >> 6;
    auto at = [&](int ck, int w) {
        int f = (ck + lo_pad) >> 6;
        const int fmax = nwords - nbb;
        f = f < 0 ? 0 : f > fmax ? fmax : f;
        int rel = w - f;
        rel = rel < 0 ? 0 : rel >= nbb ? nbb - 1 : rel; // blocks a staged column does not hold are never walked
        return (int64_t)ck * nbb + rel;
    };
    int ip = rows - 1, c = L - 1, ins = 0, cnt = 0; // the walker's state (lane g == 0)
    bool walking = valid && g == 0 && ip >= pad && c >= 0;
    while (__any(walking)) {
        const int src = lane - g;
        const int wip = __shfl(ip, src, 64), wc = __shfl(c, src, 64);
        const bool wwalking = __shfl((int)walking, src, 64) != 0;
        int wlo = (wip - 32) >> 6;
        wlo = wlo > nwords - 2 ? nwords - 2 : wlo;
        wlo = wlo < 0 ? 0 : wlo;
        const int w1 = wlo + 1 < nwords ? wlo + 1 : wlo;
        if (wwalking && g < GL) {
            ulonglong2 buf[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int e = g + k * GL;
                const int ck = wc - (e >> 1) > 0 ? wc - (e >> 1) : 0;
                buf[k] = tr[at(ck, (e & 1) ? w1 : wlo)];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) tile[g + k * GL] = buf[k];
        }
        __syncthreads();
        if (walking) {
            const int c_hi = c, win_lo = c - TC + 1;
            while (ip >= pad && c >= 0 && c >= win_lo) {
                const int w = ip >> 6;
                if (w < wlo || w > wlo + 1) break;
                const ulonglong2 e = tile[(c_hi - c) * 2 + (w - wlo)];
                const uint64_t bit = ((uint64_t)1) << (ip & 63);
                int code;
                if (e.x & bit) { code = 1; ins++; ip--; }
                else if (e.y & bit) { code = 2; c--; }
                else { code = (ops && q[ip - pad] != t[c]) ? 3 : 0; ip--; c--; }
                if (ops && cnt < ops_cap) ops[cnt] = (uint8_t)code;
                cnt++;
            }
            walking = ip >= pad && c >= 0;
        }
        __syncthreads(); // the tile is overwritten in the next round
    }
    if (valid && g == 0) {
        if (ip >= pad) { // left boundary reached: the remaining query bases are insertions
            const int r = ip - pad + 1;
            if (ops) for (int u = 0; u < r; u++) if (cnt + u < ops_cap) ops[cnt + u] = 1;
            ins += r; cnt += r;
        } else if (c >= 0) { // query used up: the remaining target bases are deletions
            if (ops) for (int u = 0; u <= c; u++) if (cnt + u < ops_cap) ops[cnt + u] = 2;
            cnt += c + 1;
        }
        n_ins = ins; n_ops = cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// lane-per-pair kernel: W words per lane in registers
// ------------------------------------------------------------------------------------------------
// One wavefront's worth of work: items [base, base + 64) (clipped to a.end).  lds_wave: S*W*64 words
// private to the wavefront; gwave: index of the wavefront's trace region (MODE_NW only).
template <int W, int MODE, bool HALF = false>
__device__ __forceinline__ void lane_item(const MyArgs &a, int64_t base, uint64_t *lds_wave, int64_t gwave)
{
    const int lane = threadIdx.x & 63;
    const int S = a.S;
    uint64_t *peq = lds_wave + lane; // word (sym, b) at peq[(sym*W + b)*64]
    constexpr int ROWS = 64 * W;
    {
        Item it = setup_item<MODE>(a, base + lane);
        if (MODE == MODE_FWD_EXACT || MODE == MODE_NW) {
            if (!__any(it.valid)) return;
        }
        const int pad = ROWS - it.m;
        // ---- Peq into LDS (lane-private words: no synchronisation needed) ----
        {
            const uint64_t padfill_on = (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) ? ~(uint64_t)0 : 0;
            for (int s = 0; s < S; s++) {
#pragma unroll
                for (int b = 0; b < W; b++) peq[(s * W + b) * 64] = pad_mask(pad, b) & padfill_on;
            }
            int mmax = wave_max_i32(it.valid ? it.m : 0);
            for (int i0 = 0; i0 < mmax; i0 += 8) {
                if (it.valid && i0 < it.m) {
                    uint64_t chunk = load8(it.q + i0);
                    int lim = it.m - i0; if (lim > 8) lim = 8;
                    for (int u = 0; u < lim; u++) {
                        int code = (int)((chunk >> (8 * u)) & 0xff);
                        int bit = pad + i0 + u;
                        peq[(code * W + (bit >> 6)) * 64] |= ((uint64_t)1) << (bit & 63);
                    }
                }
            }
        }
        uint64_t Pv[W], Mv[W];
#pragma unroll
        for (int b = 0; b < W; b++) { Pv[b] = ~pad_mask(pad, b); Mv[b] = 0; }
        const uint32_t top_hp = (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) ? 0u : 1u;
        int32_t score = it.m;
        int32_t best = 0x3fffffff, cnt = 0, lim = it.k, last = -1;
        int32_t *out = nullptr; int32_t out_cap = 0;
        if (MODE == MODE_FWD && it.valid) { out = a.slots + (int64_t)it.pair * a.slot_cap; out_cap = a.slot_cap; }
        if (MODE == MODE_FWD_EXACT && it.valid) { out = a.loc_end + a.loc_off[it.pair]; out_cap = a.nloc[it.pair]; }
        ulonglong2 *tT = nullptr;
        if (MODE == MODE_NW) tT = a.tr + gwave * a.region + lane;
        const int n = it.valid ? it.n : 0;
        const int nmax = wave_max_i32(n);
        // Columns every valid lane has (the pairs of a work item are sorted by target length, so this is nearly
        // all of them): run without per-lane bounds tests.  Lanes without a pair tag along on the arena's first
        // bytes with out_cap = 0; the NW pass stores every column, so it keeps the tested loop throughout.
        const int nfast = MODE == MODE_NW ? 0 : (min(-wave_max_i32(it.valid ? -n : -0x7fffffff), nmax) & ~15);
        // Eight columns per target chunk.  The chunks are double-buffered (the load of the next one is
        // issued a whole block of columns before its first use) and the match words of column j+1 are
        // read from LDS while column j is computed.
        auto read_eq = [&](uint64_t *eq, uint64_t chunk, int u, bool ok) {
            const int code = ok ? (int)((chunk >> (8 * u)) & 0xff) : 0;
            const uint64_t *pe = peq + (size_t)code * (W * 64);
#pragma unroll
            for (int b = 0; b < W; b++) eq[b] = pe[b * 64];
        };
        auto block8 = [&](uint64_t chunk, int j0, auto tested) {
            constexpr bool TESTED = decltype(tested)::value;
            constexpr bool FWD = MODE == MODE_FWD || MODE == MODE_FWD_EXACT;
            uint64_t eq[W], eqn[W];
            // Forward pass: the eight bottom-row scores of the block stay in registers and only their minimum is
            // tested against the threshold (one v_min3 per two columns); the end-column bookkeeping below runs for
            // the few blocks in which some lane comes within reach of its best - not per column for every decoy.
            int32_t sc[8];
            read_eq(eq, chunk, 0, !TESTED || j0 < n);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = j0 + u;
                if (u < 7) read_eq(eqn, chunk, u + 1, !TESTED || j + 1 < n);
                if (FWD) sc[u] = 0x3fffffff;
                if (!TESTED || j < n) {
                    uint32_t hp = top_hp, hm = 0;
#pragma unroll
                    for (int b = 0; b < W; b++) {
                        uint64_t ph0;
                        if (FWD && HALF && b == 0) { word_step_hi(Pv[0], Mv[0], eq[0], hp, hm); continue; } // HALF: see lane_forward
                        word_step(Pv[b], Mv[b], eq[b], hp, hm, ph0);
                        if (MODE == MODE_NW) {
                            ulonglong2 e; e.x = Pv[b]; e.y = ph0;
                            tT[((int64_t)j * W + b) * 64] = e;
                        }
                    }
                    score += (int32_t)hp - (int32_t)hm;
                    if (FWD) sc[u] = score;
                    else if (MODE == MODE_REV) {
                        if (score <= it.best) last = j;
                    }
                }
#pragma unroll
                for (int b = 0; b < W; b++) eq[b] = eqn[b];
            }
            if (FWD) {
                const int32_t bm = min(min(min(sc[0], sc[1]), min(sc[2], sc[3])), min(min(sc[4], sc[5]), min(sc[6], sc[7])));
                if (bm <= lim) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (sc[u] <= lim) {
                            if (sc[u] < best) { best = sc[u]; cnt = 0; lim = sc[u]; }
                            if (cnt < out_cap) out[cnt] = j0 + u;
                            cnt++;
                        }
                    }
                }
            }
        };
        uint64_t ca = load8(it.t), cb = 0;
        int j0 = 0;
        for (; j0 < nfast; j0 += 16) {
            cb = load8(it.t + j0 + 8);
            block8(ca, j0, std::false_type{});
            ca = load8(it.t + j0 + 16); // at most 8 bytes past the slice: the arena is padded
            block8(cb, j0 + 8, std::false_type{});
        }
        for (; j0 < nmax; j0 += 16) {
            if (j0 + 8 < n) cb = load8(it.t + j0 + 8);
            block8(ca, j0, std::true_type{});
            if (j0 + 16 < n) ca = load8(it.t + j0 + 16);
            if (j0 + 8 < nmax) block8(cb, j0 + 8, std::true_type{});
        }
        if (MODE == MODE_FWD) {
            if (it.valid) { a.ed[it.pair] = (cnt > 0) ? best : -1; a.nloc[it.pair] = cnt; }
            const bool over = __any(it.valid && cnt > out_cap);
            if (a.wflag && over && lane == 0) a.wflag[atomicAdd(a.wcount, 1)] = (int32_t)blockIdx.x * a.bstride + a.boff;
        } else if (MODE == MODE_REV) {
            if (it.valid) a.loc_start[it.loc] = a.loc_end[it.loc] - last;
        } else if (MODE == MODE_NW) {
            __threadfence();
            if (it.valid) {
                uint8_t *ops = nullptr; int ops_cap = 0;
                if (a.ops) { ops = a.ops + a.ops_off[it.pair]; ops_cap = (int)(a.ops_off[it.pair + 1] - a.ops_off[it.pair]); }
                int ins, nops;
                traceback(tT, 64, (int64_t)W * 64, pad, ROWS, it.n, it.q, it.t, ops, ops_cap, ins, nops,
                          (ulonglong2 *)(lds_wave + (size_t)S * W * 64) + lane, W, 0);
                a.first_ins[it.pair] = ins;
                if (a.n_ops) a.n_ops[it.pair] = nops;
            }
            __threadfence();
        }
    }
}

// Forward pass of a lane class: when every pair of the wavefront leaves the lower half of its first word to the top
// padding (the pairs are sorted for that: k_pair_keys), that word runs on one 32-bit half.  Decided once per work item,
// outside every loop, so that each variant keeps its own instruction schedule and register allocation.
template <int W, int MODE>
__device__ __forceinline__ void lane_forward(const MyArgs &a, int64_t base, uint64_t *lds_wave)
{
    const int64_t p = base + (threadIdx.x & 63);
    const int m = p < a.end ? a.q_len[p] : 1;
    if (__builtin_amdgcn_readfirstlane((int)(__all(64 * W - m >= 32) ? 1 : 0)) != 0) lane_item<W, MODE, true>(a, base, lds_wave, 0);
    else lane_item<W, MODE, false>(a, base, lds_wave, 0);
}

template <int W, int MODE>
__global__ void __launch_bounds__(256) myers_lane(MyArgs a)
{
    extern __shared__ uint64_t lds[];
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int64_t gwave = (int64_t)blockIdx.x * waves_per_block + wave;
    const int64_t nwaves = (int64_t)gridDim.x * waves_per_block;
    uint64_t *lds_wave = lds + (size_t)wave * ((size_t)a.S * W * 64 + (MODE == MODE_NW ? (size_t)kStage * 2 * 64 * 2 : 0));
    for (int64_t base = a.begin + gwave * 64; base < a.end; base += nwaves * 64) lane_item<W, MODE>(a, base, lds_wave, gwave);
}

// ------------------------------------------------------------------------------------------------
// lane-group kernel: G lanes per pair, W words per lane, skewed (anti-diagonal) schedule
// ------------------------------------------------------------------------------------------------
// One wavefront's worth of work: items [base, base + 64/G).  slot0: first trace slot of the wavefront (MODE_NW).
// LONG (G = 64, W = 8 only): a query of any length, taken in strips of 32768 rows by the one wavefront.  A strip runs
// over all columns; the horizontal deltas that leave its last row go to a byte-per-column buffer in HBM and enter the
// next strip's first row (two buffers, alternating).  Only the last strip scores.  The top padding lies in strip 0.
template <int G, int W, int MODE, bool LONG = false>
__device__ __forceinline__ void group_item(const MyArgs &a, int64_t base, uint64_t *lds_wave, int64_t slot0)
{
    static_assert(!LONG || (G == 64 && W == 8), "the long class is the widest wavefront-per-pair class, looped");
    constexpr int P = 64 / G;    // pairs per wave
    constexpr int NB = G * W;    // words per pair (and strip)
    constexpr int ROWS = 64 * NB;
    const int lane = threadIdx.x & 63;
    const int g = lane % G, p = lane / G;
    const int S = a.S;
    uint64_t *peq = lds_wave + (size_t)p * S * NB; // word (sym, blk) at peq[sym*NB + blk]
    {
        Item it = setup_item<MODE>(a, base + p);
        if (MODE == MODE_FWD_EXACT || MODE == MODE_NW) {
            if (!__any(it.valid)) return;
        }
        const int nstrip = LONG ? __builtin_amdgcn_readfirstlane((it.m + ROWS - 1) / ROWS) : 1;
        const int NBT = NB * nstrip;          // blocks of the whole (padded) query
        const int pad = ROWS * nstrip - it.m; // < ROWS
        const uint64_t padfill_on = (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) ? ~(uint64_t)0 : 0;
        uint64_t Pv[W], Mv[W];
        const uint32_t top_hp = (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) ? 0u : 1u;
        int32_t score = it.m;
        int32_t best = 0x3fffffff, cnt = 0, lim = it.k, last = -1;
        int32_t *out = nullptr; int32_t out_cap = 0;
        if (MODE == MODE_FWD && it.valid) { out = a.slots + (int64_t)it.pair * a.slot_cap; out_cap = a.slot_cap; }
        if (MODE == MODE_FWD_EXACT && it.valid) { out = a.loc_end + a.loc_off[it.pair]; out_cap = a.nloc[it.pair]; }
        ulonglong2 *tT = nullptr;
        if (MODE == MODE_NW) tT = a.tr + (slot0 + p) * a.region;
        // NW pass with a known distance: only the blocks that the ed + 1 reachable diagonals cross are stored
        // (myers_band explains the range); nbb == NBT stores everything.
        const int nbb = (MODE == MODE_NW && a.nbb > 0 && a.nbb < NBT) ? a.nbb : NBT;
        int lo_pad = 0;
        if (MODE == MODE_NW) {
            const int delta = it.m - it.n;
            const int hh = (it.best - (delta < 0 ? -delta : delta)) >> 1;
            lo_pad = (delta < 0 ? delta : 0) - hh - 1 + pad;
        }
        const int n = it.valid ? it.n : 0;
        const int nmax = wave_max_i32(n);
        const int steps = nmax + G - 1;
        uint8_t *hb0 = nullptr, *hb1 = nullptr; // LONG: 64 readable bytes in front of each, 16 behind column n
        if (LONG) {
            const int64_t hi = MODE == MODE_NW ? slot0 : base - a.hbase;
            hb0 = a.hbuf + (2 * hi) * a.hstride + 64;
            hb1 = hb0 + a.hstride;
        }
      for (int strip = 0; strip < nstrip; strip++) {
        const bool last_strip = strip == nstrip - 1;
        const uint8_t *hin = (strip & 1) ? hb0 : hb1; // what the strip before left
        uint8_t *hout = (strip & 1) ? hb1 : hb0;
        if (LONG && strip > 0) __threadfence();       // the strip before wrote hin through other lanes
#pragma unroll
        for (int w = 0; w < W; w++) {
            const int blk = g * W + w, gblk = strip * NB + blk;
            for (int s = 0; s < S; s++) peq[s * NB + blk] = pad_mask(pad, gblk) & padfill_on;
            if (it.valid) {
                for (int bit = 0; bit < 64; bit++) {
                    int i = 64 * gblk + bit - pad;
                    if (i >= 0 && i < it.m) peq[(int)it.q[i] * NB + blk] |= ((uint64_t)1) << bit;
                }
            }
            Pv[w] = ~pad_mask(pad, gblk); Mv[w] = 0;
        }
        score = it.m;
        uint32_t h_out = 0; // bit 0: +1 carry out of this lane's last word, bit 1: -1 carry
        if (MODE != MODE_NW) {
            // Forward / start passes.  Three things keep the per-step overhead down:
            //  * every lane refills its 8 target bytes at the same steps (t % 8 == 0, from its own column t - g on:
            //    an unaligned 8-byte load), so the refill is a uniform branch and the byte window shifts unconditionally;
            //  * between the ramp-up and the shortest target of the wavefront every lane is inside its pair: those steps
            //    run without range tests (lanes without a pair tag along on garbage with out_cap = 0, as in lane_item);
            //  * only the last lane of a pair scores, by giving the others thresholds that never fire - no g == G-1 branch.
            const bool lastlane = g == G - 1;
            const int act_n = it.valid ? n : nmax;
            const int nmin = -wave_max_i32(-act_n);
            const uint8_t *tp = it.valid ? it.t : a.seq + 64; // 64 readable bytes lie in front of every target
            int32_t limv = (lastlane && last_strip) ? lim : (int32_t)0x80000000;
            const int32_t revbest = (lastlane && last_strip) ? it.best : (int32_t)0x80000000;
            uint64_t tw = 0, twn = load8(tp - g);
            uint64_t hw = 0, hwn = 0; // LONG: the deltas entering lane 0's first row, eight columns at a time like the target
            if (LONG && strip > 0) hwn = load8(hin - g);
            auto step = [&](int t, auto tested) {
                constexpr bool TESTED = decltype(tested)::value;
                uint32_t h_in = G <= 16 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h_out, 0x111 /* row_shr:1 */, 0xf, 0xf, false)
                                        : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if (g == 0) h_in = top_hp;
                if (LONG && strip > 0) { if (g == 0) h_in = (uint32_t)(hw & 0xff); hw >>= 8; }
                uint32_t hp = h_in & 1u, hm = h_in >> 1;
                const int col = t - g;
                const int code = (int)(tw & 0xff);
                tw >>= 8;
                if (!TESTED || (col >= 0 && col < act_n)) {
                    const uint64_t *pe = peq + (size_t)code * NB + g * W;
#pragma unroll
                    for (int w = 0; w < W; w++) {
                        uint64_t ph0;
                        word_step(Pv[w], Mv[w], pe[w], hp, hm, ph0);
                    }
                    h_out = hp | (hm << 1);
                    if (LONG && !last_strip && lastlane) hout[col] = (uint8_t)h_out;
                    score += (int32_t)hp - (int32_t)hm;
                    if (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) {
                        if (score <= limv) {
                            if (score < best) { best = score; cnt = 0; limv = score; }
                            if (cnt < out_cap) out[cnt] = col;
                            cnt++;
                        }
                    } else {
                        if (score <= revbest) last = col;
                    }
                }
            };
            int t = 0;
            auto refill = [&](int tt) {
                tw = twn;
                int pos = tt - g + 8;
                pos = pos < act_n ? pos : act_n;
                twn = load8(tp + pos);
                if (LONG && strip > 0) { hw = hwn; hwn = load8(hin + pos); }
            };
            const int t_fast0 = G - 1, t_fast1 = nmin > G - 1 ? nmin : G - 1;
            for (; t < t_fast0 && t < steps; t++) { if ((t & 7) == 0) refill(t); step(t, std::true_type{}); }
            for (; t < t_fast1; t++) { if ((t & 7) == 0) refill(t); step(t, std::false_type{}); }
            for (; t < steps; t++) { if ((t & 7) == 0) refill(t); step(t, std::true_type{}); }
        } else {
        uint64_t tw = 0;    // this lane's next target bytes (its column lags g steps behind lane 0)
        uint64_t twn = n > 0 ? load8(it.t) : 0; // the chunk after it, loaded eight columns ahead of its use
        uint64_t hw = 0;    // LONG: the deltas entering lane 0's first row
        for (int t = 0; t < steps; t++) {
            // the carry moves one lane down per step: a DPP shift (inside a 16-lane row, or across the whole wavefront)
            uint32_t h_in = G <= 16 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h_out, 0x111 /* row_shr:1 */, 0xf, 0xf, false)
                                    : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            if (g == 0) h_in = top_hp;
            const int col = t - g;
            if (col >= 0 && col < n) {
                if ((col & 7) == 0) { tw = twn; if (col + 8 < n) twn = load8(it.t + col + 8); }
                if (LONG && strip > 0) {
                    if ((col & 7) == 0) hw = load8(hin + col);
                    if (g == 0) h_in = (uint32_t)(hw & 0xff);
                    hw >>= 8;
                }
                uint32_t hp = h_in & 1u, hm = h_in >> 1;
                const int code = (int)(tw & 0xff);
                tw >>= 8;
                const uint64_t *pe = peq + (size_t)code * NB + g * W;
#pragma unroll
                for (int w = 0; w < W; w++) {
                    uint64_t ph0;
                    word_step(Pv[w], Mv[w], pe[w], hp, hm, ph0);
                    ulonglong2 e; e.x = Pv[w]; e.y = ph0;
                    int f = (col + lo_pad) >> 6;
                    f = f < 0 ? 0 : f > NBT - nbb ? NBT - nbb : f;
                    const int rel = strip * NB + g * W + w - f;
                    if (rel >= 0 && rel < nbb) tT[(int64_t)col * nbb + rel] = e;
                }
                h_out = hp | (hm << 1);
                if (LONG && !last_strip && g == G - 1) hout[col] = (uint8_t)h_out;
            }
        }
        }
      } // strips
        if (MODE == MODE_FWD) {
            if (it.valid && g == G - 1) { a.ed[it.pair] = (cnt > 0) ? best : -1; a.nloc[it.pair] = cnt; }
            const bool over = __any(it.valid && g == G - 1 && cnt > out_cap);
            if (a.wflag && over && lane == 0) a.wflag[atomicAdd(a.wcount, 1)] = (int32_t)blockIdx.x * a.bstride + a.boff;
        } else if (MODE == MODE_REV) {
            if (it.valid && g == G - 1) a.loc_start[it.loc] = a.loc_end[it.loc] - last;
        } else if (MODE == MODE_NW) {
            __threadfence();
            {
                uint8_t *ops = nullptr; int ops_cap = 0;
                if (a.ops && it.valid) { ops = a.ops + a.ops_off[it.pair]; ops_cap = (int)(a.ops_off[it.pair + 1] - a.ops_off[it.pair]); }
                int ins = 0, nops = 0;
                constexpr int TILE = 16 * (G < 8 ? G : 8); // entries of a pair's tile: P * TILE <= 1024 = the staging area
                traceback_group<G>(tT, nbb, lo_pad, pad, ROWS * nstrip, it.n, it.valid, it.q, it.t, ops, ops_cap, ins, nops,
                                   (ulonglong2 *)(lds_wave + (size_t)S * NB * P) + (size_t)p * TILE, lane, g);
                if (it.valid && g == 0) {
                    a.first_ins[it.pair] = ins;
                    if (a.n_ops) a.n_ops[it.pair] = nops;
                }
            }
            __threadfence();
        }
    }
}

template <int G, int W, int MODE, bool LONG = false>
__global__ void __launch_bounds__(64) myers_group(MyArgs a)
{
    if (MODE == MODE_NW) __builtin_amdgcn_s_setprio(3); // few, long wavefronts beside the band kernels (see myers_band)
    extern __shared__ uint64_t lds[];
    constexpr int P = 64 / G;
    for (int64_t base = a.begin + (int64_t)blockIdx.x * P; base < a.end; base += (int64_t)gridDim.x * P)
        group_item<G, W, MODE, LONG>(a, base, lds, (int64_t)blockIdx.x * P);
}

// ------------------------------------------------------------------------------------------------
// one launch per pass: every wavefront takes one work item (64 pairs of one size class)
// ------------------------------------------------------------------------------------------------
// The size classes differ 100x in work per wavefront.  Launching them one after the other leaves the
// chip nearly empty while the few long-running wavefronts of the big classes finish, so a pass is
// ONE grid per register bucket: work items sorted heaviest first (workgroups are dispatched in
// index order), one wavefront per workgroup, each jumping to the code of its class.
template <int MODE, int BUCKET>
__global__ void __launch_bounds__(64) myers_mega(MyArgs a, const WorkItem *list, int32_t n)
{
    extern __shared__ uint64_t lds[];
    const int32_t item = (int32_t)blockIdx.x * a.bstride + a.boff;
    if (item >= n) return;
    WorkItem wi;
    if (MODE == MODE_FWD_EXACT) wi = list[a.wflag[item]]; // the forward launch over the same list left these
    else if (list) wi = list[item];
    else {
        // constant indices only: a dynamically indexed member would move the whole argument block into scratch memory
        int32_t cls = a.rr[0].cls, first = a.rr[0].first_block;
        int64_t rb = a.rr[0].b, re = a.rr[0].e;
#pragma unroll
        for (int t = 1; t < 5; t++)
            if (t < a.n_rr && item >= a.rr[t].first_block) { cls = a.rr[t].cls; first = a.rr[t].first_block; rb = a.rr[t].b; re = a.rr[t].e; }
        wi.cls = cls;
        const int per = wi.cls < 4 ? 64 : wi.cls < 6 ? 32 : wi.cls < 8 ? 16 : wi.cls < 10 ? 8 : 1; // 64 / G of the class
        wi.base = rb + (int64_t)(item - first) * per;
        wi.end = re;
        wi.cost = 0;
    }
    a.end = wi.end;
    constexpr bool FWD = MODE == MODE_FWD || MODE == MODE_FWD_EXACT;
    if (BUCKET == 0) {
        if (FWD) { if (wi.cls == 0) lane_forward<1, MODE>(a, wi.base, lds); else lane_forward<2, MODE>(a, wi.base, lds); }
        else if (wi.cls == 0) lane_item<1, MODE>(a, wi.base, lds, 0); else lane_item<2, MODE>(a, wi.base, lds, 0);
    } else if (BUCKET == 1) {
        if (FWD) { if (wi.cls == 2) lane_forward<3, MODE>(a, wi.base, lds); else lane_forward<4, MODE>(a, wi.base, lds); }
        else if (wi.cls == 2) lane_item<3, MODE>(a, wi.base, lds, 0); else lane_item<4, MODE>(a, wi.base, lds, 0);
    } else if (BUCKET == 2) {
        if (wi.cls == 4) group_item<2, 3, MODE>(a, wi.base, lds, 0); else group_item<2, 4, MODE>(a, wi.base, lds, 0);
    } else if (BUCKET == 3) {
        switch (wi.cls) {
        case 6: group_item<4, 3, MODE>(a, wi.base, lds, 0); break;
        case 7: group_item<4, 4, MODE>(a, wi.base, lds, 0); break;
        case 8: group_item<8, 3, MODE>(a, wi.base, lds, 0); break;
        default: group_item<8, 4, MODE>(a, wi.base, lds, 0); break;
        }
    } else {
        switch (wi.cls) {
        case 10: group_item<64, 1, MODE>(a, wi.base, lds, 0); break;
        case 11: group_item<64, 2, MODE>(a, wi.base, lds, 0); break;
        case 12: group_item<64, 4, MODE>(a, wi.base, lds, 0); break;
        case 13: group_item<64, 8, MODE>(a, wi.base, lds, 0); break;
        default: group_item<64, 8, MODE, true>(a, wi.base, lds, 0); break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// banded traceback kernel: lane-per-pair, diagonal band of 64*WB rows
// ------------------------------------------------------------------------------------------------
// The first location's global alignment has a KNOWN distance ed (forward + reverse passes), so every
// cell of an optimal path lies within |row - column| <= ed of the main diagonal.  The window of
// 64*WB rows slides down one row per column (Hyyro's diagonal formulation: the new vertical deltas
// take D0 >> 1), which makes the cost per column independent of the query length, and the stored
// traceback state {Pv, Ph} is WB words per column instead of ceil(m/64).  No Peq table: the window's
// match vector is rebuilt from NP bit-planes of the query codes that slide with the window.
// Window of column c: bit b <-> query row c - TOP + off + b, TOP = 32*WB - 1, off = per-pair shift towards the
// diagonals the path can reach (below); needs ed <= 64*WB - 10.
//
// Traceback state never goes through HBM as whole columns: the forward sweep only leaves a
// checkpoint (Pv, Mv, bit-planes) every K columns; the walk then goes through the chunks backwards,
// recomputing one chunk's K columns from its checkpoint into LDS ({Pv, Ph} per column) and walking
// inside LDS.  Twice the (cheap, banded) arithmetic instead of 16*WB bytes per column and lane of
// HBM traffic plus a dependent load per walk step.
template <int WB> struct BandK { static constexpr int K = WB <= 1 ? band_chunk(1) : 8; }; // columns per chunk

template <int WB, int NP>
__global__ void __launch_bounds__(64) myers_band(MyArgs a, const int32_t *items, int64_t n_items)
{
    constexpr int TOP = 32 * WB - 1;
    constexpr int K = BandK<WB>::K;
    constexpr int CPW = (2 + NP) * WB; // 64-bit words per checkpoint and lane
    // The traceback launches are bound by their longest wavefronts, the wider the band the longer: let those win
    // the issue slots of a SIMD they share with the many short wavefronts of the narrow classes.
    __builtin_amdgcn_s_setprio(WB >= 6 ? 3 : WB >= 4 ? 2 : WB >= 2 ? 1 : 0);
    const int lane = threadIdx.x;
    const int64_t gwave = blockIdx.x, nwaves = gridDim.x;
    uint64_t *CP = (uint64_t *)a.tr + gwave * a.region + lane; // checkpoint (chunk h, word f) at CP[(h * CPW + f) * 64]
    extern __shared__ ulonglong2 band_lds[];
    ulonglong2 *col = band_lds + lane; // (column k of the chunk, word w) at col[(k * WB + w) * 64]
    for (int64_t base = gwave * 64; base < n_items; base += nwaves * 64) {
        bool valid = base + lane < n_items;
        int p = 0, m = 0, L = 0, ed = 0;
        const uint8_t *q = a.seq, *t = a.seq;
        if (valid) {
            p = items[base + lane];
            m = a.q_len[p];
            ed = a.ed[p];
            const int64_t lo = a.loc_off[p];
            const int s0 = a.loc_start[lo];
            L = a.loc_end[lo] - s0 + 1;
            q = a.seq + a.q_off[p];
            t = a.seq + a.t_off[p] + s0;
            if (tb_by_arithmetic(ed, m, L)) { valid = false; p = m = L = ed = 0; q = t = a.seq; } // answered by k_band_keys
        }
        // Diagonals (row - column) an optimal path can touch: it starts on diagonal 0, ends on delta = m - L and
        // pays one edit per diagonal change, so |d| + |delta - d| <= ed, i.e. d in [min(0, delta) - h, max(0, delta) + h]
        // with h = (ed - |delta|) / 2: ed + 1 diagonals, not 2 ed + 1.  The window (bit b <-> diagonal b - TOP + off)
        // is shifted per lane so that its lowest diagonal sits just above that range; off is kept a multiple of 8
        // so that the 8-byte query window refills together with the target's.
        const int delta = m - L;
        const int hh = (ed - (delta < 0 ? -delta : delta)) >> 1;
        const int off = ((delta < 0 ? delta : 0) - hh - 1 + TOP) & ~7;
        uint64_t PL[NP][WB], VP[WB], VN[WB];
        // Target and query bytes travel one chunk of K columns at a time (K / 8 loads of 8 bytes each, issued together
        // a whole chunk before their first use; see myers_band32).  Column c consumes target byte c and query row
        // TOP + 1 + off + c, both multiples of 8 together.  These kernels run at one or two wavefronts per SIMD:
        // nothing else hides a load.
        constexpr int NB = K / 8;
        uint64_t qw = 0, tw = 0, qwb[NB], twb[NB], qwn[NB], twn[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) { qwb[i] = twb[i] = qwn[i] = twn[i] = 0; }
        int pref_c = -0x40000000; // chunk the blocks in twn / qwn belong to
        auto qblock = [&](int r) { return (r >= 0 && r < m) ? load8(q + r) : (uint64_t)0; };
        auto prefetch = [&](int cn) {
            pref_c = cn;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                twn[i] = (cn + 8 * i >= 0 && cn + 8 * i < L) ? load8(t + cn + 8 * i) : 0;
                qwn[i] = qblock(TOP + 1 + off + cn + 8 * i);
            }
        };
        auto push_row = [&](uint32_t code1) { // code + 1 of the row entering at the bottom; 0 = outside the query: matches nothing
#pragma unroll
            for (int k = 0; k < NP; k++) {
#pragma unroll
                for (int w = 0; w < WB - 1; w++) PL[k][w] = shr1_in(PL[k][w], (uint32_t)PL[k][w + 1]);
                PL[k][WB - 1] = shr1_in(PL[k][WB - 1], code1 >> k);
            }
        };
        // one column of the banded recurrence; STORE: leave {Pv, Ph} of the column in LDS slot k.
        // next_c: the column group to prefetch when this column opens a group
        auto column = [&](int c, int k, bool store, int next_c) {
            if (k == 0) { // k is a compile-time constant in the unrolled chunk loops
                if (pref_c != c) prefetch(c); // not staged ahead: first chunk of a sweep, or chunks were skipped
#pragma unroll
                for (int i = 0; i < NB; i++) { twb[i] = twn[i]; qwb[i] = qwn[i]; }
                prefetch(next_c);
            }
            if ((k & 7) == 0) { tw = twb[k >> 3]; qw = qwb[k >> 3]; }
            const uint32_t tc1 = (uint32_t)(tw & 0xff) + 1u;
            tw >>= 8;
            // match vector: bit set where every plane of the query code equals the target's bit (XNOR per plane,
            // the target's bits as per-lane all-0 / all-1 masks); three-input ops on 32-bit halves throughout
            uint64_t Eq[WB];
            {
                uint32_t mk[NP];
#pragma unroll
                for (int kk = 0; kk < NP; kk++) mk[kk] = ((tc1 >> kk) & 1u) - 1u; // bit set -> 0, clear -> ~0
#pragma unroll
                for (int w = 0; w < WB; w++) {
                    uint32_t el = (uint32_t)PL[0][w] ^ mk[0], eh = (uint32_t)(PL[0][w] >> 32) ^ mk[0];
#pragma unroll
                    for (int kk = 1; kk < NP; kk++) {
                        el = bop3<0x60>(el, (uint32_t)PL[kk][w], mk[kk]);         // a & (b ^ c)
                        eh = bop3<0x60>(eh, (uint32_t)(PL[kk][w] >> 32), mk[kk]);
                    }
                    Eq[w] = pack64(el, eh);
                }
            }
            uint64_t D0[WB], HP[WB], HN[WB];
            uint64_t carry = 0;
#pragma unroll
            for (int w = 0; w < WB; w++) {
                const uint64_t x = Eq[w] & VP[w];
                const uint64_t s1 = x + VP[w];
                const uint64_t s2 = s1 + carry;
                carry = (uint64_t)((s1 < x) | (s2 < s1));
                D0[w] = bop3_64<0xBE>(s2, VP[w], Eq[w]) | VN[w]; // (s2 ^ VP) | Eq | VN
                HP[w] = bop3_64<0xF1>(VN[w], D0[w], VP[w]);      // VN | ~(D0 | VP)
                HN[w] = D0[w] & VP[w];
            }
#pragma unroll
            for (int w = 0; w < WB; w++) {
                const uint64_t X = shr1_in(D0[w], w + 1 < WB ? (uint32_t)D0[w + 1] : 0u);
                VN[w] = X & HP[w];
                VP[w] = bop3_64<0xF1>(HN[w], X, HP[w]);          // HN | ~(X | HP)
                if (store) {
                    ulonglong2 e; e.x = VP[w];
                    e.y = shr1_in(HP[w], w + 1 < WB ? (uint32_t)HP[w + 1] : 0u); // Ph in this column's alignment
                    col[(k * WB + w) * 64] = e;
                }
            }
            {
                const int r = TOP + 1 + off + c; // the row that enters the window for column c + 1
                push_row((r >= 0 && r < m) ? (uint32_t)(qw & 0xff) + 1u : 0u);
                qw >>= 8;
            }
        };
#pragma unroll
        for (int k = 0; k < NP; k++)
#pragma unroll
            for (int w = 0; w < WB; w++) PL[k][w] = 0;
        // the window of column -1 holds rows off - TOP - 1 .. off + TOP; those above row 0 are virtual (planes 0)
        {
            const int r_end = TOP + 1 + off, r0 = off - TOP - 1;
            int ri = r0 < 0 ? (r_end < 0 ? r_end : 0) : r0; // multiples of 8
            uint64_t cur = 0, nxt = qblock(ri);
            while (ri < r_end) {
                if ((ri & 7) == 0) { cur = nxt; nxt = qblock(ri + 8); }
                push_row((ri >= 0 && ri < m) ? (uint32_t)(cur & 0xff) + 1u : 0u);
                cur >>= 8;
                ri++;
            }
        }
        // column -1: D[r][-1] = r + 1 for query rows, 0 for the virtual rows above row 0
#pragma unroll
        for (int w = 0; w < WB; w++) {
            // bit b <-> row -1 - TOP + off + b >= 0  <=>  b >= TOP + 1 - off
            const int z = TOP + 1 - off - 64 * w;
            VP[w] = z <= 0 ? ~(uint64_t)0 : z >= 64 ? 0 : (~(uint64_t)0) << z;
            VN[w] = 0;
        }
        const int Lmax = wave_max_i32(L);
        const int nchunk = (Lmax + K - 1) / K;
        // ---- forward sweep: checkpoints only ----
        for (int h = 0; h < nchunk; h++) {
            if (h * K < L) {
                uint64_t *cp = CP + (int64_t)h * CPW * 64;
#pragma unroll
                for (int w = 0; w < WB; w++) { cp[w * 64] = VP[w]; cp[(WB + w) * 64] = VN[w]; }
#pragma unroll
                for (int k = 0; k < NP; k++)
#pragma unroll
                    for (int w = 0; w < WB; w++) cp[((2 + k) * WB + w) * 64] = PL[k][w];
            }
            if (h + 1 < nchunk) // the last chunk is recomputed first anyway
#pragma unroll
                for (int k = 0; k < K; k++) column(h * K + k, k, false, h * K + K);
        }
        // ---- backwards through the chunks: recompute into LDS, walk inside LDS ----
        int r = m - 1, c = L - 1, ins = 0;
        bool walking = valid;
        for (int h = nchunk - 1; h >= 0; h--) {
            const int c0 = h * K;
            const bool act = walking && c0 < L;
            if (!__any(act)) continue;
            if (act) {
                const uint64_t *cp = CP + (int64_t)h * CPW * 64;
#pragma unroll
                for (int w = 0; w < WB; w++) { VP[w] = cp[w * 64]; VN[w] = cp[(WB + w) * 64]; }
#pragma unroll
                for (int k = 0; k < NP; k++)
#pragma unroll
                    for (int w = 0; w < WB; w++) PL[k][w] = cp[((2 + k) * WB + w) * 64];
            }
            // next group in processing order: the rest of this chunk, then the chunk before it
#pragma unroll
            for (int k = 0; k < K; k++) column(c0 + k, k, true, c0 - K);
            if (act) {
                while (r >= 0 && c >= c0) {
                    const int b = r - c + TOP - off;
                    const ulonglong2 e = col[((c - c0) * WB + (b >> 6)) * 64];
                    const uint64_t bit = ((uint64_t)1) << (b & 63);
                    if (e.x & bit) { ins++; r--; }
                    else if (e.y & bit) c--;
                    else { r--; c--; }
                }
                if (r < 0) walking = false; // query used up: only deletions remain
            }
        }
        if (valid) {
            if (r >= 0) ins += r + 1; // target used up: the remaining query bases are insertions
            a.first_ins[p] = ins;
        }
    }
}

// The wide classes with G lanes per hit: the window's 64*WB rows are split over G neighbouring lanes (WBL words
// each, lane g of a hit holds words g*WBL ..), so a column costs a lane 1/G of the instructions, the critical path
// of a launch - its longest hit, twice - shrinks accordingly, and bands far wider than a lane's registers hold
// become possible (16 words: distances up to 1014, the long reads contained in long exons that otherwise take the
// full-height kernels with their whole-matrix trace).  What crosses lanes per column: the carry of the one addition
// (each lane adds with carry-in 0 and reports carry-out and "all ones"; a lane's carry-in follows from the two
// wave-wide ballots: the nearest lower lane of the group that generates or stops a carry decides), and the bits that
// shift in from the lane above (D0, Ph, the bit-planes).  All lanes of a hit walk the same path (the stored columns
// are in LDS, any lane reads any lane's words); lane 0 of the group reports.  Cross-lane reads are never executed
// under a lane-dependent condition: the source lane would be inactive.
template <int G, int WBL, int NP>
__global__ void __launch_bounds__(64) myers_bandg(MyArgs a, const int32_t *items, int64_t n_items)
{
    constexpr int WB = G * WBL;
    constexpr int TOP = 32 * WB - 1;
    constexpr int K = 8;
    constexpr int HITS = 64 / G;                // hits per wavefront (G = 3: lane 63 idles)
    constexpr int CPW = (2 + NP) * WBL;         // 64-bit words per checkpoint and lane
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const int g = lane % G, h = lane / G;
    const uint64_t below_me = (lane == 0 ? 0ull : (~0ull >> (64 - lane))) & ~((lane - g) == 0 ? 0ull : (~0ull >> (64 - (lane - g)))); // lanes of my hit below me
    const int64_t gwave = blockIdx.x, nwaves = gridDim.x;
    uint64_t *CP = (uint64_t *)a.tr + gwave * a.region + lane; // checkpoint (chunk c, word f) at CP[(c * CPW + f) * 64]
    extern __shared__ ulonglong2 band_lds[];
    ulonglong2 *col = band_lds; // (column k of the chunk, word w of lane l) at col[(k * WBL + w) * 64 + l]
    for (int64_t base = gwave * HITS; base < n_items; base += nwaves * HITS) {
        bool valid = h < HITS && base + h < n_items;
        int p = 0, m = 0, L = 0, ed = 0;
        const uint8_t *q = a.seq, *t = a.seq;
        if (valid) {
            p = items[base + h];
            m = a.q_len[p];
            ed = a.ed[p];
            const int64_t lo = a.loc_off[p];
            const int s0 = a.loc_start[lo];
            L = a.loc_end[lo] - s0 + 1;
            q = a.seq + a.q_off[p];
            t = a.seq + a.t_off[p] + s0;
            if (tb_by_arithmetic(ed, m, L)) { valid = false; p = m = L = ed = 0; q = t = a.seq; } // answered by k_band_keys
        }
        const int delta = m - L;
        const int hh = (ed - (delta < 0 ? -delta : delta)) >> 1;
        const int off = ((delta < 0 ? delta : 0) - hh - 1 + TOP) & ~7; // see myers_band
        uint64_t PL[NP][WBL], VP[WBL], VN[WBL];
        uint64_t qw = 0, tw = 0, qwn = 0, twn = 0;
        int pref_c = -0x40000000;
        auto qblock = [&](int r) { return (r >= 0 && r < m) ? load8(q + r) : (uint64_t)0; };
        auto prefetch = [&](int cn) {
            pref_c = cn;
            twn = (cn >= 0 && cn < L) ? load8(t + cn) : 0;
            qwn = qblock(TOP + 1 + off + cn);
        };
        // value of lane + 1 as a DPP move (a VALU instruction, not a trip through the LDS crossbar): groups of 2, 4 and 8
        // lanes never straddle a row of 16, groups of 3 do
        auto from_next_lane = [&](int v) {
            return (G == 3 || G == 6 || G == 12 || G == 32) ? __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false)
                          : __builtin_amdgcn_update_dpp(0, v, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
        };
        auto push_row = [&](uint32_t code1) { // the whole window moves down one row; the top lane takes the new row's code
            uint32_t low = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) low |= ((uint32_t)PL[k][0] & 1u) << k;
            const uint32_t above = (uint32_t)from_next_lane((int)low); // bit 0 of the next lane's planes, before they move
            const uint32_t in = g == G - 1 ? code1 : above;
#pragma unroll
            for (int k = 0; k < NP; k++) {
#pragma unroll
                for (int w = 0; w < WBL - 1; w++) PL[k][w] = shr1_in(PL[k][w], (uint32_t)PL[k][w + 1]);
                PL[k][WBL - 1] = shr1_in(PL[k][WBL - 1], in >> k);
            }
        };
        auto column = [&](int c, int k, bool store, int next_c) {
            if (k == 0) {
                if (pref_c != c) prefetch(c);
                tw = twn; qw = qwn;
                prefetch(next_c);
            }
            const uint32_t tc1 = (uint32_t)(tw & 0xff) + 1u;
            tw >>= 8;
            uint64_t Eq[WBL];
            {
                uint32_t mk[NP];
#pragma unroll
                for (int kk = 0; kk < NP; kk++) mk[kk] = ((tc1 >> kk) & 1u) - 1u;
#pragma unroll
                for (int w = 0; w < WBL; w++) {
                    uint32_t el = (uint32_t)PL[0][w] ^ mk[0], eh = (uint32_t)(PL[0][w] >> 32) ^ mk[0];
#pragma unroll
                    for (int kk = 1; kk < NP; kk++) {
                        el = bop3<0x60>(el, (uint32_t)PL[kk][w], mk[kk]);
                        eh = bop3<0x60>(eh, (uint32_t)(PL[kk][w] >> 32), mk[kk]);
                    }
                    Eq[w] = pack64(el, eh);
                }
            }
            // (Eq & VP) + VP over the whole window: first with carry-in 0 per lane ...
            uint64_t SUM[WBL];
            uint64_t carry = 0, ones = ~(uint64_t)0;
#pragma unroll
            for (int w = 0; w < WBL; w++) {
                const uint64_t x = Eq[w] & VP[w];
                const uint64_t s1 = x + VP[w];
                const uint64_t s2 = s1 + carry;
                carry = (uint64_t)((s1 < x) | (s2 < s1));
                SUM[w] = s2;
                ones &= s2;
            }
            // ... then the carry-in: the nearest lower lane of the hit that generates a carry (1) or stops one (0)
            {
                const uint64_t gen = __ballot(carry != 0);
                const uint64_t pass = __ballot(ones == ~(uint64_t)0);
                const uint64_t decides = (gen | ~pass) & below_me;
                uint64_t cin = 0;
                if (decides) cin = (gen >> (63 - __clzll((long long)decides))) & 1ull;
#pragma unroll
                for (int w = 0; w < WBL; w++) {
                    const uint64_t sx = SUM[w] + cin;
                    cin = (uint64_t)(sx < cin);
                    SUM[w] = sx;
                }
            }
            uint64_t D0[WBL], HP[WBL], HN[WBL];
#pragma unroll
            for (int w = 0; w < WBL; w++) {
                D0[w] = bop3_64<0xBE>(SUM[w], VP[w], Eq[w]) | VN[w];
                HP[w] = bop3_64<0xF1>(VN[w], D0[w], VP[w]);
                HN[w] = D0[w] & VP[w];
            }
            // bit 0 of D0 and HP of the lane above shift into this lane's top word (nothing above the window)
            const int lowbits = (int)(((uint32_t)D0[0] & 1u) | (((uint32_t)HP[0] & 1u) << 1));
            const int from_above = from_next_lane(lowbits);
            const int up = g == G - 1 ? 0 : from_above;
#pragma unroll
            for (int w = 0; w < WBL; w++) {
                const uint64_t X = shr1_in(D0[w], w + 1 < WBL ? (uint32_t)D0[w + 1] : (uint32_t)(up & 1));
                VN[w] = X & HP[w];
                VP[w] = bop3_64<0xF1>(HN[w], X, HP[w]);
                if (store) {
                    ulonglong2 e; e.x = VP[w];
                    e.y = shr1_in(HP[w], w + 1 < WBL ? (uint32_t)HP[w + 1] : (uint32_t)((up >> 1) & 1));
                    col[(k * WBL + w) * 64 + lane] = e;
                }
            }
            {
                const int r = TOP + 1 + off + c;
                push_row((r >= 0 && r < m) ? (uint32_t)(qw & 0xff) + 1u : 0u);
                qw >>= 8;
            }
        };
#pragma unroll
        for (int k = 0; k < NP; k++)
#pragma unroll
            for (int w = 0; w < WBL; w++) PL[k][w] = 0;
        {
            // every lane of the wavefront takes part in push_row's exchange the same number of times: the rows a
            // lane pushes beyond its own are virtual (above row 0) and fall out of the window again
            const int r_end = TOP + 1 + off, r0 = off - TOP - 1;
            const int mine = r0 < 0 ? (r_end < 0 ? r_end : 0) : r0;
            const int steps = wave_max_i32(r_end - mine);
            int ri = r_end - steps; // a multiple of 8, never below the window's first row
            uint64_t cur = 0, nxt = qblock(ri);
            for (int sidx = 0; sidx < steps; sidx++) {
                if ((ri & 7) == 0) { cur = nxt; nxt = qblock(ri + 8); }
                push_row((ri >= 0 && ri < m) ? (uint32_t)(cur & 0xff) + 1u : 0u);
                cur >>= 8;
                ri++;
            }
        }
#pragma unroll
        for (int w = 0; w < WBL; w++) {
            const int z = TOP + 1 - off - 64 * (g * WBL + w);
            VP[w] = z <= 0 ? ~(uint64_t)0 : z >= 64 ? 0 : (~(uint64_t)0) << z;
            VN[w] = 0;
        }
        const int Lmax = wave_max_i32(L);
        const int nchunk = (Lmax + K - 1) / K;
        for (int ch = 0; ch < nchunk; ch++) {
            if (ch * K < L) {
                uint64_t *cp = CP + (int64_t)ch * CPW * 64;
#pragma unroll
                for (int w = 0; w < WBL; w++) { cp[w * 64] = VP[w]; cp[(WBL + w) * 64] = VN[w]; }
#pragma unroll
                for (int k = 0; k < NP; k++)
#pragma unroll
                    for (int w = 0; w < WBL; w++) cp[((2 + k) * WBL + w) * 64] = PL[k][w];
            }
            if (ch + 1 < nchunk) {
#pragma unroll
                for (int k = 0; k < K; k++) column(ch * K + k, k, false, ch * K + K);
            }
        }
        int r = m - 1, cc = L - 1, ins = 0;
        bool walking = valid;
        for (int ch = nchunk - 1; ch >= 0; ch--) {
            const int c0 = ch * K;
            const bool act = walking && c0 < L;
            if (!__any(act)) continue;
            if (act) {
                const uint64_t *cp = CP + (int64_t)ch * CPW * 64;
#pragma unroll
                for (int w = 0; w < WBL; w++) { VP[w] = cp[w * 64]; VN[w] = cp[(WBL + w) * 64]; }
#pragma unroll
                for (int k = 0; k < NP; k++)
#pragma unroll
                    for (int w = 0; w < WBL; w++) PL[k][w] = cp[((2 + k) * WBL + w) * 64];
            }
#pragma unroll
            for (int k = 0; k < K; k++) column(c0 + k, k, true, c0 - K);
            __syncthreads(); // one wavefront per workgroup: the group's LDS writes before the walk's reads
            if (act) {
                while (r >= 0 && cc >= c0) {
                    const int b = r - cc + TOP - off;
                    const int gw = b >> 6; // word of the window, over the group's lanes
                    const ulonglong2 e = col[((cc - c0) * WBL + gw % WBL) * 64 + (lane - g + gw / WBL)];
                    const uint64_t bit = ((uint64_t)1) << (b & 63);
                    if (e.x & bit) { ins++; r--; }
                    else if (e.y & bit) cc--;
                    else { r--; cc--; }
                }
                if (r < 0) walking = false;
            }
            __syncthreads();
        }
        if (valid && g == 0) {
            if (r >= 0) ins += r + 1;
            a.first_ins[p] = ins;
        }
    }
}

// The same kernel on a 32-row window (bit b <-> query row c - 15 + off + b), for hits with ed <= 22: every
// quantity is one 32-bit register, a checkpoint is 2 + NP dwords, a stored column 8 bytes per lane, so twice as
// many wavefronts fit beside each other (the traceback launches are bound by latency, not by arithmetic).
template <int NP>
__global__ void __launch_bounds__(64) myers_band32(MyArgs a, const int32_t *items, int64_t n_items)
{
    constexpr int TOP = 15;
    constexpr int K = band_chunk(0);
    constexpr int CPW = 2 + NP; // dwords per checkpoint and lane
    const int lane = threadIdx.x;
    const int64_t gwave = blockIdx.x, nwaves = gridDim.x;
    uint32_t *CP = (uint32_t *)((uint64_t *)a.tr + gwave * a.region) + lane; // checkpoint (chunk h, dword f) at CP[(h * CPW + f) * 64]
    extern __shared__ uint2 band32_lds[];
    uint2 *col = band32_lds + lane; // column k of the chunk at col[k * 64]
    for (int64_t base = gwave * 64; base < n_items; base += nwaves * 64) {
        bool valid = base + lane < n_items;
        int p = 0, m = 0, L = 0, ed = 0;
        const uint8_t *q = a.seq, *t = a.seq;
        if (valid) {
            p = items[base + lane];
            m = a.q_len[p];
            ed = a.ed[p];
            const int64_t lo = a.loc_off[p];
            const int s0 = a.loc_start[lo];
            L = a.loc_end[lo] - s0 + 1;
            q = a.seq + a.q_off[p];
            t = a.seq + a.t_off[p] + s0;
            if (tb_by_arithmetic(ed, m, L)) { valid = false; p = m = L = ed = 0; q = t = a.seq; } // answered by k_band_keys
        }
        const int delta = m - L;
        const int hh = (ed - (delta < 0 ? -delta : delta)) >> 1;
        const int off = ((delta < 0 ? delta : 0) - hh - 1 + TOP) & ~7; // see myers_band
        uint32_t PL[NP], VP, VN = 0;
        // Target and query bytes travel one chunk of K columns at a time (K / 8 loads of 8 bytes, issued together a
        // whole chunk before their first use): a lane's single 8-byte loads each pulled a 64-byte line that had left
        // the L2 again by the time the lane came back for the next 8 bytes (profiles/r01: 4.3 GB fetched per launch
        // against 0.35 GB written), and the group of eight columns between a load and its use was shorter than the
        // load's latency.
        constexpr int NB = K / 8;
        uint64_t qw = 0, tw = 0, qwb[NB], twb[NB], qwn[NB], twn[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) { qwb[i] = twb[i] = qwn[i] = twn[i] = 0; }
        int pref_c = -0x40000000;
        auto qblock = [&](int r) { return (r >= 0 && r < m) ? load8(q + r) : (uint64_t)0; };
        auto prefetch = [&](int cn) {
            pref_c = cn;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                twn[i] = (cn + 8 * i >= 0 && cn + 8 * i < L) ? load8(t + cn + 8 * i) : 0;
                qwn[i] = qblock(TOP + 1 + off + cn + 8 * i);
            }
        };
        auto push_row = [&](uint32_t code1) {
#pragma unroll
            for (int k = 0; k < NP; k++) PL[k] = __builtin_amdgcn_alignbit(code1 >> k, PL[k], 1);
        };
        auto column = [&](int c, int k, bool store, int next_c) {
            if (k == 0) { // k is a compile-time constant in the unrolled chunk loops
                if (pref_c != c) prefetch(c);
#pragma unroll
                for (int i = 0; i < NB; i++) { twb[i] = twn[i]; qwb[i] = qwn[i]; }
                prefetch(next_c);
            }
            if ((k & 7) == 0) { tw = twb[k >> 3]; qw = qwb[k >> 3]; }
            const uint32_t tc1 = (uint32_t)(tw & 0xff) + 1u;
            tw >>= 8;
            uint32_t Eq = PL[0] ^ (((tc1 >> 0) & 1u) - 1u);
#pragma unroll
            for (int kk = 1; kk < NP; kk++) Eq = bop3<0x60>(Eq, PL[kk], ((tc1 >> kk) & 1u) - 1u);
            const uint32_t sum = (Eq & VP) + VP;
            const uint32_t D0 = bop3<0xBE>(sum, VP, Eq) | VN;
            const uint32_t HP = bop3<0xF1>(VN, D0, VP);
            const uint32_t HN = D0 & VP;
            const uint32_t X = D0 >> 1;
            VN = X & HP;
            VP = bop3<0xF1>(HN, X, HP);
            if (store) col[k * 64] = make_uint2(VP, HP >> 1);
            const int r = TOP + 1 + off + c;
            push_row((r >= 0 && r < m) ? (uint32_t)(qw & 0xff) + 1u : 0u);
            qw >>= 8;
        };
#pragma unroll
        for (int k = 0; k < NP; k++) PL[k] = 0;
        {
            const int r_end = TOP + 1 + off, r0 = off - TOP - 1;
            int ri = r0 < 0 ? (r_end < 0 ? r_end : 0) : r0;
            uint64_t cur = 0, nxt = qblock(ri);
            while (ri < r_end) {
                if ((ri & 7) == 0) { cur = nxt; nxt = qblock(ri + 8); }
                push_row((ri >= 0 && ri < m) ? (uint32_t)(cur & 0xff) + 1u : 0u);
                cur >>= 8;
                ri++;
            }
        }
        {
            const int z = TOP + 1 - off; // bit b <-> row -1 - TOP + off + b >= 0
            VP = z <= 0 ? ~0u : z >= 32 ? 0u : (~0u) << z;
        }
        const int Lmax = wave_max_i32(L);
        const int nchunk = (Lmax + K - 1) / K;
        for (int h = 0; h < nchunk; h++) {
            if (h * K < L) {
                uint32_t *cp = CP + (int64_t)h * CPW * 64;
                cp[0] = VP; cp[64] = VN;
#pragma unroll
                for (int k = 0; k < NP; k++) cp[(2 + k) * 64] = PL[k];
            }
            if (h + 1 < nchunk)
#pragma unroll
                for (int k = 0; k < K; k++) column(h * K + k, k, false, h * K + K);
        }
        int r = m - 1, c = L - 1, ins = 0;
        bool walking = valid;
        // The checkpoint of the chunk below is fetched while this chunk is recomputed: taken when the chunk begins, it
        // held the wavefront for a trip to HBM per chunk with nothing else to issue (two to three wavefronts per SIMD).
        uint32_t nVP = 0, nVN = 0, nPL[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) nPL[k] = 0;
        int staged = -1; // chunk whose checkpoint nVP / nVN / nPL hold (of the lanes that take part in it)
        auto stage_checkpoint = [&](int h) {
            staged = h;
            if (h >= 0 && walking && h * K < L) {
                const uint32_t *cp = CP + (int64_t)h * CPW * 64;
                nVP = cp[0]; nVN = cp[64];
#pragma unroll
                for (int k = 0; k < NP; k++) nPL[k] = cp[(2 + k) * 64];
            }
        };
        for (int h = nchunk - 1; h >= 0; h--) {
            const int c0 = h * K;
            const bool act = walking && c0 < L;
            if (!__any(act)) continue;
            if (staged != h) stage_checkpoint(h);
            if (act) {
                VP = nVP; VN = nVN;
#pragma unroll
                for (int k = 0; k < NP; k++) PL[k] = nPL[k];
            }
            stage_checkpoint(h - 1);
#pragma unroll
            for (int k = 0; k < K; k++) column(c0 + k, k, true, c0 - K);
            if (act) {
#if ULTRA_WALK_BY_COLUMN
                // The walk visits the columns of the chunk one after the other (a move up stays in its column, the two others
                // go one to the left), so it is written over the columns, not over the moves: the K stored columns are read
                // in a fixed order - loads that do not depend on the walk and are issued ahead of it - and a lane acts on
                // column k when that is where it stands.  A run of moves up is the run of set Pv bits below the current
                // one: counted, not stepped.  (Move by move, every step waited for an LDS read whose address the step
                // before had produced: about half of the kernel's time at two to three wavefronts per SIMD.)
#pragma unroll
                for (int k = K - 1; k >= 0; k--) {
                    const uint2 e = col[k * 64];
                    if (r >= 0 && c == c0 + k) {
                        const int b = (r - c + TOP - off) & 31;
                        int ups = __clz((int)~(e.x << (31 - b))); // set bits b, b - 1, ...: that many insertions
                        ups = ups < r + 1 ? ups : r + 1;
                        ins += ups; r -= ups;
                        if (r >= 0) {
                            if (!(e.y & (1u << ((b - ups) & 31)))) r--; // not a deletion: diagonal
                            c--;
                        }
                    }
                }
#else
                while (r >= 0 && c >= c0) {
                    const int b = r - c + TOP - off;
                    const uint2 e = col[(c - c0) * 64];
                    const uint32_t bit = 1u << (b & 31);
                    if (e.x & bit) { ins++; r--; }
                    else if (e.y & bit) c--;
                    else { r--; c--; }
                }
#endif
                if (r < 0) walking = false;
            }
        }
        if (valid) {
            if (r >= 0) ins += r + 1;
            a.first_ins[p] = ins;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// banded start pass: lane-per-location, reversed strings, the same sliding window
// ------------------------------------------------------------------------------------------------
// edlib finds the start of a location by aligning the reversed query against the reversed target prefix that ends
// at the location's end column and taking the LAST column whose bottom-row score equals the known distance
// (edlib.cpp, obtainAlignment -> SHW on the reversed sequences; the full-height version of this is
// myers_mega<MODE_REV>).  The distance `best` is known here, so only cells within |row - column| <= best matter:
// 2 best + 1 diagonals.  The window slides exactly as in the traceback kernels; nothing is stored.  The bottom-row
// score is tracked without ever holding a column of scores: S = D[c][c] follows diagonal 0 (one D0 bit per
// column); at the first column that can hold an end (c0 = m - 1 - best) the vertical deltas between diagonal 0 and
// row m - 1 are summed once (population counts), and from then on D[m-1][c] follows the horizontal delta of that row.
// Values outside the band are over-estimates, never under-estimates: the test D <= best is exact.
// Needs 2 best <= rows - 10; the locations beyond go through myers_mega<MODE_REV> (rev_items).
// Only the narrow classes (32, 64, 128 rows: distance <= 59): a lane that carries 4 or 6 words through every
// column runs longer than the lane groups of the full-height kernels, which split a column over 4 - 64 lanes
// (measured: 3.0 ms for the 384-row class against 1.9 ms for the whole m = 513..2048 bucket at full height).
constexpr int kStartBandClasses = 3;
__device__ __forceinline__ int band_start_class(int best) // 0..2, kBandFallback = none
{
    int c = 0;
    while (c < kStartBandClasses && 2 * best > band_rows(c) - 10) c++;
    return c < kStartBandClasses ? c : kBandFallback;
}

template <int NP>
__global__ void __launch_bounds__(64) myers_bstart32(MyArgs a, const int32_t *items, int64_t n_items)
{
    constexpr int TOP = 15;
    const int lane = threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n_items; base += (int64_t)gridDim.x * 64) {
        const bool valid = base + lane < n_items;
        int loc = 0, m = 1, L = 0, best = 0, e = 0;
        const uint8_t *q = a.rseq, *t = a.rseq;
        if (valid) {
            loc = items[base + lane];
            const int p = a.loc_pair[loc];
            e = a.loc_end[loc];
            m = a.q_len[p]; best = a.ed[p];
            L = m + best < e + 1 ? m + best : e + 1;
            q = a.rseq + (a.N - a.q_off[p] - m);
            t = a.rseq + (a.N - a.t_off[p] - (int64_t)(e + 1));
        }
        const int off = (-best - 1 + TOP) & ~7;
        const int bt = TOP - off;                 // window bit of diagonal 0 (in the alignment of the updated column)
        uint32_t PL[NP], VP, VN = 0;
        uint64_t qw = 0, tw = 0, qwn = 0, twn = 0;
        auto qblock = [&](int r) { return (r >= 0 && r < m) ? load8(q + r) : (uint64_t)0; };
        auto prefetch = [&](int cn) {
            twn = (cn >= 0 && cn < L) ? load8(t + cn) : 0;
            qwn = qblock(TOP + 1 + off + cn);
        };
        auto push_row = [&](uint32_t code1) {
#pragma unroll
            for (int k = 0; k < NP; k++) PL[k] = __builtin_amdgcn_alignbit(code1 >> k, PL[k], 1);
        };
#pragma unroll
        for (int k = 0; k < NP; k++) PL[k] = 0;
        {
            const int r_end = TOP + 1 + off, r0 = off - TOP - 1;
            int ri = r0 < 0 ? (r_end < 0 ? r_end : 0) : r0;
            uint64_t cur = 0, nxt = qblock(ri);
            while (ri < r_end) {
                if ((ri & 7) == 0) { cur = nxt; nxt = qblock(ri + 8); }
                push_row((ri >= 0 && ri < m) ? (uint32_t)(cur & 0xff) + 1u : 0u);
                cur >>= 8;
                ri++;
            }
        }
        {
            const int z = TOP + 1 - off; // bit b <-> row -1 - TOP + off + b >= 0
            VP = z <= 0 ? ~0u : z >= 32 ? 0u : (~0u) << z;
        }
        const int c0 = m - 1 - best > 0 ? m - 1 - best : 0;
        int S = 0, Dm = 0, last = 0;
        const int Lmax = wave_max_i32(L);
        prefetch(0);
        for (int c = 0; c < Lmax; c++) {
            if ((c & 7) == 0) { tw = twn; qw = qwn; prefetch(c + 8); }
            const uint32_t tc1 = (uint32_t)(tw & 0xff) + 1u;
            tw >>= 8;
            uint32_t Eq = PL[0] ^ (((tc1 >> 0) & 1u) - 1u);
#pragma unroll
            for (int kk = 1; kk < NP; kk++) Eq = bop3<0x60>(Eq, PL[kk], ((tc1 >> kk) & 1u) - 1u);
            const uint32_t sum = (Eq & VP) + VP;
            const uint32_t D0 = bop3<0xBE>(sum, VP, Eq) | VN;
            const uint32_t HP = bop3<0xF1>(VN, D0, VP);
            const uint32_t HN = D0 & VP;
            const uint32_t X = D0 >> 1;
            VN = X & HP;
            VP = bop3<0xF1>(HN, X, HP);
            // D0 / HP / HN sit one bit higher than the updated column's alignment
            S += 1 - (int)((D0 >> ((bt + 1) & 31)) & 1u);
            const int bm = bt + (m - 1 - c);     // window bit of row m - 1 in this column
            if (c == c0) {
                const uint32_t mask = (bm >= 31 ? ~0u : ((2u << (bm & 31)) - 1u)) & ~((2u << (bt & 31)) - 1u); // bits bt + 1 .. bm
                Dm = S + __popc(VP & mask) - __popc(VN & mask);
            } else if (c > c0) {
                Dm += (int)((HP >> ((bm + 1) & 31)) & 1u) - (int)((HN >> ((bm + 1) & 31)) & 1u);
            }
            if (c >= c0 && c < L && Dm <= best) last = c;
            const int r = TOP + 1 + off + c;
            push_row((r >= 0 && r < m) ? (uint32_t)(qw & 0xff) + 1u : 0u);
            qw >>= 8;
        }
        if (valid) a.loc_start[loc] = e - last;
    }
}

template <int WB, int NP>
__global__ void __launch_bounds__(64) myers_bstart(MyArgs a, const int32_t *items, int64_t n_items)
{
    constexpr int TOP = 32 * WB - 1;
    const int lane = threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n_items; base += (int64_t)gridDim.x * 64) {
        const bool valid = base + lane < n_items;
        int loc = 0, m = 1, L = 0, best = 0, e = 0;
        const uint8_t *q = a.rseq, *t = a.rseq;
        if (valid) {
            loc = items[base + lane];
            const int p = a.loc_pair[loc];
            e = a.loc_end[loc];
            m = a.q_len[p]; best = a.ed[p];
            L = m + best < e + 1 ? m + best : e + 1;
            q = a.rseq + (a.N - a.q_off[p] - m);
            t = a.rseq + (a.N - a.t_off[p] - (int64_t)(e + 1));
        }
        const int off = (-best - 1 + TOP) & ~7;
        const int bt = TOP - off;
        uint64_t PL[NP][WB], VP[WB], VN[WB];
        uint64_t qw = 0, tw = 0, qwn = 0, twn = 0;
        auto qblock = [&](int r) { return (r >= 0 && r < m) ? load8(q + r) : (uint64_t)0; };
        auto prefetch = [&](int cn) {
            twn = (cn >= 0 && cn < L) ? load8(t + cn) : 0;
            qwn = qblock(TOP + 1 + off + cn);
        };
        auto push_row = [&](uint32_t code1) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
#pragma unroll
                for (int w = 0; w < WB - 1; w++) PL[k][w] = shr1_in(PL[k][w], (uint32_t)PL[k][w + 1]);
                PL[k][WB - 1] = shr1_in(PL[k][WB - 1], code1 >> k);
            }
        };
        // bit i of a window vector (i < 64 WB), without indexing registers dynamically
        auto bit_of = [&](const uint64_t (&v)[WB], int i) {
            uint64_t wsel = v[0];
#pragma unroll
            for (int w = 1; w < WB; w++) wsel = (i >> 6) == w ? v[w] : wsel;
            return (int)((wsel >> (i & 63)) & 1);
        };
#pragma unroll
        for (int k = 0; k < NP; k++)
#pragma unroll
            for (int w = 0; w < WB; w++) PL[k][w] = 0;
        {
            const int r_end = TOP + 1 + off, r0 = off - TOP - 1;
            int ri = r0 < 0 ? (r_end < 0 ? r_end : 0) : r0;
            uint64_t cur = 0, nxt = qblock(ri);
            while (ri < r_end) {
                if ((ri & 7) == 0) { cur = nxt; nxt = qblock(ri + 8); }
                push_row((ri >= 0 && ri < m) ? (uint32_t)(cur & 0xff) + 1u : 0u);
                cur >>= 8;
                ri++;
            }
        }
#pragma unroll
        for (int w = 0; w < WB; w++) {
            const int z = TOP + 1 - off - 64 * w;
            VP[w] = z <= 0 ? ~(uint64_t)0 : z >= 64 ? 0 : (~(uint64_t)0) << z;
            VN[w] = 0;
        }
        const int c0 = m - 1 - best > 0 ? m - 1 - best : 0;
        int S = 0, Dm = 0, last = 0;
        const int Lmax = wave_max_i32(L);
        prefetch(0);
        for (int c = 0; c < Lmax; c++) {
            if ((c & 7) == 0) { tw = twn; qw = qwn; prefetch(c + 8); }
            const uint32_t tc1 = (uint32_t)(tw & 0xff) + 1u;
            tw >>= 8;
            uint64_t Eq[WB];
            {
                uint32_t mk[NP];
#pragma unroll
                for (int kk = 0; kk < NP; kk++) mk[kk] = ((tc1 >> kk) & 1u) - 1u;
#pragma unroll
                for (int w = 0; w < WB; w++) {
                    uint32_t el = (uint32_t)PL[0][w] ^ mk[0], eh = (uint32_t)(PL[0][w] >> 32) ^ mk[0];
#pragma unroll
                    for (int kk = 1; kk < NP; kk++) {
                        el = bop3<0x60>(el, (uint32_t)PL[kk][w], mk[kk]);
                        eh = bop3<0x60>(eh, (uint32_t)(PL[kk][w] >> 32), mk[kk]);
                    }
                    Eq[w] = pack64(el, eh);
                }
            }
            uint64_t D0[WB], HP[WB], HN[WB];
            uint64_t carry = 0;
#pragma unroll
            for (int w = 0; w < WB; w++) {
                const uint64_t x = Eq[w] & VP[w];
                const uint64_t s1 = x + VP[w];
                const uint64_t s2 = s1 + carry;
                carry = (uint64_t)((s1 < x) | (s2 < s1));
                D0[w] = bop3_64<0xBE>(s2, VP[w], Eq[w]) | VN[w];
                HP[w] = bop3_64<0xF1>(VN[w], D0[w], VP[w]);
                HN[w] = D0[w] & VP[w];
            }
#pragma unroll
            for (int w = 0; w < WB; w++) {
                const uint64_t X = shr1_in(D0[w], w + 1 < WB ? (uint32_t)D0[w + 1] : 0u);
                VN[w] = X & HP[w];
                VP[w] = bop3_64<0xF1>(HN[w], X, HP[w]);
            }
            S += 1 - bit_of(D0, bt + 1);
            const int bm = bt + (m - 1 - c);
            if (c == c0) {
                int acc = 0;
#pragma unroll
                for (int w = 0; w < WB; w++) { // bits bt + 1 .. bm of the window
                    const int lo = bt + 1 - 64 * w, hi = bm - 64 * w; // inclusive, relative to this word
                    if (hi < 0 || lo > 63) continue;
                    const uint64_t above = lo <= 0 ? ~(uint64_t)0 : (~(uint64_t)0) << lo;
                    const uint64_t below = hi >= 63 ? ~(uint64_t)0 : (((uint64_t)2 << hi) - 1);
                    const uint64_t mask = above & below;
                    acc += __popcll(VP[w] & mask) - __popcll(VN[w] & mask);
                }
                Dm = S + acc;
            } else if (c > c0) {
                Dm += bit_of(HP, bm + 1) - bit_of(HN, bm + 1);
            }
            if (c >= c0 && c < L && Dm <= best) last = c;
            const int r = TOP + 1 + off + c;
            push_row((r >= 0 && r < m) ? (uint32_t)(qw & 0xff) + 1u : 0u);
            qw >>= 8;
        }
        if (valid) a.loc_start[loc] = e - last;
    }
}

// Sort key of every location for the start pass: band class by 2 x distance, then the number of columns, longest
// first; locations beyond every band class are keyed by the size class of their query (dense lists for
// myers_mega<MODE_REV>).
__global__ void __launch_bounds__(256) k_start_keys(MyArgs a, int64_t n_loc, uint32_t *keys, int32_t *idx);

// Every hit is routed by its distance (band rows needed: 2*ed + 3) and, inside its band class, sorted by the length
// of its first location, longest first - the lanes of a wavefront then run equally long.  Hits beyond every band
// class are keyed by the size class of their query instead: each full-height kernel then gets a dense list of
// exactly its hits.  (kBandLBits / kFullLBits: key layout of the banded start pass, k_start_keys.)
constexpr uint32_t kBandLBits = 24, kFullLBits = 20;
constexpr int kNumSizeClasses = 15; // 14 = the long class: wavefront per pair, strips of 32768 rows
constexpr uint32_t kSkipClass = 15;  // sort key of the pairs answered without a launch (k_pair_keys)
constexpr int kLongClass = 14;
constexpr int kBandBounds = kBandClasses + kNumSizeClasses + 1; // first sorted position of each group, then the end
__device__ __forceinline__ uint32_t size_class_of(int m)
{
    return m <= 256 ? (uint32_t)((m - 1) >> 6) : m <= 384 ? 4u : m <= 512 ? 5u : m <= 768 ? 6u : m <= 1024 ? 7u : m <= 1536 ? 8u
         : m <= 2048 ? 9u : m <= 4096 ? 10u : m <= 8192 ? 11u : m <= 16384 ? 12u : m <= 32768 ? 13u : 14u;
}
// Traceback routing keys, 14 bits (two radix passes instead of the four a 28-bit key with the exact length took):
// band class (by distance), then a bin of the first location's length, longest first - exact below 256 columns,
// 32 bins per octave above (the lanes of a wavefront then differ by 3 % at most).  Hits beyond every band class:
// size class of the query, then 32 coarse bins.
constexpr int kTbBins = 1024, kTbFullBins = 32, kTbKeyBits = 14;
static_assert(kBandClasses * kTbBins + kNumSizeClasses * kTbFullBins <= (1 << kTbKeyBits), "routing key bits");
__device__ __forceinline__ uint32_t tb_length_bin(uint32_t L) // 0 .. 767
{
    if (L < 256) return L;
    const int e = 31 - __builtin_clz(L); // 8 .. 23 (a location is shorter than 2^24 columns)
    const uint32_t b = 256u + (uint32_t)(e - 8) * 32u + ((L >> (e - 5)) & 31u);
    return b < 767u ? b : 767u;
}
// COUNT: what the host needs to lay the launches out - hits per group (info[g]: the band classes, then the size classes
// of the hits no band class takes) and info[kBandBounds + g] = the longest first location of a band class / the largest
// distance of a size class.  It only needs the forward pass (the class follows from the distance), so it runs in front
// of the start pass; the first location's start is not known then, and its length is bounded by m + ed (an alignment
// of m query bases with ed edits) and by the target.  !COUNT: the sort keys, with the exact length, after the start pass.
template <bool COUNT>
__global__ void __launch_bounds__(256) k_band_keys(MyArgs a, int64_t n_hits, uint32_t *keys, int32_t *info)
{
    // the maxima and counts go through LDS: the hits come in the order of the sorted pairs, lengths growing, and a
    // maximum kept with one global atomic per hit would take one almost every time
    __shared__ int32_t top[kBandClasses + kNumSizeClasses], cnt[kBandClasses + kNumSizeClasses];
    if (COUNT) {
        if (threadIdx.x < kBandClasses + kNumSizeClasses) { top[threadIdx.x] = 0; cnt[threadIdx.x] = 0; }
        __syncthreads();
    }
    // COUNT: a few hundred blocks, each over many hits (every block ends with atomics on the same few global counters)
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n_hits; i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    int slot = -1, v = 0;
    if (i < n_hits) {
        const int32_t p = a.hit_pairs[i];
        const int ed = a.ed[p];
        uint32_t L;
        if (COUNT) {
            const int32_t mq = a.q_len[p], tl = a.t_len[p];
            L = (uint32_t)(mq + ed < tl ? mq + ed : tl);
        } else {
            const int64_t lo = a.loc_off[p];
            L = (uint32_t)(a.loc_end[lo] - a.loc_start[lo] + 1);
        }
        // ed + 1 diagonals, one spare on either side, up to 7 lost to the window's 8-row alignment: ed <= 64 WB - 10
        uint32_t c = 0;
        while (c < kBandFallback && ed > band_rows((int)c) - 10) c++;
        if (c < kBandFallback) {
            if (!COUNT) {
                const int mq = a.q_len[p];
                if (tb_by_arithmetic(ed, mq, (int)L)) { // no traceback: behind the class's last real hit (bin of length 0)
                    a.first_ins[p] = mq > (int)L ? mq - (int)L : 0;
                    keys[i] = c * kTbBins + (kTbBins - 1);
                } else keys[i] = c * kTbBins + (kTbBins - 1 - tb_length_bin(L));
            }
            slot = (int)c; v = (int)L;
        } else {
            const uint32_t sc = size_class_of(a.q_len[p]);
            if (!COUNT) keys[i] = kBandClasses * kTbBins + sc * kTbFullBins + (kTbFullBins - 1 - (tb_length_bin(L) >> 5));
            slot = kBandClasses + (int)sc; v = ed;
        }
    }
    if (!COUNT) continue;
    {
        // nearly every lane of a wavefront has the slot of its first lane: one LDS atomic for them all (every lane takes
        // part in the exchange, the ones past the end with no slot), the others on their own
        const int slot0 = __shfl(slot, 0, 64);
        int vm = slot == slot0 ? v : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(vm, o, 64); vm = t > vm ? t : vm; }
        const int same = __popcll(__ballot(slot == slot0 && slot >= 0));
        if ((threadIdx.x & 63) == 0 && slot0 >= 0) { if (vm > top[slot0]) atomicMax(&top[slot0], vm); atomicAdd(&cnt[slot0], same); }
        if (slot >= 0 && slot != slot0) { if (v > top[slot]) atomicMax(&top[slot], v); atomicAdd(&cnt[slot], 1); }
    }
    }
    if (!COUNT) return;
    __syncthreads();
    if (threadIdx.x < kBandClasses + kNumSizeClasses) {
        const int32_t v = top[threadIdx.x];
        int32_t *g = info + kBandBounds + threadIdx.x;
        if (v > *(volatile int32_t *)g) atomicMax(g, v);
        if (cnt[threadIdx.x]) atomicAdd(info + threadIdx.x, cnt[threadIdx.x]);
    }
}
// keys of the banded start pass (k_start_keys below):
// info[g] = first sorted position of group g (the band classes, then the 14 size classes of the
// rest, then the end); info[kBandBounds + c] = longest first location of band class c
__global__ void k_band_bounds(const uint32_t *keys, int64_t n, int32_t *info)
{
    const int g = threadIdx.x;
    if (g >= kBandBounds) return;
    const uint32_t thr = g < kBandClasses ? ((uint32_t)g << kBandLBits)
                       : g < kBandBounds - 1 ? (((uint32_t)kBandFallback << kBandLBits) | ((uint32_t)(g - kBandClasses) << kFullLBits))
                                             : 0xffffffffu;
    int64_t lo = 0, hi = n;
    if (g == kBandBounds - 1) lo = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < thr) lo = mid + 1; else hi = mid;
    }
    info[g] = (int32_t)lo;
    if (g < kBandClasses) {
        int32_t maxl = 0;
        if (lo < n && (keys[lo] >> kBandLBits) == (uint32_t)g) maxl = (int32_t)(((1u << kBandLBits) - 1u) - (keys[lo] & ((1u << kBandLBits) - 1u)));
        info[kBandBounds + g] = maxl;
    }
}

__global__ void __launch_bounds__(256) k_start_keys(MyArgs a, int64_t n_loc, uint32_t *keys, int32_t *idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_loc) return;
    const int32_t p = a.loc_pair[i];
    const int best = a.ed[p], m = a.q_len[p];
    const uint32_t L = (uint32_t)(m + best); // columns of the reverse pass (an end column close to the target's start has fewer)
    const uint32_t c = (uint32_t)band_start_class(best);
    if (c < kBandFallback) keys[i] = (c << kBandLBits) | (((1u << kBandLBits) - 1u) - (L < (1u << kBandLBits) ? L : (1u << kBandLBits) - 1u));
    else keys[i] = (c << kBandLBits) | (size_class_of(m) << kFullLBits) | (((1u << kFullLBits) - 1u) - (L < (1u << kFullLBits) ? L : (1u << kFullLBits) - 1u));
    idx[i] = (int32_t)i;
}

// ------------------------------------------------------------------------------------------------
// small helper kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_reverse(const uint8_t *src, uint8_t *dst, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[n - 1 - i];
}

// packed[i] = (hit << kHitShift) | nloc  so one scan yields both CSR offsets: hits in the high half, locations in the low
// one (location indices are int32 everywhere else; ultra_myers_run_impl refuses a batch with more).  The split was
// 24 | 40 bits at first: a batch with more than 2^24 hits (a million cDNA reads) wrapped its hit offsets and lost
// tracebacks without a word.  overflow[b] is set when a pair of register
// bucket b (pairs [bucket_first[b], bucket_first[b + 1]) of the sorted order) has more end columns than first-pass slots.
constexpr int kHitShift = 32;
struct BucketFirst { int32_t first[6]; };
__global__ void k_pack(const int32_t *ed, const int32_t *nloc, uint64_t *packed, int32_t n, int slot_cap, int64_t *overflow, BucketFirst bf)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        packed[i] = (ed[i] >= 0 ? (((uint64_t)1) << kHitShift) : 0) | (uint64_t)(uint32_t)nloc[i];
        if (nloc[i] > slot_cap) { // rare; every writer stores the same value
            int b = 0;
            while (b < 4 && i >= bf.first[b + 1]) b++;
            overflow[b] = 1;
        }
    }
}

constexpr int SCAN_BLOCK = 1024;
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_block(const uint64_t *in, uint64_t *out, uint64_t *blocksum, int64_t n)
{
    __shared__ uint64_t sh[SCAN_BLOCK];
    int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint64_t v = i < n ? in[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
        uint64_t add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (i < n) out[i] = sh[threadIdx.x] - v; // exclusive
    if (threadIdx.x == SCAN_BLOCK - 1 && blocksum) blocksum[blockIdx.x] = sh[threadIdx.x];
}
__global__ void k_scan_add(uint64_t *out, const uint64_t *blockoff, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (i < n) out[i] += blockoff[blockIdx.x];
}
// splits the scanned packed values; entry n holds the totals
__global__ void k_unpack(const uint64_t *scanned, const uint64_t *packed, int64_t *loc_off, int64_t *hit_off, int32_t n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t LOW = (((uint64_t)1) << kHitShift) - 1;
    if (i < n) {
        loc_off[i] = (int64_t)(scanned[i] & LOW);
        hit_off[i] = (int64_t)(scanned[i] >> kHitShift);
        if (i == n - 1) {
            uint64_t tot = scanned[i] + packed[i];
            loc_off[n] = (int64_t)(tot & LOW);
            hit_off[n] = (int64_t)(tot >> kHitShift);
        }
    }
}
constexpr size_t kRouteBytes = 4096; // MyersState::h_route: routing counts (256 ints), then the pass boundaries
struct BoundPos { int64_t v[kNumSizeClasses + 2]; }; // first pair of every size class present, then the end
__global__ void k_gather_bounds(const int64_t *loc_off, const int64_t *hit_off, BoundPos pos, int64_t *out, int np)
{
    // constant indices: a dynamically indexed by-value argument would go through scratch memory
    int64_t p = 0;
#pragma unroll
    for (int t = 0; t < kNumSizeClasses + 2; t++) if (t == (int)threadIdx.x) p = pos.v[t];
    const int i = threadIdx.x;
    if (i < np) { out[2 * i] = loc_off[p]; out[2 * i + 1] = hit_off[p]; }
}
// CSR fill: end columns kept in the first-pass slots, owner pair of every location, hit list
// loc_cap: entries the location arrays hold (the launch that runs ahead of the host's knowledge of the total)
__global__ void k_fill(const int32_t *ed, const int32_t *nloc, const int32_t *slots, int slot_cap,
                       const int64_t *loc_off, const int64_t *hit_off,
                       int32_t *loc_end, int32_t *loc_pair, int32_t *hit_pairs, int32_t n, int64_t loc_cap)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = nloc[i];
    int64_t o = loc_off[i];
    if (o + c > loc_cap) c = o < loc_cap ? (int)(loc_cap - o) : 0; // the host fills again once the arrays have grown
    const bool in_slots = nloc[i] <= slot_cap;
    for (int a = 0; a < c; a++) {
        loc_pair[o + a] = i;
        if (in_slots) loc_end[o + a] = slots[(int64_t)i * slot_cap + a];
    }
    if (ed[i] >= 0) hit_pairs[hit_off[i]] = i;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct ClassDef { int G, W; };
// rows = 64*G*W ascending
const ClassDef kClasses[] = {
    {1, 1}, {1, 2}, {1, 3}, {1, 4},                   // <= 256 rows: lane per pair
    {2, 3}, {2, 4}, {4, 3}, {4, 4}, {8, 3}, {8, 4},   // 384, 512, 768, 1024, 1536, 2048 rows: 2..8 lanes per pair
    {64, 1}, {64, 2}, {64, 4}, {64, 8},               // 4096 .. 32768 rows: wavefront per pair
    {64, 8},                                          // beyond: the same, in strips of 32768 rows (group_item<.., LONG>)
};
static_assert(sizeof(kClasses) / sizeof(kClasses[0]) == kNumSizeClasses, "size_class_of() and kClasses must agree");
constexpr int kStripRows = 64 * 64 * 8;

template <int W, int MODE> hipError_t launch_lane(hipStream_t stream, const MyArgs &a, int64_t items, int max_waves)
{
    int S = a.S;
    size_t per_wave = (size_t)S * W * 64 * sizeof(uint64_t) + (MODE == MODE_NW ? (size_t)kStage * 2 * 64 * sizeof(ulonglong2) : 0);
    int wpb = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(64 * 1024) / per_wave));
    size_t shmem = per_wave * wpb;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)myers_lane<W, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    int64_t waves = (items + 63) / 64;
    if (max_waves > 0 && waves > max_waves) waves = max_waves;
    int64_t blocks = (waves + wpb - 1) / wpb;
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL((myers_lane<W, MODE>), dim3((unsigned)blocks), dim3(64 * wpb), shmem, stream, a);
    return hipGetLastError();
}

template <int G, int W, int MODE, bool LONG = false> hipError_t launch_group(hipStream_t stream, const MyArgs &a, int64_t items, int max_blocks)
{
    constexpr int P = 64 / G;
    size_t shmem = (size_t)P * a.S * G * W * sizeof(uint64_t) + (MODE == MODE_NW ? (size_t)kStage * 2 * 64 * sizeof(ulonglong2) : 0);
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)myers_group<G, W, MODE, LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    int64_t blocks = (items + P - 1) / P;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL((myers_group<G, W, MODE, LONG>), dim3((unsigned)blocks), dim3(64), shmem, stream, a);
    return hipGetLastError();
}

template <int MODE> hipError_t launch_class(hipStream_t ctx, int cls, const MyArgs &a, int64_t items, int max_par)
{
    switch (cls) {
    case 0: return launch_lane<1, MODE>(ctx, a, items, max_par);
    case 1: return launch_lane<2, MODE>(ctx, a, items, max_par);
    case 2: return launch_lane<3, MODE>(ctx, a, items, max_par);
    case 3: return launch_lane<4, MODE>(ctx, a, items, max_par);
    case 4: return launch_group<2, 3, MODE>(ctx, a, items, max_par);
    case 5: return launch_group<2, 4, MODE>(ctx, a, items, max_par);
    case 6: return launch_group<4, 3, MODE>(ctx, a, items, max_par);
    case 7: return launch_group<4, 4, MODE>(ctx, a, items, max_par);
    case 8: return launch_group<8, 3, MODE>(ctx, a, items, max_par);
    case 9: return launch_group<8, 4, MODE>(ctx, a, items, max_par);
    case 10: return launch_group<64, 1, MODE>(ctx, a, items, max_par);
    case 11: return launch_group<64, 2, MODE>(ctx, a, items, max_par);
    case 12: return launch_group<64, 4, MODE>(ctx, a, items, max_par);
    case 13: return launch_group<64, 8, MODE>(ctx, a, items, max_par);
    case 14: return launch_group<64, 8, MODE, true>(ctx, a, items, max_par);
    }
    return hipErrorInvalidValue;
}

constexpr int kBuckets = 5;
inline int bucket_of(int cls) { return cls < 2 ? 0 : cls < 4 ? 1 : cls < 6 ? 2 : cls < 10 ? 3 : 4; } // the long class rides in bucket 4
inline int bucket_wmax(int b) { return b == 0 ? 2 : b < 4 ? 4 : 8; } // words per lane: LDS per wavefront = S * wmax * 512 B

// Development knob: a comma-separated list of small integers from the environment (lane assignments of a pass), read
// once; absent or malformed = the built-in assignment.
struct DevList { int n = 0; int v[16] = {}; };
DevList dev_list(const char *name)
{
    DevList d;
    const char *e = getenv(name);
    if (!e || !*e) return d;
    while (*e && d.n < 16) {
        char *end = nullptr;
        const long x = strtol(e, &end, 10);
        if (end == e || x < 0 || x > 64) { d.n = 0; return d; }
        d.v[d.n++] = (int)x;
        e = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') { d.n = 0; return d; }
    }
    return d;
}

// part / parts: this launch takes every parts-th work item of the list starting with item `part`, on lane `lane_i`
template <int MODE> hipError_t launch_mega(ultra_ctx *ctx, MyArgs a, int bucket, const WorkItem *d_list, int32_t n, int part = 0, int parts = 1, int lane_i = -1)
{
    if (n <= 0) return hipSuccess;
    a.bstride = parts; a.boff = part;
    const int32_t blocks = (n - part + parts - 1) / parts;
    if (blocks <= 0) return hipSuccess;
    a.wflag = (MODE == MODE_FWD || MODE == MODE_FWD_EXACT) ? ctx->my.buckets[bucket].fwd_flag.as<int32_t>() : nullptr;
    a.wcount = (int32_t *)(ctx->my.bounds.as<int64_t>() + 120) + bucket;
    const size_t shmem = (size_t)a.S * bucket_wmax(bucket) * 64 * sizeof(uint64_t);
    hipStream_t cs = ctx->lane(lane_i >= 0 ? lane_i : bucket);
    const void *fn = bucket == 0 ? (const void *)myers_mega<MODE, 0> : bucket == 1 ? (const void *)myers_mega<MODE, 1>
                   : bucket == 2 ? (const void *)myers_mega<MODE, 2> : bucket == 3 ? (const void *)myers_mega<MODE, 3>
                                                                                     : (const void *)myers_mega<MODE, 4>;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    switch (bucket) {
    case 0: hipLaunchKernelGGL((myers_mega<MODE, 0>), dim3((unsigned)blocks), dim3(64), shmem, cs, a, d_list, n); break;
    case 1: hipLaunchKernelGGL((myers_mega<MODE, 1>), dim3((unsigned)blocks), dim3(64), shmem, cs, a, d_list, n); break;
    case 2: hipLaunchKernelGGL((myers_mega<MODE, 2>), dim3((unsigned)blocks), dim3(64), shmem, cs, a, d_list, n); break;
    case 3: hipLaunchKernelGGL((myers_mega<MODE, 3>), dim3((unsigned)blocks), dim3(64), shmem, cs, a, d_list, n); break;
    default: hipLaunchKernelGGL((myers_mega<MODE, 4>), dim3((unsigned)blocks), dim3(64), shmem, cs, a, d_list, n); break;
    }
    return hipGetLastError();
}

} // namespace

hipError_t ultra_scan_u64(ultra_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n, DevBuf &tmp, hipStream_t on)
{
    const hipStream_t ss = on ? on : ctx->stream;
    // three-level exclusive scan (n <= 2^30)
    if (n <= 0) return hipSuccess;
    int64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipError_t e;
    if ((e = tmp.ensure(sizeof(uint64_t) * (size_t)(2 * nb + 2 * SCAN_BLOCK + 2 * ((nb + SCAN_BLOCK - 1) / SCAN_BLOCK)))) != hipSuccess) return e;
    uint64_t *bs = tmp.as<uint64_t>();
    uint64_t *bs2 = bs + nb;
    hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, ss, in, out, bs, n);
    if (nb > 1) {
        if (nb > (int64_t)SCAN_BLOCK * SCAN_BLOCK) return hipErrorInvalidValue;
        int64_t nb2 = (nb + SCAN_BLOCK - 1) / SCAN_BLOCK;
        uint64_t *bs3 = bs2 + nb;
        hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb2), dim3(SCAN_BLOCK), 0, ss, bs, bs2, bs3, nb);
        if (nb2 > 1) {
            hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(SCAN_BLOCK), 0, ss, bs3, bs3 + nb2, (uint64_t *)nullptr, nb2);
            hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb2), dim3(SCAN_BLOCK), 0, ss, bs2, bs3 + nb2, nb);
        }
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, ss, out, bs2, n);
    }
    return hipGetLastError();
}

// ---- bucket sort: ids grouped by a small key --------------------------------------------------------------------
// Three launches (histogram, scan of the buckets, scatter) instead of a radix sort's dozen: the keys have a few
// thousand values at most, a block counts its 16384 elements in LDS and reserves its share of every bucket with one
// global atomic.  The order inside a bucket is arbitrary (it differs from run to run); callers only rely on the grouping.
namespace {
constexpr int kBsThreads = 1024, kBsPer = 16;
__global__ void __launch_bounds__(kBsThreads) k_bsort_hist(const uint32_t *keys, int64_t n, int nb, int32_t *hist)
{
    extern __shared__ int32_t lh[];
    for (int b = threadIdx.x; b < nb; b += kBsThreads) lh[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kBsThreads * kBsPer;
#pragma unroll
    for (int k = 0; k < kBsPer; k++) {
        const int64_t i = base + (int64_t)k * kBsThreads + threadIdx.x;
        if (i < n) atomicAdd(&lh[keys[i]], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += kBsThreads) { const int32_t c = lh[b]; if (c) atomicAdd(&hist[b], c); }
}
// one block: first[b] = cursor[b] = elements in the buckets before b (b = 0 .. nb); gfirst[g] = first[g * stride]
__global__ void __launch_bounds__(kBsThreads) k_bsort_scan(const int32_t *hist, int nb, int32_t *first, int32_t *cursor, int stride, int32_t *gfirst)
{
    __shared__ int32_t part[kBsThreads];
    const int per = (nb + kBsThreads - 1) / kBsThreads;
    const int lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    int32_t sum = 0;
    for (int b = lo; b < hi; b++) sum += hist[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < kBsThreads; d <<= 1) {
        const int32_t t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int32_t run = part[threadIdx.x] - sum;
    for (int b = lo; b < hi; b++) {
        first[b] = run; cursor[b] = run;
        if (gfirst && b % stride == 0) gfirst[b / stride] = run;
        run += hist[b];
    }
    if (threadIdx.x == kBsThreads - 1) {
        first[nb] = part[threadIdx.x];
        if (gfirst && nb % stride == 0) gfirst[nb / stride] = part[threadIdx.x];
    }
}
__global__ void __launch_bounds__(kBsThreads) k_bsort_scatter(const uint32_t *keys, const int32_t *vals, int64_t n, int nb, int32_t *cursor, int32_t *out)
{
    extern __shared__ int32_t lh[];
    for (int b = threadIdx.x; b < nb; b += kBsThreads) lh[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kBsThreads * kBsPer;
    uint32_t key[kBsPer];
    int32_t rank[kBsPer];
#pragma unroll
    for (int k = 0; k < kBsPer; k++) {
        const int64_t i = base + (int64_t)k * kBsThreads + threadIdx.x;
        key[k] = 0; rank[k] = 0;
        if (i < n) { key[k] = keys[i]; rank[k] = atomicAdd(&lh[key[k]], 1); }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += kBsThreads) { const int32_t c = lh[b]; lh[b] = c ? atomicAdd(&cursor[b], c) : 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBsPer; k++) {
        const int64_t i = base + (int64_t)k * kBsThreads + threadIdx.x;
        if (i < n) out[lh[key[k]] + rank[k]] = vals ? vals[i] : (int32_t)i;
    }
}
} // namespace

// out = vals (or 0 .. n-1) grouped by keys[i] < nb, ascending; first (nb + 1 entries, device) = first position of every
// bucket; gfirst (optional, nb / stride + 1 entries) = the same for every stride-th bucket.  n < 2^31, nb <= 8192.
hipError_t ultra_bucket_sort(hipStream_t s, const uint32_t *keys, const int32_t *vals, int64_t n, int nb, int32_t *out,
                             int32_t *first, int stride, int32_t *gfirst, DevBuf &tmp)
{
    if (nb <= 0 || nb > 8192 || n < 0 || n > 0x7fffffff) return hipErrorInvalidValue;
    hipError_t e;
    if ((e = tmp.ensure((size_t)nb * 8 + 64)) != hipSuccess) return e;
    int32_t *hist = tmp.as<int32_t>(), *cursor = hist + nb;
    if ((e = hipMemsetAsync(hist, 0, (size_t)nb * 4, s)) != hipSuccess) return e;
    const unsigned blocks = (unsigned)((n + kBsThreads * kBsPer - 1) / (kBsThreads * kBsPer));
    const size_t lds = (size_t)nb * 4;
    if (blocks) hipLaunchKernelGGL(k_bsort_hist, dim3(blocks), dim3(kBsThreads), lds, s, keys, n, nb, hist);
    hipLaunchKernelGGL(k_bsort_scan, dim3(1), dim3(kBsThreads), 0, s, hist, nb, first, cursor, stride > 0 ? stride : 1, gfirst);
    if (blocks) hipLaunchKernelGGL(k_bsort_scatter, dim3(blocks), dim3(kBsThreads), lds, s, keys, vals, n, nb, cursor, out);
    return hipGetLastError();
}

void ultra_myers_free(ultra_ctx *ctx)
{
    MyersState &st = ctx->my;
    DevBuf *bufs[] = {&st.seq, &st.rseq, &st.qoff, &st.qlen, &st.toff, &st.tlen, &st.k, &st.ed, &st.nloc, &st.slots,
                      &st.packed, &st.scanned, &st.blocksum, &st.locoff, &st.hitoff, &st.locend, &st.locstart,
                      &st.locpair, &st.hitpairs, &st.firstins, &st.nops, &st.trace, &st.bounds, &st.ops, &st.opsoff,
                      &st.bandcls, &st.banditems, &st.bandinfo, &st.sorttmp, &st.trace2, &st.startitems, &st.startinfo,
                      &st.hbuf, &st.raw, &st.lut, &st.u_qoff, &st.u_qlen, &st.u_toff, &st.u_tlen, &st.u_k, &st.u_rec, &st.skeys, &st.perm_d, &st.inv_d, &st.stats, &st.worktmp};
    for (DevBuf *b : bufs) b->release();
    for (auto &e : st.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (st.h_route) { (void)hipHostFree(st.h_route); st.h_route = nullptr; }
    for (auto &bk : st.buckets) {
        bk.fwd_list.release(); bk.rev_list.release(); bk.fwd_flag.release();
        if (bk.ev0) { (void)hipEventDestroy(bk.ev0); bk.ev0 = nullptr; }
        if (bk.ev1) { (void)hipEventDestroy(bk.ev1); bk.ev1 = nullptr; }
    }
    st.classes.clear();
}

// ---- staging ---------------------------------------------------------------------------------------------------
// Everything past the raw copies happens on the device: alphabet compaction (presence mask -> dense codes, the encoded
// arena and its reverse written by one kernel), the order of the pairs (radix sort by size class and target length, so
// that the 64 pairs of a wavefront run equally long; stable, i.e. the caller's order inside equal keys) and the
// forward work lists (one item per wavefront, heaviest first within a register bucket).  The host only sees two
// small read-backs: the presence mask and the per-class counts.
namespace {

__global__ void __launch_bounds__(256) k_presence(const uint8_t *raw, int64_t n, uint32_t *mask)
{
    __shared__ uint32_t sm[8];
    if (threadIdx.x < 8) sm[threadIdx.x] = 0;
    __syncthreads();
    // 16 bytes per load; the letters (64 .. 127: all a sequence file normally holds) collect in one 64-bit register,
    // anything else goes to LDS directly (rare)
    uint64_t mid = 0;
    auto one = [&](uint32_t b) {
        if ((b >> 6) == 1u) mid |= (uint64_t)1 << (b & 63);
        else atomicOr(&sm[b >> 5], 1u << (b & 31));
    };
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    int64_t head = (int64_t)((16 - ((uintptr_t)raw & 15)) & 15);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / 16;
    const uint4 *v = (const uint4 *)(raw + head);
    for (int64_t i = tid; i < nvec; i += nt) {
        const uint4 x = v[i];
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int sft = 0; sft < 32; sft += 8) one((w[j] >> sft) & 0xffu);
    }
    for (int64_t i = tid; i < head; i += nt) one(raw[i]);
    for (int64_t i = head + nvec * 16 + tid; i < n; i += nt) one(raw[i]);
    if ((uint32_t)mid) atomicOr(&sm[2], (uint32_t)mid);
    if ((uint32_t)(mid >> 32)) atomicOr(&sm[3], (uint32_t)(mid >> 32));
    __syncthreads();
    if (threadIdx.x < 8 && sm[threadIdx.x]) atomicOr(&mask[threadIdx.x], sm[threadIdx.x]);
}

// seq[at + i] = lut[raw[i]] and the mirror image rseq[N - 1 - (at + i)]
__global__ void __launch_bounds__(256) k_encode(const uint8_t *raw, int64_t n, const uint8_t *lut, uint8_t *seq, uint8_t *rseq, int64_t at, int64_t N)
{
    __shared__ uint8_t sl[256];
    sl[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    // eight bytes per thread and step: one load, eight table look-ups, one store each way (the mirror image byte-swapped)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    const int64_t n8 = n / 8;
    for (int64_t i = tid; i < n8; i += nt) {
        uint64_t x;
        __builtin_memcpy(&x, raw + 8 * i, 8);
        uint64_t y = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) y |= (uint64_t)sl[(x >> (8 * j)) & 0xff] << (8 * j);
        __builtin_memcpy(seq + at + 8 * i, &y, 8);
        const uint64_t z = __builtin_bswap64(y);
        __builtin_memcpy(rseq + (N - (at + 8 * i) - 8), &z, 8);
    }
    for (int64_t i = 8 * n8 + tid; i < n; i += nt) {
        const uint8_t c = sl[raw[i]];
        seq[at + i] = c;
        rseq[N - 1 - (at + i)] = c;
    }
}

// A pair whose query is more than k bases longer than its target cannot be a hit: in HW mode every alignment of the
// whole query leaves at least m - n query bases unmatched, so the distance is at least m - n > k and edlib answers -1
// (it gives up within a few columns: its band never reaches the last row).  With the reference's k = 0.4 * min(m, n)
// (classify_read_with_mams.py:312) that is every flank and exon segment longer than 1.4 x the read - the tallest,
// slowest size classes.  They sort behind every class and get their answer at stage time; no launch sees them.
__device__ __forceinline__ bool pair_cannot_hit(int m, int t, int k)
{
    const int ke = (k < 0 || k > m) ? m : k; // as setup_item clamps it
    return m - t > ke;
}
// stats, kStatN = 16 entries each: pairs per size class ([15] = skipped), forward cells per class ([15] = the cells the
// skipped pairs would have cost), longest target per class, longest query per class
constexpr int kStatN = 16;
template <bool PACKED>
__global__ void __launch_bounds__(256) k_pair_keys(const int32_t *q_len, const int32_t *t_len, const int32_t *k, const PairRec *rec, int32_t n, uint32_t *keys,
                                                   int32_t *idx, unsigned long long *stats)
{
    // 16 copies of the counters, a thread uses copy (thread & 15): the pairs of a wavefront mostly fall into two or three
    // classes, and 64 lanes adding to one LDS address take 64 turns
    constexpr int kCopies = 16;
    __shared__ unsigned long long shc[kCopies][4 * kStatN];
    for (int i = threadIdx.x; i < kCopies * 4 * kStatN; i += blockDim.x) (&shc[0][0])[i] = 0;
    __syncthreads();
    unsigned long long *sh = shc[threadIdx.x & (kCopies - 1)];
    // a few thousand blocks, each over many pairs: every block ends with up to 64 atomics on the same global counters
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        int m, t, kp;
        if (PACKED) { const PairRec r = rec[p]; m = r.q_len; t = r.t_len; kp = r.k; }
        else { m = q_len[p]; t = t_len[p]; kp = k[p]; }
        const uint32_t c = pair_cannot_hit(m, t, kp) ? kSkipClass : size_class_of(m);
        // Lane classes (m <= 256): queries that leave the lower half of their first word empty sort together, so that
        // whole wavefronts can run that word on one 32-bit half (lane_forward).
        const uint32_t half = (c < 4 && ((c + 1) * 64 - (uint32_t)m) >= 32) ? 1u : 0u;
        keys[p] = (c << 25) | (half << 24) | (uint32_t)(t < 0xFFFFFF ? t : 0xFFFFFF);
        idx[p] = p;
        atomicAdd(&sh[c], 1ull);
        atomicAdd(&sh[kStatN + c], (unsigned long long)m * (unsigned long long)t);
        atomicMax(&sh[2 * kStatN + c], (unsigned long long)t);
        atomicMax(&sh[3 * kStatN + c], (unsigned long long)m);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * kStatN; i += blockDim.x) {
        unsigned long long v = 0;
        for (int c2 = 0; c2 < kCopies; c2++) v = i < 2 * kStatN ? v + shc[c2][i] : (shc[c2][i] > v ? shc[c2][i] : v);
        if (!v) continue;
        if (i < 2 * kStatN) atomicAdd(&stats[i], v); else atomicMax(&stats[i], v);
    }
}

template <bool PACKED>
__global__ void __launch_bounds__(256) k_gather_pairs(const int32_t *perm, int32_t n, int64_t shift,
                                                      const int64_t *uq, const int32_t *uql, const int64_t *ut, const int32_t *utl, const int32_t *uk,
                                                      const PairRec *rec,
                                                      int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len, int32_t *k, int32_t *inv,
                                                      int32_t *ed, int32_t *nloc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = perm[i];
    PairRec r;
    if (PACKED) r = rec[p]; // one 32-byte access instead of five lines
    else { r.q_off = uq[p]; r.t_off = ut[p]; r.q_len = uql[p]; r.t_len = utl[p]; r.k = uk[p]; }
    q_off[i] = r.q_off + shift; t_off[i] = r.t_off + shift;
    q_len[i] = r.q_len; t_len[i] = r.t_len; k[i] = r.k;
    inv[p] = i;
    // the pairs no launch takes (pair_cannot_hit) are answered here; every run of the staged batch leaves them alone
    if (pair_cannot_hit(r.q_len, r.t_len, r.k)) { ed[i] = -1; nloc[i] = 0; }
}

// The forward work lists of all register buckets in one go: one launch makes the items, ONE sort orders them by
// (bucket, cost descending), one launch deals them into the buckets' lists (five rounds of ~12 tiny launches before).
struct WorkRanges { int32_t n; struct { int32_t cls, G, W, first_item, bucket; int64_t pb, pe; } r[kNumSizeClasses + 1]; };
constexpr uint32_t kWorkCostMask = 0x0FFFFFFFu; // sort key = bucket << 28 | (mask - min(cost, mask))
__global__ void __launch_bounds__(256) k_work_items(WorkRanges wr, const int32_t *q_len, const int32_t *t_len, int32_t n_items, WorkItem *items, uint32_t *keys, int32_t *idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    int r = 0;
    while (r + 1 < wr.n && i >= wr.r[r + 1].first_item) r++;
    const int per = 64 / wr.r[r].G;
    WorkItem wi;
    wi.base = wr.r[r].pb + (int64_t)(i - wr.r[r].first_item) * per;
    wi.end = wr.r[r].pe;
    wi.cls = wr.r[r].cls;
    const int64_t last = (wi.base + per < wi.end ? wi.base + per : wi.end) - 1;
    int64_t cost = (int64_t)wr.r[r].W * ((int64_t)t_len[last] + wr.r[r].G); // ~ words per lane x columns
    if (wr.r[r].cls == kLongClass) cost *= (q_len[last] + kStripRows - 1) / kStripRows; // one pair per item, strip after strip
    wi.cost = (int32_t)(cost < 0x7fffffff ? cost : 0x7fffffff);
    items[i] = wi;
    const uint32_t c28 = (uint32_t)(cost < (int64_t)kWorkCostMask ? cost : (int64_t)kWorkCostMask);
    keys[i] = ((uint32_t)wr.r[r].bucket << 28) | (kWorkCostMask - c28);
    idx[i] = i;
}
struct WorkOut { WorkItem *out[kBuckets]; int32_t first[kBuckets + 1]; };
__global__ void __launch_bounds__(256) k_gather_work(const WorkItem *items, const int32_t *order, int32_t n, WorkOut wo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = 0;
    while (b + 1 < kBuckets && i >= wo.first[b + 1]) b++;
    wo.out[b][i - wo.first[b]] = items[order[i]];
}

} // namespace

void ultra_launch_presence(hipStream_t s, const uint8_t *raw, int64_t n, uint32_t *mask)
{
    if (n <= 0) return;
    const unsigned gb = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_presence, dim3(gb), dim3(256), 0, s, raw, n, mask);
}
void ultra_launch_encode(hipStream_t s, const uint8_t *raw, int64_t n, const uint8_t *lut, uint8_t *seq, uint8_t *rseq, int64_t at, int64_t N)
{
    if (n <= 0) return;
    const unsigned gb = (unsigned)std::min<int64_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(k_encode, dim3(gb), dim3(256), 0, s, raw, n, lut, seq, rseq, at, N);
}

// Orders the n_pairs pairs that lie UNSORTED in st.u_* (offsets relative to the arena's first byte) and builds the
// forward work lists.  The arena (st.seq / st.rseq, st.N, st.S) must have been filled on ctx->stream before.
int ultra_myers_stage_pairs(ultra_ctx *ctx, int32_t n_pairs)
{
    MyersState &st = ctx->my;
    hipStream_t s = ctx->stream;
    st.n_pairs = n_pairs;
    st.perm.clear(); st.perm_on_host = false;
    st.classes.clear();
    st.cells = 0; st.cells_skipped = 0; st.n_skipped = 0;
    const size_t np = (size_t)std::max(n_pairs, 1);
    ULTRA_HIP_CHECK(ctx, st.qoff.ensure(np * 8)); ULTRA_HIP_CHECK(ctx, st.toff.ensure(np * 8));
    ULTRA_HIP_CHECK(ctx, st.qlen.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.tlen.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.k.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, st.ed.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.nloc.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, st.firstins.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.nops.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, st.hitpairs.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, st.slots.ensure(np * 4 * (size_t)ctx->cfg.loc_slots));
    ULTRA_HIP_CHECK(ctx, st.packed.ensure(np * 8)); ULTRA_HIP_CHECK(ctx, st.scanned.ensure(np * 8));
    ULTRA_HIP_CHECK(ctx, st.locoff.ensure((np + 1) * 8)); ULTRA_HIP_CHECK(ctx, st.hitoff.ensure((np + 1) * 8));
    ULTRA_HIP_CHECK(ctx, st.bounds.ensure(64 * 8 * 4));
    ULTRA_HIP_CHECK(ctx, st.skeys.ensure(np * 4 * 4)); // keys in | keys out | index in | index out (pairs, then work items)
    ULTRA_HIP_CHECK(ctx, st.perm_d.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.inv_d.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, st.stats.ensure(4 * kStatN * 8));
    for (auto &e : st.ev) if (!e) ULTRA_HIP_CHECK(ctx, hipEventCreate(&e));
    for (int b = 0; b < kBuckets; b++) {
        MyersBucket &bk = st.buckets[b];
        bk.n_fwd = 0; bk.wmax = bucket_wmax(b); bk.cells = 0;
        if (!bk.ev0) { ULTRA_HIP_CHECK(ctx, hipEventCreate(&bk.ev0)); ULTRA_HIP_CHECK(ctx, hipEventCreate(&bk.ev1)); }
    }
    if (n_pairs == 0) { ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s)); st.staged = true; return ULTRA_OK; }
    ultra_drain_guard drain(ctx);
    uint32_t *k_in = st.skeys.as<uint32_t>(), *k_out = k_in + np;
    int32_t *i_in = (int32_t *)(k_out + np), *i_out = i_in + np;
    const unsigned gp = (unsigned)((n_pairs + 255) / 256);
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.stats.p, 0, 4 * kStatN * 8, s));
    if (st.u_packed)
        hipLaunchKernelGGL(k_pair_keys<true>, dim3(std::min(gp, 2048u)), dim3(256), 0, s, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                           st.u_rec.as<PairRec>(), n_pairs, k_in, i_in, st.stats.as<unsigned long long>());
    else
        hipLaunchKernelGGL(k_pair_keys<false>, dim3(std::min(gp, 2048u)), dim3(256), 0, s, st.u_qlen.as<int32_t>(), st.u_tlen.as<int32_t>(), st.u_k.as<int32_t>(),
                           (const PairRec *)nullptr, n_pairs, k_in, i_in, st.stats.as<unsigned long long>());
    {
        size_t tmp_bytes = 0;
        ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, i_in, st.perm_d.as<int32_t>(), (size_t)n_pairs, 0, 29, s));
        ULTRA_HIP_CHECK(ctx, st.sorttmp.ensure(tmp_bytes + 64));
        ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(st.sorttmp.p, tmp_bytes, k_in, k_out, i_in, st.perm_d.as<int32_t>(), (size_t)n_pairs, 0, 29, s));
    }
    if (st.u_packed)
        hipLaunchKernelGGL(k_gather_pairs<true>, dim3(gp), dim3(256), 0, s, st.perm_d.as<int32_t>(), n_pairs, (int64_t)kArenaPad,
                           (const int64_t *)nullptr, (const int32_t *)nullptr, (const int64_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                           st.u_rec.as<PairRec>(),
                           st.qoff.as<int64_t>(), st.qlen.as<int32_t>(), st.toff.as<int64_t>(), st.tlen.as<int32_t>(), st.k.as<int32_t>(),
                           st.inv_d.as<int32_t>(), st.ed.as<int32_t>(), st.nloc.as<int32_t>());
    else
        hipLaunchKernelGGL(k_gather_pairs<false>, dim3(gp), dim3(256), 0, s, st.perm_d.as<int32_t>(), n_pairs, (int64_t)kArenaPad,
                           st.u_qoff.as<int64_t>(), st.u_qlen.as<int32_t>(), st.u_toff.as<int64_t>(), st.u_tlen.as<int32_t>(), st.u_k.as<int32_t>(),
                           (const PairRec *)nullptr,
                           st.qoff.as<int64_t>(), st.qlen.as<int32_t>(), st.toff.as<int64_t>(), st.tlen.as<int32_t>(), st.k.as<int32_t>(),
                           st.inv_d.as<int32_t>(), st.ed.as<int32_t>(), st.nloc.as<int32_t>());
    ULTRA_HIP_CHECK(ctx, hipGetLastError());
    unsigned long long h_stats[4 * kStatN];
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_stats, st.stats.p, sizeof h_stats, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    int64_t at = 0;
    for (int c = 0; c < kNumSizeClasses; c++) {
        const int64_t cnt = (int64_t)h_stats[c];
        if (cnt <= 0) continue;
        MyersClass mc; mc.cls = c; mc.G = kClasses[c].G; mc.W = kClasses[c].W; mc.pb = at; mc.pe = at + cnt;
        mc.max_t = (int32_t)h_stats[2 * kStatN + c];
        mc.max_m = (int32_t)h_stats[3 * kStatN + c];
        mc.lb = mc.le = mc.hb = mc.he = 0;
        mc.cells = (double)h_stats[kStatN + c];
        st.classes.push_back(mc);
        st.cells += mc.cells;
        st.buckets[bucket_of(c)].cells += mc.cells;
        at += cnt;
    }
    st.n_skipped = (int64_t)h_stats[kSkipClass];
    st.cells_skipped = (double)h_stats[kStatN + kSkipClass];
    st.n_run = (int32_t)at; // the skipped pairs lie behind every class
    if (at + st.n_skipped != n_pairs) return ultra_fail(ctx, ULTRA_E_HIP, "ultra_myers_stage: size-class counts do not add up");
    // forward work lists (heaviest wavefront first), one per register bucket
    {
        WorkRanges wr; wr.n = 0;
        WorkOut wo;
        int32_t n_items = 0;
        for (int b = 0; b < kBuckets; b++) {
            MyersBucket &bk = st.buckets[b];
            wo.first[b] = n_items;
            for (auto &mc : st.classes) {
                const int cls = mc.cls;
                if (bucket_of(cls) != b) continue;
                const int per = 64 / mc.G;
                wr.r[wr.n].cls = cls; wr.r[wr.n].G = mc.G; wr.r[wr.n].W = mc.W; wr.r[wr.n].first_item = n_items; wr.r[wr.n].bucket = b;
                wr.r[wr.n].pb = mc.pb; wr.r[wr.n].pe = mc.pe;
                wr.n++;
                n_items += (int32_t)((mc.pe - mc.pb + per - 1) / per);
            }
            bk.n_fwd = n_items - wo.first[b];
            ULTRA_HIP_CHECK(ctx, bk.fwd_list.ensure((size_t)std::max(bk.n_fwd, 1) * sizeof(WorkItem)));
            ULTRA_HIP_CHECK(ctx, bk.fwd_flag.ensure((size_t)std::max(bk.n_fwd, 1) * sizeof(int32_t)));
            wo.out[b] = bk.fwd_list.as<WorkItem>();
        }
        wo.first[kBuckets] = n_items;
        if (n_items > 0) {
            // (an item holds at least one pair, so the pair sort's key / index scratch is large enough)
            ULTRA_HIP_CHECK(ctx, st.worktmp.ensure((size_t)n_items * sizeof(WorkItem)));
            const unsigned gi = (unsigned)((n_items + 255) / 256);
            hipLaunchKernelGGL(k_work_items, dim3(gi), dim3(256), 0, s, wr, st.qlen.as<int32_t>(), st.tlen.as<int32_t>(), n_items, st.worktmp.as<WorkItem>(), k_in, i_in);
            size_t tmp_bytes = 0;
            ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, i_in, i_out, (size_t)n_items, 0, 31, s));
            ULTRA_HIP_CHECK(ctx, st.sorttmp.ensure(tmp_bytes + 64));
            ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(st.sorttmp.p, tmp_bytes, k_in, k_out, i_in, i_out, (size_t)n_items, 0, 31, s));
            hipLaunchKernelGGL(k_gather_work, dim3(gi), dim3(256), 0, s, st.worktmp.as<WorkItem>(), i_out, n_items, wo);
            ULTRA_HIP_CHECK(ctx, hipGetLastError());
        }
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    drain.dismiss();
    st.staged = true;
    return ULTRA_OK;
}

// caller's pair index of every sorted position, on the host (only the Myers-only entry points need it)
int ultra_myers_host_perm(ultra_ctx *ctx)
{
    MyersState &st = ctx->my;
    if (st.perm_on_host) return ULTRA_OK;
    st.perm.resize((size_t)st.n_pairs);
    if (st.n_pairs > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.perm.data(), st.perm_d.p, (size_t)st.n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    st.perm_on_host = true;
    return ULTRA_OK;
}

int ultra_myers_stage_impl(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                           const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                           const int32_t *k)
{
    MyersState &st = ctx->my;
    st.staged = false; st.ran = false;
    ctx->mam.up_valid = false; // st.raw is about to hold this batch's arena instead of an uploaded request's reads
    if (n_pairs < 0 || (n_pairs > 0 && (!arena || !q_off || !q_len || !t_off || !t_len || !k)))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_myers_stage: null pointer or negative pair count");
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // the device code trusts the descriptors: validate them here
    for (int32_t p = 0; p < n_pairs; p++) {
        if (q_len[p] < 1 || t_len[p] < 1 || q_off[p] < 0 || t_off[p] < 0 ||
            (uint64_t)q_off[p] + (uint64_t)q_len[p] > arena_len || (uint64_t)t_off[p] + (uint64_t)t_len[p] > arena_len)
            return ultra_fail(ctx, ULTRA_E_ARG, "ultra_myers_stage: pair " + std::to_string(p) +
                                                    " has an empty sequence or offsets outside the arena");
    }
    hipStream_t s = ctx->stream;
    ultra_drain_guard drain(ctx);
    st.pool_serial = 0; // this batch's arena replaces whatever resident pool the buffers held
    // ---- arena: raw bytes up, alphabet compaction (distinct bytes -> dense codes, comparisons stay byte-exact) ----
    const int64_t N = (int64_t)arena_len + 2 * (int64_t)kArenaPad;
    ULTRA_HIP_CHECK(ctx, st.raw.ensure(arena_len + 16));
    ULTRA_HIP_CHECK(ctx, st.seq.ensure((size_t)N + 16)); ULTRA_HIP_CHECK(ctx, st.rseq.ensure((size_t)N + 16));
    ULTRA_HIP_CHECK(ctx, st.lut.ensure(256 + 32));
    uint32_t *d_mask = (uint32_t *)(st.lut.as<uint8_t>() + 256);
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(d_mask, 0, 32, s));
    uint32_t h_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (arena_len > 0 && n_pairs > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.raw.p, arena, arena_len, hipMemcpyHostToDevice, s));
        const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)arena_len + 255) / 256, 4096);
        hipLaunchKernelGGL(k_presence, dim3(gb), dim3(256), 0, s, st.raw.as<uint8_t>(), (int64_t)arena_len, d_mask);
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_mask, d_mask, 32, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    }
    uint8_t h_lut[256];
    int S = 0;
    for (int i = 0; i < 256; i++) { h_lut[i] = 0; if (h_mask[i >> 5] >> (i & 31) & 1u) h_lut[i] = (uint8_t)(S < 255 ? S : 255), S++; }
    if (S > kMaxSymbols)
        return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_myers_stage: more than 36 distinct symbols in the batch");
    if (S == 0) S = 1;
    st.N = N; st.S = S;
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.seq.p, 0, kArenaPad, s));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.seq.as<uint8_t>() + kArenaPad + arena_len, 0, kArenaPad + 16, s));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.rseq.p, 0, kArenaPad, s));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.rseq.as<uint8_t>() + kArenaPad + arena_len, 0, kArenaPad + 16, s));
    if (arena_len > 0 && n_pairs > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.lut.p, h_lut, 256, hipMemcpyHostToDevice, s));
        const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)arena_len + 255) / 256, 65536);
        hipLaunchKernelGGL(k_encode, dim3(gb), dim3(256), 0, s, st.raw.as<uint8_t>(), (int64_t)arena_len, st.lut.as<uint8_t>(),
                           st.seq.as<uint8_t>(), st.rseq.as<uint8_t>(), (int64_t)kArenaPad, N);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    // ---- pair descriptors up, as they are ----
    const size_t np = (size_t)std::max(n_pairs, 1);
    ULTRA_HIP_CHECK(ctx, st.u_qoff.ensure(np * 8)); ULTRA_HIP_CHECK(ctx, st.u_toff.ensure(np * 8));
    ULTRA_HIP_CHECK(ctx, st.u_qlen.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.u_tlen.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, st.u_k.ensure(np * 4));
    if (n_pairs > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.u_qoff.p, q_off, np * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.u_toff.p, t_off, np * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.u_qlen.p, q_len, np * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.u_tlen.p, t_len, np * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.u_k.p, k, np * 4, hipMemcpyHostToDevice, s));
    }
    drain.dismiss();
    st.u_packed = false;
    return ultra_myers_stage_pairs(ctx, n_pairs);
}

// The per-pass times of the last run (events of the run's streams; the caller has synchronised).
int ultra_myers_collect_times(ultra_ctx *ctx)
{
    MyersState &st = ctx->my;
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&st.ms[0], st.ev[0], st.ev[1]));
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&st.ms[1], st.ev[1], st.ev[2]));
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&st.ms[2], st.ev[2], st.ev[3]));
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&st.ms[3], st.ev[0], st.ev[3]));
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&st.ms[4], st.ev[0], st.ev[4]));
    for (auto &bk : st.buckets) { bk.fwd_ms = 0; if (bk.n_fwd > 0) ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&bk.fwd_ms, bk.ev0, bk.ev1)); }
    st.times_pending = false;
    return ULTRA_OK;
}

// defer_sync: return with the traceback still in flight on ctx->stream when everything the caller needs on the host
// has arrived (st.times_pending tells); the caller enqueues its next kernels behind it, synchronises once at the end
// and calls ultra_myers_collect_times.
int ultra_myers_run_impl(ultra_ctx *ctx, bool want_ops, bool defer_sync)
{
    MyersState &st = ctx->my;
    if (!st.staged) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_myers_run: no staged batch");
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int32_t n = st.n_pairs;
    st.ran = false;
    st.times_pending = false;
    st.total_loc = st.total_hit = 0;
    for (auto &m : st.ms) m = 0;
    if (n == 0) {
        if (ctx->mid_hook) {
            const int hrc = ctx->mid_hook(s);
            if (hrc != ULTRA_OK) return hrc;
            ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        }
        st.ran = true;
        return ULTRA_OK;
    }
    MyArgs a;
    memset(&a, 0, sizeof a);
    a.seq = st.seq.as<uint8_t>(); a.rseq = st.rseq.as<uint8_t>(); a.N = st.N; a.S = st.S;
    a.q_off = st.qoff.as<int64_t>(); a.q_len = st.qlen.as<int32_t>(); a.t_off = st.toff.as<int64_t>();
    a.t_len = st.tlen.as<int32_t>(); a.k = st.k.as<int32_t>();
    a.ed = st.ed.as<int32_t>(); a.nloc = st.nloc.as<int32_t>(); a.slots = st.slots.as<int32_t>(); a.slot_cap = ctx->cfg.loc_slots;
    a.loc_off = st.locoff.as<int64_t>(); a.hit_pairs = st.hitpairs.as<int32_t>(); a.first_ins = st.firstins.as<int32_t>();
    a.n_ops = st.nops.as<int32_t>();

    // long class: the strips' hand-over buffers, two per item of a pass (pairs now, locations and hits later)
    const MyersClass *longc = nullptr;
    for (auto &mc : st.classes) if (mc.cls == kLongClass) longc = &mc;
    const int64_t hstride = longc ? (((int64_t)longc->max_t + 64 + 16 + 63) & ~(int64_t)63) : 0;
    auto long_buffers = [&](int64_t items) -> int {
        const uint64_t bytes = (uint64_t)2 * (uint64_t)std::max<int64_t>(items, 1) * (uint64_t)hstride;
        if (bytes > ((uint64_t)4 << 30))
            return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_myers_run: more than 4 GiB of strip buffers for queries beyond 32768 bases");
        ULTRA_HIP_CHECK(ctx, st.hbuf.ensure((size_t)bytes + 64));
        a.hbuf = st.hbuf.as<uint8_t>(); a.hstride = hstride;
        return ULTRA_OK;
    };
    if (longc) {
        const int lrc = long_buffers(longc->pe - longc->pb);
        if (lrc != ULTRA_OK) return lrc;
        a.hbase = longc->pb;
    }
    ultra_drain_guard drain(ctx); // any early return below first waits for what has been enqueued
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.firstins.p, 0, (size_t)n * 4, s));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.nops.p, 0, (size_t)n * 4, s));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.bounds.as<int64_t>() + 120, 0, kBuckets * 4, s)); // lengths of the buckets' re-run lists
    ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[0], s));
    // ---- pass 1: forward HW ----
    ULTRA_HIP_CHECK(ctx, ultra_fork_at(ctx, 0));
    static const DevList dev_order = dev_list("ULTRA_DEV_FWD_ORDER"); // buckets in launch order
    for (int bi = 0; bi < kBuckets; bi++) {
        const int b = dev_order.n == kBuckets && dev_order.v[bi] < kBuckets ? dev_order.v[bi] : kBuckets - 1 - bi;
        MyersBucket &bk = st.buckets[b];
        if (bk.n_fwd <= 0) continue;
        const int lane_b = b == kBuckets - 1 && ctx->cfg.launch_lanes >= kBuckets + 3 ? kBuckets + 2 : b;
        hipStream_t cs = ctx->lane(lane_b);
        ULTRA_HIP_CHECK(ctx, hipEventRecord(bk.ev0, cs));
        // The workgroup dispatcher deals the chip's slots about evenly over the launches in flight, not by their work:
        // the bucket of the three- and four-word lane classes (m = 129 .. 256 against long targets) finished 1.5 - 2 ms
        // after the others.  With lanes to spare its list is dealt over three launches (interleaved, so each keeps the
        // heaviest-first order) and takes three shares: the buckets then end within a millisecond of each other
        // (measured: forward phase 13.0 -> 12.4 ms; a second launch for the lane groups' bucket as well: 12.6).
        static const DevList dev_fwd = dev_list("ULTRA_DEV_FWD_LANES"); // parts of bucket 1, then lanes: b0, b1 part 0 / 1 / 2, b2, b3, b4
        if (dev_fwd.n == 8) {
            const int parts = dev_fwd.v[0] < 1 ? 1 : dev_fwd.v[0] > 3 ? 3 : dev_fwd.v[0];
            const int lane_of[5] = {dev_fwd.v[1], dev_fwd.v[2], dev_fwd.v[5], dev_fwd.v[6], dev_fwd.v[7]};
            if (b == 1 && bk.n_fwd >= 64) {
                for (int pi = 0; pi < parts; pi++)
                    ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD>(ctx, a, b, bk.fwd_list.as<WorkItem>(), bk.n_fwd, pi, parts, dev_fwd.v[2 + pi]));
            } else ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD>(ctx, a, b, bk.fwd_list.as<WorkItem>(), bk.n_fwd, 0, 1, lane_of[b]));
            ULTRA_HIP_CHECK(ctx, hipEventRecord(bk.ev1, cs));
            continue;
        }
        if (ctx->cfg.launch_lanes >= kBuckets + 3 && b == 1 && bk.n_fwd >= 64) {
            // parts on lanes 1, 4, 5 and the widest bucket (few, long wavefronts) on the last lane: the third part on lane 6
            // started only when the second had ended (12.4 -> 12.1 ms; scripts/dev/lane_variants.sh)
            ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD>(ctx, a, b, bk.fwd_list.as<WorkItem>(), bk.n_fwd, 0, 3));
            ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD>(ctx, a, b, bk.fwd_list.as<WorkItem>(), bk.n_fwd, 1, 3, kBuckets - 1));
            ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD>(ctx, a, b, bk.fwd_list.as<WorkItem>(), bk.n_fwd, 2, 3, kBuckets));
        } else ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD>(ctx, a, b, bk.fwd_list.as<WorkItem>(), bk.n_fwd, 0, 1, lane_b));
        ULTRA_HIP_CHECK(ctx, hipEventRecord(bk.ev1, cs));
    }
    ULTRA_HIP_CHECK(ctx, ultra_join_at(ctx, 1));
    ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[4], s)); // all forward launches done
    // The caller's mid-run work only needs the forward pass.  With eight lanes it goes onto the last auxiliary lane now,
    // beside the offsets below (a chain of small kernels on an otherwise idle chip); every later join takes it back in.
    bool hook_early = false;
    if (ctx->mid_hook && ctx->cfg.launch_lanes == kLanes) {
        hipStream_t hs = ctx->aux[kAuxStreams - 1];
        ULTRA_HIP_CHECK(ctx, hipEventRecord(ctx->fork_ev, s));
        ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(hs, ctx->fork_ev, 0));
        const int hrc = ctx->mid_hook(hs);
        if (hrc != ULTRA_OK) return hrc;
        ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[6], hs));
        hook_early = true;
    }
    // ---- CSR offsets ----
    int64_t *d_overflow = st.bounds.as<int64_t>() + 48; // one flag per register bucket
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(d_overflow, 0, kBuckets * 8, s));
    {
        BucketFirst bf;
        for (int b = 0; b <= kBuckets; b++) bf.first[b] = n;
        for (auto mc = st.classes.rbegin(); mc != st.classes.rend(); ++mc) {
            const int b = bucket_of(mc->cls);
            for (int t = 0; t <= b; t++) bf.first[t] = std::min<int32_t>(bf.first[t], (int32_t)mc->pb);
        }
        bf.first[0] = 0;
        for (int b = kBuckets - 1; b >= 1; b--) bf.first[b] = std::min(bf.first[b], bf.first[b + 1]);
        hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, s, a.ed, a.nloc, st.packed.as<uint64_t>(), n, a.slot_cap, d_overflow, bf);
    }
    ULTRA_HIP_CHECK(ctx, ultra_scan_u64(ctx, st.packed.as<uint64_t>(), st.scanned.as<uint64_t>(), n, st.blocksum));
    hipLaunchKernelGGL(k_unpack, dim3((n + 255) / 256), dim3(256), 0, s, st.scanned.as<uint64_t>(), st.packed.as<uint64_t>(),
                       st.locoff.as<int64_t>(), st.hitoff.as<int64_t>(), n);
    int64_t h_overflow[kBuckets] = {0, 0, 0, 0, 0};
    int32_t h_rerun[kBuckets] = {0, 0, 0, 0, 0}; // work items to run again without a limit on the end columns
    const int64_t loc_cap = (int64_t)(std::min(st.locend.cap, std::min(st.locstart.cap, st.locpair.cap)) / 4);
    {
        // class boundaries (+ totals) to the host: one small copy, the only mid-batch synchronisation
        std::vector<int64_t> pos;
        for (auto &mc : st.classes) pos.push_back(mc.pb);
        pos.push_back(n);
        int npz = (int)pos.size();
        int64_t *dpos = st.bounds.as<int64_t>();
        int64_t *dout = dpos + 64;
        if (npz > kNumSizeClasses + 2) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_myers_run: more size classes than the library knows");
        BoundPos bp;
        for (int t = 0; t < kNumSizeClasses + 2; t++) bp.v[t] = t < npz ? pos[(size_t)t] : 0;
        hipLaunchKernelGGL(k_gather_bounds, dim3(1), dim3(64), 0, s, st.locoff.as<int64_t>(), st.hitoff.as<int64_t>(), bp, dout, npz);
        // The totals land in page-locked memory: a copy into pageable memory holds the host until it has been carried out,
        // and the launch below, which is meant to run while the totals travel, went out 0.3 ms late.
        if (!st.h_route) ULTRA_HIP_CHECK(ctx, hipHostMalloc((void **)&st.h_route, kRouteBytes, hipHostMallocDefault));
        int64_t *hb = (int64_t *)((char *)st.h_route + 1024);
        int64_t *p_overflow = hb + 40;
        int32_t *p_rerun = (int32_t *)(hb + 48);
        ultra_drain_guard drain_b(ctx);
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hb, dout, npz * 16, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(p_overflow, d_overflow, kBuckets * 8, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(p_rerun, st.bounds.as<int64_t>() + 120, kBuckets * 4, hipMemcpyDeviceToHost, s));
        // The location arrays are filled while the totals travel to the host, into what the arrays hold from earlier
        // batches; the host only waits for the copies (st.ev[1] serves as their marker until it is recorded for good).
        if (loc_cap > 0) {
            ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[1], s));
            hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, s, a.ed, a.nloc, a.slots, a.slot_cap, a.loc_off,
                               st.hitoff.as<int64_t>(), st.locend.as<int32_t>(), st.locpair.as<int32_t>(), st.hitpairs.as<int32_t>(), n, loc_cap);
            ULTRA_HIP_CHECK(ctx, hipGetLastError());
            ULTRA_HIP_CHECK(ctx, hipEventSynchronize(st.ev[1]));
        } else ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        drain_b.dismiss();
        for (int b = 0; b < kBuckets; b++) { h_overflow[b] = p_overflow[b]; h_rerun[b] = p_rerun[b]; }
        for (size_t c = 0; c < st.classes.size(); c++) {
            st.classes[c].lb = hb[2 * c]; st.classes[c].le = hb[2 * (c + 1)];
            st.classes[c].hb = hb[2 * c + 1]; st.classes[c].he = hb[2 * (c + 1) + 1];
        }
        st.total_loc = hb[2 * (npz - 1)]; st.total_hit = hb[2 * (npz - 1) + 1];
    }
    if (st.total_loc > (int64_t)INT32_MAX)
        return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_myers_run: more than 2^31 - 1 locations in one batch (" + std::to_string(st.total_loc) + "): split it");
    const size_t nl = (size_t)std::max<int64_t>(st.total_loc, 1);
    if ((int64_t)nl > loc_cap) { // first batch, or more locations than any batch before: the arrays grow, fill (again)
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s)); // the launch that ran ahead still writes the old arrays
        ULTRA_HIP_CHECK(ctx, st.locend.ensure(nl * 4)); ULTRA_HIP_CHECK(ctx, st.locstart.ensure(nl * 4)); ULTRA_HIP_CHECK(ctx, st.locpair.ensure(nl * 4));
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, s, a.ed, a.nloc, a.slots, a.slot_cap, a.loc_off,
                           st.hitoff.as<int64_t>(), st.locend.as<int32_t>(), st.locpair.as<int32_t>(), st.hitpairs.as<int32_t>(), n, (int64_t)nl);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    a.loc_end = st.locend.as<int32_t>(); a.loc_start = st.locstart.as<int32_t>(); a.loc_pair = st.locpair.as<int32_t>();
    if (longc && longc->le - longc->lb > longc->pe - longc->pb) { // more locations than pairs: the start pass needs more buffers
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        const int lrc = long_buffers(longc->le - longc->lb);
        if (lrc != ULTRA_OK) return lrc;
    }
    // ---- traceback routing, first half.  Which band class a hit goes to follows from its distance, so the per-class
    //      counts the host lays the traceback launches out with only need the forward pass: they are taken now and have
    //      reached the host long before the start pass ends.  What is left behind the start pass is the sort of the hits
    //      inside their classes (by the first location's length, which needs its start): stream-ordered, no host round
    //      trip between the two passes (before: 1.0 ms from the last start kernel to the first traceback kernel). ----
    const int64_t n_hits = st.total_hit;
    // the band kernels rebuild match vectors from 3 or 5 bit-planes of (code + 1): up to 31 symbols; beyond, full height
    const bool use_band = ctx->cfg.banded_traceback != 0 && !want_ops && n_hits > 0 && st.S + 1 <= 32;
    if (!st.h_route) ULTRA_HIP_CHECK(ctx, hipHostMalloc((void **)&st.h_route, kRouteBytes, hipHostMallocDefault));
    int32_t *const h_info = st.h_route; // page-locked: the copy must not hold the host up
    if (use_band) ULTRA_HIP_CHECK(ctx, st.bandinfo.ensure(256));
    auto route_counts = [&]() -> int {
        if (!use_band) return ULTRA_OK;
        ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.bandinfo.p, 0, 256, s));
        hipLaunchKernelGGL(k_band_keys<true>, dim3((unsigned)std::min<int64_t>((n_hits + 255) / 256, 1024)), dim3(256), 0, s, a, n_hits, (uint32_t *)nullptr, st.bandinfo.as<int32_t>());
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_info, st.bandinfo.p, (kBandBounds + kBandClasses + kNumSizeClasses) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[5], s));
        return ULTRA_OK;
    };
    // ---- pass 1b: exact re-run for pairs with more end columns than first-pass slots (rare), and
    //      pass 2: start positions. ----
    const bool band_starts = ctx->cfg.banded_starts != 0 && st.total_loc > 0 && st.S + 1 <= 32;
    // With the full set of lanes the start pass's launches leave the main stream free: the counts are taken on it beside
    // the pass instead of in front of it (0.1 ms of an otherwise idle chip).
    const bool counts_beside = !band_starts && ctx->cfg.launch_lanes == kLanes;
    if (!counts_beside) { const int rrc = route_counts(); if (rrc != ULTRA_OK) return rrc; }
    ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[1], s));
    if (!band_starts) {
        // Full-height reverse pass: one launch per register bucket on the bucket's own lane.  A bucket's start pass
        // only reads end columns of the bucket's own pairs, so the re-run and the start pass follow each other in
        // the lane without a join in between: one bucket's re-run rides under the other buckets' start passes.
        ULTRA_HIP_CHECK(ctx, ultra_fork_at(ctx, 2));
        for (int b = kBuckets - 1; b >= 0; b--) {
            if (h_overflow[b]) // only the buckets that have such a pair: an idle launch would still hold the bucket's lane
                ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD_EXACT>(ctx, a, b, st.buckets[b].fwd_list.as<WorkItem>(), h_rerun[b]));
            MyArgs ab = a;
            ab.n_rr = 0;
            int32_t blocks = 0;
            for (auto mc = st.classes.rbegin(); mc != st.classes.rend(); ++mc) { // heaviest class of the bucket first
                const int cls = mc->cls;
                if (bucket_of(cls) != b || mc->le <= mc->lb) continue;
                if (cls == kLongClass) ab.hbase = mc->lb;
                const int per = 64 / mc->G;
                ab.rr[ab.n_rr].cls = cls; ab.rr[ab.n_rr].first_block = blocks; ab.rr[ab.n_rr].b = mc->lb; ab.rr[ab.n_rr].e = mc->le;
                ab.n_rr++;
                blocks += (int32_t)((mc->le - mc->lb + per - 1) / per);
            }
            st.buckets[b].n_rev = blocks;
            ULTRA_HIP_CHECK(ctx, launch_mega<MODE_REV>(ctx, ab, b, nullptr, blocks));
        }
        if (counts_beside) { const int rrc = route_counts(); if (rrc != ULTRA_OK) return rrc; }
        ULTRA_HIP_CHECK(ctx, ultra_join_at(ctx, 3));
    } else {
        // Banded reverse pass, lane per location.  The locations are sorted by (band class, columns) so that the
        // lanes of a wavefront run equally long; what no band class takes (distance > 59) goes through the
        // full-height kernels as dense per-size-class lists.
        const size_t NL = (size_t)st.total_loc;
        ULTRA_HIP_CHECK(ctx, st.startitems.ensure(NL * 4 * 4 + 64)); // keys in | keys out | locations in | locations out
        ULTRA_HIP_CHECK(ctx, st.startinfo.ensure(256));
        uint32_t *k_in = st.startitems.as<uint32_t>(), *k_out = k_in + NL;
        int32_t *v_in = (int32_t *)(k_out + NL), *v_out = v_in + NL;
        // the exact re-runs write end columns the start kernels read: they go first, beside the sort
        ULTRA_HIP_CHECK(ctx, ultra_fork(ctx));
        for (int b = kBuckets - 1; b >= 0; b--)
            if (h_overflow[b])
                ULTRA_HIP_CHECK(ctx, launch_mega<MODE_FWD_EXACT>(ctx, a, b, st.buckets[b].fwd_list.as<WorkItem>(), h_rerun[b]));
        ULTRA_HIP_CHECK(ctx, hipMemsetAsync(st.startinfo.p, 0, 256, s));
        hipLaunchKernelGGL(k_start_keys, dim3((unsigned)((NL + 255) / 256)), dim3(256), 0, s, a, (int64_t)NL, k_in, v_in);
        {
            size_t tmp_bytes = 0;
            ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, NL, 0, kBandLBits + 4, s));
            ULTRA_HIP_CHECK(ctx, st.sorttmp.ensure(tmp_bytes + 64));
            ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(st.sorttmp.p, tmp_bytes, k_in, k_out, v_in, v_out, NL, 0, kBandLBits + 4, s));
        }
        hipLaunchKernelGGL(k_band_bounds, dim3(1), dim3(64), 0, s, k_out, (int64_t)NL, st.startinfo.as<int32_t>());
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
        int32_t h_info[kBandBounds];
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_info, st.startinfo.p, sizeof h_info, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, ultra_join(ctx));
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        const int np = st.S + 1 <= 8 ? 3 : 5;
        const int lanes = ctx->cfg.launch_lanes;
        ULTRA_HIP_CHECK(ctx, ultra_fork(ctx));
        // full-height lists first (few, long wavefronts), on the lanes of their buckets
        for (int b = kBuckets - 1; b >= 0; b--) {
            MyArgs ab = a;
            ab.rev_items = v_out;
            ab.n_rr = 0;
            int32_t blocks = 0;
            for (int cls = kNumSizeClasses - 1; cls >= 0; cls--) {
                if (bucket_of(cls) != b) continue;
                const int64_t lb = h_info[kBandClasses + cls], le = h_info[kBandClasses + cls + 1];
                if (le <= lb) continue;
                const int per = 64 / kClasses[cls].G;
                ab.rr[ab.n_rr].cls = cls; ab.rr[ab.n_rr].first_block = blocks; ab.rr[ab.n_rr].b = lb; ab.rr[ab.n_rr].e = le;
                ab.n_rr++;
                if (cls == kLongClass) ab.hbase = lb;
                blocks += (int32_t)((le - lb + per - 1) / per);
            }
            st.buckets[b].n_rev = blocks;
            ULTRA_HIP_CHECK(ctx, launch_mega<MODE_REV>(ctx, ab, b, nullptr, blocks));
        }
        for (int c = kBandClasses - 1; c >= 0; c--) {
            const int64_t ni = (int64_t)h_info[c + 1] - h_info[c];
            if (ni <= 0) continue;
            const int32_t *items = v_out + h_info[c];
            const int64_t waves = (ni + 63) / 64;
            const dim3 grid((unsigned)std::min<int64_t>(waves, (int64_t)ctx->prop.multiProcessorCount * 64)), block(64);
            hipStream_t cs = ctx->lane(lanes > kBuckets ? kBuckets + c % (lanes - kBuckets) : c); // beside the buckets' full-height lists
            const int WB = band_words(c);
            if (c == 0) {
                if (np == 3) hipLaunchKernelGGL((myers_bstart32<3>), grid, block, 0, cs, a, items, ni);
                else hipLaunchKernelGGL((myers_bstart32<5>), grid, block, 0, cs, a, items, ni);
            } else if (np == 3) {
                switch (WB) {
                case 1: hipLaunchKernelGGL((myers_bstart<1, 3>), grid, block, 0, cs, a, items, ni); break;
                case 2: hipLaunchKernelGGL((myers_bstart<2, 3>), grid, block, 0, cs, a, items, ni); break;
                default: break; // wider classes are not banded in the start pass (kStartBandClasses)
                }
            } else {
                switch (WB) {
                case 1: hipLaunchKernelGGL((myers_bstart<1, 5>), grid, block, 0, cs, a, items, ni); break;
                case 2: hipLaunchKernelGGL((myers_bstart<2, 5>), grid, block, 0, cs, a, items, ni); break;
                default: break; // wider classes are not banded in the start pass (kStartBandClasses)
                }
            }
            ULTRA_HIP_CHECK(ctx, hipGetLastError());
        }
        ULTRA_HIP_CHECK(ctx, ultra_join(ctx));
    }
    ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[2], s));
    {   // from here on the run is latency-bound: another context's planner may queue up behind this event
        std::lock_guard<std::mutex> lk(ctx->tail_mu);
        if (ctx->run_active) { ctx->tail_open = true; ctx->tail_has_event = true; }
    }
    ctx->tail_cv.notify_all();
    // The caller's mid-run work on a lane beside the traceback's routing (two chains of small kernels, the chip is
    // otherwise idle): the main stream takes it back in before the routing's copy to the host.
    bool hook_beside = false;
    if (ctx->mid_hook && !hook_early) {
        hipStream_t hs = ctx->cfg.launch_lanes > 1 ? ctx->aux[0] : s;
        if (hs != s) {
            ULTRA_HIP_CHECK(ctx, hipEventRecord(ctx->fork_ev, s));
            ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(hs, ctx->fork_ev, 0));
        }
        const int hrc = ctx->mid_hook(hs);
        if (hrc != ULTRA_OK) return hrc;
        if (hs != s) { ULTRA_HIP_CHECK(ctx, hipEventRecord(ctx->join_ev[0], hs)); hook_beside = true; }
    }
    // ---- pass 3: first-location traceback ----
    a.ops = want_ops ? st.ops.as<uint8_t>() : nullptr;
    a.ops_off = want_ops ? st.opsoff.as<int64_t>() : nullptr;
    const size_t budget = (size_t)16 << 30; // trace scratch budget (bytes)
    int32_t h_band[2 * (kBandClasses + 1)]; // counts, then longest first location, per band class
    for (auto &v : h_band) v = 0;
    a.nw_items = nullptr;
    bool forked = false;
    int xtra_used = 0; // streams beyond the lanes that carry a full-height launch of this run
    bool band_xtra = false; // ... or the widest band class (the last of those streams)
    bool hook_arrived = false; // a host synchronisation has followed the caller's mid-run work on s
    int32_t full_first[kNumSizeClasses + 1], full_edmax[kNumSizeClasses];
    for (auto &v : full_first) v = 0;
    for (auto &v : full_edmax) v = 0;
    const int32_t *d_items_sorted = nullptr;
    // The banded launches are prepared first (the full-height ones need their sort) but enqueued last: the few,
    // long wavefronts of the full-height kernels and of the wide bands set the length of the phase, and launched
    // behind the thousands of narrow-band wavefronts they would wait for LDS before they even start.
    std::function<int()> launch_bands = [] { return (int)ULTRA_OK; };
    if (use_band) {
        // 3a: banded kernels for every hit whose distance fits a band of at most 1024 rows.  Second half of the routing:
        // the hits sorted inside their classes, behind the start pass and in front of the fork below
        const size_t NH = (size_t)n_hits;
        ULTRA_HIP_CHECK(ctx, st.banditems.ensure(NH * 4 * 4)); // keys in | keys out | items out
        uint32_t *k_in = st.banditems.as<uint32_t>(), *k_out = k_in + NH;
        int32_t *items_sorted = (int32_t *)(k_out + NH);
        hipLaunchKernelGGL(k_band_keys<false>, dim3((unsigned)((n_hits + 255) / 256)), dim3(256), 0, s, a, n_hits, k_in, (int32_t *)nullptr);
        {
            size_t tmp_bytes = 0;
            ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, a.hit_pairs, items_sorted, NH, 0, kTbKeyBits, s));
            ULTRA_HIP_CHECK(ctx, st.sorttmp.ensure(tmp_bytes + 64));
            ULTRA_HIP_CHECK(ctx, rocprim::radix_sort_pairs(st.sorttmp.p, tmp_bytes, k_in, k_out, a.hit_pairs, items_sorted, NH, 0, kTbKeyBits, s));
        }
        if (ultra_hostsync_mask(ctx) >> 4 & 1u) ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        ULTRA_HIP_CHECK(ctx, hipEventSynchronize(st.ev[5])); // the counts: on the host since the start pass began
        {   // hits per group -> first sorted position of every group, then the end
            int32_t acc = 0;
            for (int g = 0; g < kBandBounds; g++) { const int32_t c = g < kBandBounds - 1 ? h_info[g] : 0; h_info[g] = acc; acc += c; }
        }
        int32_t band_first[kBandClasses + 1];
        for (int c = 0; c <= kBandClasses; c++) band_first[c] = h_info[c];
        for (int c = 0; c < kBandClasses; c++) {
            h_band[c] = h_info[c + 1] - h_info[c];
            h_band[kBandClasses + 1 + c] = h_info[kBandBounds + c];
        }
        h_band[kBandFallback] = h_info[kBandBounds - 1] - h_info[kBandClasses];
        for (int c = 0; c <= kNumSizeClasses; c++) full_first[c] = h_info[kBandClasses + c];
        for (int c = 0; c < kNumSizeClasses; c++) full_edmax[c] = h_info[kBandBounds + kBandClasses + c];
        d_items_sorted = items_sorted;
        const int np = st.S + 1 <= 8 ? 3 : 5; // bit-planes of (code + 1)
        // checkpoint scratch (64-bit words), split between the band classes so that their launches can overlap
        int64_t units[kBandClasses], region[kBandClasses];
        size_t total_words = 0;
        for (int c = 0; c < kBandClasses; c++) {
            const int K = band_chunk(c);
            units[c] = region[c] = 0;
            if (h_band[c] <= 0) continue;
            const int64_t chunks = ((int64_t)h_band[kBandClasses + 1 + c] + K - 1) / K;
            region[c] = chunks * (2 + np) * band_rows(c) / band_group(c); // 64-bit words per wavefront: 64 lanes x (rows / 64 / lanes per hit) words each
            const int per_wave = 64 / band_group(c); // hits per wavefront
            const int64_t waves = ((int64_t)h_band[c] + per_wave - 1) / per_wave;
            // Far more wavefronts than fit at once (the LDS of a CU holds ten): short-lived ones keep the eight launches of
            // the phase balanced as slots come free (measured: 20 / 10 per CU 4.25 ms, 80 / 40 4.16, 160 / 80 4.11, 320 / 160
            // 4.11; 4 / 2 - about what is resident - 5.7).  The scratch budget below still caps them.
            int64_t u = std::min<int64_t>(waves, (int64_t)ctx->prop.multiProcessorCount * (c == 0 ? 160 : 80));
            u = std::min<int64_t>(u, std::max<int64_t>(1, (int64_t)((budget / kBandClasses) / (size_t)(region[c] * 8))));
            units[c] = u;
            total_words += (size_t)region[c] * (size_t)u;
        }
        ULTRA_HIP_CHECK(ctx, st.trace.ensure(total_words * 8 + 64));
        ULTRA_HIP_CHECK(ctx, ultra_fork(ctx));
        forked = true;
        std::vector<int64_t> v_units(units, units + kBandClasses), v_region(region, region + kBandClasses);
        std::vector<int32_t> v_first(band_first, band_first + kBandClasses + 1), v_count(h_band, h_band + kBandClasses);
        launch_bands = [=, &st, &band_xtra]() -> int {
        const int64_t *units = v_units.data(), *region = v_region.data();
        const int32_t *band_first = v_first.data(), *h_band = v_count.data();
        size_t at = 0;
        static const DevList dev_order = dev_list("ULTRA_DEV_TB_ORDER"); // band classes in launch order (experiments)
        for (int ci = 0; ci < kBandClasses; ci++) {
            const int c = dev_order.n == kBandClasses && dev_order.v[ci] < kBandClasses ? dev_order.v[ci] : kBandClasses - 1 - ci;
            if (h_band[c] <= 0) continue;
            MyArgs b = a;
            b.tr = (ulonglong2 *)(st.trace.as<uint64_t>() + at);
            b.region = region[c];
            at += (size_t)region[c] * (size_t)units[c];
            const int32_t *items = items_sorted + band_first[c];
            const dim3 grid((unsigned)units[c]), block(64);
            static const DevList dev_tb = dev_list("ULTRA_DEV_TB_LANES"); // lane of band class 0 .. 7
            // With eight lanes the classes go to lanes 4 .. 7, 0 .. 3: measured 0.45 ms shorter than class c on lane c, where
            // the 4-word class regularly started only when the 1-word class had ended (lanes i and i + 4 seem to share a
            // dispatch pipe; scripts/ubench/dispatch.hip, scripts/dev/lane_variants.sh)
            hipStream_t cs = ctx->lane(dev_tb.n == kBandClasses ? dev_tb.v[c] : ctx->cfg.launch_lanes == 8 ? (c + 4) % 8 : c);
            if (c >= 8 && ctx->cfg.launch_lanes == kLanes) { // the ninth class: the last of the streams beyond the lanes
                cs = ctx->xtra[kXtraStreams - 1];
                if (!(xtra_used >> (kXtraStreams - 1) & 1)) ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(cs, ctx->fork_ev, 0));
                band_xtra = true;
            }
            const int64_t ni = h_band[c];
            const int WB = band_words(c);
            const size_t band_lds = (size_t)band_chunk(c) * band_rows(c) * 16; // K columns x {Pv, Ph} of the window per lane
            if (c >= 3) {
                const size_t glds = (size_t)8 * (band_words(c) / band_group(c)) * 64 * 16; // K columns x WBL words x 64 lanes x {Pv, Ph}
#define ULTRA_BANDG(C, NPV) hipLaunchKernelGGL((myers_bandg<band_group(C), band_words(C) / band_group(C), NPV>), grid, block, glds, cs, b, items, ni)
                if (np == 3) {
                    switch (c) {
                    case 3: ULTRA_BANDG(3, 3); break;
                    case 4: ULTRA_BANDG(4, 3); break;
                    case 5: ULTRA_BANDG(5, 3); break;
                    case 6: ULTRA_BANDG(6, 3); break;
                    case 7: ULTRA_BANDG(7, 3); break;
                    default: ULTRA_BANDG(kBandClasses - 1, 3); break;
                    }
                } else {
                    switch (c) {
                    case 3: ULTRA_BANDG(3, 5); break;
                    case 4: ULTRA_BANDG(4, 5); break;
                    case 5: ULTRA_BANDG(5, 5); break;
                    case 6: ULTRA_BANDG(6, 5); break;
                    case 7: ULTRA_BANDG(7, 5); break;
                    default: ULTRA_BANDG(kBandClasses - 1, 5); break;
                    }
                }
#undef ULTRA_BANDG
            } else if (c == 0) {
                if (np == 3) hipLaunchKernelGGL((myers_band32<3>), grid, block, band_lds, cs, b, items, ni);
                else hipLaunchKernelGGL((myers_band32<5>), grid, block, band_lds, cs, b, items, ni);
            } else if (np == 3) {
                switch (WB) {
                case 1: hipLaunchKernelGGL((myers_band<1, 3>), grid, block, band_lds, cs, b, items, ni); break;
                case 2: hipLaunchKernelGGL((myers_band<2, 3>), grid, block, band_lds, cs, b, items, ni); break;
                default: break; // 4 words and up: myers_bandg above
                }
            } else {
                switch (WB) {
                case 1: hipLaunchKernelGGL((myers_band<1, 5>), grid, block, band_lds, cs, b, items, ni); break;
                case 2: hipLaunchKernelGGL((myers_band<2, 5>), grid, block, band_lds, cs, b, items, ni); break;
                default: break; // 4 words and up: myers_bandg above
                }
            }
            ULTRA_HIP_CHECK(ctx, hipGetLastError());
        }
        return ULTRA_OK;
        };
    }
    // 3b: full-height kernels for what is left (distance > 1014, or edit scripts wanted, or banding off).
    //     Their own scratch buffer, split between the classes that have work; the launches overlap with
    //     each other and with the banded ones (these are bound by their critical path, not by the ALUs).
    if (hook_beside) { ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(s, ctx->join_ev[0], 0)); hook_beside = false; }
    const bool need_full = !use_band || h_band[kBandFallback] > 0;
    if (need_full) {
        struct FullPlan { int cls; int64_t region, items, begin, end, Lmax; int max_par; size_t entries; int nbb; };
        std::vector<FullPlan> plans;
        for (auto &mc : st.classes) {
            if (mc.he <= mc.hb) continue;
            FullPlan fp;
            fp.cls = mc.cls;
            fp.begin = mc.hb; fp.end = mc.he;
            if (use_band) { fp.begin = full_first[fp.cls]; fp.end = full_first[fp.cls + 1]; } // dense list of what the banded kernels left
            fp.items = fp.end - fp.begin;
            if (fp.items <= 0) continue;
            plans.push_back(fp);
        }
        size_t total = 0;
        for (auto &fp : plans) {
            const int G = kClasses[fp.cls].G, W = kClasses[fp.cls].W;
            int32_t max_t = 0, max_m = 0;
            for (auto &mc : st.classes) if (mc.cls == fp.cls) { max_t = mc.max_t; max_m = mc.max_m; }
            const int NB = fp.cls == kLongClass ? G * W * ((max_m + kStripRows - 1) / kStripRows) : G * W; // blocks of the tallest query
            // first location: L = end - start + 1 <= m + ed <= m + k, and k <= m (clamped in setup_item)
            fp.Lmax = std::min<int64_t>(max_t, (int64_t)2 * 64 * NB);
            const size_t share = budget / plans.size();
            fp.nbb = 0;
            if (G == 1) {
                fp.region = fp.Lmax * W * 64;
                int64_t waves = (fp.items + 63) / 64;
                int64_t units = std::min<int64_t>(waves, std::max<int64_t>(1, (int64_t)(share / (size_t)(fp.region * 16))));
                units = std::min<int64_t>(units, (int64_t)ctx->prop.multiProcessorCount * 16);
                fp.max_par = (int)units;
                fp.entries = (size_t)fp.region * (size_t)(units + 4); // the grid is rounded up to whole blocks
            } else {
                const int P = 64 / G;
                // distance known (banded mode): blocks crossed by ed + 3 diagonals, + 1 for their alignment
                fp.nbb = use_band ? std::min<int>(NB, (full_edmax[fp.cls] + 3 + 63) / 64 + 1) : NB;
                fp.region = fp.Lmax * fp.nbb;
                int64_t blocks = (fp.items + P - 1) / P;
                int64_t b = std::min<int64_t>(blocks, std::max<int64_t>(1, (int64_t)(share / (size_t)(fp.region * 16 * P))));
                b = std::min<int64_t>(b, (int64_t)ctx->prop.multiProcessorCount * 32);
                fp.max_par = (int)b;
                fp.entries = (size_t)fp.region * (size_t)(b * P);
            }
            total += fp.entries;
        }
        ULTRA_HIP_CHECK(ctx, st.trace2.ensure(total * 16 + 64));
        if (!forked) { ULTRA_HIP_CHECK(ctx, ultra_fork(ctx)); forked = true; }
        size_t at = 0;
        int q = 0;
        for (auto it = plans.rbegin(); it != plans.rend(); ++it, ++q) {
            MyArgs b = a;
            b.tr = st.trace2.as<ulonglong2>() + at;
            at += it->entries;
            b.region = it->region; b.Lmax = (int32_t)it->Lmax; b.nbb = it->nbb;
            b.begin = it->begin; b.end = it->end;
            b.nw_items = use_band ? d_items_sorted : nullptr;
            // a lane of their own while there are lanes; beyond that beside the 1-word band and up, never behind the
            // 32-row band's launch (lane 0, the longest of the phase).  With the full set of lanes (16 hardware queues) and
            // the band classes on them, they go to streams of their own instead.
            int li = kBandClasses + q;
            const int lanes = ctx->cfg.launch_lanes;
            static const bool no_xtra = getenv("ULTRA_DEV_NO_XTRA") != nullptr;
            if (use_band && lanes == kLanes && !no_xtra) {
                const int xi = q % (kXtraStreams - 1); // the last one belongs to the widest band class
                if (!(xtra_used >> xi & 1)) { ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->xtra[xi], ctx->fork_ev, 0)); xtra_used |= 1 << xi; }
                ULTRA_HIP_CHECK(ctx, launch_class<MODE_NW>(ctx->xtra[xi], it->cls, b, it->items, it->max_par));
                continue;
            }
            if (li >= lanes && lanes > 1) li = 1 + (li - lanes) % (lanes - 1);
            ULTRA_HIP_CHECK(ctx, launch_class<MODE_NW>(ctx->lane(li), it->cls, b, it->items, it->max_par));
        }
    }
    {
        const int rc = launch_bands();
        if (rc != ULTRA_OK) return rc;
    }
    if (forked) ULTRA_HIP_CHECK(ctx, ultra_join_at(ctx, 5));
    if (band_xtra) xtra_used |= 1 << (kXtraStreams - 1);
    for (int xi = 0; xi < kXtraStreams; xi++)
        if (xtra_used >> xi & 1) {
            ULTRA_HIP_CHECK(ctx, hipEventRecord(ctx->xtra_ev[xi], ctx->xtra[xi]));
            ULTRA_HIP_CHECK(ctx, hipStreamWaitEvent(s, ctx->xtra_ev[xi], 0));
        }
    ULTRA_HIP_CHECK(ctx, hipEventRecord(st.ev[3], s));
    if (hook_early) { // joined into s with the start pass; its copies have landed once its event has
        ULTRA_HIP_CHECK(ctx, hipEventSynchronize(st.ev[6]));
        hook_arrived = true;
    }
    if (defer_sync && hook_arrived) {
        st.times_pending = true;
        drain.dismiss();
        st.ran = true;
        return ULTRA_OK;
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    {
        const int trc = ultra_myers_collect_times(ctx);
        if (trc != ULTRA_OK) return trc;
    }
    drain.dismiss();
    st.ran = true;
    return ULTRA_OK;
}
