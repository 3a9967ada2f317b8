// MAM materialisation on the device + the fused MAM-stage pipeline of libultra_hip.so (gfx950).
//
// Replaces, for a whole batch of (read, location) instances at once, everything the reference does
// between its edlib calls and its chainer call in
//   /root/reference/modules/classify_read_with_mams.py
//     add_segment_to_mam  :303-402   (which locations become MAMs, FP64 scores)
//     main                :440-555   (phase A, first/last-exon reduction, phase B with its
//                                     de-duplication, outermost-partial filter, stable sort by y)
// and then runs the chaining kernels (chain.hip) on the MAM lists without leaving the GPU.
//
// One wavefront per instance.  All MAMs of one candidate share y (= the segment's stop), so the
// reference's stable sort of MAMs by y is done as a rank computation over the instance's few
// candidates; the MAM records themselves are written once, directly at their sorted position.
#include "common.h"

#include <algorithm>
#include <string.h>

namespace {

enum { MODE_COUNT = 0, MODE_KEPT = 1, MODE_WRITE = 2 };

struct MamArgs {
    // Myers results, indexed by pair (device order)
    const int32_t *ed, *nloc;
    const int64_t *loc_off;
    const int32_t *loc_start, *loc_end, *first_ins;
    // candidates
    const uint8_t *kind;
    const int32_t *e_start, *e_stop, *pair, *pairc, *seq_len, *seq_id;
    const int64_t *cap_off; // [n_cand + 1] slots per candidate = n_loc(pair) + (contained pair ? 1 : 0)
    // instances
    const int64_t *cand_off;
    const int32_t *read_len;
    const int64_t *fps, *lps; // first_part_stop / last_part_start (classify_read_with_mams.py:194-221)
    double min_acc;
    int32_t lds_cap, n_inst;
    // scratch + outputs (slot layout: instance i owns [cap_off[cand_off[i]], cap_off[cand_off[i+1]]))
    uint8_t *flag;
    int32_t *mx, *my, *mc, *md, *mmsl, *mcand;
    double *mval;
    int32_t *n_mam;
};

struct Agg {
    long long ps, pe, mind, maxc; // outermost partial start / end, min d, max c (:530-543)
};

__device__ __forceinline__ bool start_kind(int k) { return k == 2 || k == 4; }
__device__ __forceinline__ bool end_kind(int k) { return k == 3 || k == 5; }

// edlib_alignment + cigar_to_accuracy (:71-119): '=' columns over all columns of the first location's path
__device__ __forceinline__ double pair_accuracy(const MamArgs &a, int p, int &span0)
{
    const int64_t lo = a.loc_off[p];
    span0 = a.loc_end[lo] - a.loc_start[lo] + 1;
    const int ins = a.first_ins[p];
    return (double)(span0 - a.ed[p] + ins) / (double)(span0 + ins);
}

// One candidate = one call of add_segment_to_mam (:303-402).
//   MODE_COUNT: decides which locations become MAMs (flags), returns their number, updates the aggregates
//   MODE_KEPT : number of flagged MAMs that survive the outermost-partial filter (:547)
//   MODE_WRITE: writes the surviving MAMs from slot `o` on, returns how many
// The fields of one candidate every pass over it starts from (an instance of at most 64 candidates keeps its
// lane's candidate in registers across the passes instead of re-reading it through two levels of loads).
struct CandData {
    int kind, es, ee, p, pc, nl, ed, seq_len;
    int64_t lo, cap_off;
    // accuracies of the candidate's pair and of its contained-read pair (with the latter's first span), computed once
    // per load: every pass over the candidate needs them, each behind four dependent loads and an FP64 division
    double acc, acc_c;
    int s0_c, ed_c;
};
__device__ __forceinline__ CandData load_cand(const MamArgs &a, int c)
{
    CandData d;
    d.kind = a.kind[c]; d.es = a.e_start[c]; d.ee = a.e_stop[c]; d.p = a.pair[c]; d.pc = a.pairc[c];
    d.seq_len = a.seq_len[c]; d.cap_off = a.cap_off[c];
    d.nl = 0; d.ed = -1; d.lo = 0;
    d.acc = 0.0; d.acc_c = 0.0; d.s0_c = 0; d.ed_c = -1;
    if (d.p >= 0) {
        d.nl = a.nloc[d.p]; d.ed = a.ed[d.p]; d.lo = a.loc_off[d.p];
        if (d.ed >= 0 && d.ee - d.es >= 9) { int s0; d.acc = pair_accuracy(a, d.p, s0); }
    }
    if (d.pc >= 0) {
        d.ed_c = a.ed[d.pc];
        if (d.ed_c >= 0) d.acc_c = pair_accuracy(a, d.pc, d.s0_c);
    }
    return d;
}

template <int MODE>
__device__ int walk(const MamArgs &a, const CandData &cd, int c, int read_len, Agg &g, long long ps, long long pe, int64_t o)
{
    const int kind = cd.kind;
    const int es = cd.es, ee = cd.ee;
    const int width = ee - es;
    const bool st = start_kind(kind), en = end_kind(kind);
    uint8_t *flag = a.flag + cd.cap_off;
    const int p = cd.p;
    int cnt = 0, nl = 0;
    auto take = [&](int s, int e, double val, int msl) {
        if (MODE == MODE_COUNT) {
            if (e < g.mind) g.mind = e;
            if (s > g.maxc) g.maxc = s;
            if (st && s < g.ps) g.ps = s;
            if (en && e > g.pe) g.pe = e;
            cnt++;
        } else {
            if (st && s > ps) return;
            if (en && e < pe) return;
            if (MODE == MODE_WRITE) {
                const int64_t w = o + cnt;
                a.mx[w] = es; a.my[w] = ee; a.mc[w] = s; a.md[w] = e; a.mval[w] = val; a.mmsl[w] = msl; a.mcand[w] = c;
            }
            cnt++;
        }
    };
    if (p >= 0) {
        nl = cd.nl;
        const int ed = cd.ed;
        const int64_t lo = cd.lo;
        if (width >= 9) { // approximate hit (:311-349)
            bool ok = false;
            double acc = 0.0;
            if (ed >= 0) { acc = cd.acc; ok = acc > a.min_acc; }
            double last = 0.0;
            int smax = -1;
            for (int i = 0; i < nl; i++) {
                if (!ok) { if (MODE == MODE_COUNT) flag[i] = 0; continue; }
                const int s = a.loc_start[lo + i], e = a.loc_end[lo + i];
                const int span = e - s + 1;
                const double fr = (double)(span - ed) / (double)span;
                const double score = (acc > fr ? acc : fr) * (double)span;
                if (MODE == MODE_COUNT) {
                    // "start in considered_starts": every earlier location's start is in the set
                    bool seen = false;
                    if (s == smax) seen = true;
                    else if (s < smax) { for (int t = 0; t < i; t++) if (a.loc_start[lo + t] == s) { seen = true; break; } }
                    if (s > smax) smax = s;
                    if (seen && score <= last) { flag[i] = 0; continue; }
                    flag[i] = 1;
                    last = score;
                    take(s, e, score, span);
                } else if (flag[i]) take(s, e, score, span);
            }
        } else { // 5..8 bp: exact hits only (:352-369)
            const bool ok = ed == 0;
            int smax = -1;
            for (int i = 0; i < nl; i++) {
                if (!ok) { if (MODE == MODE_COUNT) flag[i] = 0; continue; }
                const int s = a.loc_start[lo + i], e = a.loc_end[lo + i];
                if (MODE == MODE_COUNT) {
                    bool seen = false;
                    if (s == smax) seen = true;
                    else if (s < smax) { for (int t = 0; t < i; t++) if (a.loc_start[lo + t] == s) { seen = true; break; } }
                    if (s > smax) smax = s;
                    flag[i] = seen ? 0 : 1;
                    if (!seen) take(s, e, (double)cd.seq_len, 0);
                } else if (flag[i]) take(s, e, (double)cd.seq_len, 0);
            }
        }
    }
    const int pc = cd.pc;
    if (pc >= 0) { // read contained in the segment (:374-402)
        if (MODE == MODE_COUNT) flag[nl] = (cd.ed_c >= 0 && cd.acc_c > a.min_acc) ? 1 : 0;
        if (flag[nl]) take(0, read_len - 1, cd.acc_c * (double)cd.s0_c, cd.s0_c);
    }
    return cnt;
}

// reductions over the VW lanes of one instance (aligned groups of a wavefront: the butterfly stays inside)
template <int VW> __device__ __forceinline__ long long wave_min_ll(long long v)
{
#pragma unroll
    for (int o = VW / 2; o > 0; o >>= 1) { long long w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
template <int VW> __device__ __forceinline__ long long wave_max_ll(long long v)
{
#pragma unroll
    for (int o = VW / 2; o > 0; o >>= 1) { long long w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
template <int VW> __device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int o = VW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// VW lanes per instance: 32 (two instances share a wavefront; an instance has ~18 candidates) unless the largest
// instance's candidate arrays would not fit LDS twice, then 64.
// wavefronts per SIMD the register allocation aims at: 4 = 106 VGPRs, no spills.  Measured in round 3: 5 (96 VGPRs, 15
// spilled) 1.59 -> 1.92 ms per cDNA step, 6 (80 VGPRs, 69 spilled) 2.42 ms.
#ifndef ULTRA_MAM_WAVES
#define ULTRA_MAM_WAVES 4
#endif
template <int VW>
__global__ void __launch_bounds__(64, ULTRA_MAM_WAVES) k_mam_build(MamArgs a)
{
    extern __shared__ int32_t lds_all[];
    const int lane = threadIdx.x % VW, part = threadIdx.x / VW;
    const int inst_raw = blockIdx.x * (64 / VW) + part;
    const bool live = inst_raw < a.n_inst;
    const int inst = live ? inst_raw : 0;
    int32_t *lds = lds_all + (size_t)part * 7 * a.lds_cap;
    const int64_t c0 = a.cand_off[inst];
    const int C = live ? (int)(a.cand_off[inst + 1] - c0) : 0;
    int32_t *ly = lds, *lkept = lds + a.lds_cap, *lact = lds + 2 * a.lds_cap;
    // what the serial selection loop reads per candidate: kind | e_start | seq_len | seq_id (from LDS, not as a
    // chain of dependent global loads on one lane)
    int32_t *lkind = lds + 3 * a.lds_cap, *lstart = lds + 4 * a.lds_cap, *lslen = lds + 5 * a.lds_cap, *lsid = lds + 6 * a.lds_cap;
    const int read_len = a.read_len[inst];
    const long long BIG = 1LL << 32;
    Agg g = {BIG, 0, BIG, 0};
    long long miny = BIG * 4, maxx = -1;
    // at most two candidates per lane (92 % of the bench workload's instances with VW = 32): loaded once, kept in
    // registers across the passes instead of re-read through two levels of dependent loads in each
    const bool kept = C <= 2 * VW;
    CandData mine = {}, mine2 = {};
    if (kept && lane < C) mine = load_cand(a, (int)(c0 + lane));
    if (kept && lane + VW < C) mine2 = load_cand(a, (int)(c0 + lane + VW));
    auto cand = [&](int i, int c) { return kept ? (i < VW ? mine : mine2) : load_cand(a, c); };
    // ---- phase A: full segments and flanks (:441-458) ----
    for (int i = lane; i < C; i += VW) {
        const int c = (int)(c0 + i);
        const CandData cd = cand(i, c);
        const int kind = cd.kind;
        ly[i] = cd.ee;
        lkind[i] = kind; lstart[i] = cd.es; lslen[i] = cd.seq_len; lsid[i] = a.seq_id[c];
        int cnt = 0;
        if (kind <= 1) {
            cnt = walk<MODE_COUNT>(a, cd, c, read_len, g, 0, 0, 0);
            if (cnt > 0) {
                if (cd.ee < miny) miny = cd.ee;
                if (cd.es > maxx) maxx = cd.es;
            }
        }
        lkept[i] = cnt;
        lact[i] = kind <= 1 ? 1 : 0;
    }
    miny = wave_min_ll<VW>(miny);
    maxx = wave_max_ll<VW>(maxx);
    const bool any_a = maxx >= 0;
    // :469-476
    const long long fps = a.fps[inst], lps = a.lps[inst];
    const long long first_stop = any_a ? (fps > miny ? fps : miny) : (fps > -1 ? fps : -1);
    const long long last_start = any_a ? (lps < maxx ? lps : maxx) : (lps < BIG ? lps : BIG);
    __syncthreads();
    // ---- phase B selection, in candidate order (:479-523) ----
    if (lane == 0) {
        int seg0 = -1, fl0 = -1; // first candidate of the partial-segment / partial-flank loops
        for (int i = 0; i < C;) {
            const int kind = lkind[i];
            if (kind == 2) {
                if (seg0 < 0) seg0 = i;
                int pick = -1;
                if ((long long)ly[i] <= first_stop) pick = i;
                else if (last_start <= (long long)lstart[i]) pick = i + 1;
                if (pick >= 0 && lslen[pick] > 5) {
                    const int id = lsid[pick];
                    bool tried = false;
                    for (int t = seg0; t < i; t++) if (lact[t] && lsid[t] == id) { tried = true; break; }
                    if (!tried) lact[pick] = 1;
                }
                i += 2;
            } else if (kind >= 4) {
                if (fl0 < 0) fl0 = i;
                if (lslen[i] > 5) {
                    const int id = lsid[i];
                    bool tried = false;
                    for (int t = fl0; t < i; t++) if (lact[t] && lsid[t] == id) { tried = true; break; }
                    if (!tried) lact[i] = 1;
                }
                i++;
            } else i++;
        }
    }
    __syncthreads();
    // ---- phase B emission ----
    for (int i = lane; i < C; i += VW) {
        const int c = (int)(c0 + i);
        if (lkind[i] >= 2 && lact[i]) lkept[i] = walk<MODE_COUNT>(a, cand(i, c), c, read_len, g, 0, 0, 0);
    }
    // ---- outermost-partial filter (:530-547) ----
    long long ps = wave_min_ll<VW>(g.ps), pe = wave_max_ll<VW>(g.pe);
    const long long mind = wave_min_ll<VW>(g.mind), maxc = wave_max_ll<VW>(g.maxc);
    ps = ps < mind ? ps : mind;
    pe = pe > maxc ? pe : maxc;
    for (int i = lane; i < C; i += VW) {
        const int c = (int)(c0 + i);
        if (lkind[i] >= 2 && lkept[i] > 0) lkept[i] = walk<MODE_KEPT>(a, cand(i, c), c, read_len, g, ps, pe, 0);
    }
    __syncthreads();
    // ---- stable sort by y (:553) as a rank over candidates, then the MAM records ----
    const int64_t base = a.cap_off[c0];
    int total = 0;
    for (int i = lane; i < C; i += VW) {
        const int k = lkept[i];
        if (k <= 0) continue;
        total += k;
        const int yi = ly[i];
        int off = 0;
        for (int t = 0; t < C; t++) {
            const int yt = ly[t];
            if (yt < yi || (yt == yi && t < i)) off += lkept[t];
        }
        walk<MODE_WRITE>(a, cand(i, (int)(c0 + i)), (int)(c0 + i), read_len, g, ps, pe, base + off);
    }
    total = wave_sum_i32<VW>(total);
    if (lane == 0 && live) a.n_mam[inst] = total;
}

__global__ void k_cand_cap(const int32_t *pair, const int32_t *pairc, const int32_t *nloc, uint64_t *cap, int32_t n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cap[i] = (uint64_t)((pair[i] >= 0 ? nloc[pair[i]] : 0) + (pairc[i] >= 0 ? 1 : 0));
}
__global__ void k_remap_pairs(int32_t *pair, int32_t *pairc, const int32_t *inv, int32_t n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (pair[i] >= 0) pair[i] = inv[pair[i]];
    if (pairc[i] >= 0) pairc[i] = inv[pairc[i]];
}
// closes an exclusive scan: out[n] = out[n-1] + in[n-1]
__global__ void k_scan_close(const uint64_t *in, uint64_t *out, int64_t n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[n] = n > 0 ? out[n - 1] + in[n - 1] : 0;
}
__global__ void k_gather_i64(const int64_t *src, const int64_t *pos, int64_t *dst, int32_t n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[pos[i]];
}
// out[0] += sum of v (the batch's MAM total; the per-instance counts stay on the device until a fetch wants them).  out[0]
// is zeroed long before (behind k_mam_build's launch).  One block of 1024 threads took 0.1 ms at the very end of a run.
__global__ void __launch_bounds__(256) k_sum_i32(const int32_t *v, int32_t n, unsigned long long *out)
{
    __shared__ long long part[256];
    long long acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += v[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0] != 0) atomicAdd(out, (unsigned long long)part[0]);
}
__global__ void k_widen(const int32_t *in, uint64_t *out, int32_t n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint64_t)(uint32_t)in[i];
}
// compacts every instance's chain (rows picked by the chainer) into the fetch arrays
struct GatherArgs {
    const int64_t *inst_base, *chain_off;
    const int32_t *chain_idx, *chain_len;
    const int32_t *mx, *my, *mc, *md, *mmsl, *mcand;
    const double *mval;
    int32_t *cx, *cy, *cc, *cd, *cmsl, *ccand, *cj;
    double *cval;
    int32_t n_inst;
    const uint8_t *kind; const int32_t *chr; // per candidate; chr may be null
    uint8_t *ckind; int32_t *cchr;
};
__global__ void __launch_bounds__(256) k_chain_gather(GatherArgs a)
{
    const int lane = threadIdx.x & 63;
    const int inst = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (inst >= a.n_inst) return;
    const int64_t b = a.inst_base[inst], o = a.chain_off[inst];
    const int len = a.chain_len[inst];
    for (int t = lane; t < len; t += 64) {
        const int j = a.chain_idx[b + t];
        const int64_t r = b + j;
        a.cj[o + t] = j;
        a.cx[o + t] = a.mx[r]; a.cy[o + t] = a.my[r]; a.cc[o + t] = a.mc[r]; a.cd[o + t] = a.md[r];
        const int cand = a.mcand[r];
        a.cval[o + t] = a.mval[r]; a.cmsl[o + t] = a.mmsl[r]; a.ccand[o + t] = cand;
        a.ckind[o + t] = a.kind[cand];
        a.cchr[o + t] = a.chr ? a.chr[cand] : -1;
    }
}

} // namespace

void ultra_mam_free(ultra_ctx *ctx)
{
    MamState &m = ctx->mam;
    DevBuf *bufs[] = {&m.kind, &m.estart, &m.estop, &m.pair, &m.pairc, &m.seqlen, &m.seqid, &m.candoff, &m.readlen, &m.fps, &m.lps,
                      &m.cap, &m.capoff, &m.instbase, &m.scantmp, &m.flag, &m.mx, &m.my, &m.mc, &m.md, &m.mmsl, &m.mcand, &m.mval,
                      &m.nmam, &m.chainidx, &m.chainlen, &m.value, &m.lenw, &m.chainoff, &m.cx, &m.cy, &m.cc, &m.cd, &m.cmsl,
                      &m.ccand, &m.cj, &m.cval, &m.chr, &m.ckind, &m.cchr, &m.d_namoff, &m.d_readoff, &m.d_nampart, &m.d_namx, &m.d_namy,
                      &m.p_ncand, &m.p_npair, &m.p_over, &m.p_misc, &m.p_wide, &m.p_poff, &m.p_state, &m.p_state_mid, &m.p_state_big, &m.total};
    for (DevBuf *b : bufs) b->release();
    for (auto &e : m.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (m.pinned_base) { (void)hipHostFree(m.pinned_base); m.pinned_base = nullptr; m.pinned_cap = 0; }
}

void ultra_mam_remap_pairs(ultra_ctx *ctx, int32_t nc)
{
    if (nc <= 0) return;
    MamState &m = ctx->mam;
    hipLaunchKernelGGL(k_remap_pairs, dim3((nc + 255) / 256), dim3(256), 0, ctx->stream, m.pair.as<int32_t>(), m.pairc.as<int32_t>(),
                       ctx->my.inv_d.as<int32_t>(), nc);
}

extern "C" int ultra_mam_stage(ultra_ctx *ctx, const ultra_mam_input *in)
{
    if (!ctx || !in) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    m.staged = false; m.ran = false;
    const int32_t nc = in->n_cand, ni = in->n_inst, np = in->n_pairs;
    if (nc < 0 || ni < 0 || np < 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: negative count");
    if (ni > 0 && (!in->inst_cand_off || !in->inst_read_len || !in->inst_first_part_stop || !in->inst_last_part_start))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: null instance array");
    if (nc > 0 && (!in->cand_kind || !in->cand_e_start || !in->cand_e_stop || !in->cand_pair || !in->cand_pair_contained ||
                   !in->cand_seq_len || !in->cand_seq_id))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: null candidate array");
    // ---- validate the candidate lists (device code trusts them) ----
    if (ni > 0 && (in->inst_cand_off[0] != 0 || in->inst_cand_off[ni] != nc))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: inst_cand_off must run from 0 to n_cand");
    int32_t max_c = 0;
    for (int32_t i = 0; i < ni; i++) {
        const int64_t b = in->inst_cand_off[i], e = in->inst_cand_off[i + 1];
        if (e < b || e > nc) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: inst_cand_off must be non-decreasing");
        if (in->inst_read_len[i] < 1) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: read length < 1");
        max_c = std::max<int32_t>(max_c, (int32_t)std::min<int64_t>(e - b, 0x7fffffff));
        int phase = 0; // kinds must come as 0* 1* (2 3)* (4|5)*
        static const int kRank[6] = {0, 1, 2, 2, 3, 3};
        for (int64_t c = b; c < e; c++) {
            const int k = in->cand_kind[c];
            bool ok = k <= 5 && kRank[k] >= phase;
            if (ok && k == 2) ok = c + 1 < e && in->cand_kind[c + 1] == 3 && in->cand_e_start[c] == in->cand_e_start[c + 1] &&
                                   in->cand_e_stop[c] == in->cand_e_stop[c + 1];
            if (ok && k == 3) ok = c > b && in->cand_kind[c - 1] == 2;
            if (!ok) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: candidate kinds of instance " + std::to_string(i) +
                                                             " are not in the order full segments, full flanks, (start,end) partial segment pairs, partial flanks");
            phase = kRank[k];
            if (in->cand_pair[c] < -1 || in->cand_pair[c] >= np || in->cand_pair_contained[c] < -1 || in->cand_pair_contained[c] >= np)
                return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: candidate " + std::to_string(c) + " refers to a pair outside the batch");
            if (in->cand_pair[c] >= 0 && in->cand_e_stop[c] - in->cand_e_start[c] < 5)
                return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_stage: candidate " + std::to_string(c) + " is narrower than 5 bp but has a pair");
        }
    }
    if ((size_t)max_c * 28 > (size_t)160 * 1024 - 64)
        return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_mam_stage: more than 5849 candidates in one instance");
    int rc = ultra_myers_stage_impl(ctx, in->arena, in->arena_len, np, in->q_off, in->q_len, in->t_off, in->t_len, in->k);
    if (rc != ULTRA_OK) return rc;
    hipStream_t s = ctx->stream;
    const size_t NC = (size_t)std::max(nc, 1), NI = (size_t)std::max(ni, 1);
    ULTRA_HIP_CHECK(ctx, m.kind.ensure(NC)); ULTRA_HIP_CHECK(ctx, m.estart.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.estop.ensure(NC * 4));
    ULTRA_HIP_CHECK(ctx, m.pair.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.pairc.ensure(NC * 4));
    ULTRA_HIP_CHECK(ctx, m.seqlen.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.seqid.ensure(NC * 4)); ULTRA_HIP_CHECK(ctx, m.chr.ensure(NC * 4));
    ULTRA_HIP_CHECK(ctx, m.cap.ensure((NC + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.capoff.ensure((NC + 1) * 8));
    ULTRA_HIP_CHECK(ctx, m.candoff.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.readlen.ensure(NI * 4));
    ULTRA_HIP_CHECK(ctx, m.fps.ensure(NI * 8)); ULTRA_HIP_CHECK(ctx, m.lps.ensure(NI * 8));
    ULTRA_HIP_CHECK(ctx, m.instbase.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.nmam.ensure(NI * 4));
    ULTRA_HIP_CHECK(ctx, m.chainlen.ensure(NI * 4 + 16)); ULTRA_HIP_CHECK(ctx, m.value.ensure(NI * 8));
    ULTRA_HIP_CHECK(ctx, m.lenw.ensure((NI + 1) * 8)); ULTRA_HIP_CHECK(ctx, m.chainoff.ensure((NI + 1) * 8));
    if (nc > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.kind.p, in->cand_kind, NC, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.estart.p, in->cand_e_start, NC * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.estop.p, in->cand_e_stop, NC * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.pair.p, in->cand_pair, NC * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.pairc.p, in->cand_pair_contained, NC * 4, hipMemcpyHostToDevice, s));
        ultra_mam_remap_pairs(ctx, nc); // candidate -> device pair order (the pairs were sorted on the device)
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.seqlen.p, in->cand_seq_len, NC * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.seqid.p, in->cand_seq_id, NC * 4, hipMemcpyHostToDevice, s));
        if (in->cand_chr) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.chr.p, in->cand_chr, NC * 4, hipMemcpyHostToDevice, s));
    }
    m.has_chr = in->cand_chr != nullptr;
    if (ni > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.candoff.p, in->inst_cand_off, ((size_t)ni + 1) * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.readlen.p, in->inst_read_len, (size_t)ni * 4, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.fps.p, in->inst_first_part_stop, (size_t)ni * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.lps.p, in->inst_last_part_start, (size_t)ni * 8, hipMemcpyHostToDevice, s));
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    for (auto &e : m.ev) if (!e) ULTRA_HIP_CHECK(ctx, hipEventCreate(&e));
    m.n_cand = nc; m.n_inst = ni; m.max_cand = max_c;
    m.staged = true;
    return ULTRA_OK;
}

namespace {

void tail_state(ultra_ctx *ctx, bool active, bool open, bool has_event)
{
    {
        std::lock_guard<std::mutex> lk(ctx->tail_mu);
        ctx->run_active = active; ctx->tail_open = open; ctx->tail_has_event = has_event;
    }
    ctx->tail_cv.notify_all();
}

// Everything of a run is enqueued; what is left for mam_run_finish is the last synchronisation and the closing counts.
int mam_run_enqueue(ultra_ctx *ctx, double min_acc)
{
    MamState &m = ctx->mam;
    m.ran = false;
    m.total_cap = m.total_chain = m.total_mam = 0;
    for (auto &v : m.ms) v = 0;
    const int32_t nc = m.n_cand, ni = m.n_inst;
    MyersState &my = ctx->my;
    hipStream_t s = ctx->stream;
    if (m.pinned_cap < (size_t)ni + 1 + 3) {
        if (m.pinned_base) { (void)hipHostFree(m.pinned_base); m.pinned_base = nullptr; m.pinned_cap = 0; }
        ULTRA_HIP_CHECK(ctx, hipHostMalloc((void **)&m.pinned_base, ((size_t)ni + 1 + 64) * 8, hipHostMallocDefault));
        m.pinned_cap = (size_t)ni + 1 + 64;
    }
    m.h_instbase.clear();
    // ---- slot layout: one slot per location of a candidate's pair (+1 for its contained-read pair).  It only needs
    //      the forward pass, so it is enqueued from inside the Myers run (behind the start pass, on the stream the run
    //      names: a lane beside the traceback's routing) and its totals come back with that run's next host
    //      synchronisation. ----
    ctx->mid_hook = [ctx, &m, &my, nc, ni](hipStream_t s) -> int {
        if (ni == 0) return ULTRA_OK;
        if (nc > 0) {
            hipLaunchKernelGGL(k_cand_cap, dim3((nc + 255) / 256), dim3(256), 0, s, m.pair.as<int32_t>(), m.pairc.as<int32_t>(),
                               my.nloc.as<int32_t>(), m.cap.as<uint64_t>(), nc);
            ULTRA_HIP_CHECK(ctx, ultra_scan_u64(ctx, m.cap.as<uint64_t>(), m.capoff.as<uint64_t>(), nc, m.scantmp, s));
            hipLaunchKernelGGL(k_scan_close, dim3(1), dim3(64), 0, s, m.cap.as<uint64_t>(), m.capoff.as<uint64_t>(), (int64_t)nc);
            hipLaunchKernelGGL(k_gather_i64, dim3((ni + 1 + 255) / 256), dim3(256), 0, s, m.capoff.as<int64_t>(), m.candoff.as<int64_t>(),
                               m.instbase.as<int64_t>(), ni + 1);
            ULTRA_HIP_CHECK(ctx, hipGetLastError());
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.pinned_base, m.instbase.p, ((size_t)ni + 1) * 8, hipMemcpyDeviceToHost, s));
        } else {
            ULTRA_HIP_CHECK(ctx, hipMemsetAsync(m.instbase.p, 0, ((size_t)ni + 1) * 8, s));
            ULTRA_HIP_CHECK(ctx, hipMemsetAsync(m.capoff.p, 0, 8, s));
            memset(m.pinned_base, 0, ((size_t)ni + 1) * 8); // no candidates: no slots, and no copy that would say so
        }
        return ULTRA_OK;
    };
    int rc = ultra_myers_run_impl(ctx, false, true); // returns with the traceback in flight: k_mam_build queues behind it
    ctx->mid_hook = nullptr;
    if (rc != ULTRA_OK) return rc;
    ultra_drain_guard drain(ctx);
    if (ni == 0) {
        drain.dismiss();
        return ULTRA_OK;
    }
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ULTRA_HIP_CHECK(ctx, hipEventRecord(m.ev[0], s));
    m.total_cap = m.pinned_base[ni];
    const size_t T = (size_t)std::max<int64_t>(m.total_cap, 1);
    ULTRA_HIP_CHECK(ctx, m.flag.ensure(T + 16));
    ULTRA_HIP_CHECK(ctx, m.mx.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.my.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.mc.ensure(T * 4));
    ULTRA_HIP_CHECK(ctx, m.md.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.mmsl.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.mcand.ensure(T * 4));
    ULTRA_HIP_CHECK(ctx, m.mval.ensure(T * 8)); ULTRA_HIP_CHECK(ctx, m.chainidx.ensure(T * 4));
    ULTRA_HIP_CHECK(ctx, m.cx.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.cy.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.cc.ensure(T * 4));
    ULTRA_HIP_CHECK(ctx, m.cd.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.cmsl.ensure(T * 4)); ULTRA_HIP_CHECK(ctx, m.ccand.ensure(T * 4));
    ULTRA_HIP_CHECK(ctx, m.cval.ensure(T * 8)); ULTRA_HIP_CHECK(ctx, m.cj.ensure(T * 4));
    ULTRA_HIP_CHECK(ctx, m.ckind.ensure(T)); ULTRA_HIP_CHECK(ctx, m.cchr.ensure(T * 4));
    // ---- MAM lists ----
    MamArgs a;
    memset(&a, 0, sizeof a);
    a.ed = my.ed.as<int32_t>(); a.nloc = my.nloc.as<int32_t>(); a.loc_off = my.locoff.as<int64_t>();
    a.loc_start = my.locstart.as<int32_t>(); a.loc_end = my.locend.as<int32_t>(); a.first_ins = my.firstins.as<int32_t>();
    a.kind = m.kind.as<uint8_t>(); a.e_start = m.estart.as<int32_t>(); a.e_stop = m.estop.as<int32_t>();
    a.pair = m.pair.as<int32_t>(); a.pairc = m.pairc.as<int32_t>(); a.seq_len = m.seqlen.as<int32_t>(); a.seq_id = m.seqid.as<int32_t>();
    a.cap_off = m.capoff.as<int64_t>(); a.cand_off = m.candoff.as<int64_t>(); a.read_len = m.readlen.as<int32_t>();
    a.fps = m.fps.as<int64_t>(); a.lps = m.lps.as<int64_t>(); a.min_acc = min_acc;
    a.lds_cap = std::max<int32_t>(m.max_cand, 1);
    a.n_inst = ni;
    a.flag = m.flag.as<uint8_t>();
    a.mx = m.mx.as<int32_t>(); a.my = m.my.as<int32_t>(); a.mc = m.mc.as<int32_t>(); a.md = m.md.as<int32_t>();
    a.mmsl = m.mmsl.as<int32_t>(); a.mcand = m.mcand.as<int32_t>(); a.mval = m.mval.as<double>(); a.n_mam = m.nmam.as<int32_t>();
    {
        size_t shmem = (size_t)a.lds_cap * 28;
        if (shmem * 2 <= 64 * 1024) {
            hipLaunchKernelGGL(k_mam_build<32>, dim3((unsigned)((ni + 1) / 2)), dim3(64), shmem * 2, s, a);
        } else {
            if (shmem > 64 * 1024)
                ULTRA_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)k_mam_build<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL(k_mam_build<64>, dim3((unsigned)ni), dim3(64), shmem, s, a);
        }
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
    }
    ULTRA_HIP_CHECK(ctx, hipEventRecord(m.ev[1], s));
    ULTRA_HIP_CHECK(ctx, m.total.ensure(8));
    ULTRA_HIP_CHECK(ctx, hipMemsetAsync(m.total.p, 0, 8, s)); // the MAM total is summed into it at the end of the run
    // ---- chaining (solver switch of classify_read_with_mams.py:560-575) ----
    // The host sorts the instances into launches by size while k_mam_build runs.  An instance cannot hold more MAMs
    // than it has slots, and the slot counts are on the host already: the launches are made up from these bounds, an
    // instance whose bound passes the solver switch is listed for both solvers and the kernels pick it up by its
    // actual count on the device.  Only when a limit fails on the bounds are the exact counts read back first.
    m.h_nmam.resize((size_t)ni);
    m.h_slots.resize((size_t)ni);
    for (int32_t i = 0; i < ni; i++) m.h_slots[(size_t)i] = (int32_t)(m.pinned_base[i + 1] - m.pinned_base[i]);
    m.exact_needed = false;
    ULTRA_HIP_CHECK(ctx, hipEventRecord(m.ev[2], s));
    int32_t *d_status = nullptr;
    rc = ultra_chain_launch(ctx, ni, m.h_slots.data(), m.instbase.as<int64_t>(), m.nmam.as<int32_t>(),
                            a.mx, a.my, a.mc, a.md, a.mval,
                            0, -1, -1, m.chainidx.as<int32_t>(), m.chainlen.as<int32_t>(), m.value.as<double>(), &d_status, nullptr, nullptr, true);
    if (rc == kChainNeedExact) {
        m.exact_needed = true;
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.h_nmam.data(), m.nmam.p, (size_t)ni * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        ULTRA_HIP_CHECK(ctx, hipEventRecord(m.ev[2], s));
        rc = ultra_chain_launch(ctx, ni, m.h_nmam.data(), m.instbase.as<int64_t>(), m.nmam.as<int32_t>(),
                                a.mx, a.my, a.mc, a.md, a.mval,
                                0, -1, -1, m.chainidx.as<int32_t>(), m.chainlen.as<int32_t>(), m.value.as<double>(), &d_status);
    }
    if (rc != ULTRA_OK) { (void)hipStreamSynchronize(s); return rc; }
    ULTRA_HIP_CHECK(ctx, hipEventRecord(m.ev[3], s));
    // ---- compact the chains for the fetch ----
    hipLaunchKernelGGL(k_widen, dim3((ni + 255) / 256), dim3(256), 0, s, m.chainlen.as<int32_t>(), m.lenw.as<uint64_t>(), ni);
    ULTRA_HIP_CHECK(ctx, ultra_scan_u64(ctx, m.lenw.as<uint64_t>(), m.chainoff.as<uint64_t>(), ni, m.scantmp));
    hipLaunchKernelGGL(k_scan_close, dim3(1), dim3(64), 0, s, m.lenw.as<uint64_t>(), m.chainoff.as<uint64_t>(), (int64_t)ni);
    GatherArgs g;
    g.inst_base = m.instbase.as<int64_t>(); g.chain_off = m.chainoff.as<int64_t>();
    g.chain_idx = m.chainidx.as<int32_t>(); g.chain_len = m.chainlen.as<int32_t>();
    g.mx = a.mx; g.my = a.my; g.mc = a.mc; g.md = a.md; g.mmsl = a.mmsl; g.mcand = a.mcand; g.mval = a.mval;
    g.cx = m.cx.as<int32_t>(); g.cy = m.cy.as<int32_t>(); g.cc = m.cc.as<int32_t>(); g.cd = m.cd.as<int32_t>();
    g.cmsl = m.cmsl.as<int32_t>(); g.ccand = m.ccand.as<int32_t>(); g.cval = m.cval.as<double>(); g.cj = m.cj.as<int32_t>(); g.n_inst = ni;
    g.kind = m.kind.as<uint8_t>(); g.chr = m.has_chr ? m.chr.as<int32_t>() : nullptr; g.ckind = m.ckind.as<uint8_t>(); g.cchr = m.cchr.as<int32_t>();
    hipLaunchKernelGGL(k_chain_gather, dim3((ni + 3) / 4), dim3(256), 0, s, g);
    ULTRA_HIP_CHECK(ctx, hipGetLastError());
    ULTRA_HIP_CHECK(ctx, hipEventRecord(m.ev[4], s));
    hipLaunchKernelGGL(k_sum_i32, dim3((unsigned)std::min(256, (ni + 255) / 256)), dim3(256), 0, s, m.nmam.as<int32_t>(), ni, m.total.as<unsigned long long>());
    ULTRA_HIP_CHECK(ctx, hipGetLastError());
    // the closing counts land in page-locked memory behind the slot offsets (a copy into pageable memory would hold the
    // host here until the whole run has been carried out)
    int64_t *h_close = m.pinned_base + (size_t)ni + 1;
    h_close[0] = h_close[1] = h_close[2] = 0;
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_close[0], m.chainoff.as<int64_t>() + ni, 8, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_close[1], m.total.p, 8, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(&h_close[2], d_status, 4, hipMemcpyDeviceToHost, s));
    drain.dismiss();
    return ULTRA_OK;
}

int mam_run_finish(ultra_ctx *ctx)
{
    MamState &m = ctx->mam;
    MyersState &my = ctx->my;
    const int32_t ni = m.n_inst;
    ultra_drain_guard drain(ctx);
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (my.times_pending) { const int trc = ultra_myers_collect_times(ctx); if (trc != ULTRA_OK) return trc; }
    drain.dismiss();
    if (ni == 0) { m.ran = true; return ULTRA_OK; }
    const int64_t *h_close = m.pinned_base + (size_t)ni + 1;
    m.total_mam = h_close[1];
    m.nmam_on_host = m.exact_needed; // otherwise the counts are copied when a fetch asks for the MAM lists
    if ((int32_t)h_close[2] != 0)
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_run: an instance reached the reference's 'BUG' state in the n log n chainer");
    m.total_chain = h_close[0];
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&m.ms[0], m.ev[0], m.ev[1]));
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&m.ms[1], m.ev[2], m.ev[3]));
    ULTRA_HIP_CHECK(ctx, hipEventElapsedTime(&m.ms[2], m.ev[3], m.ev[4]));
    m.ran = true;
    return ULTRA_OK;
}

} // namespace

// A run in two halves.  ultra_mam_run_begin returns when the whole run has been ENQUEUED (the host has the forward pass's
// counts by then; the start pass, the traceback, the MAM build and the chainers are still in flight): the caller's thread
// is free for the next batch, e.g. ultra_mam_plan_resident_behind on another context.  ultra_mam_run_end waits for the
// device and closes the run; nothing of the results may be touched in between.
static int mam_run_begin(ultra_ctx *ctx, double min_acc, bool defer_tail)
{
    if (!ctx) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    ctx->defer_tail = defer_tail;
    if (!m.staged || !ctx->my.staged) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_run: no staged batch");
    if (m.begun) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_run_begin: the previous run has not been closed (ultra_mam_run_end)");
    tail_state(ctx, true, false, false);
    const int rc = mam_run_enqueue(ctx, min_acc);
    if (rc != ULTRA_OK) { tail_state(ctx, false, false, false); return rc; }
    {
        // a run without a start pass (no pairs) never opened its tail: nobody must wait for it
        std::lock_guard<std::mutex> lk(ctx->tail_mu);
        ctx->tail_open = true;
    }
    ctx->tail_cv.notify_all();
    m.begun = true;
    return ULTRA_OK;
}

extern "C" int ultra_mam_run_end(ultra_ctx *ctx)
{
    if (!ctx) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    if (!m.begun) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_run_end: no run in flight (ultra_mam_run_begin)");
    m.begun = false;
    ctx->defer_tail = false;
    const int rc = mam_run_finish(ctx);
    tail_state(ctx, false, false, false);
    return rc;
}

extern "C" int ultra_mam_run_begin(ultra_ctx *ctx, double min_acc) { return mam_run_begin(ctx, min_acc, true); }

extern "C" int ultra_mam_run(ultra_ctx *ctx, double min_acc)
{
    const int rc = mam_run_begin(ctx, min_acc, false);
    if (rc != ULTRA_OK) return rc;
    return ultra_mam_run_end(ctx);
}

extern "C" int ultra_mam_sizes(const ultra_ctx *ctx, int64_t *n_chain_total, int64_t *n_mam_total)
{
    if (!ctx || !ctx->mam.ran) return ULTRA_E_STATE;
    if (n_chain_total) *n_chain_total = ctx->mam.total_chain;
    if (n_mam_total) *n_mam_total = ctx->mam.total_mam;
    return ULTRA_OK;
}

extern "C" int ultra_mam_counts(const ultra_ctx *ctx, int32_t *n_inst, int32_t *n_cand, int32_t *n_pairs)
{
    if (!ctx || !ctx->mam.staged) return ULTRA_E_STATE;
    if (n_inst) *n_inst = ctx->mam.n_inst;
    if (n_cand) *n_cand = ctx->mam.n_cand;
    if (n_pairs) *n_pairs = ctx->my.n_pairs;
    return ULTRA_OK;
}

// The staged batch as the device holds it (pairs in their sorted order, cand_pair pointing into that order, offsets
// relative to the arena's first byte): lets a test compare a device-planned batch with a host-planned one by content.
extern "C" int ultra_mam_fetch_plan(ultra_ctx *ctx, int64_t *inst_cand_off, int32_t *inst_read_len, int64_t *inst_fps, int64_t *inst_lps,
                                    uint8_t *cand_kind, int32_t *cand_e_start, int32_t *cand_e_stop, int32_t *cand_pair, int32_t *cand_pair_contained,
                                    int32_t *cand_seq_len, int32_t *cand_seq_id, int32_t *cand_chr,
                                    int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len, int32_t *k)
{
    if (!ctx) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    MyersState &st = ctx->my;
    if (!m.staged || !st.staged) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_fetch_plan: no staged batch");
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t ni = (size_t)m.n_inst, nc = (size_t)m.n_cand, np = (size_t)st.n_pairs;
    ultra_drain_guard drain(ctx);
    auto down = [&](void *dst, const DevBuf &src, size_t bytes) -> hipError_t {
        if (!dst || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, s);
    };
    ULTRA_HIP_CHECK(ctx, down(inst_cand_off, m.candoff, (ni + 1) * 8));
    ULTRA_HIP_CHECK(ctx, down(inst_read_len, m.readlen, ni * 4));
    ULTRA_HIP_CHECK(ctx, down(inst_fps, m.fps, ni * 8)); ULTRA_HIP_CHECK(ctx, down(inst_lps, m.lps, ni * 8));
    ULTRA_HIP_CHECK(ctx, down(cand_kind, m.kind, nc)); ULTRA_HIP_CHECK(ctx, down(cand_e_start, m.estart, nc * 4));
    ULTRA_HIP_CHECK(ctx, down(cand_e_stop, m.estop, nc * 4)); ULTRA_HIP_CHECK(ctx, down(cand_pair, m.pair, nc * 4));
    ULTRA_HIP_CHECK(ctx, down(cand_pair_contained, m.pairc, nc * 4)); ULTRA_HIP_CHECK(ctx, down(cand_seq_len, m.seqlen, nc * 4));
    ULTRA_HIP_CHECK(ctx, down(cand_seq_id, m.seqid, nc * 4));
    if (cand_chr && !m.has_chr) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_fetch_plan: the batch was staged without cand_chr");
    ULTRA_HIP_CHECK(ctx, down(cand_chr, m.chr, nc * 4));
    ULTRA_HIP_CHECK(ctx, down(q_off, st.qoff, np * 8)); ULTRA_HIP_CHECK(ctx, down(t_off, st.toff, np * 8));
    ULTRA_HIP_CHECK(ctx, down(q_len, st.qlen, np * 4)); ULTRA_HIP_CHECK(ctx, down(t_len, st.tlen, np * 4)); ULTRA_HIP_CHECK(ctx, down(k, st.k, np * 4));
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    drain.dismiss();
    if (q_off) for (size_t i = 0; i < np; i++) q_off[i] -= (int64_t)kArenaPad;
    if (t_off) for (size_t i = 0; i < np; i++) t_off[i] -= (int64_t)kArenaPad;
    return ULTRA_OK;
}

extern "C" int ultra_mam_last_timing(const ultra_ctx *ctx, float ms[3])
{
    if (!ctx || !ctx->mam.ran) return ULTRA_E_STATE;
    if (ms) for (int i = 0; i < 3; i++) ms[i] = ctx->mam.ms[i];
    return ULTRA_OK;
}

extern "C" int ultra_mam_fetch(ultra_ctx *ctx, ultra_mam_output *out)
{
    if (!ctx || !out) return ULTRA_E_ARG;
    MamState &m = ctx->mam;
    if (!m.ran) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_fetch: no completed run");
    const int32_t ni = m.n_inst;
    if (ni == 0) { if (out->chain_off) out->chain_off[0] = 0; if (out->mam_off) out->mam_off[0] = 0; return ULTRA_OK; }
    if (!out->value || !out->n_mam || !out->chain_off) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_fetch: null per-instance output");
    if (out->chain_cap < m.total_chain)
        return ultra_fail(ctx, ULTRA_E_CAPACITY, "ultra_mam_fetch: chain arrays too small, need " + std::to_string(m.total_chain));
    const size_t TC = (size_t)m.total_chain;
    if (TC > 0 && (!out->chain_x || !out->chain_y || !out->chain_c || !out->chain_d || !out->chain_val || !out->chain_msl || !out->chain_cand || !out->chain_j))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_fetch: null chain array");
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    ultra_drain_guard drain(ctx); // host vectors below must outlive the copies, also on an error return
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->value, m.value.p, (size_t)ni * 8, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_off, m.chainoff.p, ((size_t)ni + 1) * 8, hipMemcpyDeviceToHost, s));
    if (TC > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_x, m.cx.p, TC * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_y, m.cy.p, TC * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_c, m.cc.p, TC * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_d, m.cd.p, TC * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_val, m.cval.p, TC * 8, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_msl, m.cmsl.p, TC * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_cand, m.ccand.p, TC * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_j, m.cj.p, TC * 4, hipMemcpyDeviceToHost, s));
        if (out->chain_kind) ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_kind, m.ckind.p, TC, hipMemcpyDeviceToHost, s));
        if (out->chain_chr) {
            if (!m.has_chr) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_mam_fetch: chain_chr asked for, but the batch was staged without cand_chr");
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->chain_chr, m.cchr.p, TC * 4, hipMemcpyDeviceToHost, s));
        }
    }
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(out->n_mam, m.nmam.p, (size_t)ni * 4, hipMemcpyDeviceToHost, s));
    // optional: every instance's full, filtered and sorted MAM list (tests compare MAM sets)
    if (out->mam_off) {
        if (out->mam_cap < m.total_mam)
            return ultra_fail(ctx, ULTRA_E_CAPACITY, "ultra_mam_fetch: MAM arrays too small, need " + std::to_string(m.total_mam));
        if (m.total_mam > 0 && (!out->mam_x || !out->mam_y || !out->mam_c || !out->mam_d || !out->mam_val || !out->mam_msl || !out->mam_cand))
            return ultra_fail(ctx, ULTRA_E_ARG, "ultra_mam_fetch: null MAM array");
        const size_t T = (size_t)m.total_cap;
        std::vector<int32_t> hx(T), hy(T), hc(T), hd(T), hm(T), hk(T);
        std::vector<double> hv(T);
        ultra_drain_guard drain_all(ctx); // after the vectors it protects
        if (T > 0) {
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hx.data(), m.mx.p, T * 4, hipMemcpyDeviceToHost, s));
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hy.data(), m.my.p, T * 4, hipMemcpyDeviceToHost, s));
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), m.mc.p, T * 4, hipMemcpyDeviceToHost, s));
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hd.data(), m.md.p, T * 4, hipMemcpyDeviceToHost, s));
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hm.data(), m.mmsl.p, T * 4, hipMemcpyDeviceToHost, s));
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hk.data(), m.mcand.p, T * 4, hipMemcpyDeviceToHost, s));
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(hv.data(), m.mval.p, T * 8, hipMemcpyDeviceToHost, s));
        }
        m.h_instbase.resize((size_t)ni + 1);
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.h_instbase.data(), m.instbase.p, ((size_t)ni + 1) * 8, hipMemcpyDeviceToHost, s));
        if (!m.nmam_on_host) {
            m.h_nmam.resize((size_t)ni);
            ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(m.h_nmam.data(), m.nmam.p, (size_t)ni * 4, hipMemcpyDeviceToHost, s));
        }
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
        m.nmam_on_host = true;
        int64_t w = 0;
        for (int32_t i = 0; i < ni; i++) {
            out->mam_off[i] = w;
            const int64_t b = m.h_instbase[(size_t)i];
            for (int32_t t = 0; t < m.h_nmam[(size_t)i]; t++, w++) {
                out->mam_x[w] = hx[b + t]; out->mam_y[w] = hy[b + t]; out->mam_c[w] = hc[b + t]; out->mam_d[w] = hd[b + t];
                out->mam_val[w] = hv[b + t]; out->mam_msl[w] = hm[b + t]; out->mam_cand[w] = hk[b + t];
            }
        }
        out->mam_off[ni] = w;
        drain_all.dismiss();
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    drain.dismiss();
    return ULTRA_OK;
}

extern "C" int ultra_mam_batch(ultra_ctx *ctx, const ultra_mam_input *in, double min_acc, ultra_mam_output *out)
{
    int rc = ultra_mam_stage(ctx, in);
    if (rc != ULTRA_OK) return rc;
    rc = ultra_mam_run(ctx, min_acc);
    if (rc != ULTRA_OK) return rc;
    return ultra_mam_fetch(ctx, out);
}
