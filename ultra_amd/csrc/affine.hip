// Base-level alignment behind the MAM stage for gfx950: batched semi-global
// affine-gap alignment with traceback.  Replaces, for a batch of (read, reconstructed reference) pairs,
//     help_functions.parasail_alignment(s1, s2, 2, -2, 3, 1) -> parasail.sg_trace_scan_16 / _32
//                                                    /root/reference/modules/help_functions.py:204-226
// as align.get_exact_alignment calls it               /root/reference/modules/align.py:338-354.
// parasail is a third-party C library that is absent here: the recurrences are Gotoh's, the conventions (free end gaps
// in both sequences, gap of length L = open + (L - 1) * extend, matrix_create("ACGT", match, mismatch)) parasail's
// published ones; the tie rules are this file's own and stated in include/ultra_hip.h (parity unpinned against parasail itself).
//
// Fill: one wavefront per pair, 64 query rows per strip, skewed schedule - at step t lane l computes the cell
// (row 64 s + l, column t - l), its upper neighbours come from lane l - 1 (one cross-lane move per value and step),
// the strip's last row is handed to the next strip through a per-pair boundary row.  int32 scores: what parasail's
// 16-bit kernel returns unless it saturates, and what its 32-bit rerun returns then.  One trace byte per cell
// (2 bits H source, 1 bit "E was extended", 1 bit "F was extended"), stored BY STEP so that a wavefront writes 64
// consecutive bytes.  Walk: one lane per pair.
#include "common.h"

#include <algorithm>
#include <string.h>

namespace {

constexpr int32_t kNeg = -(1 << 29);

struct AffArgs {
    const uint8_t *seq;
    const int64_t *q_off, *t_off;
    const int32_t *q_len, *t_len;
    int32_t match, mismatch, gap_open, gap_extend;
    const int64_t *tr_off, *bd_off;   // per pair: first trace byte / first boundary entry (2 x n int32 per pair)
    uint8_t *tr;
    int32_t *bd;
    int32_t *score, *end_q, *end_t;
    uint8_t *ops; const int64_t *ops_off; int32_t *n_ops;
    int32_t first, count;             // pairs [first, first + count) of the batch are in this launch
};

__device__ __forceinline__ int sub_code(uint32_t c)
{
    c &= ~0x20u; // upper case
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

__global__ void __launch_bounds__(64) k_affine_fill(AffArgs a)
{
    const int lane = threadIdx.x;
    const int p = a.first + (int)blockIdx.x;
    const int m = a.q_len[p], n = a.t_len[p];
    const uint8_t *q = a.seq + a.q_off[p], *t = a.seq + a.t_off[p];
    uint8_t *tr = a.tr ? a.tr + a.tr_off[blockIdx.x] : nullptr; // null: score and end cell only
    int32_t *Hb = a.bd + a.bd_off[blockIdx.x], *Fb = Hb + n;
    const int go = a.gap_open, ge = a.gap_extend;
    const int strips = (m + 63) >> 6;
    int32_t best_col = kNeg, best_i = 0, best_row = kNeg, best_j = 0;
    for (int s = 0; s < strips; s++) {
        const int i = 64 * s + lane;
        const bool row_ok = i < m;
        const int qc = row_ok ? sub_code(q[i]) : -2;
        int32_t Hleft = 0, E = kNeg, Hdiag = 0, Hout = 0, Fout = kNeg;
        int32_t bH = 0, bF = kNeg;
        int tc = -3;
        uint8_t *trs = tr ? tr + (int64_t)s * (n + 64) * 64 : nullptr;
        const int steps = n + 63;
        for (int tt = 0; tt < steps; tt++) {
            if ((tt & 63) == 0) { // this strip's upper boundary for columns tt .. tt + 63 (row above the strip)
                const int ci = tt + lane;
                bH = (s > 0 && ci < n) ? Hb[ci] : 0;
                bF = (s > 0 && ci < n) ? Fb[ci] : kNeg;
            }
            int32_t upH = __shfl_up(Hout, 1, 64), upF = __shfl_up(Fout, 1, 64);
            const int32_t b0H = __shfl(bH, tt & 63, 64), b0F = __shfl(bF, tt & 63, 64);
            int tcn = __shfl_up(tc, 1, 64);
            if (lane == 0) { upH = b0H; upF = b0F; tcn = tt < n ? sub_code(t[tt]) : -3; }
            tc = tcn;
            const int c = tt - lane;
            if (row_ok && c >= 0 && c < n) {
                const int32_t e_ext = E - ge, e_opn = Hleft - go;
                const bool e_is_ext = e_ext >= e_opn;
                E = e_is_ext ? e_ext : e_opn;
                const int32_t f_ext = upF - ge, f_opn = upH - go;
                const bool f_is_ext = f_ext >= f_opn;
                const int32_t F = f_is_ext ? f_ext : f_opn;
                // matrix_create("ACGT", match, mismatch): a character outside the alphabet maps to the matrix's wildcard
                // row / column, whose entries are 0 (N against anything neither earns nor costs)
                int32_t h = Hdiag + ((qc >= 0 && tc >= 0) ? (qc == tc ? a.match : a.mismatch) : 0);
                int src = 0;
                if (F > h) { h = F; src = 1; }
                if (E > h) { h = E; src = 2; }
                if (trs) trs[(int64_t)tt * 64 + lane] = (uint8_t)(src | ((int)e_is_ext << 2) | ((int)f_is_ext << 3));
                Hdiag = upH; Hleft = h; Hout = h; Fout = F;
                if (lane == 63) { Hb[c] = h; Fb[c] = F; }
                if (c == n - 1 && h > best_col) { best_col = h; best_i = i; }
                if (i == m - 1 && h > best_row) { best_row = h; best_j = c; }
            }
        }
        __syncthreads(); // the boundary row written by lane 63 is read by every lane of the next strip
    }
    // end cell: last row left to right, then last column top to bottom, strictly better replaces (include/ultra_hip.h)
    int32_t vr = best_row, jr = best_j;
    {
        const int src = (m - 1) & 63;
        vr = __shfl(vr, src, 64); jr = __shfl(jr, src, 64);
    }
    int32_t vc = best_col, ic = best_i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t ov = __shfl_xor(vc, o, 64), oi = __shfl_xor(ic, o, 64);
        if (ov > vc || (ov == vc && oi < ic)) { vc = ov; ic = oi; }
    }
    if (lane == 0) {
        if (vc > vr) { a.score[p] = vc; a.end_q[p] = ic; a.end_t[p] = n - 1; }
        else { a.score[p] = vr; a.end_q[p] = m - 1; a.end_t[p] = jr; }
    }
}

// one lane per pair: edit script written BACKWARDS (the host turns it round)
__global__ void __launch_bounds__(64) k_affine_walk(AffArgs a)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= a.count) return;
    const int p = a.first + idx;
    const int m = a.q_len[p], n = a.t_len[p];
    const uint8_t *q = a.seq + a.q_off[p], *t = a.seq + a.t_off[p];
    const uint8_t *tr = a.tr + a.tr_off[idx];
    uint8_t *ops = a.ops + a.ops_off[p];
    const int bi = a.end_q[p], bj = a.end_t[p];
    int cnt = 0;
    for (int i = m - 1; i > bi; i--) ops[cnt++] = 1;
    for (int j = n - 1; j > bj; j--) ops[cnt++] = 2;
    int i = bi, j = bj, state = 0;
    const int64_t strip_bytes = (int64_t)(n + 64) * 64;
    while (i >= 0 && j >= 0) {
        const uint8_t b = tr[(int64_t)(i >> 6) * strip_bytes + (int64_t)(j + (i & 63)) * 64 + (i & 63)];
        if (state == 0) state = b & 3;
        if (state == 0) { const int qc = sub_code(q[i]); ops[cnt++] = (qc >= 0 && qc == sub_code(t[j])) ? 0 : 3; i--; j--; }
        else if (state == 1) { ops[cnt++] = 1; if (!((b >> 3) & 1)) state = 0; i--; }
        else { ops[cnt++] = 2; if (!((b >> 2) & 1)) state = 0; j--; }
    }
    for (; i >= 0; i--) ops[cnt++] = 1;
    for (; j >= 0; j--) ops[cnt++] = 2;
    a.n_ops[p] = cnt;
}

} // namespace

void ultra_affine_free(ultra_ctx *ctx)
{
    for (DevBuf *b : {&ctx->aff.seq, &ctx->aff.qoff, &ctx->aff.toff, &ctx->aff.qlen, &ctx->aff.tlen, &ctx->aff.troff, &ctx->aff.bdoff,
                      &ctx->aff.tr, &ctx->aff.bd, &ctx->aff.score, &ctx->aff.endq, &ctx->aff.endt, &ctx->aff.ops, &ctx->aff.opsoff, &ctx->aff.nops})
        b->release();
}

extern "C" int ultra_sg_affine_batch(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                                     const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                                     int32_t match, int32_t mismatch, int32_t gap_open, int32_t gap_extend,
                                     int32_t *score, int32_t *end_query, int32_t *end_ref,
                                     uint8_t *ops, const int64_t *ops_off, int32_t *n_ops)
{
    if (!ctx) return ULTRA_E_ARG;
    if (n_pairs < 0 || (n_pairs > 0 && (!arena || !q_off || !q_len || !t_off || !t_len || !score || !end_query || !end_ref)))
        return ultra_fail(ctx, ULTRA_E_ARG, "ultra_sg_affine_batch: null pointer or negative pair count");
    if (ops && (!ops_off || !n_ops)) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_sg_affine_batch: ops given without ops_off / n_ops");
    if (gap_open < 0 || gap_extend < 0) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_sg_affine_batch: gap penalties are given as positive numbers");
    if (n_pairs == 0) return ULTRA_OK;
    const size_t budget = (size_t)12 << 30; // trace bytes in flight
    std::vector<int64_t> h_ops_off;
    for (int32_t p = 0; p < n_pairs; p++) {
        if (q_len[p] < 1 || t_len[p] < 1 || q_off[p] < 0 || t_off[p] < 0 || (uint64_t)q_off[p] + (uint64_t)q_len[p] > arena_len ||
            (uint64_t)t_off[p] + (uint64_t)t_len[p] > arena_len)
            return ultra_fail(ctx, ULTRA_E_ARG, "ultra_sg_affine_batch: pair " + std::to_string(p) + " has an empty sequence or offsets outside the arena");
        if ((int64_t)std::max(q_len[p], t_len[p]) * std::max(std::abs(match), std::abs(mismatch)) > (1 << 28))
            return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_sg_affine_batch: scores of pair " + std::to_string(p) + " would leave the int32 range");
        const size_t trace = (size_t)((q_len[p] + 63) / 64) * (size_t)(t_len[p] + 64) * 64;
        if (trace > budget) return ultra_fail(ctx, ULTRA_E_UNSUPPORTED, "ultra_sg_affine_batch: pair " + std::to_string(p) + " needs more than 12 GiB of trace");
        if (ops && ops_off[p + 1] - ops_off[p] < (int64_t)q_len[p] + t_len[p])
            return ultra_fail(ctx, ULTRA_E_CAPACITY, "ultra_sg_affine_batch: edit-script slot of pair " + std::to_string(p) + " is smaller than m + n");
    }
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    AffineState &af = ctx->aff;
    const size_t np = (size_t)n_pairs;
    ultra_drain_guard drain(ctx);
    ULTRA_HIP_CHECK(ctx, af.seq.ensure(arena_len + 16));
    ULTRA_HIP_CHECK(ctx, af.qoff.ensure(np * 8)); ULTRA_HIP_CHECK(ctx, af.toff.ensure(np * 8));
    ULTRA_HIP_CHECK(ctx, af.qlen.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, af.tlen.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, af.score.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, af.endq.ensure(np * 4)); ULTRA_HIP_CHECK(ctx, af.endt.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, af.nops.ensure(np * 4));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.seq.p, arena, arena_len, hipMemcpyHostToDevice, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.qoff.p, q_off, np * 8, hipMemcpyHostToDevice, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.toff.p, t_off, np * 8, hipMemcpyHostToDevice, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.qlen.p, q_len, np * 4, hipMemcpyHostToDevice, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.tlen.p, t_len, np * 4, hipMemcpyHostToDevice, s));
    AffArgs a;
    memset(&a, 0, sizeof a);
    a.seq = af.seq.as<uint8_t>(); a.q_off = af.qoff.as<int64_t>(); a.t_off = af.toff.as<int64_t>();
    a.q_len = af.qlen.as<int32_t>(); a.t_len = af.tlen.as<int32_t>();
    a.match = match; a.mismatch = mismatch; a.gap_open = gap_open; a.gap_extend = gap_extend;
    a.score = af.score.as<int32_t>(); a.end_q = af.endq.as<int32_t>(); a.end_t = af.endt.as<int32_t>(); a.n_ops = af.nops.as<int32_t>();
    if (ops) {
        const size_t total_ops = (size_t)ops_off[n_pairs];
        ULTRA_HIP_CHECK(ctx, af.ops.ensure(total_ops + 16)); ULTRA_HIP_CHECK(ctx, af.opsoff.ensure((np + 1) * 8));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.opsoff.p, ops_off, (np + 1) * 8, hipMemcpyHostToDevice, s));
        a.ops = af.ops.as<uint8_t>(); a.ops_off = af.opsoff.as<int64_t>();
    }
    // chunks of pairs whose trace fits the budget
    std::vector<int64_t> h_troff, h_bdoff;
    for (int32_t first = 0; first < n_pairs;) {
        h_troff.clear(); h_bdoff.clear();
        size_t tr_bytes = 0, bd_entries = 0;
        int32_t cnt = 0;
        while (first + cnt < n_pairs) {
            const int32_t p = first + cnt;
            const size_t trace = ops ? (size_t)((q_len[p] + 63) / 64) * (size_t)(t_len[p] + 64) * 64 : 64;
            if (cnt > 0 && tr_bytes + trace > budget) break;
            h_troff.push_back((int64_t)tr_bytes); h_bdoff.push_back((int64_t)bd_entries);
            tr_bytes += trace; bd_entries += 2 * (size_t)t_len[p];
            cnt++;
        }
        ULTRA_HIP_CHECK(ctx, af.tr.ensure(tr_bytes + 64)); ULTRA_HIP_CHECK(ctx, af.bd.ensure(bd_entries * 4 + 64));
        ULTRA_HIP_CHECK(ctx, af.troff.ensure((size_t)cnt * 8)); ULTRA_HIP_CHECK(ctx, af.bdoff.ensure((size_t)cnt * 8));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.troff.p, h_troff.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(af.bdoff.p, h_bdoff.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, s));
        a.tr = ops ? af.tr.as<uint8_t>() : nullptr; a.bd = af.bd.as<int32_t>(); a.tr_off = af.troff.as<int64_t>(); a.bd_off = af.bdoff.as<int64_t>();
        a.first = first; a.count = cnt;
        hipLaunchKernelGGL(k_affine_fill, dim3((unsigned)cnt), dim3(64), 0, s, a);
        if (ops) hipLaunchKernelGGL(k_affine_walk, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, s, a);
        ULTRA_HIP_CHECK(ctx, hipGetLastError());
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s)); // the offset vectors are reused by the next chunk
        first += cnt;
    }
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(score, af.score.p, np * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(end_query, af.endq.p, np * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(end_ref, af.endt.p, np * 4, hipMemcpyDeviceToHost, s));
    if (ops) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(n_ops, af.nops.p, np * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(ops, af.ops.p, (size_t)ops_off[n_pairs], hipMemcpyDeviceToHost, s));
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    drain.dismiss();
    if (ops)
        for (int32_t p = 0; p < n_pairs; p++) std::reverse(ops + ops_off[p], ops + ops_off[p] + n_ops[p]);
    return ULTRA_OK;
}
