// Shared host-side plumbing for libultra_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ultra_hip.h"

// Debug canaries (there is no compute-sanitizer on ROCm): with ULTRA_CANARIES=1 in the environment
// every device buffer of the library is allocated at exactly the size asked for, between two guard regions of
// kCanaryBytes filled with a pattern; ultra_debug_canaries() reads the guards of every live buffer back.  An
// out-of-bounds store of a kernel then shows up as a damaged guard instead of a corrupted neighbour (csrc/api.hip).
constexpr size_t kCanaryBytes = 256;
bool ultra_canaries_on();
void ultra_canary_note(void *base, size_t bytes);
void ultra_canary_drop(void *base);

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *base = nullptr; // what hipMalloc returned (= p unless the buffer sits between canaries)
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        if (ultra_canaries_on()) {
            hipError_t e = hipMalloc(&base, bytes + 2 * kCanaryBytes);
            if (e != hipSuccess) { base = nullptr; return e; }
            e = hipMemset(base, 0xA5, kCanaryBytes);
            if (e == hipSuccess) e = hipMemset((char *)base + kCanaryBytes + bytes, 0xA5, kCanaryBytes);
            if (e != hipSuccess) { (void)hipFree(base); base = nullptr; return e; }
            p = (char *)base + kCanaryBytes;
            cap = bytes;
            ultra_canary_note(base, bytes);
            return hipSuccess;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        base = p;
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (base) {
            if (base != p) ultra_canary_drop(base);
            (void)hipFree(base);
        }
        p = base = nullptr; cap = 0;
    }
    template <typename T> T *as() const { return (T *)p; }
};

// One size class of the Myers kernels: G lanes per pair, W 64-row blocks per lane.
struct MyersClass {
    int cls;            // index into the class table (14 = the long class: strips of 32768 rows)
    int G, W;           // rows covered = 64*G*W
    int64_t pb, pe;     // pair range (sorted order)
    int32_t max_t;      // longest target in the class
    int32_t max_m;      // longest query in the class
    int64_t lb, le;     // location (CSR) range, known after the forward pass
    int64_t hb, he;     // hit (ed >= 0) range
    double cells;       // forward-pass cells of the class (sum of q_len * t_len)
};

// One work item of a pass = one wavefront: items [base, min(base + per_wave, end)) of size class cls.
struct WorkItem {
    int64_t base, end;
    int32_t cls;
    int32_t cost;
};

// Register bucket of size classes: one launch per pass and bucket (myers_mega<MODE, BUCKET>).
struct MyersBucket {
    DevBuf fwd_list, rev_list, fwd_flag;
    int32_t n_fwd = 0, n_rev = 0;
    int wmax = 0;
    double cells = 0;   // forward-pass cells (sum of q_len * t_len) of the bucket's pairs
    float fwd_ms = 0;   // device time of the bucket's forward launch in the last run
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// A pair as the device planner writes it (one 32-byte record: the gather into sorted order then costs one random access
// per pair instead of five).  The host entry points upload five arrays (u_qoff ...) instead: MyersState::u_packed tells.
struct PairRec { int64_t q_off, t_off; int32_t q_len, t_len, k, pad; };

struct MyersState {
    bool staged = false, ran = false;
    int32_t n_pairs = 0;
    int64_t N = 0;      // arena length
    int S = 0;          // alphabet size of the batch (distinct bytes)
    std::vector<int32_t> perm;      // sorted position -> caller's pair index (host copy, fetched on demand)
    bool perm_on_host = false;
    std::vector<MyersClass> classes;
    MyersBucket buckets[5];
    double cells = 0;            // forward cells of the pairs that run
    double cells_skipped = 0;    // cells of the pairs answered at stage time (query more than k longer than the target)
    int64_t n_skipped = 0;       // their number; they are the last pairs of the sorted order
    int32_t n_run = 0;           // pairs in front of them
    float ms[5] = {0, 0, 0, 0, 0};
    int64_t total_loc = 0, total_hit = 0;
    DevBuf seq, rseq, qoff, qlen, toff, tlen, k, ed, nloc, slots, packed, scanned, blocksum, locoff, hitoff,
        locend, locstart, locpair, hitpairs, firstins, nops, trace, bounds, ops, opsoff, bandcls, banditems, bandinfo, sorttmp, trace2, startitems, startinfo, hbuf;
    // staging: raw bytes, byte -> code table (+ presence mask), the pairs as the caller ordered them, sort scratch,
    // sorted position -> caller's index and its inverse, per-class statistics, unsorted work items
    bool u_packed = false;
    DevBuf u_rec;
    DevBuf raw, lut, u_qoff, u_qlen, u_toff, u_tlen, u_k, skeys, perm_d, inv_d, stats, worktmp;
    // resident-pool mode (ultra_mam_stage_nams): arena = [pad | index pool | reads of the batch (reads_cap) | pad]; the pool
    // part is encoded once per (context, device index); byte -> code assignments only ever grow
    uint64_t pool_serial = 0;
    int64_t reads_cap = 0;
    uint8_t h_lut[256] = {};
    bool h_seen[256] = {};
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // [5] traceback counts on the host, [6] the caller's mid-run work done
    int32_t *h_route = nullptr;  // page-locked landing place of the traceback routing's counts
    bool times_pending = false;  // the last run returned before its final synchronisation (ultra_myers_run_impl, defer_sync)
};

struct ChainState {
    uint8_t *fixed = nullptr;        // device address of chain.hip's fixed scratch array on this context's GPU
    DevBuf ioff, off, x, y, c, d, val, idx, len, value, work, namC, namtrace, intron, sorttmp;
    std::vector<int32_t> h_list;     // kept alive until the stream has consumed them
    std::vector<int64_t> h_workoff;
};

struct MamState {
    bool staged = false, ran = false;
    int32_t n_cand = 0, n_inst = 0, max_cand = 0;
    int64_t total_cap = 0, total_chain = 0, total_mam = 0;
    float ms[3] = {0, 0, 0}; // MAM build, chaining, chain gather
    std::vector<int64_t> h_instbase;
    std::vector<int32_t> h_nmam;
    bool nmam_on_host = false;       // h_nmam holds the last run's counts (read back only when something needs them)
    int64_t *pinned_base = nullptr;  // pinned copy of instbase (first MAM slot of every instance, then the total), read
    size_t pinned_cap = 0;           // back while the Myers passes still run; entries.  The three words behind the total take
                                     // the run's closing counts (chain rows, MAMs, chainer status): ultra_mam_run_end reads them
    bool begun = false;              // ultra_mam_run_begin has enqueued a run that ultra_mam_run_end has not closed yet
    bool exact_needed = false;       // that run read the exact MAM counts back (h_nmam is current)
    std::vector<int32_t> h_slots;    // slots per instance = upper bound of its MAM count
    bool up_valid = false;           // an uploaded request (ultra_mam_upload_nams) lies in d_nam* / my.raw
    int32_t up_ni = 0;
    int64_t up_nnam = 0, up_read_bytes = 0;
    bool has_chr = false;            // chr[] holds the chromosome id of every candidate (needed for chain_chr in the fetch)
    DevBuf chr, ckind, cchr;
    // device planner (dplan.hip): the request on the device, per-instance counts, overflow list, counters, scan scratch
    DevBuf d_namoff, d_readoff, d_nampart, d_namx, d_namy, p_ncand, p_npair, p_over, p_misc, p_wide, p_poff, p_state, p_state_mid, p_state_big, total;
    DevBuf kind, estart, estop, pair, pairc, seqlen, seqid, candoff, readlen, fps, lps, cap, capoff, instbase, scantmp, flag,
        mx, my, mc, md, mmsl, mcand, mval, nmam, chainidx, chainlen, value, lenw, chainoff, cx, cy, cc, cd, cmsl, ccand, cj, cval;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

struct AffineState {
    DevBuf seq, qoff, toff, qlen, tlen, troff, bdoff, tr, bd, score, endq, endt, ops, opsoff, nops;
};

// Launch lanes = HIP streams the independent launches of a pass are dealt over.  Streams beyond the process's
// hardware queues (GPU_MAX_HW_QUEUES, 4 unless the environment says otherwise before the first HIP call) share a
// queue and serialise, so the default lane count follows that variable (ultra_ctx_create).
constexpr int kAuxStreams = 7;
constexpr int kXtraStreams = 4;
#ifndef ULTRA_HOSTSYNC_DEFAULT
#define ULTRA_HOSTSYNC_DEFAULT 0x54u
#endif
constexpr int kLanes = kAuxStreams + 1;

// Distinct bytes of a batch (dense codes, byte-exact comparison).  The match table of a wavefront is S x words x 512 B of
// LDS: up to 16 symbols the kernels keep their usual occupancy, beyond that (soft-masked sequence + IUPAC codes: 33 bytes)
// they run with fewer wavefronts per CU; 36 is what still fits the 160 KB of a CU for the widest class (32768 rows).
constexpr int kMaxSymbols = 36;
constexpr size_t kArenaPad = 64;

struct ultra_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // size classes of one pass are independent: they are launched round-robin on these streams so
    // that short launches overlap (fork = aux streams wait for `stream`, join = `stream` waits for them)
    hipStream_t aux[kAuxStreams] = {};
    // Streams beyond the lanes, for the traceback's full-height launches (hits whose distance no band class takes: long
    // reads against the reference's generous k): on a lane they stood in front of a band class's launch, which then
    // started when they had ended (dRNA: 4.7 + 3.5 ms in a row, Iso-Seq 2.2 + 3.1).  Joined by ultra_join_xtra.
    hipStream_t xtra[kXtraStreams] = {};
    hipEvent_t xtra_ev[kXtraStreams] = {};
    hipEvent_t fork_ev = nullptr, join_ev[kAuxStreams] = {};
    // lane i of a forked section: the auxiliary streams in use, then the main stream itself
    hipStream_t lane(int i) const { i %= cfg.launch_lanes; return i < cfg.launch_lanes - 1 ? aux[i] : stream; }
    // enqueued by the Myers run on the stream it names, right after the start pass has been launched: work of the caller
    // that only needs the forward results rides under the remaining passes and shares their next host synchronisation
    std::function<int(hipStream_t)> mid_hook;
    // How far the enqueueing of a run in flight has come.  Another context's planner (ultra_mam_plan_resident_behind)
    // orders its kernels behind this run's start pass (my.ev[2]): they then share the chip with the latency-bound tail of
    // the run (traceback, MAM build, chaining) instead of crawling beside the forward and start passes, which fill it.
    std::mutex tail_mu;
    std::condition_variable tail_cv;
    bool run_active = false;     // between ultra_mam_run_begin and the end of ultra_mam_run_end
    bool tail_open = false;      // the start pass of that run has been enqueued (or the run has none)
    bool tail_has_event = false; // ... and my.ev[2] marks its end on the device
    bool defer_tail = false;     // the run in flight was begun by ultra_mam_run_begin itself: no host waits behind the start pass
    ultra_cfg cfg;
    std::string err;
    hipDeviceProp_t prop;
    MyersState my;
    ChainState ch;
    MamState mam;
    AffineState aff;
};

#define ULTRA_HIP_CHECK(ctx, call)                                                                      \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            char b__[512];                                                                              \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                     __LINE__);                                                                         \
            (ctx)->err = b__;                                                                           \
            return ULTRA_E_HIP;                                                                         \
        }                                                                                               \
    } while (0)

static inline hipError_t ultra_fork(ultra_ctx *ctx)
{
    hipError_t e = hipEventRecord(ctx->fork_ev, ctx->stream);
    for (int i = 0; i < ctx->cfg.launch_lanes - 1 && e == hipSuccess; i++) e = hipStreamWaitEvent(ctx->aux[i], ctx->fork_ev, 0);
    return e;
}
static inline hipError_t ultra_join(ultra_ctx *ctx)
{
    hipError_t e = hipSuccess;
    for (int i = 0; i < ctx->cfg.launch_lanes - 1 && e == hipSuccess; i++) {
        e = hipEventRecord(ctx->join_ev[i], ctx->aux[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->join_ev[i], 0);
    }
    return e;
}

// Host waits in front of a pass's launches (bit = point: 0 forward fork, 1 forward join, 2 start-pass fork, 3 start-pass
// join, 4 in front of the traceback's launches, 5 traceback join, 6 chaining fork, 7 chaining join): at a fork the host
// first waits for the main stream, at a join it waits for the lanes instead of the main stream waiting for their events on
// the device.  Launches enqueued behind a pending event all start at once, in no order; enqueued onto a free stream one
// after the other, the few long wavefronts go first.  Default 0x54 (measured); ULTRA_DEV_HOSTSYNC
// overrides it for experiments.  A run begun with ultra_mam_run_begin (ctx->defer_tail) leaves out the waits behind the
// start pass: that call promises to return with the tail in flight.
static inline unsigned ultra_hostsync_mask(const ultra_ctx *ctx)
{
    static const unsigned m = [] { const char *e = getenv("ULTRA_DEV_HOSTSYNC"); return e ? (unsigned)strtoul(e, nullptr, 0) : ULTRA_HOSTSYNC_DEFAULT; }();
    return ctx->defer_tail ? m & ~0xF0u : m;
}
static inline hipError_t ultra_fork_at(ultra_ctx *ctx, int point)
{
    if (ultra_hostsync_mask(ctx) >> point & 1u) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) return e; }
    return ultra_fork(ctx);
}
static inline hipError_t ultra_join_at(ultra_ctx *ctx, int point)
{
    if (!(ultra_hostsync_mask(ctx) >> point & 1u)) return ultra_join(ctx);
    hipError_t e = hipSuccess;
    for (int i = 0; i < ctx->cfg.launch_lanes - 1 && e == hipSuccess; i++) e = hipStreamSynchronize(ctx->aux[i]);
    return e;
}

// Error paths must not return with work in flight: host vectors that async copies still read or write would be
// destroyed under the DMA, and the next call would race with the aux lanes.  Armed after the first enqueue of a
// call, dismissed on its normal return; on any other return it waits (best effort) for every lane.
struct ultra_drain_guard {
    ultra_ctx *ctx;
    bool armed = true;
    explicit ultra_drain_guard(ultra_ctx *c) : ctx(c) {}
    void dismiss() { armed = false; }
    ~ultra_drain_guard()
    {
        if (!armed) return;
        for (int i = 0; i < kAuxStreams; i++) if (ctx->aux[i]) (void)hipStreamSynchronize(ctx->aux[i]);
        for (int i = 0; i < kXtraStreams; i++) if (ctx->xtra[i]) (void)hipStreamSynchronize(ctx->xtra[i]);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    }
};

static inline int ultra_fail(ultra_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// NAM mode of ultra_chain_launch: per-instance intron limits in, DP vectors out (device pointers)
struct ultra_nam_chain {
    const int64_t *d_max_intron;
    int64_t *d_C;
    int32_t *d_trace;
};

int ultra_myers_stage_impl(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                           const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                           const int32_t *k);
int ultra_myers_stage_pairs(ultra_ctx *ctx, int32_t n_pairs);
// presence mask (8 x u32, accumulated with atomicOr) of n raw bytes; seq[at + i] = lut[raw[i]] with the mirror image in rseq
void ultra_launch_presence(hipStream_t s, const uint8_t *raw, int64_t n, uint32_t *mask);
void ultra_launch_encode(hipStream_t s, const uint8_t *raw, int64_t n, const uint8_t *lut, uint8_t *seq, uint8_t *rseq, int64_t at, int64_t N);
void ultra_mam_remap_pairs(ultra_ctx *ctx, int32_t n_cand);
int ultra_myers_host_perm(ultra_ctx *ctx);
int ultra_myers_run_impl(ultra_ctx *ctx, bool want_ops, bool defer_sync = false);
int ultra_myers_collect_times(ultra_ctx *ctx);
// caller-provided device workspace of ultra_chain_launch (instead of the context's growable buffers)
struct ultra_chain_ws {
    uint8_t *aux; size_t aux_cap;   // work offsets | launch list | status word: ultra_chain_aux_bytes(n_inst)
    uint8_t *work; size_t work_cap; // search-tree arrays of the n log n solver
};
static inline size_t ultra_chain_aux_bytes(int32_t n_inst) { return ((size_t)n_inst + 1) * 8 + (size_t)n_inst * 4 + 64; }
int ultra_chain_launch(ultra_ctx *ctx, int32_t n_inst, const int32_t *h_cnt, const int64_t *d_off, const int32_t *d_cnt,
                       const int32_t *d_x, const int32_t *d_y, const int32_t *d_c, const int32_t *d_d, const double *d_val,
                       int32_t solver, int32_t overlap_small, int32_t overlap_large,
                       int32_t *d_idx, int32_t *d_len, double *d_value, int32_t **d_status_out,
                       const ultra_nam_chain *nam = nullptr, const ultra_chain_ws *ws = nullptr, bool cnt_is_bound = false);
constexpr int kChainNeedExact = -1000; // ultra_chain_launch with cnt_is_bound: a limit failed on the bounds, nothing was enqueued
int ultra_chain_scratch_init(ultra_ctx *ctx);
// exclusive scan of n 64-bit values on `on` (ctx->stream when null); tmp: scratch, grown as needed
hipError_t ultra_scan_u64(ultra_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n, DevBuf &tmp, hipStream_t on = nullptr);
// ids (or vals) grouped by a small key: three launches on s (myers.hip)
hipError_t ultra_bucket_sort(hipStream_t s, const uint32_t *keys, const int32_t *vals, int64_t n, int nb, int32_t *out,
                             int32_t *first, int stride, int32_t *gfirst, DevBuf &tmp);
void ultra_myers_free(ultra_ctx *ctx);
void ultra_chain_free(ultra_ctx *ctx);
void ultra_mam_free(ultra_ctx *ctx);
void ultra_affine_free(ultra_ctx *ctx);
