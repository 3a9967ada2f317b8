// C ABI of libultra_hip.so: context management and the Myers entry points (see include/ultra_hip.h).
#include "common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>

// ---- debug canaries (common.h) ----
namespace {
std::mutex g_canary_mu;
std::map<void *, std::pair<size_t, int>> g_canaries; // hipMalloc base -> (payload bytes, device)
}

bool ultra_canaries_on()
{
    static const bool on = [] { const char *e = getenv("ULTRA_CANARIES"); return e && *e && strcmp(e, "0") != 0; }();
    return on;
}

void ultra_canary_note(void *base, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_canary_mu);
    g_canaries[base] = {bytes, dev};
}

void ultra_canary_drop(void *base)
{
    std::lock_guard<std::mutex> lk(g_canary_mu);
    g_canaries.erase(base);
}

extern "C" {

// Reads back the guard regions of every live device buffer of the process (all contexts, indexes, pipes).  Synchronises
// the devices first.  self_test != 0: additionally allocates a buffer, overruns it by one byte on purpose and requires the
// check to see it (answers ULTRA_E_STATE when it does not).
int ultra_debug_canaries(int64_t *n_buffers, int64_t *n_damaged, int self_test)
{
    if (n_buffers) *n_buffers = 0;
    if (n_damaged) *n_damaged = 0;
    if (!ultra_canaries_on()) return ULTRA_E_STATE;
    int cur = 0;
    (void)hipGetDevice(&cur);
    auto damaged = [](void *base, size_t bytes) -> int {
        uint8_t g[2 * kCanaryBytes];
        if (hipMemcpy(g, base, kCanaryBytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (hipMemcpy(g + kCanaryBytes, (char *)base + kCanaryBytes + bytes, kCanaryBytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        for (size_t i = 0; i < 2 * kCanaryBytes; i++) if (g[i] != 0xA5) return 1;
        return 0;
    };
    int rc = ULTRA_OK;
    if (self_test) {
        DevBuf t;
        if (t.ensure(1000) != hipSuccess) return ULTRA_E_HIP;
        const bool clean = damaged(t.base, 1000) == 0;
        (void)hipMemset(t.p, 0, 1001); // one byte too many
        const bool seen = damaged(t.base, 1000) == 1;
        t.release();
        if (!clean || !seen) return ULTRA_E_STATE;
    }
    std::lock_guard<std::mutex> lk(g_canary_mu);
    int64_t nb = 0, nd = 0;
    for (auto &kv : g_canaries) {
        if (hipSetDevice(kv.second.second) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { rc = ULTRA_E_HIP; break; }
        const int d = damaged(kv.first, kv.second.first);
        if (d < 0) { rc = ULTRA_E_HIP; break; }
        nb++;
        if (d) { nd++; fprintf(stderr, "ultra_debug_canaries: guard of a %zu-byte device buffer damaged\n", kv.second.first); }
    }
    (void)hipSetDevice(cur);
    if (n_buffers) *n_buffers = nb;
    if (n_damaged) *n_damaged = nd;
    return rc;
}

void ultra_cfg_default(ultra_cfg *cfg)
{
    if (!cfg) return;
    cfg->chain_switch_n = 200; // classify_read_with_mams.py:561
    cfg->overlap_small = 20;   // classify_read_with_mams.py:575
    cfg->overlap_large = 5;    // classify_read_with_mams.py:563
    cfg->loc_slots = 32;
    cfg->banded_traceback = 1;
    cfg->nam_switch_n = 90;    // align.py:448
    cfg->launch_lanes = 0;
    cfg->banded_starts = 0;    // measured slower than the full-height pass on the bench workload (dead ends)
}

int ultra_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ultra_ctx_create(int device, const ultra_cfg *cfg, ultra_ctx **out)
{
    if (!out) return ULTRA_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ULTRA_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return ULTRA_E_NODEVICE;
    ultra_ctx *ctx = new ultra_ctx();
    ctx->device = device;
    if (cfg) ctx->cfg = *cfg; else ultra_cfg_default(&ctx->cfg);
    if (ctx->cfg.loc_slots < 1 || ctx->cfg.loc_slots > 1024 || ctx->cfg.launch_lanes < 0 || ctx->cfg.launch_lanes > kLanes) { delete ctx; return ULTRA_E_ARG; }
    if (ctx->cfg.launch_lanes == 0) {
        // one lane per hardware queue of the process: HIP's default is 4, GPU_MAX_HW_QUEUES raises it
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        long nq = 4;
        if (q && *q) {
            char *endp = nullptr;
            const long v = strtol(q, &endp, 10);
            if (endp != q && *endp == 0 && v >= 1) nq = v; // anything else: HIP's own default
        }
        ctx->cfg.launch_lanes = nq > kLanes ? kLanes : (int)nq;
    }
    if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) { delete ctx; return ULTRA_E_HIP; }
    // the code object is built for gfx950 only; refuse anything else loudly
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) { delete ctx; return ULTRA_E_NODEVICE; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ULTRA_E_HIP; }
    bool ok = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kAuxStreams && ok; i++)
        ok = hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < kXtraStreams && ok; i++)
        ok = hipStreamCreateWithFlags(&ctx->xtra[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->xtra_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { ultra_ctx_destroy(ctx); return ULTRA_E_HIP; }
    {
        // HIP binds a stream to one of the process's hardware queues at the stream's first command (the least loaded
        // queue then).  First commands in creation order, here, give the lanes of one context queues of their own; left
        // to the first batch, the order depended on which launches that batch happened to have, and two lanes of a
        // context could end up behind each other on one queue for the life of the context.
        void *warm = nullptr;
        static const bool no_bind = getenv("ULTRA_DEV_NO_BIND") != nullptr;
        ok = no_bind || hipMalloc(&warm, 256) == hipSuccess;
        if (!no_bind) {
        if (ok) ok = hipMemsetAsync(warm, 0, 4, ctx->stream) == hipSuccess;
        for (int i = 0; i < kAuxStreams && ok; i++) ok = hipMemsetAsync((char *)warm + 8 + 4 * i, 0, 4, ctx->aux[i]) == hipSuccess;
        for (int i = 0; i < kXtraStreams && ok; i++) ok = hipMemsetAsync((char *)warm + 64 + 4 * i, 0, 4, ctx->xtra[i]) == hipSuccess;
        if (ok) ok = hipDeviceSynchronize() == hipSuccess;
        }
        if (warm) (void)hipFree(warm);
        if (!ok) { ultra_ctx_destroy(ctx); return ULTRA_E_HIP; }
    }
    if (ultra_chain_scratch_init(ctx) != ULTRA_OK) { ultra_ctx_destroy(ctx); return ULTRA_E_HIP; }
    *out = ctx;
    return ULTRA_OK;
}

void ultra_ctx_destroy(ultra_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ultra_myers_free(ctx);
    ultra_chain_free(ctx);
    ultra_mam_free(ctx);
    ultra_affine_free(ctx);
    for (int i = 0; i < kAuxStreams; i++) {
        if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); }
        if (ctx->join_ev[i]) (void)hipEventDestroy(ctx->join_ev[i]);
    }
    for (int i = 0; i < kXtraStreams; i++) {
        if (ctx->xtra[i]) { (void)hipStreamSynchronize(ctx->xtra[i]); (void)hipStreamDestroy(ctx->xtra[i]); }
        if (ctx->xtra_ev[i]) (void)hipEventDestroy(ctx->xtra_ev[i]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *ultra_last_error(const ultra_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int ultra_host_alloc(size_t bytes, void **out)
{
    if (!out) return ULTRA_E_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return ULTRA_E_HIP; }
    return ULTRA_OK;
}

void ultra_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int ultra_device_info(const ultra_ctx *ctx, char *name, int name_cap, int *n_cu, int *clock_khz)
{
    if (!ctx) return ULTRA_E_ARG;
    if (name && name_cap > 0) { strncpy(name, ctx->prop.gcnArchName, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = ctx->prop.clockRate;
    return ULTRA_OK;
}

int ultra_myers_stage(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                      const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                      const int32_t *k)
{
    if (!ctx) return ULTRA_E_ARG;
    // the MAM stage's candidate arrays are permuted for the pair batch it staged itself: a new pair batch
    // invalidates them (ultra_mam_run then answers ULTRA_E_STATE instead of reading stale indices)
    ctx->mam.staged = false; ctx->mam.ran = false;
    return ultra_myers_stage_impl(ctx, arena, arena_len, n_pairs, q_off, q_len, t_off, t_len, k);
}

int ultra_myers_run(ultra_ctx *ctx)
{
    if (!ctx) return ULTRA_E_ARG;
    return ultra_myers_run_impl(ctx, false);
}

int ultra_myers_last_timing(const ultra_ctx *ctx, double *cells, float ms[5])
{
    if (!ctx || !ctx->my.ran) return ULTRA_E_STATE;
    if (cells) *cells = ctx->my.cells;
    if (ms) for (int i = 0; i < 5; i++) ms[i] = ctx->my.ms[i];
    return ULTRA_OK;
}

int ultra_myers_cell_counts(const ultra_ctx *ctx, double *cells_computed, double *cells_skipped, int64_t *pairs_skipped)
{
    if (!ctx || !ctx->my.staged) return ULTRA_E_STATE;
    if (cells_computed) *cells_computed = ctx->my.cells;
    if (cells_skipped) *cells_skipped = ctx->my.cells_skipped;
    if (pairs_skipped) *pairs_skipped = ctx->my.n_skipped;
    return ULTRA_OK;
}

int ultra_myers_class_timing(const ultra_ctx *ctx, int32_t cap, int32_t *G, int32_t *W, double *cells, float *fwd_ms,
                             int32_t *n_classes)
{
    if (!ctx || !ctx->my.ran) return ULTRA_E_STATE;
    if (!n_classes || cap < 0 || (cap > 0 && (!G || !W || !cells || !fwd_ms))) return ULTRA_E_ARG;
    int32_t n = 0;
    for (int b = 0; b < 5; b++) {
        const MyersBucket &bk = ctx->my.buckets[b];
        if (bk.n_fwd <= 0) continue;
        if (n < cap) { G[n] = b; W[n] = bk.wmax; cells[n] = bk.cells; fwd_ms[n] = bk.fwd_ms; }
        n++;
    }
    *n_classes = n;
    return ULTRA_OK;
}

static int fetch_impl(ultra_ctx *ctx, int32_t *ed, int32_t *n_loc, int64_t *loc_off,
                      int32_t *loc_start, int32_t *loc_end, int64_t loc_cap, int64_t *loc_need,
                      int32_t *first_ins, uint8_t *ops, const int64_t *ops_off, int32_t *n_ops)
{
    MyersState &st = ctx->my;
    if (!st.ran) return ultra_fail(ctx, ULTRA_E_STATE, "ultra_myers_fetch: no completed run");
    const int32_t n = st.n_pairs;
    if (loc_need) *loc_need = st.total_loc;
    if (n == 0) { if (loc_off) loc_off[0] = 0; return ULTRA_OK; }
    { const int prc = ultra_myers_host_perm(ctx); if (prc != ULTRA_OK) return prc; }
    if (!ed || !n_loc || !loc_off || !first_ins) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_myers_fetch: null output pointer");
    if (st.total_loc > loc_cap || (st.total_loc > 0 && (!loc_start || !loc_end))) {
        return ultra_fail(ctx, ULTRA_E_CAPACITY, "ultra_myers_fetch: location arrays too small, need " + std::to_string(st.total_loc));
    }
    ULTRA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<int32_t> h_ed(n), h_nl(n), h_fi(n), h_no;
    std::vector<int64_t> h_lo((size_t)n + 1);
    std::vector<int32_t> h_ls((size_t)st.total_loc), h_le((size_t)st.total_loc);
    std::vector<uint8_t> h_ops;
    std::vector<int64_t> s_opsoff; // sorted-order offsets used on the device
    ultra_drain_guard drain(ctx); // declared after the vectors: they must outlive every copy enqueued below, also on an error return
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_ed.data(), st.ed.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_nl.data(), st.nloc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_fi.data(), st.firstins.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_lo.data(), st.locoff.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s));
    if (st.total_loc > 0) {
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_ls.data(), st.locstart.p, (size_t)st.total_loc * 4, hipMemcpyDeviceToHost, s));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_le.data(), st.locend.p, (size_t)st.total_loc * 4, hipMemcpyDeviceToHost, s));
    }
    if (ops) {
        h_no.resize(n);
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_no.data(), st.nops.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        s_opsoff.resize((size_t)n + 1);
        s_opsoff[0] = 0;
        for (int32_t i = 0; i < n; i++) { int32_t p = st.perm[i]; s_opsoff[i + 1] = s_opsoff[i] + (ops_off[p + 1] - ops_off[p]); }
        h_ops.resize((size_t)std::max<int64_t>(s_opsoff[n], 1));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(h_ops.data(), st.ops.p, (size_t)s_opsoff[n], hipMemcpyDeviceToHost, s));
    }
    ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(s));
    drain.dismiss();
    // un-permute: results are stored in (class, target length) order on the device
    for (int32_t i = 0; i < n; i++) { int32_t p = st.perm[i]; ed[p] = h_ed[i]; n_loc[p] = h_nl[i]; first_ins[p] = h_fi[i]; }
    loc_off[0] = 0;
    for (int32_t p = 0; p < n; p++) loc_off[p + 1] = loc_off[p] + n_loc[p];
    for (int32_t i = 0; i < n; i++) {
        int32_t p = st.perm[i];
        int64_t src = h_lo[i], dst = loc_off[p];
        for (int32_t a = 0; a < h_nl[i]; a++) { loc_start[dst + a] = h_ls[src + a]; loc_end[dst + a] = h_le[src + a]; }
    }
    if (ops) {
        for (int32_t i = 0; i < n; i++) {
            int32_t p = st.perm[i];
            int64_t cap = ops_off[p + 1] - ops_off[p];
            int32_t cnt = h_no[i];
            if (n_ops) n_ops[p] = cnt;
            if (h_ed[i] < 0) continue;
            if (cnt > cap) return ultra_fail(ctx, ULTRA_E_CAPACITY, "ultra_myers_hw_batch: edit-script slot of pair " + std::to_string(p) + " too small");
            memcpy(ops + ops_off[p], h_ops.data() + s_opsoff[i], (size_t)cnt);
        }
    }
    return ULTRA_OK;
}

int ultra_myers_fetch(ultra_ctx *ctx, int32_t *ed, int32_t *n_loc, int64_t *loc_off,
                      int32_t *loc_start, int32_t *loc_end, int64_t loc_cap, int64_t *loc_need,
                      int32_t *first_ins)
{
    if (!ctx) return ULTRA_E_ARG;
    return fetch_impl(ctx, ed, n_loc, loc_off, loc_start, loc_end, loc_cap, loc_need, first_ins, nullptr, nullptr, nullptr);
}

int ultra_myers_hw_batch(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                         const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                         const int32_t *k, int32_t *ed, int32_t *n_loc, int64_t *loc_off,
                         int32_t *loc_start, int32_t *loc_end, int64_t loc_cap, int64_t *loc_need,
                         int32_t *first_ins, uint8_t *ops, const int64_t *ops_off, int32_t *n_ops)
{
    if (!ctx) return ULTRA_E_ARG;
    ctx->mam.staged = false; ctx->mam.ran = false; // see ultra_myers_stage
    int rc = ultra_myers_stage_impl(ctx, arena, arena_len, n_pairs, q_off, q_len, t_off, t_len, k);
    if (rc != ULTRA_OK) return rc;
    bool want_ops = ops != nullptr;
    if (want_ops) {
        if (!ops_off) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_myers_hw_batch: ops given without ops_off");
        MyersState &st = ctx->my;
        { const int prc = ultra_myers_host_perm(ctx); if (prc != ULTRA_OK) return prc; }
        std::vector<int64_t> s_off((size_t)n_pairs + 1);
        s_off[0] = 0;
        for (int32_t i = 0; i < n_pairs; i++) {
            int32_t p = st.perm[i];
            if (ops_off[p + 1] < ops_off[p]) return ultra_fail(ctx, ULTRA_E_ARG, "ultra_myers_hw_batch: ops_off must be non-decreasing");
            s_off[i + 1] = s_off[i] + (ops_off[p + 1] - ops_off[p]);
        }
        ULTRA_HIP_CHECK(ctx, st.ops.ensure((size_t)s_off[n_pairs] + 16));
        ULTRA_HIP_CHECK(ctx, st.opsoff.ensure(((size_t)n_pairs + 1) * 8));
        ULTRA_HIP_CHECK(ctx, hipMemcpyAsync(st.opsoff.p, s_off.data(), ((size_t)n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        ULTRA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    rc = ultra_myers_run_impl(ctx, want_ops);
    if (rc != ULTRA_OK) return rc;
    return fetch_impl(ctx, ed, n_loc, loc_off, loc_start, loc_end, loc_cap, loc_need, first_ins, ops, ops_off, n_ops);
}

} // extern "C"
