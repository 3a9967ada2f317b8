// Batch pipeline of one GPU: what replaces the N-worker fan-out of /root/reference/modules/pc.py:78-83 on the device
// side.  The reference overlaps reads by running `nr_cores` align_single processes on 1000-read batches
// (modules/pc.py:40, modules/align.py:416-427); here a GPU takes batches of 10^5 reads and the overlap is between
// the PHASES of consecutive batches:
//
//     slot 0:  [ H2D + device plan + sort ][      kernels      ][ D2H ]
//     slot 1:                    [ H2D + device plan + sort ][      kernels      ][ D2H ]
//
// `depth` slots, each a context of its own (streams, device buffers, pinned result buffers) with a host thread that
// walks its batch through stage -> run -> fetch.  The kernels of different slots do not overlap (a pass already fills
// the chip; measured without gain in round 1): `run` is serialised by a mutex, everything else rides beside it.
// The packed index (ultra_dindex) is resident in HBM once and shared by the slots.
#include "common.h"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string.h>
#include <thread>

struct PipeSlot {
    ultra_ctx *ctx = nullptr;
    std::thread th;
    // job
    ultra_plan_request rq;
    double min_acc = 0;
    int state = 0; // 0 idle, 1 submitted, 2 done
    int rc = ULTRA_OK;
    std::string err;
    // results in pinned host memory (grown as needed)
    int32_t n_inst = 0;
    int64_t n_chain = 0;
    void *pin = nullptr;
    size_t pin_cap = 0;
    ultra_mam_output view;
    float ms_stage = 0, ms_run = 0, ms_fetch = 0, ms_wait = 0;
    double cells = 0;
    float myers_ms[5] = {0, 0, 0, 0, 0}, mam_ms[3] = {0, 0, 0};
};

struct ultra_pipe {
    int device = 0;
    int depth = 0;
    const ultra_index *ix = nullptr;
    ultra_dindex *dix = nullptr;
    std::vector<PipeSlot> slots;
    std::mutex mu, run_mu;
    std::condition_variable cv;
    std::deque<int> order; // slots in submission order, oldest first
    int next = 0;
    bool stop = false;
    int held = -1;         // slot whose results the caller is looking at
    std::string err;
};

namespace {

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int slot_fetch(ultra_pipe *p, PipeSlot &sl)
{
    ultra_ctx *ctx = sl.ctx;
    int64_t nchain = 0, nmam = 0;
    int rc = ultra_mam_sizes(ctx, &nchain, &nmam);
    if (rc != ULTRA_OK) return rc;
    const size_t ni = (size_t)sl.rq.n_inst, tc = (size_t)std::max<int64_t>(nchain, 1);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t need = up(ni * 8) + up(ni * 4) + up((ni + 1) * 8) + 7 * up(tc * 4) + up(tc * 8) + up(tc) + up(tc * 4) + 4096;
    if (need > sl.pin_cap) {
        if (sl.pin) (void)hipHostFree(sl.pin);
        sl.pin = nullptr; sl.pin_cap = 0;
        if (hipHostMalloc(&sl.pin, need + need / 4, hipHostMallocDefault) != hipSuccess) { ctx->err = "ultra_pipe: hipHostMalloc failed"; return ULTRA_E_HIP; }
        sl.pin_cap = need + need / 4;
    }
    uint8_t *at = (uint8_t *)sl.pin;
    auto carve = [&](size_t bytes) { uint8_t *r = at; at += up(bytes); return r; };
    ultra_mam_output &o = sl.view;
    memset(&o, 0, sizeof o);
    o.value = (double *)carve(ni * 8); o.n_mam = (int32_t *)carve(ni * 4); o.chain_off = (int64_t *)carve((ni + 1) * 8);
    o.chain_cap = (int64_t)tc;
    o.chain_x = (int32_t *)carve(tc * 4); o.chain_y = (int32_t *)carve(tc * 4); o.chain_c = (int32_t *)carve(tc * 4); o.chain_d = (int32_t *)carve(tc * 4);
    o.chain_val = (double *)carve(tc * 8);
    o.chain_msl = (int32_t *)carve(tc * 4); o.chain_cand = (int32_t *)carve(tc * 4); o.chain_j = (int32_t *)carve(tc * 4);
    o.chain_kind = (uint8_t *)carve(tc); o.chain_chr = (int32_t *)carve(tc * 4);
    rc = ultra_mam_fetch(ctx, &o);
    if (rc != ULTRA_OK) return rc;
    sl.n_inst = sl.rq.n_inst; sl.n_chain = nchain;
    (void)p;
    return ULTRA_OK;
}

void slot_main(ultra_pipe *p, int si)
{
    PipeSlot &sl = p->slots[(size_t)si];
    (void)hipSetDevice(p->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || sl.state == 1; });
            if (p->stop && sl.state != 1) return;
        }
        const double t0 = now_ms();
        // (Measured in round 3: ordering the planner's kernels behind the running slot's start pass -
        // ultra_mam_plan_resident_behind - changes nothing here: 9.5 against 9.7 M reads/s.  The run's tail is short of
        // wavefront slots too, the planner takes 8 ms beside it instead of 4.)
        int rc = ultra_mam_stage_nams(sl.ctx, p->dix, &sl.rq);
        if (rc == ULTRA_E_UNSUPPORTED) {
            // an instance beyond the device planner's envelope (or a 17th symbol): plan this batch on the host
            ultra_plan *pl = nullptr;
            rc = ultra_plan_create(p->ix, &sl.rq, &pl);
            if (rc == ULTRA_OK) { rc = ultra_mam_stage(sl.ctx, ultra_plan_input(pl)); ultra_plan_destroy(pl); }
            else sl.ctx->err = ultra_plan_last_error();
        }
        const double t1 = now_ms();
        double t2 = t1, t3 = t1;
        if (rc == ULTRA_OK) {
            std::lock_guard<std::mutex> run(p->run_mu);
            t2 = now_ms();
            rc = ultra_mam_run(sl.ctx, sl.min_acc);
            t3 = now_ms();
        }
        if (rc == ULTRA_OK) rc = slot_fetch(p, sl);
        const double t4 = now_ms();
        if (rc == ULTRA_OK) {
            (void)ultra_myers_last_timing(sl.ctx, &sl.cells, sl.myers_ms);
            (void)ultra_mam_last_timing(sl.ctx, sl.mam_ms);
        }
        sl.ms_stage = (float)(t1 - t0); sl.ms_wait = (float)(t2 - t1); sl.ms_run = (float)(t3 - t2); sl.ms_fetch = (float)(t4 - t3);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            sl.rc = rc;
            if (rc != ULTRA_OK) sl.err = sl.ctx->err;
            sl.state = 2;
        }
        p->cv.notify_all();
    }
}

} // namespace

extern "C" {

void ultra_pipe_destroy(ultra_pipe *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    for (auto &sl : p->slots) if (sl.th.joinable()) sl.th.join();
    (void)hipSetDevice(p->device);
    for (auto &sl : p->slots) {
        if (sl.ctx) ultra_ctx_destroy(sl.ctx);
        if (sl.pin) (void)hipHostFree(sl.pin);
    }
    if (p->dix) ultra_dindex_destroy(p->dix);
    delete p;
}

int ultra_pipe_create(int device, const ultra_cfg *cfg, const ultra_index *ix, int32_t depth, ultra_pipe **out)
{
    if (!out || !ix) return ULTRA_E_ARG;
    *out = nullptr;
    if (depth < 1 || depth > 4) return ULTRA_E_ARG;
    ultra_pipe *p = new ultra_pipe();
    p->device = device; p->depth = depth; p->ix = ix;
    int rc = ultra_dindex_create(device, ix, &p->dix);
    if (rc != ULTRA_OK) { delete p; return rc; }
    p->slots.resize((size_t)depth);
    for (int i = 0; i < depth; i++) {
        rc = ultra_ctx_create(device, cfg, &p->slots[(size_t)i].ctx);
        if (rc != ULTRA_OK) { ultra_pipe_destroy(p); return rc; }
    }
    for (int i = 0; i < depth; i++) p->slots[(size_t)i].th = std::thread(slot_main, p, i);
    *out = p;
    return ULTRA_OK;
}

const char *ultra_pipe_last_error(const ultra_pipe *p) { return p ? p->err.c_str() : "null pipe"; }

int ultra_pipe_submit(ultra_pipe *p, const ultra_plan_request *rq, double min_acc)
{
    if (!p || !rq) return ULTRA_E_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    if ((int)p->order.size() >= p->depth) { p->err = "ultra_pipe_submit: every slot holds a batch - collect one first (ultra_pipe_wait / ultra_pipe_release)"; return ULTRA_E_STATE; }
    // next idle slot in ring order
    int si = -1;
    for (int t = 0; t < p->depth; t++) { const int c = (p->next + t) % p->depth; if (p->slots[(size_t)c].state == 0 && c != p->held) { si = c; break; } }
    if (si < 0) { p->err = "ultra_pipe_submit: no idle slot"; return ULTRA_E_STATE; }
    p->next = (si + 1) % p->depth;
    PipeSlot &sl = p->slots[(size_t)si];
    sl.rq = *rq; sl.min_acc = min_acc; sl.state = 1; sl.rc = ULTRA_OK;
    p->order.push_back(si);
    lk.unlock();
    p->cv.notify_all();
    return ULTRA_OK;
}

int ultra_pipe_pending(const ultra_pipe *p) { return p ? (int)p->order.size() : 0; }

int ultra_pipe_wait(ultra_pipe *p, int32_t *n_inst, int64_t *n_chain, ultra_mam_output *view)
{
    if (!p) return ULTRA_E_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->held >= 0) { p->err = "ultra_pipe_wait: release the previous batch first"; return ULTRA_E_STATE; }
    if (p->order.empty()) { p->err = "ultra_pipe_wait: nothing submitted"; return ULTRA_E_STATE; }
    const int si = p->order.front();
    PipeSlot &sl = p->slots[(size_t)si];
    p->cv.wait(lk, [&] { return sl.state == 2; });
    p->order.pop_front();
    if (sl.rc != ULTRA_OK) { p->err = sl.err; sl.state = 0; return sl.rc; }
    p->held = si;
    if (n_inst) *n_inst = sl.n_inst;
    if (n_chain) *n_chain = sl.n_chain;
    if (view) *view = sl.view;
    return ULTRA_OK;
}

int ultra_pipe_release(ultra_pipe *p)
{
    if (!p) return ULTRA_E_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->held < 0) return ULTRA_E_STATE;
    p->slots[(size_t)p->held].state = 0;
    p->held = -1;
    return ULTRA_OK;
}

int ultra_pipe_timing(const ultra_pipe *p, float ms[4], double *cells, float myers_ms[5], float mam_ms[3])
{
    if (!p || p->held < 0) return ULTRA_E_STATE;
    const PipeSlot &sl = p->slots[(size_t)p->held];
    if (ms) { ms[0] = sl.ms_stage; ms[1] = sl.ms_wait; ms[2] = sl.ms_run; ms[3] = sl.ms_fetch; }
    if (cells) *cells = sl.cells;
    if (myers_ms) for (int i = 0; i < 5; i++) myers_ms[i] = sl.myers_ms[i];
    if (mam_ms) for (int i = 0; i < 3; i++) mam_ms[i] = sl.mam_ms[i];
    return ULTRA_OK;
}

} // extern "C"
