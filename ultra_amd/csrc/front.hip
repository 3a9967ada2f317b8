// The per-read steps in front of the MAM stage, for a whole batch of reads, natively (host code, worker threads):
//
//   get_mems_from_input                 /root/reference/modules/align.py:357-388   hit lines -> NAMs per chromosome, sorted by y
//   annotate_guaranteed_optimal_bound   align.py:112-188                           instances split at the intron limit, coverage bound
//   the chaining loop of align_single   align.py:440-475                           order by bound, break rule, solver switch, every
//                                                                                  co-optimal chain, sort by (-score, span), dropoff / max_loc
//   remove_read_polyA_ends              help_functions.py:41-73 (sequence part)
//   reverse_complement                  help_functions.py:93-97
//
// followed by ONE ultra_chain_nams_batch launch for every instance of the batch and ONE device-planned MAM stage over
// every selected (read, location).  The Python of ultra_amd/align.py that did this per read (33 hit lines parsed,
// sorted and wrapped into namedtuples per read: 100+ us) stays as the reference-named shim the parity tests compare
// this file with (tests/test_front_native.py, tests/golden/align_golden.json).
//
// The reference chains the instances of a read one by one and stops at the first whose bound falls below
// dropoff * best value so far (:443); here every instance is chained first and the same rule is applied afterwards in
// the same order, which selects the same set.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string.h>
#include <thread>

namespace {

struct Nam { int32_t x, y, c, d, p0, p1, chr; };
struct Inst { int64_t first; int32_t n; int32_t chr; int32_t read; uint8_t rc; int64_t bound; int64_t limit; };
struct Loc { int32_t read, chr; uint8_t rc; int64_t score; const int64_t *sol; int32_t sol_n; int64_t inst_first; };

// bytes that are written right after they are reserved: no value-initialisation as std::vector::resize would do
struct RawBytes {
    uint8_t *p = nullptr; size_t cap = 0, n = 0;
    RawBytes() = default;
    RawBytes(const RawBytes &) = delete;
    RawBytes &operator=(const RawBytes &) = delete;
    RawBytes(RawBytes &&o) noexcept : p(o.p), cap(o.cap), n(o.n) { o.p = nullptr; o.cap = o.n = 0; }
    ~RawBytes() { free(p); }
    uint8_t *room(size_t more)
    {
        if (n + more > cap) { cap = (n + more) * 2 + 4096; p = (uint8_t *)realloc(p, cap); }
        return p + n;
    }
};

// An array the device copies from or into: page-locked (a copy from pageable memory goes through a staging buffer at a
// third of the rate), grown without keeping its contents (every user writes what it reads), ordinary memory when the
// runtime refuses.  Only the members the code below uses on std::vector.
template <typename T> struct HostBuf {
    T *p = nullptr; size_t n = 0; bool locked = false;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { release(); }
    void release() { if (p) { if (locked) (void)hipHostFree(p); else free(p); } p = nullptr; n = 0; locked = false; }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void resize(size_t m)
    {
        if (m <= n) return;
        release();
        void *q = nullptr;
        if (hipHostMalloc(&q, m * sizeof(T), hipHostMallocDefault) == hipSuccess) locked = true;
        else { (void)hipGetLastError(); q = malloc(m * sizeof(T)); }
        if (!q) throw std::bad_alloc();
        p = (T *)q; n = m;
    }
};

struct Part { // what one worker thread produced for its range of reads (copied to its place in read order afterwards)
    std::vector<Nam> nams;          // instance after instance, in the read's processing order
    std::vector<Inst> insts;        // first = position in this part's nams
    std::vector<int32_t> inst_per_read;
    RawBytes mod;                   // poly-A-trimmed reads
    std::vector<int64_t> mod_len;
    std::string err;
    int err_read = -1;
    void reset() { nams.clear(); insts.clear(); inst_per_read.clear(); mod.n = 0; mod_len.clear(); err.clear(); err_read = -1; }
};
// selection phase, per worker thread
struct SelPart {
    std::vector<Loc> locs; std::vector<int32_t> per_read; std::vector<int64_t> sol; std::vector<int64_t> limit; int64_t nq = 0, nl = 0;
    void reset() { locs.clear(); per_read.clear(); sol.clear(); limit.clear(); nq = nl = 0; }
};

inline bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); } // str.split() on ASCII

// int(token): optional sign, decimal digits (what namfinder writes); anything else is the reference's ValueError
bool parse_int(const uint8_t *b, const uint8_t *e, int64_t &out)
{
    if (b >= e) return false;
    bool neg = false;
    if (*b == '+' || *b == '-') { neg = *b == '-'; b++; }
    if (b >= e || e - b > 18) return false;
    int64_t v = 0;
    for (; b < e; b++) {
        if (*b < '0' || *b > '9') return false;
        v = v * 10 + (*b - '0');
    }
    out = neg ? -v : v;
    return true;
}

// stable; insertion sort for the short lists a read has (std::stable_sort allocates a buffer per call)
template <typename T, typename Less> void stable_small_sort(T *b, T *e, Less less)
{
    if (e - b > 48) { std::stable_sort(b, e, less); return; }
    for (T *i = b + 1; i < e; i++) {
        if (!less(*i, *(i - 1))) continue;
        T v = *i;
        T *j = i;
        while (j > b && less(v, *(j - 1))) { *j = *(j - 1); j--; }
        *j = v;
    }
}

// one strand's hit lines [b, e) -> NAMs grouped by chromosome in order of first appearance, each group sorted by y (stable)
struct ParseScratch { std::vector<Nam> raw; std::vector<int32_t> chr_order; };
bool parse_hits(const uint8_t *b, const uint8_t *e, std::vector<Nam> &out, std::vector<std::pair<int32_t, int32_t>> &groups, ParseScratch &S,
                std::string &err)
{
    std::vector<Nam> &raw = S.raw;
    std::vector<int32_t> &chr_order = S.chr_order;
    raw.clear(); chr_order.clear();
    while (b < e) {
        const uint8_t *le = (const uint8_t *)memchr(b, '\n', (size_t)(e - b));
        if (!le) le = e;
        const uint8_t *tok[4][2];
        int nt = 0;
        const uint8_t *p = b;
        while (p < le && nt < 4) {
            while (p < le && is_space(*p)) p++;
            if (p >= le) break;
            tok[nt][0] = p;
            while (p < le && !is_space(*p)) p++;
            tok[nt][1] = p;
            nt++;
        }
        if (nt < 4) { err = "a hit line has fewer than four fields (IndexError in align.py:365-371)"; return false; }
        // exon_part_id = "chr^start^stop": exactly three fields (:366)
        const uint8_t *c1 = (const uint8_t *)memchr(tok[0][0], '^', (size_t)(tok[0][1] - tok[0][0]));
        const uint8_t *c2 = c1 ? (const uint8_t *)memchr(c1 + 1, '^', (size_t)(tok[0][1] - c1 - 1)) : nullptr;
        if (!c1 || !c2 || memchr(c2 + 1, '^', (size_t)(tok[0][1] - c2 - 1))) { err = "exon_part_id is not chr^start^stop (ValueError in align.py:366)"; return false; }
        int64_t chr, p0, p1, rpos, qpos, len;
        if (!parse_int(tok[0][0], c1, chr) || !parse_int(c1 + 1, c2, p0) || !parse_int(c2 + 1, tok[0][1], p1) ||
            !parse_int(tok[1][0], tok[1][1], rpos) || !parse_int(tok[2][0], tok[2][1], qpos) || !parse_int(tok[3][0], tok[3][1], len)) {
            err = "a hit line holds something that is not an integer (ValueError in align.py:368-374)"; return false;
        }
        const int64_t x = p0 + (rpos - 1), y = x + len - 1, c = qpos - 1, d = c + len - 1;
        const int64_t LIM = 0x7fffffff;
        if (chr < 0 || chr > LIM || c < 0 || d < c || d >= (1 << 30) || x < 0 || y >= LIM || p0 < 0 || p0 > LIM || p1 < 0 || p1 > LIM) {
            err = "hit coordinates outside the supported range (read positions < 2^30, reference positions < 2^31 - 1)"; return false;
        }
        Nam n;
        n.x = (int32_t)x; n.y = (int32_t)y; n.c = (int32_t)c; n.d = (int32_t)d; n.p0 = (int32_t)p0; n.p1 = (int32_t)p1; n.chr = (int32_t)chr;
        raw.push_back(n);
        if (chr_order.empty() || (chr_order.back() != n.chr && std::find(chr_order.begin(), chr_order.end(), n.chr) == chr_order.end())) chr_order.push_back(n.chr);
        b = le < e ? le + 1 : e;
    }
    for (int32_t chr : chr_order) {
        const int32_t first = (int32_t)out.size();
        for (const Nam &n : raw) if (n.chr == chr) out.push_back(n);
        stable_small_sort(out.data() + first, out.data() + out.size(), [](const Nam &a, const Nam &b2) { return a.y < b2.y; });
        groups.push_back({first, (int32_t)out.size() - first});
    }
    return true;
}

// size of the union of the inclusive read intervals [c, d] (the sweep of align.py:157-182)
int64_t covered(const Nam *n, int32_t cnt, std::vector<std::pair<int32_t, int32_t>> &tmp)
{
    tmp.clear();
    bool sorted = true;
    for (int32_t i = 0; i < cnt; i++) {
        tmp.push_back({n[i].c, n[i].d});
        if (i > 0 && tmp[(size_t)i] < tmp[(size_t)i - 1]) sorted = false;
    }
    if (!sorted) stable_small_sort(tmp.data(), tmp.data() + tmp.size(), [](const std::pair<int32_t, int32_t> &a, const std::pair<int32_t, int32_t> &b) { return a < b; });
    int64_t total = 0, end = -1;
    for (auto &iv : tmp)
        if (iv.second > end) { total += iv.second - std::max<int64_t>(iv.first, end + 1) + 1; end = iv.second; }
    return total;
}

struct RcTable {
    uint8_t t[256]; bool ok[256];
    RcTable()
    {
        const char *from = "ACGTUNRYKMSWBVDH", *to = "TGCAANYRMKSWVBHD";
        for (int i = 0; i < 256; i++) { t[i] = 0; ok[i] = false; }
        for (int i = 0; from[i]; i++) {
            t[(uint8_t)from[i]] = (uint8_t)to[i]; ok[(uint8_t)from[i]] = true;
            t[(uint8_t)(from[i] | 0x20)] = (uint8_t)(to[i] | 0x20); ok[(uint8_t)(from[i] | 0x20)] = true;
        }
        t[(uint8_t)'X'] = 'X'; ok[(uint8_t)'X'] = true; // upper case only, as in the reference's table
    }
};
const RcTable kRc;

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <typename F> void parallel_ranges(int32_t n, int n_threads, F &&fn)
{
    const int T = std::max(1, std::min<int>(n_threads, n > 0 ? n : 1));
    if (T == 1) { fn(0, 0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) {
        const int32_t b = (int32_t)((int64_t)n * t / T), e = (int32_t)((int64_t)n * (t + 1) / T);
        th.emplace_back([&fn, t, b, e] { fn(t, b, e); });
    }
    for (auto &x : th) x.join();
}

} // namespace

struct ultra_front {
    std::string err;
    int n_threads = 0;
    // phase 1
    std::vector<Nam> nams;
    std::vector<Inst> insts;
    // worker-thread buffers live as long as the front: memory that is malloc'ed and touched for the first time in every batch
    // costs more than the parsing itself (page faults of a multi-threaded process serialise in the kernel)
    std::vector<Part> parts;
    std::vector<SelPart> sel_parts;
    size_t n_nam = 0, n_inst = 0, n_loc = 0; // the vectors below only grow (no re-initialisation per batch): these are the counts in use
    std::vector<int64_t> read_inst_off;     // instances of read r: [off[r], off[r + 1]) in the reference's processing order
    std::vector<uint8_t> mod;
    std::vector<int64_t> mod_off;
    // phase 2: chaining inputs / outputs
    std::vector<int64_t> inst_off, intron, C, value;
    std::vector<int32_t> ny, nc, nd, trace;
    // phase 3
    std::vector<Loc> locs;
    std::vector<int64_t> loc_off, last_limit;
    std::vector<int32_t> loc_chr;
    std::vector<uint8_t> loc_rc;
    std::vector<int64_t> loc_score;
    // phase 4: the plan request and the MAM stage's results
    // what travels to and from the device lies in page-locked memory (HostBuf)
    HostBuf<int64_t> rq_nam_off, rq_part, rq_x, rq_y, rq_read_off;
    std::vector<int32_t> sol_c, sol_d, sol_j;
    HostBuf<uint8_t> rq_reads;
    HostBuf<double> m_value, m_cval;
    HostBuf<int32_t> m_nmam, m_cx, m_cy, m_cc, m_cd, m_msl, m_cand, m_cj, m_cchr;
    HostBuf<int64_t> m_chain_off;
    HostBuf<uint8_t> m_ckind;
    float ms[6] = {0, 0, 0, 0, 0, 0};
    int64_t n_quadratic = 0, n_nlogn = 0;
};

namespace {

int front_fail(ultra_front *f, int code, const std::string &m) { f->err = m; return code; }

template <typename V> void grow(V &v, size_t n) { if (v.size() < n) v.resize(n + n / 8 + 16); }

int front_prepare(ultra_front *f, const ultra_front_input *in)
{
    const int32_t n = in->n_reads;
    if (n < 0 || (n > 0 && (!in->reads || !in->read_off || !in->fw_begin || !in->fw_end || !in->rc_begin || !in->rc_end)) || in->n_chr < 0 ||
        (in->n_chr > 0 && !in->max_intron_chr))
        return front_fail(f, ULTRA_E_ARG, "ultra_front: null array or negative count");
    const int T = in->n_threads > 0 ? in->n_threads : (f->n_threads > 0 ? f->n_threads : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    const int NP = std::max(1, std::min<int>(T, std::max(n, 1)));
    std::vector<Part> &parts = f->parts;
    if ((int)parts.size() < NP) parts.resize((size_t)NP);
    for (auto &P : parts) P.reset();
    parallel_ranges(n, NP, [&](int t, int32_t rb, int32_t re) {
        Part &P = parts[(size_t)t];
        std::vector<std::pair<int32_t, int32_t>> groups, tmp;
        std::vector<Nam> local[2];
        std::vector<Inst> order;
        ParseScratch scratch;
        const int64_t span = re > rb ? (in->fw_end[re - 1] - in->fw_begin[rb]) + (in->rc_end[re - 1] - in->rc_begin[rb]) : 0;
        if (span > 0) P.nams.reserve((size_t)(span / 16));       // a hit line is at least that long
        P.insts.reserve((size_t)(re - rb) * 2); P.inst_per_read.reserve((size_t)(re - rb)); P.mod_len.reserve((size_t)(re - rb));
        if (re > rb) (void)P.mod.room((size_t)(in->read_off[re] - in->read_off[rb]));
        for (int32_t r = rb; r < re && P.err.empty(); r++) {
            order.clear();
            for (int strand = 0; strand < 2; strand++) {
                const int64_t b = strand ? in->rc_begin[r] : in->fw_begin[r], e = strand ? in->rc_end[r] : in->fw_end[r];
                std::vector<Nam> &loc = local[strand];
                loc.clear(); groups.clear();
                if (e > b && !parse_hits(in->seed_text + b, in->seed_text + e, loc, groups, scratch, P.err)) { P.err_read = r; break; }
                for (auto &g : groups) {
                    const Nam *gn = loc.data() + g.first;
                    const int32_t chr = gn[0].chr;
                    if (chr >= in->n_chr || in->max_intron_chr[chr] < 0) { P.err = "chromosome id without an entry in max_intron_chr (KeyError in align.py:124)"; P.err_read = r; break; }
                    const int64_t limit = std::min<int64_t>(in->max_intron_chr[chr] + 20000, in->max_global_intron);
                    int32_t s = 0;
                    for (int32_t i = 1; i <= g.second; i++) {
                        if (i == g.second || (int64_t)gn[i].x - (int64_t)gn[i - 1].y > limit) { // cut wherever two consecutive seeds are further apart (:133)
                            Inst it;
                            it.first = g.first + s; it.n = i - s; it.chr = chr; it.read = r; it.rc = (uint8_t)strand; it.limit = limit;
                            it.bound = covered(gn + s, i - s, tmp);
                            order.push_back(it);
                            s = i;
                        }
                    }
                }
                if (!P.err.empty()) break;
            }
            if (!P.err.empty()) break;
            // sorted(list(ub.items()) + list(ub_rc.items()), key=bound, reverse=True): stable (:442)
            if (order.size() > 1) stable_small_sort(order.data(), order.data() + order.size(), [](const Inst &a, const Inst &b2) { return a.bound > b2.bound; });
            P.inst_per_read.push_back((int32_t)order.size());
            for (Inst &it : order) {        // the NAMs follow that order: instance i's NAMs are contiguous and ascending in the part
                const Nam *src = local[it.rc].data() + it.first;
                it.first = (int64_t)P.nams.size();
                P.nams.insert(P.nams.end(), src, src + it.n);
                P.insts.push_back(it);
            }
            // remove_read_polyA_ends(seq, qual, threshold, 1)[0]
            const uint8_t *seq = in->reads + in->read_off[r];
            const int64_t len = in->read_off[r + 1] - in->read_off[r];
            const int64_t window = std::min<int64_t>(len / 2, 100);
            const int64_t head = window == 0 ? 0 : len - window;      // seq[:-0] is '' and seq[-0:] the whole read
            uint8_t *dst = P.mod.room((size_t)len);                    // the trimmed read is never longer
            memcpy(dst, seq, (size_t)head);
            int64_t w = head;
            for (int64_t i = head; i < len;) {
                int64_t j = i + 1;
                while (j < len && seq[j] == seq[i]) j++;
                const int64_t run = j - i;
                if (run > in->polyA_threshold && (seq[i] == 'A' || seq[i] == 'T')) dst[w++] = seq[i];
                else { memcpy(dst + w, seq + i, (size_t)run); w += run; }
                i = j;
            }
            P.mod.n += (size_t)w;
            P.mod_len.push_back(w);
        }
    });
    for (auto &P : parts)
        if (!P.err.empty()) return front_fail(f, ULTRA_E_ARG, "ultra_front: read " + std::to_string(P.err_read) + ": " + P.err);
    // ---- every part to its place in read order (the chaining launch's arrays: instances in processing order, NAMs contiguous) ----
    std::vector<size_t> nam_base((size_t)NP + 1, 0), inst_base((size_t)NP + 1, 0), mod_base((size_t)NP + 1, 0), read_base((size_t)NP + 1, 0);
    for (int t = 0; t < NP; t++) {
        nam_base[(size_t)t + 1] = nam_base[(size_t)t] + parts[(size_t)t].nams.size();
        inst_base[(size_t)t + 1] = inst_base[(size_t)t] + parts[(size_t)t].insts.size();
        mod_base[(size_t)t + 1] = mod_base[(size_t)t] + parts[(size_t)t].mod.n;
        read_base[(size_t)t + 1] = read_base[(size_t)t] + parts[(size_t)t].inst_per_read.size();
    }
    const size_t tn = nam_base[(size_t)NP], NI = inst_base[(size_t)NP];
    f->n_nam = tn; f->n_inst = NI;
    grow(f->nams, tn + 1); grow(f->insts, NI + 1); grow(f->mod, mod_base[(size_t)NP] + 16);
    grow(f->read_inst_off, (size_t)n + 1); grow(f->mod_off, (size_t)n + 1);
    grow(f->inst_off, NI + 1); grow(f->intron, NI + 1);
    grow(f->ny, tn + 1); grow(f->nc, tn + 1); grow(f->nd, tn + 1);
    f->read_inst_off[0] = 0; f->mod_off[0] = 0; f->inst_off[0] = 0;
    parallel_ranges(NP, NP, [&](int, int32_t pb, int32_t pe) {
        for (int32_t t = pb; t < pe; t++) {
            const Part &P = parts[(size_t)t];
            const size_t nb = nam_base[(size_t)t], ib = inst_base[(size_t)t], mb = mod_base[(size_t)t], rb = read_base[(size_t)t];
            if (!P.nams.empty()) memcpy(f->nams.data() + nb, P.nams.data(), P.nams.size() * sizeof(Nam));
            for (size_t k = 0; k < P.nams.size(); k++) { f->ny[nb + k] = P.nams[k].y; f->nc[nb + k] = P.nams[k].c; f->nd[nb + k] = P.nams[k].d; }
            for (size_t i = 0; i < P.insts.size(); i++) {
                Inst it = P.insts[i];
                it.first += (int64_t)nb;
                f->insts[ib + i] = it;
                f->inst_off[ib + i + 1] = it.first + it.n;        // the next instance starts where this one ends
                f->intron[ib + i] = it.limit;
            }
            if (P.mod.n) memcpy(f->mod.data() + mb, P.mod.p, P.mod.n);
            int64_t ic = (int64_t)ib, mc = (int64_t)mb;
            for (size_t r = 0; r < P.inst_per_read.size(); r++) {
                ic += P.inst_per_read[r]; mc += P.mod_len[r];
                f->read_inst_off[rb + r + 1] = ic; f->mod_off[rb + r + 1] = mc;
            }
        }
    });
    return ULTRA_OK;
}

// every co-optimal chain of every instance a read's loop reaches, sorted, cut at dropoff / max_loc (align.py:440-475)
int front_select(ultra_front *f, const ultra_front_input *in, int n_threads)
{
    const int32_t n = in->n_reads;
    const int NP = std::max(1, std::min<int>(n_threads, std::max(n, 1)));
    std::vector<SelPart> &parts = f->sel_parts;
    if ((int)parts.size() < NP) parts.resize((size_t)NP);
    for (auto &P : parts) P.reset();
    parallel_ranges(n, NP, [&](int t, int32_t rb, int32_t re) {
        SelPart &P = parts[(size_t)t];
        struct Ch { int64_t inst; int32_t end; int64_t span, value; };
        std::vector<Ch> all;
        std::vector<std::pair<size_t, int32_t>> pending;     // (first sol entry, count) of the part's locations: pointers once sol stops growing
        if (re > rb) {
            const int64_t nn = f->inst_off[(size_t)f->read_inst_off[re]] - f->inst_off[(size_t)f->read_inst_off[rb]];
            P.sol.reserve((size_t)nn + 64); P.locs.reserve((size_t)(re - rb) * 2); P.per_read.reserve((size_t)(re - rb)); P.limit.reserve((size_t)(re - rb));
        }
        for (int32_t r = rb; r < re; r++) {
            all.clear();
            int64_t best = 0, limit = -1;
            for (int64_t i = f->read_inst_off[r]; i < f->read_inst_off[r + 1]; i++) {
                const Inst &it = f->insts[(size_t)i];
                if ((double)it.bound < (double)best * in->dropoff) break;                      // :443
                limit = it.limit;                                                              // :446, as the loop leaves it
                if (it.n < 90) P.nq++; else P.nl++;                                            // :448 (cfg.nam_switch_n's default)
                const int64_t base = f->inst_off[(size_t)i] + i;                               // n + 1 DP entries per instance
                const int64_t v = f->value[(size_t)i];
                if (v > best) best = v;
                const Nam *nm = f->nams.data() + it.first;
                for (int32_t j = 1; j <= it.n; j++) {
                    if (f->C[(size_t)(base + j)] != v) continue;
                    int32_t idx = j, first = j;
                    while (idx > 0) { first = idx; idx = f->trace[(size_t)(base + idx)]; }
                    all.push_back({i, j, (int64_t)nm[j - 1].y - (int64_t)nm[first - 1].x, v});
                }
            }
            P.limit.push_back(limit);
            if (all.size() > 1) stable_small_sort(all.data(), all.data() + all.size(), [](const Ch &a, const Ch &b) { return a.value != b.value ? a.value > b.value : a.span < b.span; });
            int32_t kept = 0;
            for (size_t k = 0; k < all.size(); k++) {
                if ((double)all[k].value / (double)all[0].value < in->dropoff || (int32_t)k >= in->max_loc) break;   // :472
                const Inst &it = f->insts[(size_t)all[k].inst];
                const int64_t base = f->inst_off[(size_t)all[k].inst] + all[k].inst;
                Loc L;
                L.read = r; L.chr = it.chr; L.rc = it.rc; L.score = all[k].value; L.sol = nullptr; L.inst_first = it.first;
                const size_t first = P.sol.size();
                int32_t idx = all[k].end, cnt = 0;
                while (idx > 0) { P.sol.push_back(it.first + idx - 1); idx = f->trace[(size_t)(base + idx)]; cnt++; }
                std::reverse(P.sol.begin() + (int64_t)first, P.sol.end());
                L.sol_n = cnt;
                pending.push_back({first, cnt});
                P.locs.push_back(L);
                kept++;
            }
            P.per_read.push_back(kept);
        }
        for (size_t l = 0; l < P.locs.size(); l++) P.locs[l].sol = P.sol.data() + pending[l].first;
    });
    // ---- one MAM-stage instance per location: offsets first (serial, two additions per location), then the parts in parallel ----
    std::vector<size_t> loc_base((size_t)NP + 1, 0), read_base((size_t)NP + 1, 0);
    f->n_quadratic = f->n_nlogn = 0;
    for (int t = 0; t < NP; t++) {
        loc_base[(size_t)t + 1] = loc_base[(size_t)t] + parts[(size_t)t].locs.size();
        read_base[(size_t)t + 1] = read_base[(size_t)t] + parts[(size_t)t].per_read.size();
        f->n_quadratic += parts[(size_t)t].nq; f->n_nlogn += parts[(size_t)t].nl;
    }
    const size_t NL = loc_base[(size_t)NP];
    f->n_loc = NL;
    grow(f->locs, NL + 1); grow(f->loc_off, (size_t)n + 1); grow(f->last_limit, (size_t)n + 1);
    grow(f->loc_chr, NL + 1); grow(f->loc_rc, NL + 1); grow(f->loc_score, NL + 1);
    grow(f->rq_nam_off, NL + 1); grow(f->rq_read_off, NL + 1);
    f->loc_off[0] = 0; f->rq_nam_off[0] = 0; f->rq_read_off[0] = 0;
    {
        size_t l = 0, r = 0;
        for (int t = 0; t < NP; t++) {
            const SelPart &P = parts[(size_t)t];
            for (const Loc &L : P.locs) {
                f->locs[l] = L;
                f->rq_nam_off[l + 1] = f->rq_nam_off[l] + L.sol_n;
                f->rq_read_off[l + 1] = f->rq_read_off[l] + (f->mod_off[(size_t)L.read + 1] - f->mod_off[(size_t)L.read]);
                l++;
            }
            for (size_t k = 0; k < P.per_read.size(); k++, r++) { f->loc_off[r + 1] = f->loc_off[r] + P.per_read[k]; f->last_limit[r] = P.limit[k]; }
        }
    }
    const size_t NS = (size_t)f->rq_nam_off[NL];
    grow(f->rq_part, 3 * NS + 3); grow(f->rq_x, NS + 1); grow(f->rq_y, NS + 1);
    grow(f->sol_c, NS + 1); grow(f->sol_d, NS + 1); grow(f->sol_j, NS + 1);
    grow(f->rq_reads, (size_t)f->rq_read_off[NL] + 16);
    std::atomic<int32_t> bad_read(-1);
    parallel_ranges((int32_t)NL, n_threads, [&](int, int32_t lb, int32_t le) {
        for (int32_t l = lb; l < le; l++) {
            const Loc &L = f->locs[(size_t)l];
            f->loc_chr[(size_t)l] = L.chr; f->loc_rc[(size_t)l] = L.rc; f->loc_score[(size_t)l] = L.score;
            for (int32_t k = 0; k < L.sol_n; k++) {
                const int64_t gi = L.sol[k];
                const Nam &nm = f->nams[(size_t)gi];
                const size_t o = (size_t)f->rq_nam_off[(size_t)l] + (size_t)k;
                f->sol_j[o] = (int32_t)(gi - L.inst_first);
                f->rq_part[3 * o] = nm.chr; f->rq_part[3 * o + 1] = nm.p0; f->rq_part[3 * o + 2] = nm.p1;
                f->rq_x[o] = nm.x; f->rq_y[o] = nm.y; f->sol_c[o] = nm.c; f->sol_d[o] = nm.d;
            }
            const uint8_t *src = f->mod.data() + f->mod_off[(size_t)L.read];
            const int64_t len = f->mod_off[(size_t)L.read + 1] - f->mod_off[(size_t)L.read];
            uint8_t *dst = f->rq_reads.data() + f->rq_read_off[(size_t)l];
            if (!L.rc) memcpy(dst, src, (size_t)len);
            else for (int64_t i = 0; i < len; i++) {
                const uint8_t ch = src[len - 1 - i];
                if (!kRc.ok[ch]) bad_read.store(L.read);
                dst[i] = kRc.t[ch];
            }
        }
    });
    if (bad_read.load() >= 0)
        return front_fail(f, ULTRA_E_ARG, "ultra_front: read " + std::to_string(bad_read.load()) +
                                              ": a character outside reverse_complement's table (KeyError in help_functions.py:96)");
    return ULTRA_OK;
}

} // namespace

extern "C" {

int ultra_front_create(ultra_front **out)
{
    if (!out) return ULTRA_E_ARG;
    *out = new ultra_front();
    return ULTRA_OK;
}

void ultra_front_destroy(ultra_front *f) { delete f; }

const char *ultra_front_last_error(const ultra_front *f) { return f ? f->err.c_str() : "null front"; }

static void fill_view(ultra_front *f, ultra_front_view *v, bool with_locs, bool with_mam)
{
    memset(v, 0, sizeof *v);
    v->n_inst = (int64_t)f->n_inst;
    v->read_inst_off = f->read_inst_off.data();
    v->n_nam = (int64_t)f->n_nam;
    v->inst_nam_off = f->inst_off.data();
    v->inst_intron = f->intron.data();
    v->nam_y = f->ny.data(); v->nam_c = f->nc.data(); v->nam_d = f->nd.data();
    v->read_mod = f->mod.data(); v->read_mod_off = f->mod_off.data();
    for (int i = 0; i < 6; i++) v->ms[i] = f->ms[i];
    if (!with_locs) return;
    v->n_loc = (int64_t)f->n_loc;
    v->loc_off = f->loc_off.data(); v->last_limit = f->last_limit.data();
    v->loc_chr = f->loc_chr.data(); v->loc_rc = f->loc_rc.data(); v->loc_score = f->loc_score.data();
    v->sol_off = f->rq_nam_off.data(); v->sol_part = f->rq_part.data(); v->sol_x = f->rq_x.data(); v->sol_y = f->rq_y.data();
    v->sol_c = f->sol_c.data(); v->sol_d = f->sol_d.data(); v->sol_j = f->sol_j.data();
    v->loc_reads = f->rq_reads.data(); v->loc_read_off = f->rq_read_off.data();
    v->n_quadratic = f->n_quadratic; v->n_nlogn = f->n_nlogn;
    if (!with_mam) return;
    ultra_mam_output &o = v->mam;
    o.value = f->m_value.data(); o.n_mam = f->m_nmam.data(); o.chain_off = f->m_chain_off.data();
    o.chain_cap = (int64_t)f->m_cx.size();
    o.chain_x = f->m_cx.data(); o.chain_y = f->m_cy.data(); o.chain_c = f->m_cc.data(); o.chain_d = f->m_cd.data();
    o.chain_val = f->m_cval.data(); o.chain_msl = f->m_msl.data(); o.chain_cand = f->m_cand.data(); o.chain_j = f->m_cj.data();
    o.chain_kind = f->m_ckind.data(); o.chain_chr = f->m_cchr.data();
}

// phase 1 alone (no GPU): hit lines -> instances in processing order with their bounds, trimmed reads.  For parity tests.
int ultra_front_prepare(ultra_front *f, const ultra_front_input *in, ultra_front_view *view)
{
    if (!f || !in || !view) return ULTRA_E_ARG;
    const double t0 = now_ms();
    const int rc = front_prepare(f, in);
    if (rc != ULTRA_OK) return rc;
    f->ms[0] = (float)(now_ms() - t0);
    fill_view(f, view, false, false);
    view->inst_chr_rc_bound = nullptr;
    // per instance: chromosome, strand, bound, read - packed for the test as four int64 each
    grow(f->C, 4 * f->n_inst + 4);
    for (size_t i = 0; i < f->n_inst; i++) {
        f->C[4 * i] = f->insts[i].chr; f->C[4 * i + 1] = f->insts[i].rc; f->C[4 * i + 2] = f->insts[i].bound; f->C[4 * i + 3] = f->insts[i].read;
    }
    view->inst_chr_rc_bound = f->C.data();
    return ULTRA_OK;
}

int ultra_front_run(ultra_front *f, ultra_ctx *ctx, const ultra_dindex *dix, const ultra_index *ix, const ultra_front_input *in,
                    int32_t with_mam_stage, ultra_front_view *view)
{
    if (!f || !ctx || !in || !view || (with_mam_stage && !dix)) return ULTRA_E_ARG;
    const int T = in->n_threads > 0 ? in->n_threads : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    double t0 = now_ms();
    int rc = front_prepare(f, in);
    if (rc != ULTRA_OK) return rc;
    double t1 = now_ms();
    f->ms[0] = (float)(t1 - t0);
    // ---- ONE NAM chaining launch for every instance of the batch (align.py:448-453) ----
    const size_t NI = f->n_inst, NN = f->n_nam;
    grow(f->C, NN + NI + 1); grow(f->trace, NN + NI + 1); grow(f->value, NI + 1);  // every entry in use is written by the launch
    if (NI > 0) {
        rc = ultra_chain_nams_batch(ctx, (int32_t)NI, f->inst_off.data(), f->ny.data(), f->nc.data(), f->nd.data(), f->intron.data(), 0,
                                    f->C.data(), f->trace.data(), f->value.data());
        if (rc != ULTRA_OK) return front_fail(f, rc, std::string("ultra_front: NAM chaining: ") + ultra_last_error(ctx));
    }
    double t2 = now_ms();
    f->ms[1] = (float)(t2 - t1);
    rc = front_select(f, in, T);
    if (rc != ULTRA_OK) return rc;
    double t3 = now_ms();
    f->ms[2] = (float)(t3 - t2);
    f->ms[3] = f->ms[4] = f->ms[5] = 0;
    if (!with_mam_stage) { fill_view(f, view, true, false); return ULTRA_OK; }
    // ---- ONE device-planned MAM stage over every (read, location) (:487-489) ----
    ultra_plan_request rq;
    memset(&rq, 0, sizeof rq);
    rq.n_inst = (int32_t)f->n_loc;
    rq.inst_nam_off = f->rq_nam_off.data(); rq.nam_part = f->rq_part.data(); rq.nam_x = f->rq_x.data(); rq.nam_y = f->rq_y.data();
    rq.reads = f->rq_reads.data(); rq.read_off = f->rq_read_off.data();
    {
        static const bool trace = getenv("ULTRA_FRONT_TRACE") != nullptr; // development aid: where the stage time goes
        const double u0 = now_ms();
        rc = ultra_mam_upload_nams(ctx, &rq);
        const double u1 = now_ms();
        if (rc == ULTRA_OK) rc = ultra_mam_plan_resident(ctx, dix);
        if (trace) fprintf(stderr, "ultra_front: upload %.2f ms (page-locked: %d), plan %.2f ms\n", u1 - u0, (int)f->rq_reads.locked, now_ms() - u1);
    }
    if (rc == ULTRA_E_UNSUPPORTED && ix) { // an instance beyond the device planner's envelope: plan this batch on the host
        ultra_plan *pl = nullptr;
        rc = ultra_plan_create(ix, &rq, &pl);
        if (rc == ULTRA_OK) { rc = ultra_mam_stage(ctx, ultra_plan_input(pl)); ultra_plan_destroy(pl); }
        else return front_fail(f, rc, std::string("ultra_front: host plan: ") + ultra_plan_last_error());
    }
    if (rc != ULTRA_OK) return front_fail(f, rc, std::string("ultra_front: MAM stage: ") + ultra_last_error(ctx));
    double t4 = now_ms();
    f->ms[3] = (float)(t4 - t3);
    rc = ultra_mam_run(ctx, in->min_acc);
    if (rc != ULTRA_OK) return front_fail(f, rc, std::string("ultra_front: MAM run: ") + ultra_last_error(ctx));
    double t5 = now_ms();
    f->ms[4] = (float)(t5 - t4);
    int64_t nchain = 0, nmam = 0;
    rc = ultra_mam_sizes(ctx, &nchain, &nmam);
    if (rc != ULTRA_OK) return front_fail(f, rc, ultra_last_error(ctx));
    const size_t ni = std::max<size_t>(f->n_loc, 1), tc = (size_t)std::max<int64_t>(nchain, 1);
    grow(f->m_value, ni); grow(f->m_nmam, ni); grow(f->m_chain_off, ni + 1);
    grow(f->m_cx, tc); grow(f->m_cy, tc); grow(f->m_cc, tc); grow(f->m_cd, tc); grow(f->m_cval, tc); grow(f->m_msl, tc);
    grow(f->m_cand, tc); grow(f->m_cj, tc); grow(f->m_ckind, tc); grow(f->m_cchr, tc);
    fill_view(f, view, true, true);
    rc = ultra_mam_fetch(ctx, &view->mam);
    if (rc != ULTRA_OK) return front_fail(f, rc, std::string("ultra_front: fetch: ") + ultra_last_error(ctx));
    f->ms[5] = (float)(now_ms() - t5);
    for (int i = 0; i < 6; i++) view->ms[i] = f->ms[i];
    return ULTRA_OK;
}

} // extern "C"
