// Host side of SURVEY.md section 8 row f4: the packed index and the candidate enumeration in front of the MAM stage.
//
// Replaces, for a whole batch of (NAM chain, read) instances, the dictionary-driven part of
//     classify_read_with_mams.main                      /root/reference/modules/classify_read_with_mams.py:406-523
//       get_unique_exon_and_flank_locations             :180-253
//       get_unique_segment_and_flank_choordinates       :256-300
//       the candidate loops and the edlib call shapes   :303-377, :441-523
// over flat copies of the five index dictionaries the reference unpickles per worker (modules/align.py:53-110):
// parts_to_segments, segment_to_gene, gene_to_small_segments, ref_segment_sequences, ref_flank_sequences.
// The result is exactly the ultra_mam_input that ultra_mam_stage takes - no Python object per candidate.
// No GPU code in this file; it is part of libultra_hip.so because its output feeds the device stage directly.
#include "common.h"

#include <algorithm>
#include <string.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "plan.h"
#include "pyset.h"

struct ultra_plan {
    std::vector<uint8_t> arena;
    std::vector<int64_t> q_off, t_off;
    std::vector<int32_t> q_len, t_len, k;
    std::vector<uint8_t> kind;
    std::vector<int32_t> e_start, e_stop, pair, pair_contained, seq_len, seq_id, chr, ident_chr;
    std::vector<int64_t> cand_off, fps, lps;
    std::vector<int32_t> read_len;
    ultra_mam_input in;
    double cells = 0;
};

namespace {

thread_local std::string g_plan_err;
int plan_fail(int code, const std::string &msg) { g_plan_err = msg; return code; }

// Python's s[start:stop] on a sequence of length n (either bound may be absent)
inline void py_slice(int64_t n, bool has_start, int64_t start, bool has_stop, int64_t stop, int64_t &b, int64_t &e)
{
    if (!has_start) start = 0;
    else { if (start < 0) start += n; if (start < 0) start = 0; if (start > n) start = n; }
    if (!has_stop) stop = n;
    else { if (stop < 0) stop += n; if (stop < 0) stop = 0; if (stop > n) stop = n; }
    b = start; e = stop < start ? start : stop;
}

inline bool overlaps(int64_t a0, int64_t a1, int64_t b0, int64_t b1) // classify_read_with_mams.py:175-178 (is_overlapping)
{
    return (a0 <= b0 && b0 <= a1) || (a0 <= b1 && b1 <= a1) || (b0 <= a0 && a0 <= a1 && a1 <= b1);
}

struct Loc {
    int64_t chr, s0, s1;
    bool operator==(const Loc &o) const { return chr == o.chr && s0 == o.s0 && s1 == o.s1; }
};
// The reference's sorted(set(hits), key=start) on the distinct hits in arrival order: by start, equal starts in the
// iteration order of the Python set (pyset.h).
void sort_hits_like_the_reference(std::vector<Loc> &hits)
{
    const size_t n = hits.size();
    std::vector<std::pair<std::pair<int64_t, uint32_t>, uint32_t>> keyed(n); // ((start, slot), arrival index)
    bool tie = false;
    {
        std::vector<int64_t> starts(n);
        for (size_t i = 0; i < n; i++) starts[i] = hits[i].s0;
        std::sort(starts.begin(), starts.end());
        tie = std::adjacent_find(starts.begin(), starts.end()) != starts.end();
    }
    std::vector<uint32_t> slot(n, 0);
    if (tie) {
        std::vector<uint64_t> hash(n);
        for (size_t i = 0; i < n; i++) hash[i] = pyset::tuple3_hash((uint64_t)hits[i].chr, (uint64_t)hits[i].s0, (uint64_t)hits[i].s1);
        std::vector<uint32_t> table(pyset::final_size((int)n)), list(n);
        const uint32_t size = pyset::build(hash.data(), (int)n, table.data(), list.data());
        for (uint32_t s = 0; s < size; s++) if (table[s]) slot[table[s] - 1] = s;
    }
    for (size_t i = 0; i < n; i++) keyed[i] = {{hits[i].s0, slot[i]}, (uint32_t)i};
    std::sort(keyed.begin(), keyed.end());
    std::vector<Loc> out(n);
    for (size_t i = 0; i < n; i++) out[i] = hits[keyed[i].second];
    hits.swap(out);
}

struct Cand {
    uint8_t kind;
    Loc loc;
    int64_t ident_chr;
    Slice seq; // into the index pool
};

// arena of one batch: every distinct (pool slice | read) once, pairs de-duplicated by (query, target, k).  A pair
// contains its instance's read, so equal pairs can only meet inside one instance: a short linear list, no hashing.
struct PairTable {
    ultra_plan &p;
    std::unordered_map<Key3, int64_t, Key3Hash> where;  // {source tag, offset, length} -> arena offset
    size_t inst_first = 0;                              // first pair of the current instance
    explicit PairTable(ultra_plan &pl) : p(pl) {}
    void next_instance() { inst_first = p.k.size(); }
    int64_t put(int64_t tag, int64_t off, int64_t len, const uint8_t *bytes)
    {
        const Key3 key{tag, off, len};
        auto it = where.find(key);
        if (it != where.end()) return it->second;
        const int64_t at = (int64_t)p.arena.size();
        p.arena.insert(p.arena.end(), bytes, bytes + len);
        where.emplace(key, at);
        return at;
    }
    int32_t add(int64_t qo, int64_t ql, int64_t to, int64_t tl, int32_t kk)
    {
        for (size_t i = inst_first; i < p.k.size(); i++)
            if (p.q_off[i] == qo && p.t_off[i] == to && p.q_len[i] == ql && p.t_len[i] == tl && p.k[i] == kk) return (int32_t)i;
        const int32_t id = (int32_t)p.k.size();
        p.q_off.push_back(qo); p.q_len.push_back((int32_t)ql);
        p.t_off.push_back(to); p.t_len.push_back((int32_t)tl);
        p.k.push_back(kk);
        p.cells += (double)ql * (double)tl;
        return id;
    }
};

} // namespace

extern "C" const char *ultra_plan_last_error(void) { return g_plan_err.c_str(); }

extern "C" int ultra_pyset_order(const int64_t *triples, int32_t n, int32_t *order)
{
    if (n < 0 || (n > 0 && (!triples || !order))) return ULTRA_E_ARG;
    std::vector<uint64_t> hash((size_t)n);
    for (int32_t i = 0; i < n; i++) {
        for (int t = 0; t < 3; t++) if (triples[3 * i + t] < 0 || triples[3 * i + t] >= ((int64_t)1 << 61) - 1) return ULTRA_E_ARG;
        hash[(size_t)i] = pyset::tuple3_hash((uint64_t)triples[3 * i], (uint64_t)triples[3 * i + 1], (uint64_t)triples[3 * i + 2]);
    }
    std::vector<uint32_t> table(pyset::final_size(n)), list((size_t)n);
    const uint32_t size = pyset::build(hash.data(), n, table.data(), list.data());
    int32_t m = 0;
    for (uint32_t s = 0; s < size; s++) if (table[s]) order[m++] = (int32_t)table[s] - 1;
    return ULTRA_OK;
}

extern "C" int ultra_index_create(const ultra_index_input *in, ultra_index **out)
{
    if (!in || !out) return plan_fail(ULTRA_E_ARG, "ultra_index_create: null argument");
    if (in->n_parts < 0 || in->n_seg_gene < 0 || in->n_genes < 0 || in->n_seg_seq < 0 || in->n_flank_seq < 0 || in->pool_len < 0)
        return plan_fail(ULTRA_E_ARG, "ultra_index_create: negative count");
    ultra_index *ix = new ultra_index();
    ix->pool.assign(in->pool, in->pool + in->pool_len);
    for (int64_t i = 0; i < in->n_parts; i++) {
        const int64_t b = in->part_seg_off[i], e = in->part_seg_off[i + 1];
        ix->part_segs[Key3{in->part_key[3 * i], in->part_key[3 * i + 1], in->part_key[3 * i + 2]}] = {b, e};
    }
    if (in->n_parts > 0) ix->seg_flat.assign(in->part_seg, in->part_seg + 3 * in->part_seg_off[in->n_parts]);
    for (int64_t i = 0; i < in->n_seg_gene; i++)
        ix->seg_genes[Key3{in->seg_gene_key[3 * i], in->seg_gene_key[3 * i + 1], in->seg_gene_key[3 * i + 2]}] =
            {in->seg_gene_off[i], in->seg_gene_off[i + 1]};
    if (in->n_seg_gene > 0) ix->gene_flat.assign(in->seg_gene, in->seg_gene + in->seg_gene_off[in->n_seg_gene]);
    for (int64_t g = 0; g < in->n_genes; g++) ix->gene_small.push_back({in->gene_small_off[g], in->gene_small_off[g + 1]});
    if (in->n_genes > 0) ix->small_flat.assign(in->gene_small, in->gene_small + 3 * in->gene_small_off[in->n_genes]);
    for (int which = 0; which < 2; which++) {
        const int64_t n = which ? in->n_flank_seq : in->n_seg_seq;
        const int64_t *key = which ? in->flank_seq_key : in->seg_seq_key;
        const int64_t *off = which ? in->flank_seq_off : in->seg_seq_off;
        auto &map = which ? ix->flank_seq : ix->seg_seq;
        for (int64_t i = 0; i < n; i++) {
            if (off[i] < 0 || off[i + 1] < off[i] || off[i + 1] > in->pool_len) {
                delete ix;
                return plan_fail(ULTRA_E_ARG, "ultra_index_create: sequence offsets outside the pool");
            }
            map[Key3{key[3 * i], key[3 * i + 1], key[3 * i + 2]}] = Slice{off[i], off[i + 1] - off[i]};
        }
    }
    for (int32_t g : ix->gene_flat)
        if (g < 0 || g >= in->n_genes) { delete ix; return plan_fail(ULTRA_E_ARG, "ultra_index_create: gene id out of range"); }
    *out = ix;
    return ULTRA_OK;
}

extern "C" void ultra_index_destroy(ultra_index *ix) { delete ix; }

extern "C" int ultra_plan_create(const ultra_index *ix, const ultra_plan_request *rq, ultra_plan **out)
{
    if (!ix || !rq || !out) return plan_fail(ULTRA_E_ARG, "ultra_plan_create: null argument");
    if (rq->n_inst < 0) return plan_fail(ULTRA_E_ARG, "ultra_plan_create: negative instance count");
    ultra_plan *pl = new ultra_plan();
    PairTable pairs(*pl);
    {
        const size_t ni = (size_t)rq->n_inst, nc = ni * 24, np = ni * 20;
        pairs.where.reserve(np);
        pl->arena.reserve((size_t)(rq->n_inst > 0 ? rq->read_off[rq->n_inst] - rq->read_off[0] : 0) + np * 160);
        for (auto *v : {&pl->q_len, &pl->t_len, &pl->k}) v->reserve(np);
        pl->q_off.reserve(np); pl->t_off.reserve(np);
        for (auto *v : {&pl->e_start, &pl->e_stop, &pl->pair, &pl->pair_contained, &pl->seq_len, &pl->seq_id, &pl->chr, &pl->ident_chr}) v->reserve(nc);
        pl->kind.reserve(nc);
    }
    pl->cand_off.push_back(0);
    std::vector<Loc> seg_hits, flank_hits, full_order, flank_order;
    std::unordered_map<Key3, std::pair<int64_t, int64_t>, Key3Hash> seg_span, flank_span; // loc -> (first NAM's x, last NAM's y)
    std::unordered_map<Key3, int64_t, Key3Hash> seg_id, full;                              // loc -> chromosome field of its identifier
    std::vector<Cand> cands;
    std::vector<int32_t> genes;
    auto k3 = [](const Loc &l) { return Key3{l.chr, l.s0, l.s1}; };
    for (int32_t inst = 0; inst < rq->n_inst; inst++) {
        seg_hits.clear(); flank_hits.clear(); seg_span.clear(); flank_span.clear(); seg_id.clear(); full.clear();
        cands.clear(); genes.clear(); full_order.clear(); flank_order.clear();
        int64_t first_part_stop = (int64_t)1 << 32, last_part_start = 0;
        // ---- get_unique_exon_and_flank_locations (:180-253) ----
        for (int64_t u = rq->inst_nam_off[inst]; u < rq->inst_nam_off[inst + 1]; u++) {
            const int64_t chr = rq->nam_part[3 * u], p0 = rq->nam_part[3 * u + 1], p1 = rq->nam_part[3 * u + 2];
            const int64_t x = rq->nam_x[u], y = rq->nam_y[u];
            auto part = ix->part_segs.find(Key3{chr, p0, p1});
            if (part != ix->part_segs.end()) {
                if (p1 <= first_part_stop) first_part_stop = p1;
                if (p0 >= last_part_start) last_part_start = p0;
                for (int64_t s = part->second.first; s < part->second.second; s++) {
                    const int64_t t_chr = ix->seg_flat[3 * s], s0 = ix->seg_flat[3 * s + 1], s1 = ix->seg_flat[3 * s + 2];
                    if (!overlaps(s0, s1, x, y)) continue;
                    const Loc loc{chr, s0, s1};
                    seg_id[k3(loc)] = t_chr;
                    auto sp = seg_span.find(k3(loc));
                    if (sp == seg_span.end()) { seg_span[k3(loc)] = {x, y}; seg_hits.push_back(loc); } // distinct, in arrival order
                    else sp->second.second = y;
                }
            } else {
                const Loc loc{chr, p0, p1};
                flank_hits.push_back(loc);
                auto sp = flank_span.find(k3(loc));
                if (sp == flank_span.end()) flank_span[k3(loc)] = {x, y}; else sp->second.second = y;
            }
        }
        sort_hits_like_the_reference(seg_hits);
        // ---- get_unique_segment_and_flank_choordinates (:256-300) ----
        if (!seg_hits.empty()) {
            for (const Loc &loc : seg_hits) {
                auto g = ix->seg_genes.find(Key3{seg_id[k3(loc)], loc.s0, loc.s1});
                if (g == ix->seg_genes.end()) {
                    delete pl;
                    return plan_fail(ULTRA_E_ARG, "ultra_plan_create: instance " + std::to_string(inst) + ": hit segment without a gene "
                                                  "(KeyError in segment_to_gene, classify_read_with_mams.py:264)");
                }
                for (int64_t v = g->second.first; v < g->second.second; v++) genes.push_back(ix->gene_flat[v]);
            }
            std::sort(genes.begin(), genes.end());
            genes.erase(std::unique(genes.begin(), genes.end()), genes.end());
            for (int32_t g : genes)
                for (int64_t v = ix->gene_small[g].first; v < ix->gene_small[g].second; v++) {
                    const Loc loc{ix->small_flat[3 * v], ix->small_flat[3 * v + 1], ix->small_flat[3 * v + 2]};
                    if (full.emplace(k3(loc), loc.chr).second) full_order.push_back(loc);
                }
        }
        for (const Loc &loc : seg_hits)
            if (full.emplace(k3(loc), seg_id[k3(loc)]).second) full_order.push_back(loc);
        for (const Loc &loc : flank_hits) {
            bool seen = false;
            for (const Loc &o : flank_order) if (o == loc) { seen = true; break; }
            if (!seen) flank_order.push_back(loc);
        }
        // ---- candidate list in the reference's emission order (:441-523) ----
        auto seq_of = [&](const std::unordered_map<Key3, Slice, Key3Hash> &map, const Loc &loc, Slice &s) {
            auto it = map.find(k3(loc));
            if (it == map.end()) return false;
            s = it->second;
            return true;
        };
        bool missing = false;
        std::stable_sort(full_order.begin(), full_order.end(), [](const Loc &x, const Loc &y) { return x.s0 < y.s0; });
        for (const Loc &loc : full_order) {
            Cand c; c.kind = 0; c.loc = loc; c.ident_chr = full[k3(loc)];
            if (!seq_of(ix->seg_seq, loc, c.seq)) { missing = true; break; }
            cands.push_back(c);
        }
        std::vector<Loc> flank_sorted = flank_order;
        std::stable_sort(flank_sorted.begin(), flank_sorted.end(), [](const Loc &x, const Loc &y) { return x.s0 < y.s0; });
        for (const Loc &loc : flank_sorted) {
            if (missing) break;
            Cand c; c.kind = 1; c.loc = loc; c.ident_chr = loc.chr;
            if (!seq_of(ix->flank_seq, loc, c.seq)) { missing = true; break; }
            cands.push_back(c);
        }
        auto partial = [&](const Loc &loc, int64_t g0, int64_t g1, int64_t ident_chr, const Slice &whole, uint8_t kind_start) {
            int64_t b, e;
            Cand c; c.loc = loc; c.ident_chr = ident_chr;
            py_slice(whole.len, true, g0 - loc.s0, false, 0, b, e);                            // seq[g0 - start:]
            c.kind = kind_start; c.seq = Slice{whole.off + b, e - b}; cands.push_back(c);
            py_slice(whole.len, false, 0, true, whole.len - (loc.s1 - (g1 + 1)), b, e);        // seq[:len - (stop - (g1 + 1))]
            c.kind = (uint8_t)(kind_start + 1); c.seq = Slice{whole.off + b, e - b}; cands.push_back(c);
        };
        for (const Loc &loc : seg_hits) {
            if (missing) break;
            Slice whole;
            if (!seq_of(ix->seg_seq, loc, whole)) { missing = true; break; }
            const auto &sp = seg_span[k3(loc)];
            partial(loc, sp.first, sp.second, seg_id[k3(loc)], whole, 2);
        }
        for (const Loc &loc : flank_order) {
            if (missing) break;
            const auto &sp = flank_span[k3(loc)];
            const double width = (double)(loc.s1 - loc.s0);
            if (!((double)(sp.first - loc.s0) > 0.05 * width || (double)(loc.s1 - sp.second) > 0.05 * width)) continue; // :288-295
            Slice whole;
            if (!seq_of(ix->flank_seq, loc, whole)) { missing = true; break; }
            partial(loc, sp.first, sp.second, loc.chr, whole, 4);
        }
        if (missing) {
            delete pl;
            return plan_fail(ULTRA_E_ARG, "ultra_plan_create: instance " + std::to_string(inst) + ": candidate without a sequence in the index "
                                          "(KeyError in ref_segment_sequences / ref_flank_sequences)");
        }
        // ---- the edlib calls add_segment_to_mam would make (:304-377) + sequence identity classes (:479-523) ----
        const int64_t r0 = rq->read_off[inst], lr = rq->read_off[inst + 1] - r0;
        if (lr < 0) { delete pl; return plan_fail(ULTRA_E_ARG, "ultra_plan_create: read offsets not ascending"); }
        int64_t read_at = -1;
        const size_t first_cand = pl->kind.size();
        pairs.next_instance();
        for (const Cand &c : cands) {
            int32_t pr = -1, prc = -1;
            const int64_t width = c.loc.s1 - c.loc.s0, ls = c.seq.len;
            const bool skip = ls == 0 || lr == 0 || (c.kind >= 2 && ls <= 5); // pieces of at most 5 bases are never aligned (:488-520)
            if (!skip) {
                if (read_at < 0) read_at = pairs.put(1, r0, lr, rq->reads + r0);
                const int64_t seq_at = pairs.put(0, c.seq.off, ls, ix->pool.data() + c.seq.off);
                const int32_t kk = (int32_t)(0.4 * (double)std::min(lr, ls)); // the binding truncates the float
                if (width >= 5) pr = pairs.add(seq_at, ls, read_at, lr, width >= 9 ? kk : 0);
                if ((double)width >= 0.8 * (double)lr) prc = pairs.add(read_at, lr, seq_at, ls, kk);
            }
            // equal sequences (as strings) share an id within the instance
            int32_t sid = -1;
            for (size_t o = first_cand; o < pl->kind.size() && sid < 0; o++) {
                const Cand &oc = cands[o - first_cand];
                if (oc.seq.len == ls && (oc.seq.off == c.seq.off || memcmp(ix->pool.data() + oc.seq.off, ix->pool.data() + c.seq.off, (size_t)ls) == 0))
                    sid = pl->seq_id[o];
            }
            if (sid < 0) {
                sid = 0;
                for (size_t o = first_cand; o < pl->kind.size(); o++) sid = std::max(sid, pl->seq_id[o] + 1);
            }
            pl->kind.push_back(c.kind);
            pl->e_start.push_back((int32_t)c.loc.s0); pl->e_stop.push_back((int32_t)c.loc.s1);
            pl->pair.push_back(pr); pl->pair_contained.push_back(prc);
            pl->seq_len.push_back((int32_t)ls); pl->seq_id.push_back(sid);
            pl->chr.push_back((int32_t)c.loc.chr); pl->ident_chr.push_back((int32_t)c.ident_chr);
        }
        pl->cand_off.push_back((int64_t)pl->kind.size());
        pl->read_len.push_back((int32_t)lr);
        pl->fps.push_back(first_part_stop); pl->lps.push_back(last_part_start);
    }
    if (pl->arena.empty()) pl->arena.push_back(0);
    ultra_mam_input &in = pl->in;
    memset(&in, 0, sizeof in);
    in.arena = pl->arena.data(); in.arena_len = pl->arena.size();
    in.n_pairs = (int32_t)pl->k.size();
    in.q_off = pl->q_off.data(); in.q_len = pl->q_len.data(); in.t_off = pl->t_off.data(); in.t_len = pl->t_len.data(); in.k = pl->k.data();
    in.n_cand = (int32_t)pl->kind.size();
    in.cand_kind = pl->kind.data(); in.cand_e_start = pl->e_start.data(); in.cand_e_stop = pl->e_stop.data();
    in.cand_pair = pl->pair.data(); in.cand_pair_contained = pl->pair_contained.data();
    in.cand_seq_len = pl->seq_len.data(); in.cand_seq_id = pl->seq_id.data();
    in.n_inst = rq->n_inst;
    in.inst_cand_off = pl->cand_off.data(); in.inst_read_len = pl->read_len.data();
    in.inst_first_part_stop = pl->fps.data(); in.inst_last_part_start = pl->lps.data();
    in.cand_chr = pl->chr.data();
    *out = pl;
    return ULTRA_OK;
}

extern "C" const ultra_mam_input *ultra_plan_input(const ultra_plan *pl) { return pl ? &pl->in : nullptr; }

extern "C" int ultra_plan_cands(const ultra_plan *pl, const int32_t **chr, const int32_t **ident_chr, double *cells)
{
    if (!pl) return plan_fail(ULTRA_E_ARG, "ultra_plan_cands: null plan");
    if (chr) *chr = pl->chr.data();
    if (ident_chr) *ident_chr = pl->ident_chr.data();
    if (cells) *cells = pl->cells;
    return ULTRA_OK;
}

extern "C" void ultra_plan_destroy(ultra_plan *pl) { delete pl; }
