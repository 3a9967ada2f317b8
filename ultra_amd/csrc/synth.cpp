// libultra_synth.so - synthetic long-read generator for bench.py and the scale tests (plain C++, no GPU code).
//
// NOT part of the product path: GRCh38 / GENCODE / a read simulator / namfinder do not exist offline, so the
// BASELINE.json configs are run on self-generated data of the same shape.  The Python
// generator (ultra_amd/synth.py) costs about a millisecond per read; a million-read workload needs this one.
// Same model as synth.simulate_instance: a read is a (possibly truncated) transcript with the reference simulator's
// error mix - of the erroneous positions 50 % deletions / 30 % substitutions / 20 % insertions, geometric
// extension p = 0.5 (/root/reference/evaluation/simulate_reads.py:77-102) - and its NAM chain is the set of exact
// runs of at least min_nam bases, cut at exon-part borders, in the coordinates align.py hands to the MAM stage
// (/root/reference/modules/align.py:357-388).  Deterministic in (seed, read index): threads only split the range.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

extern "C" {

typedef struct synth_world {
    int64_t n_tr;
    const int64_t *tr_off;      // [n_tr + 1] -> tr_piece (exon parts of a transcript, ascending)
    const int64_t *tr_piece;    // piece ids
    const int64_t *tr_up, *tr_down; // flank piece ids per transcript
    const int64_t *piece_chr, *piece_start, *piece_stop, *piece_seq; // [n_pieces]; piece_seq = offset into pool
    const uint8_t *pool;
} synth_world;

typedef struct synth_params {
    uint64_t seed;
    int64_t n_reads;
    double err;
    int32_t len_model;          // 0: normal(len_a, len_b) clipped below at len_min; 1: log-normal(median len_a, sigma len_b) clipped to [len_min, len_max]
    double len_a, len_b;
    int32_t len_min, len_max;
    double bias3p;              // share of truncated reads that keep the 3' end (dRNA), the rest start uniformly
    double flank_start_prob, flank_end_prob;
    int32_t min_nam;
    int32_t n_threads;
} synth_params;

typedef struct synth_out {
    std::vector<uint8_t> reads;
    std::vector<int64_t> read_off, nam_off, nam_part, nam_x, nam_y, nam_c, nam_d;
    std::vector<int32_t> chr;
} synth_out;

} // extern "C"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) { next(); next(); }
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    int64_t below(int64_t n) { return n <= 1 ? 0 : (int64_t)(next() % (uint64_t)n); }
    double normal()
    {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct Piece { int64_t id, off, len; }; // piece id, offset inside the piece, length taken

void one_read(const synth_world &w, const synth_params &p, uint64_t index, synth_out &o)
{
    static const char kBases[] = "ACGT";
    std::vector<Piece> pieces, cut;
    std::vector<uint8_t> ref, read;
    std::vector<int64_t> src;
    for (uint64_t attempt = 0;; attempt++) {
        Rng rng(p.seed * 0x100000001B3ull + index * 0x9E3779B97F4A7C15ull + attempt * 0xD1B54A32D192ED03ull);
        const int64_t t = rng.below(w.n_tr);
        pieces.clear();
        if (rng.uni() < p.flank_start_prob) {
            const int64_t f = w.tr_up[t], fl = w.piece_stop[f] - w.piece_start[f];
            const int64_t ext = std::min<int64_t>(20 + rng.below(380), fl);
            if (ext > 0) pieces.push_back({f, fl - ext, ext});
        }
        for (int64_t u = w.tr_off[t]; u < w.tr_off[t + 1]; u++) {
            const int64_t id = w.tr_piece[u];
            pieces.push_back({id, 0, w.piece_stop[id] - w.piece_start[id]});
        }
        if (rng.uni() < p.flank_end_prob) {
            const int64_t f = w.tr_down[t], fl = w.piece_stop[f] - w.piece_start[f];
            const int64_t ext = std::min<int64_t>(20 + rng.below(380), fl);
            if (ext > 0) pieces.push_back({f, 0, ext});
        }
        int64_t total = 0;
        for (auto &pc : pieces) total += pc.len;
        int64_t want;
        if (p.len_model == 1) want = (int64_t)(p.len_a * std::exp(p.len_b * rng.normal()));
        else want = (int64_t)(p.len_a + p.len_b * rng.normal());
        want = std::max<int64_t>(want, p.len_min);
        if (p.len_max > 0) want = std::min<int64_t>(want, p.len_max);
        if (want < total) {
            const int64_t off = rng.uni() < p.bias3p ? total - want : rng.below(total - want + 1);
            const int64_t lo = off, hi = off + want;
            cut.clear();
            int64_t acc = 0;
            for (auto &pc : pieces) {
                const int64_t s = std::max(lo, acc), e = std::min(hi, acc + pc.len);
                if (e > s) cut.push_back({pc.id, pc.off + (s - acc), e - s});
                acc += pc.len;
            }
            pieces.swap(cut);
        }
        ref.clear();
        for (auto &pc : pieces) ref.insert(ref.end(), w.pool + w.piece_seq[pc.id] + pc.off, w.pool + w.piece_seq[pc.id] + pc.off + pc.len);
        // ---- errors ----
        read.clear(); src.clear();
        const int64_t n = (int64_t)ref.size();
        for (int64_t i = 0; i < n;) {
            if (rng.uni() < p.err) {
                const double r = rng.uni();
                if (r < 0.5) { // deletion, geometric length
                    int64_t dl = 1;
                    while (rng.uni() >= 0.5) dl++;
                    i += dl;
                    continue;
                } else if (r < 0.8) { // substitution (may re-draw the same base, as the reference's simulator does)
                    read.push_back((uint8_t)kBases[rng.below(4)]); src.push_back(-1);
                } else { // insertion after the base
                    read.push_back(ref[(size_t)i]); src.push_back(i);
                    read.push_back((uint8_t)kBases[rng.below(4)]); src.push_back(-1);
                    while (rng.uni() >= 0.5) { read.push_back((uint8_t)kBases[rng.below(4)]); src.push_back(-1); }
                }
            } else { read.push_back(ref[(size_t)i]); src.push_back(i); }
            i++;
        }
        if ((int64_t)read.size() < 30) continue;
        // ---- NAMs: maximal runs of copied, consecutive bases, cut at piece borders ----
        const size_t nam0 = o.nam_x.size();
        const int64_t rl = (int64_t)read.size();
        int64_t run = -1;
        size_t pi = 0;
        int64_t pbeg = 0; // start of pieces[pi] on the spliced reference
        auto flush = [&](int64_t r0, int64_t r1) {
            int64_t g0 = src[(size_t)r0];
            while (r0 <= r1) {
                while (pi + 1 < pieces.size() && g0 >= pbeg + pieces[pi].len) { pbeg += pieces[pi].len; pi++; }
                const Piece &pc = pieces[pi];
                const int64_t take = std::min<int64_t>(r1 - r0 + 1, pbeg + pc.len - g0);
                if (take >= p.min_nam) {
                    const int64_t x = w.piece_start[pc.id] + pc.off + (g0 - pbeg);
                    o.nam_x.push_back(x); o.nam_y.push_back(x + take - 1);
                    o.nam_c.push_back(r0); o.nam_d.push_back(r0 + take - 1);
                    o.nam_part.push_back(w.piece_chr[pc.id]); o.nam_part.push_back(w.piece_start[pc.id]); o.nam_part.push_back(w.piece_stop[pc.id]);
                }
                r0 += take; g0 += take;
            }
        };
        for (int64_t r = 0; r <= rl; r++) {
            const bool ok = r < rl && src[(size_t)r] >= 0 && (run < 0 || src[(size_t)r] == src[(size_t)r - 1] + 1);
            if (run >= 0 && !ok) { flush(run, r - 1); run = -1; }
            if (run < 0 && r < rl && src[(size_t)r] >= 0) run = r;
        }
        if (o.nam_x.size() == nam0) continue; // no seed: the read would never reach the MAM stage
        // NAMs are produced in read order = reference order (one transcript, forward strand): already sorted by y
        o.reads.insert(o.reads.end(), read.begin(), read.end());
        o.read_off.push_back((int64_t)o.reads.size());
        o.nam_off.push_back((int64_t)o.nam_x.size());
        o.chr.push_back((int32_t)w.piece_chr[pieces[0].id]);
        return;
    }
}

} // namespace

extern "C" {

synth_out *synth_generate(const synth_world *w, const synth_params *p)
{
    if (!w || !p || w->n_tr <= 0 || p->n_reads < 0) return nullptr;
    const int nt = std::max(1, std::min<int>(p->n_threads, 64));
    std::vector<synth_out> parts((size_t)nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) {
        th.emplace_back([&, t]() {
            const int64_t b = p->n_reads * t / nt, e = p->n_reads * (t + 1) / nt;
            synth_out &o = parts[(size_t)t];
            o.read_off.push_back(0); o.nam_off.push_back(0);
            for (int64_t i = b; i < e; i++) one_read(*w, *p, (uint64_t)i, o);
        });
    }
    for (auto &x : th) x.join();
    synth_out *out = new synth_out();
    out->read_off.push_back(0); out->nam_off.push_back(0);
    for (auto &o : parts) {
        const int64_t rb = (int64_t)out->reads.size(), nb = (int64_t)out->nam_x.size();
        out->reads.insert(out->reads.end(), o.reads.begin(), o.reads.end());
        for (size_t i = 1; i < o.read_off.size(); i++) out->read_off.push_back(o.read_off[i] + rb);
        for (size_t i = 1; i < o.nam_off.size(); i++) out->nam_off.push_back(o.nam_off[i] + nb);
        out->nam_part.insert(out->nam_part.end(), o.nam_part.begin(), o.nam_part.end());
        out->nam_x.insert(out->nam_x.end(), o.nam_x.begin(), o.nam_x.end());
        out->nam_y.insert(out->nam_y.end(), o.nam_y.begin(), o.nam_y.end());
        out->nam_c.insert(out->nam_c.end(), o.nam_c.begin(), o.nam_c.end());
        out->nam_d.insert(out->nam_d.end(), o.nam_d.begin(), o.nam_d.end());
        out->chr.insert(out->chr.end(), o.chr.begin(), o.chr.end());
        o = synth_out();
    }
    return out;
}

void synth_sizes(const synth_out *o, int64_t *n_reads, int64_t *read_bytes, int64_t *n_nams)
{
    *n_reads = (int64_t)o->read_off.size() - 1; *read_bytes = (int64_t)o->reads.size(); *n_nams = (int64_t)o->nam_x.size();
}

void synth_copy(const synth_out *o, uint8_t *reads, int64_t *read_off, int64_t *nam_off, int64_t *nam_part, int64_t *nam_x,
                int64_t *nam_y, int64_t *nam_c, int64_t *nam_d, int32_t *chr)
{
    memcpy(reads, o->reads.data(), o->reads.size());
    memcpy(read_off, o->read_off.data(), o->read_off.size() * 8);
    memcpy(nam_off, o->nam_off.data(), o->nam_off.size() * 8);
    memcpy(nam_part, o->nam_part.data(), o->nam_part.size() * 8);
    memcpy(nam_x, o->nam_x.data(), o->nam_x.size() * 8);
    memcpy(nam_y, o->nam_y.data(), o->nam_y.size() * 8);
    memcpy(nam_c, o->nam_c.data(), o->nam_c.size() * 8);
    memcpy(nam_d, o->nam_d.data(), o->nam_d.size() * 8);
    memcpy(chr, o->chr.data(), o->chr.size() * 4);
}

void synth_free(synth_out *o) { delete o; }

// ---- (query, target, k) pairs of the Myers micro-benchmarks ------------------------------------
// Targets are random sequences shared by `per_target` queries each; a query is either planted (a slice of its target
// with the simulator's error mix at rate err) or an unrelated decoy (the reference aligns every small segment of a hit
// gene, so most calls are decoys); k = (2 * min(m, n)) / 5 as classify_read_with_mams.py:312 passes it.
//   q_model 0: length uniform in [q_a, q_b]; 1: log-normal (median q_a, sigma q_sigma) clipped to [q_min, q_max];
//           2: q_a or q_b with equal probability
//   t_len > 0: fixed target length; t_len <= 0: target = t_factor x the query's nominal length (one query per target)
typedef struct synth_pair_params {
    uint64_t seed;
    int64_t n_pairs;
    int32_t q_model, q_a, q_b, q_min, q_max;
    double q_sigma;
    int32_t t_len;
    double t_factor;
    int32_t per_target;
    double err, planted;
} synth_pair_params;

typedef struct synth_pairs_out {
    std::vector<uint8_t> arena;
    std::vector<int64_t> q_off, t_off;
    std::vector<int32_t> q_len, t_len, k;
} synth_pairs_out;

synth_pairs_out *synth_pairs(const synth_pair_params *p)
{
    static const char kBases[] = "ACGT";
    if (!p || p->n_pairs < 0) return nullptr;
    synth_pairs_out *o = new synth_pairs_out();
    Rng rng(p->seed * 0x9E3779B97F4A7C15ull + 12345);
    auto qlen = [&]() -> int64_t {
        if (p->q_model == 0) return p->q_a + rng.below((int64_t)p->q_b - p->q_a + 1);
        if (p->q_model == 1) {
            int64_t v = (int64_t)(p->q_a * std::exp(p->q_sigma * rng.normal()));
            return std::max<int64_t>(p->q_min, std::min<int64_t>(p->q_max, v));
        }
        return rng.uni() < 0.5 ? p->q_a : p->q_b;
    };
    std::vector<uint8_t> q;
    int64_t done = 0;
    while (done < p->n_pairs) {
        int64_t m0 = qlen();
        const int64_t n = p->t_len > 0 ? p->t_len : std::max<int64_t>(1, (int64_t)(p->t_factor * (double)m0));
        const int64_t t_at = (int64_t)o->arena.size();
        for (int64_t i = 0; i < n; i++) o->arena.push_back((uint8_t)kBases[rng.below(4)]);
        const int64_t per = p->t_len > 0 ? std::max<int32_t>(1, p->per_target) : 1;
        for (int64_t j = 0; j < per && done < p->n_pairs; j++, done++) {
            if (j > 0) m0 = qlen();
            q.clear();
            if (rng.uni() < p->planted && m0 <= n) {
                const int64_t a = rng.below(n - m0 + 1);
                for (int64_t i = a; i < a + m0;) {
                    const uint8_t c = o->arena[(size_t)(t_at + i)];
                    if (rng.uni() < p->err) {
                        const double r = rng.uni();
                        if (r < 0.5) { int64_t dl = 1; while (rng.uni() >= 0.5) dl++; i += dl; continue; }
                        else if (r < 0.8) q.push_back((uint8_t)kBases[rng.below(4)]);
                        else { q.push_back(c); q.push_back((uint8_t)kBases[rng.below(4)]); while (rng.uni() >= 0.5) q.push_back((uint8_t)kBases[rng.below(4)]); }
                    } else q.push_back(c);
                    i++;
                }
                if (q.empty()) q.push_back((uint8_t)'A');
            } else
                for (int64_t i = 0; i < m0; i++) q.push_back((uint8_t)kBases[rng.below(4)]);
            const int64_t m = (int64_t)q.size();
            o->q_off.push_back((int64_t)o->arena.size()); o->q_len.push_back((int32_t)m);
            o->arena.insert(o->arena.end(), q.begin(), q.end());
            o->t_off.push_back(t_at); o->t_len.push_back((int32_t)n);
            o->k.push_back((int32_t)((2 * std::min(m, n)) / 5));
        }
    }
    return o;
}

void synth_pairs_sizes(const synth_pairs_out *o, int64_t *n_pairs, int64_t *arena_bytes)
{
    *n_pairs = (int64_t)o->k.size(); *arena_bytes = (int64_t)o->arena.size();
}

void synth_pairs_copy(const synth_pairs_out *o, uint8_t *arena, int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len, int32_t *k)
{
    memcpy(arena, o->arena.data(), o->arena.size());
    memcpy(q_off, o->q_off.data(), o->q_off.size() * 8); memcpy(q_len, o->q_len.data(), o->q_len.size() * 4);
    memcpy(t_off, o->t_off.data(), o->t_off.size() * 8); memcpy(t_len, o->t_len.data(), o->t_len.size() * 4);
    memcpy(k, o->k.data(), o->k.size() * 4);
}

void synth_pairs_free(synth_pairs_out *o) { delete o; }

// The seed text namfinder would write for reads with these NAM chains (modules/seed_wrapper.py:139-179 format), as hit
// lines only: read r's lines are text[line_off[r] .. line_off[r + 1]).  Returns the bytes written, or the bytes needed
// (negative) when cap is too small.
int64_t synth_seed_text(int64_t n_reads, const int64_t *nam_off, const int64_t *nam_part, const int64_t *nam_x, const int64_t *nam_y,
                        const int64_t *nam_c, uint8_t *text, int64_t cap, int64_t *line_off)
{
    auto put = [](uint8_t *at, int64_t v) {
        char tmp[24]; int n = 0;
        if (v == 0) tmp[n++] = '0';
        bool neg = v < 0; if (neg) v = -v;
        while (v > 0) { tmp[n++] = (char)('0' + v % 10); v /= 10; }
        int w = 0;
        if (neg) at[w++] = '-';
        while (n > 0) at[w++] = (uint8_t)tmp[--n];
        return w;
    };
    const int64_t need = nam_off[n_reads] * 100;
    if (need > cap) return -need;
    int64_t at = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        line_off[r] = at;
        for (int64_t u = nam_off[r]; u < nam_off[r + 1]; u++) {
            at += put(text + at, nam_part[3 * u]); text[at++] = '^';
            at += put(text + at, nam_part[3 * u + 1]); text[at++] = '^';
            at += put(text + at, nam_part[3 * u + 2]); text[at++] = '\t';
            at += put(text + at, nam_x[u] - nam_part[3 * u + 1] + 1); text[at++] = '\t';
            at += put(text + at, nam_c[u] + 1); text[at++] = '\t';
            at += put(text + at, nam_y[u] - nam_x[u] + 1); text[at++] = '\n';
        }
    }
    line_off[n_reads] = at;
    return at;
}

} // extern "C"
