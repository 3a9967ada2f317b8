// Host-side packed index, shared by the host planner (plan.hip) and the device
// planner (dplan.hip), which flattens it once more into device tables.
#pragma once
#include <stdint.h>

#include <unordered_map>
#include <utility>
#include <vector>

namespace uplan {

struct Key3 {
    int64_t a, b, c;
    bool operator==(const Key3 &o) const { return a == o.a && b == o.b && c == o.c; }
};
struct Key3Hash {
    size_t operator()(const Key3 &k) const
    {
        uint64_t h = (uint64_t)k.a * 0x9E3779B97F4A7C15ull;
        h ^= (uint64_t)k.b + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= (uint64_t)k.c + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        return (size_t)h;
    }
};
struct Slice { int64_t off, len; }; // into the index pool

} // namespace uplan
using uplan::Key3;
using uplan::Key3Hash;
using uplan::Slice;

struct ultra_index {
    std::vector<uint8_t> pool;
    std::unordered_map<Key3, std::pair<int64_t, int64_t>, Key3Hash> part_segs; // part -> [begin, end) in seg_flat (triples)
    std::vector<int64_t> seg_flat;
    std::unordered_map<Key3, std::pair<int64_t, int64_t>, Key3Hash> seg_genes; // segment -> [begin, end) in gene_flat
    std::vector<int32_t> gene_flat;
    std::vector<std::pair<int64_t, int64_t>> gene_small;                       // gene -> [begin, end) in small_flat (triples)
    std::vector<int64_t> small_flat;
    std::unordered_map<Key3, Slice, Key3Hash> seg_seq, flank_seq;
};

