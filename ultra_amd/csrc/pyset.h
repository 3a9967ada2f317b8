// Iteration order of a CPython set of (chr, start, stop) tuples of small non-negative ints.
//
// Why this exists: the reference orders the hit segments of an instance with
//     sorted(set(segment_hit_locations), key=lambda x: x[1])      (classify_read_with_mams.py:180-253)
// and Python's sort is stable, so two segments that START at the same coordinate - frequent in real annotations:
// alternative 3' ends, the widened segments of create_augmented_gene.py:121-192 - keep the order in which the SET
// hands them over.  That order decides which of two identical '_partial_segment_end' pieces is aligned (the `tried`
// filter of :479-500) and with it the stop coordinate of the reported MAM.  It is not random: the hash of an int is the
// int, the hash of a tuple is a fixed xxHash-style mix, so the order is a pure function of the inserted keys and
// their insertion sequence.  This header restates that function (CPython 3.8 ... 3.12: Objects/tupleobject.c tuplehash,
// Objects/setobject.c set_add_entry / set_insert_clean / set_table_resize) for keys that are pairwise distinct and
// never deleted; tests/test_abi.py checks it against the interpreter's own sets.
#pragma once
#include <stdint.h>

#ifndef __host__
#define __host__
#define __device__
#endif

namespace pyset {

constexpr int kLinearProbes = 9;
constexpr int kPerturbShift = 5;

__host__ __device__ inline uint64_t tuple3_hash(uint64_t a, uint64_t b, uint64_t c)
{
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
    uint64_t acc = P5;
    const uint64_t lane[3] = {a, b, c}; // hash(int) == int below 2^61 - 1
    for (int i = 0; i < 3; i++) {
        acc += lane[i] * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 3ull ^ (P5 ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}

// slot a new key lands in; table entries: 0 = empty, else (insertion index + 1)
template <typename T> __host__ __device__ inline uint32_t free_slot(const T *table, uint32_t mask, uint64_t hash)
{
    uint64_t perturb = hash, i = hash & mask;
    for (;;) {
        if (!table[i]) return (uint32_t)i;
        if (i + kLinearProbes <= mask)
            for (int j = 1; j <= kLinearProbes; j++)
                if (!table[i + j]) return (uint32_t)(i + j);
        perturb >>= kPerturbShift;
        i = (i * 5 + 1 + perturb) & mask;
    }
}

// table size after n insertions (8 -> 32 -> 128 -> 512 -> 2048 -> 8192 ...)
__host__ __device__ constexpr uint32_t final_size(int n)
{
    uint32_t size = 8;
    for (int fill = 1; fill <= n; fill++)
        if ((uint64_t)fill * 5 >= (uint64_t)(size - 1) * 3) {
            const uint64_t want = fill > 50000 ? (uint64_t)fill * 2 : (uint64_t)fill * 4;
            uint32_t ns = 8;
            while (ns <= want) ns <<= 1;
            size = ns;
        }
    return size;
}

// Inserts keys 0 .. n-1 (hash[k] each) in that order.  `table` must hold final_size(n) entries, `list` n entries.
// Returns the table's size; table[slot] = k + 1 for occupied slots: ascending slot = iteration order.
template <typename T> __host__ __device__ inline uint32_t build(const uint64_t *hash, int n, T *table, T *list)
{
    uint32_t size = 8;
    for (uint32_t s = 0; s < size; s++) table[s] = 0;
    for (int k = 0; k < n; k++) {
        table[free_slot(table, size - 1, hash[k])] = (T)(k + 1);
        const int fill = k + 1;
        if ((uint64_t)fill * 5 < (uint64_t)(size - 1) * 3) continue;
        const uint64_t want = fill > 50000 ? (uint64_t)fill * 2 : (uint64_t)fill * 4;
        uint32_t ns = 8;
        while (ns <= want) ns <<= 1;
        int m = 0;
        for (uint32_t s = 0; s < size; s++) if (table[s]) list[m++] = table[s];   // old table in slot order
        for (uint32_t s = 0; s < ns; s++) table[s] = 0;
        for (int t = 0; t < m; t++) table[free_slot(table, ns - 1, hash[list[t] - 1])] = list[t];
        size = ns;
    }
    return size;
}

} // namespace pyset
