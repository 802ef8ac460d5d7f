// Host-only check of the placement rules in raytracer_amd/csrc/rtg_pair_slots.hpp: random binary trees in pre-order,
// leaders classified by the rule the placements follow (a leader at every second child and at every node of depth
// kThreadedTopLevels, leader and follower alternating along a chain of first children), slots taken from PairSlots by
// the counts of leaders before -- and then: every pair has a slot of its own, the slots are exactly 0 .. n_pairs - 1,
// the top block comes first level by level, a follower sits right after its leader, and wherever the lines can be
// aligned a leader and its follower share a 128-byte line.  Prints "ok <trees> <pairs>" or the first violation.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../raytracer_amd/csrc/rtg_pair_slots.hpp"

using rtg::PairSlots;

struct Tree {
    std::vector<uint32_t> link;      // pre-order: index after the subtree
    bool inner(uint32_t s) const { return s + 1 < link[s]; }
};

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 7;
    rng_state ^= rng_state >> 9;
    return (uint32_t)(rng_state >> 16);
}

// a random binary tree over `leaves` leaves; `skew` per mille of the leaves go left at every split (0: random)
static void grow(Tree &t, uint32_t leaves, uint32_t skew) {
    const uint32_t s = (uint32_t)t.link.size();
    t.link.push_back(0);
    if (leaves > 1) {
        uint32_t left = skew ? (uint32_t)((uint64_t)leaves * skew / 1000) : 1 + rnd() % (leaves - 1);
        if (left < 1) left = 1;
        if (left > leaves - 1) left = leaves - 1;
        grow(t, left, skew);
        grow(t, leaves - left, skew);
    }
    t.link[s] = (uint32_t)t.link.size();
}

static int fail(const char *what, uint32_t leaves, uint32_t skew, uint32_t s) {
    std::printf("FAILED: %s (leaves %u, skew %u, node %u)\n", what, leaves, skew, s);
    return 1;
}

static int check(uint32_t leaves, uint32_t skew, uint64_t &pairs_seen) {
    Tree t;
    grow(t, leaves, skew);
    const uint32_t n = (uint32_t)t.link.size();
    std::vector<uint32_t> depth(n, 0), chain(n, 0), level_count;
    for (uint32_t s = 0; s < n; s++) {
        if (!t.inner(s)) continue;
        const uint32_t l = s + 1, r = t.link[l];
        depth[l] = depth[r] = depth[s] + 1;
        chain[r] = r;
        chain[l] = depth[s] >= rtg::kThreadedTopLevels ? chain[s] : l;
        if (depth[s] < rtg::kThreadedTopLevels) {
            if (level_count.size() <= depth[s]) level_count.resize(depth[s] + 1, 0);
            level_count[depth[s]]++;
        }
    }
    std::vector<uint32_t> level_first(level_count.size() + 1, 0);
    for (size_t d = 0; d < level_count.size(); d++) level_first[d + 1] = level_first[d] + level_count[d];
    PairSlots ps;
    ps.n_top = level_first.back();
    ps.n_pairs = (n - 1) / 2;
    ps.n_double = ps.n_single = 0;
    auto leader = [&](uint32_t s) { return t.inner(s) && depth[s] >= rtg::kThreadedTopLevels && ((s - chain[s]) & 1u) == 0; };
    for (uint32_t s = 0; s < n; s++)
        if (leader(s)) (t.inner(s + 1) ? ps.n_double : ps.n_single)++;
    std::vector<uint32_t> slot_of(n, 0xFFFFFFFFu), opener(ps.n_single / 2 + 1, 0), level_seen(level_count.size(), 0);
    uint32_t doubles = 0, singles = 0;
    for (uint32_t s = 0; s < n; s++) {
        if (!t.inner(s)) continue;
        if (depth[s] < rtg::kThreadedTopLevels) {
            slot_of[s] = level_first[depth[s]] + level_seen[depth[s]]++;
            continue;
        }
        if (!leader(s)) continue;
        uint32_t slot, opens;
        if (t.inner(s + 1)) {
            slot = ps.double_slot(doubles, singles);
            slot_of[s + 1] = slot + 1;
            doubles++;
        } else {
            if (ps.single_slot(doubles, singles, slot, opens)) {
                if (opens != rtg::kNoOpener) opener[opens] = slot;
            } else {
                slot = opener[opens] + 1;
            }
            singles++;
        }
        slot_of[s] = slot;
    }
    std::vector<uint8_t> taken(ps.n_pairs, 0);
    for (uint32_t s = 0; s < n; s++) {
        if (!t.inner(s)) continue;
        const uint32_t g = slot_of[s];
        if (g == 0xFFFFFFFFu) return fail("an inner node without a slot", leaves, skew, s);
        if (g >= ps.n_pairs) return fail("a slot beyond the last pair", leaves, skew, s);
        if (taken[g]) return fail("two pairs in one slot", leaves, skew, s);
        taken[g] = 1;
        if ((depth[s] < rtg::kThreadedTopLevels) != (g < ps.n_top)) return fail("top block and deep pairs mixed", leaves, skew, s);
        pairs_seen++;
    }
    // a double and its follower share a line (slots 2j-1, 2j) whenever the deep lines can be aligned at all
    const bool alignable = (ps.n_top & 1u) != 0 || ps.n_single != 0;
    for (uint32_t s = 0; s < n; s++)
        if (leader(s) && t.inner(s + 1)) {
            if (slot_of[s + 1] != slot_of[s] + 1) return fail("follower not behind its leader", leaves, skew, s);
            if (alignable && ((64u + 64u * slot_of[s]) >> 7) != ((64u + 64u * slot_of[s + 1]) >> 7))
                return fail("leader and follower in different lines", leaves, skew, s);
        }
    return 0;
}

int main(int argc, char **argv) {
    const uint32_t rounds = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 200;
    uint64_t pairs = 0, trees = 0;
    for (uint32_t leaves = 2; leaves <= 40; leaves++)
        for (uint32_t skew : {0u, 0u, 0u, 10u, 500u, 990u}) {
            if (check(leaves, skew, pairs)) return 1;
            trees++;
        }
    for (uint32_t k = 0; k < rounds; k++) {
        const uint32_t leaves = 2 + rnd() % (k % 10 == 0 ? 200000u : 20000u);
        const uint32_t skew = (k % 4 == 0) ? 0u : (k % 4 == 1 ? 500u : (k % 4 == 2 ? 10u + rnd() % 50 : 950u + rnd() % 45));
        if (check(leaves, skew, pairs)) return 1;
        trees++;
    }
    std::printf("ok %llu %llu\n", (unsigned long long)trees, (unsigned long long)pairs);
    return 0;
}
