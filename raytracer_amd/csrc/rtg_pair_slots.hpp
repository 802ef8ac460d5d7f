// rtg_pair_slots.hpp -- where the sibling pairs of a threaded tree go (shared by the host placement, the device
// builder and a host-only test: no HIP headers here).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define RTG_HD __host__ __device__
#else
#define RTG_HD
#endif

namespace rtg {

constexpr uint32_t kThreadedTopLevels = 12;

// Where the sibling pairs of a binary tree go below the top block.  Pair slot g holds positions 2 g and 2 g + 1; hot
// node i sits at byte 64 + 32 i of its slab, so slots 2 j - 1 and 2 j share a 128-byte line.  A walk that passes a
// node visits the node's first child next and, when that passes too, the first child's children: a line therefore
// holds the pair under a node X (the "leader") and the pair under X's first child (the "follower") whenever that
// child has children (a "double"); leaders whose first child is a leaf ("singles") share lines two by two.  Along a
// chain of first children leader and follower alternate, starting with a leader at every second child and at every
// node of depth kThreadedTopLevels.  Leaders are placed in pre-order: with `doubles` and `singles` the leaders of
// either kind before a leader, its slot follows from the counts alone -- the host placement and the device builder
// share these rules.
constexpr uint32_t kNoOpener = 0xFFFFFFFFu;
struct PairSlots {
    uint32_t n_top;        // pairs of the top block: the deep pairs take slots n_top .. n_pairs - 1
    uint32_t n_pairs;
    uint32_t n_double, n_single;
    // The first slot that starts a line.  When the deep slots begin in the middle of a line, the lone half slot goes to
    // the first single (if there is one; else the lines stay misaligned, which costs time, not correctness).
    RTG_HD bool lone_first() const { return (n_top & 1u) == 0 && n_single != 0; }
    RTG_HD uint32_t first_line() const { return n_top + (lone_first() ? 1u : 0u); }
    // singles that pair up in lines of their own; an odd one out takes the last slot, which is half a line
    RTG_HD uint32_t paired() const {
        const uint32_t j = n_single - (lone_first() ? 1u : 0u);
        return j & ~1u;
    }
    RTG_HD uint32_t lines_of_singles_before(uint32_t singles) const {
        uint32_t j = singles - (lone_first() && singles ? 1u : 0u);
        if (j > paired()) j = paired();
        return (j + 1) / 2;
    }
    RTG_HD uint32_t double_slot(uint32_t doubles, uint32_t singles) const {
        return first_line() + 2 * (doubles + lines_of_singles_before(singles));
    }
    // true: `slot` is the single's slot, and when `opens` is not kNoOpener the single opens line number `opens` among
    // the singles' lines (the next single joins it); false: the single takes the slot after the one that opened
    // line `opens`.
    RTG_HD bool single_slot(uint32_t doubles, uint32_t singles, uint32_t &slot, uint32_t &opens) const {
        opens = kNoOpener;
        if (lone_first() && singles == 0) {
            slot = n_top;
            return true;
        }
        const uint32_t j = singles - (lone_first() ? 1u : 0u);
        if (j >= paired()) {              // the odd one out
            slot = n_pairs - 1;
            return true;
        }
        opens = j / 2;
        if (j & 1u) return false;
        slot = first_line() + 2 * (doubles + j / 2);
        return true;
    }
};
//

} // namespace rtg
