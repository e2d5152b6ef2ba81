#include "lds_program.hpp"

#include <algorithm>
#include <stdexcept>

using zkgpu::LdsOp;

namespace zki {

namespace {

// rows know and / xor only: `not a` = a xor ONES, copy = a xor ZERO (constant slots behind the scratch slots)
int row_kind(uint32_t kind) {   // 1 xor, 0 and, -1 not a row kind
  switch (kind) {
    case TK_XOR: case TK_NOT: case TK_COPY: return 1;
    case TK_AND: return 0;
    default: return -1;
  }
}

LdsOp encode(const DevOp& d) {
  LdsOp o;
  o.kind = (unsigned short)d.kind;
  o.dst = (unsigned short)d.dst;
  o.a = (unsigned short)d.a;
  o.b = (unsigned short)d.b;
  if (d.kind == TK_INSTANCE || d.kind == TK_WITNESS) {  // 32-bit position in a | b << 16
    o.a = (unsigned short)(d.a & 0xFFFF);
    o.b = (unsigned short)(d.a >> 16);
  } else if (d.kind == TK_ASSERT) {                      // 32-bit sequence in dst | b << 16
    o.dst = (unsigned short)(d.b & 0xFFFF);
    o.b = (unsigned short)(d.b >> 16);
  } else if (d.kind == TK_CONST && d.a > 0xFFFF) {
    throw std::runtime_error("Engine: too many constants for the LDS program encoding");
  }
  return o;
}

}  // namespace

bool lds_program_fits(const Schedule& s) {
  constexpr uint64_t kLdsBytes = 160 * 1024;
  return ((uint64_t)s.n_slots + zkgpu::kLdsExtraSlots) * 4 + 64 <= kLdsBytes &&
         s.n_slots + zkgpu::kLdsExtraSlots < 0xFFFF &&   // + scratch and constant slots, 16-bit slot numbers
         // block headers carry 32-bit byte offsets into the row stream: 6 bytes per op + the padding of every kind-run of a
         // level to whole rows (copy, and, xor, not: up to four runs per launch, each less than one row of 12288 bytes) +
         // the slack rows behind the last block.  build_lds_program checks the real offsets again as it writes them.
         (uint64_t)s.ops.size() * 6 + s.launches.size() * (4ull * zkgpu::kLdsRowOps * 6) + (1u << 22) < (1ull << 32);
}

LdsProgram build_lds_program(const Schedule& s, const std::vector<uint32_t>& block_sizes, uint32_t forced_block_rows) {
  LdsProgram P;
  P.n_slots = s.n_slots + zkgpu::kLdsExtraSlots;
  std::vector<LdsOp>& lo = P.ops;
  std::vector<uint16_t>& lo6 = P.rows;
  std::vector<uint32_t>& blocks = P.blocks;
  std::vector<uint32_t>& ln = P.chunks;
  lo.reserve(s.ops.size() / 8 + 4096);
  lo6.reserve(s.ops.size() * 3 + 3 * 4096 * s.launches.size());
  const unsigned short scratch = (unsigned short)s.n_slots;  // extra slots: targets of the padding ops
  const unsigned short zero = (unsigned short)(scratch + zkgpu::kLdsZeroSlot), ones = (unsigned short)(scratch + zkgpu::kLdsOnesSlot);

  // Rows per block: every block fetches that many rows of the stream whatever it holds (the kernel counts its loads)
  // -- pick the size that fetches least for this program (C4: a level is 9 rows; 12-row blocks would fetch a third
  // more).  A block costs about a row of time by itself.
  uint32_t block_rows = block_sizes.empty() ? (uint32_t)zkgpu::kLdsMaxBlockRows : block_sizes.back();
  {
    std::vector<uint32_t> level_rows;   // rows of every maximal sequence of xor / and / not / copy rows
    for (const Launch& L : s.launches) {
      if (L.sequential) continue;
      uint32_t k = 0, open = 0;
      while (k < L.count) {
        const uint32_t kind = s.ops[L.first + k].kind;
        uint32_t e = k;
        while (e < L.count && s.ops[L.first + e].kind == kind) ++e;
        if (row_kind(kind) >= 0) {
          open += (e - k + zkgpu::kLdsRowOps - 1) / zkgpu::kLdsRowOps;
        } else if (open) {
          level_rows.push_back(open);
          open = 0;
        }
        k = e;
      }
      if (open) level_rows.push_back(open);
    }
    uint64_t best = ~0ull;
    for (uint32_t br : block_sizes) {
      uint64_t cost = 0;
      for (uint32_t n : level_rows) cost += (uint64_t)((n + br - 1) / br) * (br + 1);
      if (cost < best) {
        best = cost;
        block_rows = br;
      }
    }
  }
  if (forced_block_rows && std::find(block_sizes.begin(), block_sizes.end(), forced_block_rows) != block_sizes.end())
    block_rows = forced_block_rows;
  P.block_rows = block_rows;

  // rows of the level being emitted that are not in a block yet
  std::vector<uint32_t> open_kinds;
  uint32_t open_first = 0;
  auto close_blocks = [&](bool barrier) {
    for (size_t r = 0; r < open_kinds.size(); r += block_rows) {
      const size_t n = std::min<size_t>(block_rows, open_kinds.size() - r);
      uint32_t desc = (uint32_t)n;
      for (size_t j = 0; j < n; ++j) desc |= open_kinds[r + j] << (zkgpu::kLdsBlockKindShift + j);
      if (barrier && r + n >= open_kinds.size()) desc |= 1u << 4;
      {   // a full block of `a` and-rows followed by xor-rows?
        size_t a = 0;
        while (a < n && open_kinds[r + a] == 0) ++a;
        bool sorted = true;
        for (size_t j = a; j < n; ++j) sorted = sorted && open_kinds[r + j] == 1;
        if (sorted && n == block_rows) desc |= (uint32_t)(a + 1) << zkgpu::kLdsBlockAndShift;   // full blocks only
      }
      const uint32_t id = (uint32_t)(blocks.size() / 2);
      blocks.push_back(desc);
      const uint64_t byte_offset = ((uint64_t)open_first + (uint64_t)r * 1024u) * 12u;
      if (byte_offset + (uint64_t)block_rows * zkgpu::kLdsRowOps * 6 >= (1ull << 32))   // (lds_program_fits bounds it; never wrap silently)
        throw std::runtime_error("Engine: the row stream of the LDS-resident GF(2) program exceeds 32-bit byte offsets");
      blocks.push_back((uint32_t)byte_offset);
      // consecutive blocks form one run (one chunk)
      if (ln.size() >= 4 && (ln[ln.size() - 2] & zkgpu::kLdsChunkBlocks) && ln[ln.size() - 4] + ln[ln.size() - 1] == id)
        ++ln[ln.size() - 1];
      else
        ln.insert(ln.end(), {id, 0u, zkgpu::kLdsChunkBlocks, 1u});
    }
    open_kinds.clear();
  };
  for (const Launch& L : s.launches) {
    if (L.sequential) {
      const uint32_t first = (uint32_t)lo.size();
      for (uint32_t k = 0; k < L.count; ++k) lo.push_back(encode(s.ops[L.first + k]));
      ln.insert(ln.end(), {first, L.count, zkgpu::kLdsChunkSequential | zkgpu::kLdsChunkBarrier, 0u});
      continue;
    }
    // ops of a level arrive sorted by kind (schedule.cpp): whole rows per kind
    uint32_t k = 0;
    while (k < L.count) {
      const uint32_t kind = s.ops[L.first + k].kind;
      uint32_t e = k;
      while (e < L.count && s.ops[L.first + e].kind == kind) ++e;
      const bool last_kind = e >= L.count;
      const int rk = row_kind(kind);
      if (rk >= 0) {
        if (open_kinds.empty()) {
          if (lo6.size() / 6 > 0xFFFFFFFFull) throw std::runtime_error("Engine: the row stream of the LDS-resident GF(2) program is too long");
          open_first = (uint32_t)(lo6.size() / 6);   // in 12-byte thread records
        }
        size_t n = 0;
        auto put6 = [&](unsigned short dst, unsigned short a, unsigned short b) {
          lo6.push_back(dst);
          lo6.push_back(a);
          lo6.push_back(b);
          ++n;
        };
        for (uint32_t q = k; q < e; ++q) {
          const LdsOp o = encode(s.ops[L.first + q]);
          put6(o.dst, o.a, kind == TK_NOT ? ones : kind == TK_COPY ? zero : o.b);
        }
        // padding ops write the scratch slots -- 16 pairs, one per pair of banks: the op pair of thread t (positions 2t,
        // 2t + 1) goes to pair (t % 16), so the 16 lanes one ds_write_b64 cycle serves hit 16 different bank pairs
        const size_t run_start = lo6.size() - 3 * n;
        while (n % zkgpu::kLdsRowOps) {
          const unsigned short pair_slot = (unsigned short)(scratch + 2 * ((n / 2) % (zkgpu::kLdsScratchSlots / 2)) + (n & 1));
          // (an odd run: the partner of its last op is the unused half of that op's own pair)
          put6((n & 1) ? (unsigned short)(lo6[lo6.size() - 3] + 1) : pair_slot, zero, zero);
        }
        // the even op of every thread names the slot PAIR: its result and its neighbour's are the two halves
        for (size_t q = 0; q < n; q += 2) {
          unsigned short& d0 = lo6[run_start + 3 * q];
          const unsigned short d1 = lo6[run_start + 3 * (q + 1)];
          if ((d0 & 1) || d1 != d0 + 1)
            throw std::runtime_error("Engine: the results of a row are not allocated as aligned slot pairs (scheduler / LDS program mismatch)");
          d0 = (unsigned short)(d0 >> 1);
        }
        for (size_t r = 0; r < n / zkgpu::kLdsRowOps; ++r) open_kinds.push_back((uint32_t)rk);
        if (last_kind) close_blocks(true);
      } else {
        close_blocks(false);   // rows so far run before this kind's chunk (same level: no barrier needed)
        const LdsOp pad{scratch, 0, 0, (unsigned short)TK_NOP};
        if (lo.size() & 1) lo.push_back(pad);  // 16-B aligned rows
        const uint32_t first = (uint32_t)lo.size();
        for (uint32_t q = k; q < e; ++q) lo.push_back(encode(s.ops[L.first + q]));
        while ((lo.size() - first) % zkgpu::kLdsRowOps) lo.push_back(pad);
        const uint32_t rows = ((uint32_t)lo.size() - first) / zkgpu::kLdsRowOps;
        ln.insert(ln.end(), {first, rows, kind | (last_kind ? zkgpu::kLdsChunkBarrier : 0u), 0u});
      }
      k = e;
    }
  }
  // every block fetches block_rows rows whatever it holds: keep the rows past the last one inside the allocation
  lo6.insert(lo6.end(), (size_t)block_rows * zkgpu::kLdsRowOps * 3, scratch);
  return P;
}

}  // namespace zki
