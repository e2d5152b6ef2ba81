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
         // block headers carry 32-bit byte offsets into the row stream: 6 bytes per op + the padding of every level's row
         // sequence to whole rows (less than one row of 12288 bytes per launch) + the slack rows behind the last block.
         // build_lds_program checks the real offsets again as it writes them.
         (uint64_t)s.ops.size() * 6 + s.launches.size() * (1ull * zkgpu::kLdsRowOps * 6) + (1u << 22) < (1ull << 32);
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
    std::vector<uint32_t> level_rows;   // rows of the row sequence of every level
    for (const Launch& L : s.launches) {
      if (L.sequential) continue;
      uint32_t n = 0;
      for (uint32_t k = 0; k < L.count; ++k) n += row_kind(s.ops[L.first + k].kind) >= 0;
      if (n) level_rows.push_back((n + zkgpu::kLdsRowOps - 1) / zkgpu::kLdsRowOps);
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

  // the blocks of one level: `rows` rows starting at thread record `first_record`, the first `n_and` ops of the
  // sequence are `and`, the rest `xor`
  auto emit_blocks = [&](uint64_t first_record, uint32_t rows, uint64_t n_and) {
    const uint32_t and_rows = (uint32_t)(n_and / zkgpu::kLdsRowOps), split = (uint32_t)(n_and % zkgpu::kLdsRowOps);
    for (uint32_t r = 0; r < rows; r += block_rows) {
      const uint32_t n = std::min(block_rows, rows - r);
      const uint32_t a = and_rows <= r ? 0 : std::min(n, and_rows - r);
      uint32_t desc = n | (a << zkgpu::kLdsBlockAndShift);
      if (r + n >= rows) desc |= 1u << 4;   // the level ends with this block: barrier
      if (and_rows >= r && and_rows < r + n)   // the split row lies in this block (behind its and-rows)
        desc |= split << zkgpu::kLdsBlockSplitShift;
      const uint32_t id = (uint32_t)(blocks.size() / 2);
      blocks.push_back(desc);
      const uint64_t byte_offset = (first_record + (uint64_t)r * 1024u) * 12u;
      if (byte_offset + (uint64_t)block_rows * zkgpu::kLdsRowOps * 6 >= (1ull << 32))   // (lds_program_fits bounds it; never wrap silently)
        throw std::runtime_error("Engine: the row stream of the LDS-resident GF(2) program exceeds 32-bit byte offsets");
      blocks.push_back((uint32_t)byte_offset);
      // consecutive blocks of one shape form one run (one chunk): full blocks with the same number of and-rows, whose
      // code the kernel picks once per run, or anything else
      const uint32_t flags = zkgpu::kLdsChunkBlocks | ((n == block_rows ? a : 15u) << zkgpu::kLdsChunkAndShift);
      if (ln.size() >= 4 && ln[ln.size() - 2] == flags && ln[ln.size() - 4] + ln[ln.size() - 1] == id)
        ++ln[ln.size() - 1];
      else
        ln.insert(ln.end(), {id, 0u, flags, 1u});
    }
  };
  for (const Launch& L : s.launches) {
    if (L.sequential) {
      // a run of narrow levels: packets of 64 entries walked by one wave (lds_layout.hpp kLdsChunkWave), every level padded
      // to whole packets; a run whose levels hold two entries at most stays a plain sequence walked by one thread
      const uint32_t* lp = L.level_ptr + L.strand_levels < s.strand_level_ptr.size() ? &s.strand_level_ptr[L.level_ptr] : nullptr;
      uint32_t widest = 0;
      for (uint32_t q = 0; lp && q < L.strand_levels; ++q) widest = std::max(widest, lp[q + 1] - lp[q]);
      if (!lp || widest <= 2) {
        const uint32_t first = (uint32_t)lo.size();
        for (uint32_t k = 0; k < L.count; ++k) lo.push_back(encode(s.ops[L.first + k]));
        ln.insert(ln.end(), {first, L.count, zkgpu::kLdsChunkSequential | zkgpu::kLdsChunkBarrier, 0u});
        continue;
      }
      // The 64 lanes of a packet run in lockstep: an entry kind of its own per lane would make the wave execute every kind's
      // code one after the other.  As in the rows, `not a` is stored as a xor ONES, a copy as a xor ZERO and the padding as
      // ZERO xor ZERO into a scratch slot: a packet is and / xor entries (one select) plus, rarely, inputs, constants and
      // asserts, which the kernel runs in a second pass over the lanes that hold them.
      const uint32_t first = (uint32_t)lo.size();
      for (uint32_t q = 0; q < L.strand_levels; ++q) {
        for (uint32_t k = lp[q]; k < lp[q + 1]; ++k) {
          LdsOp o = encode(s.ops[L.first + k]);
          if (o.kind == TK_NOT) { o.kind = TK_XOR; o.b = ones; }
          else if (o.kind == TK_COPY) { o.kind = TK_XOR; o.b = zero; }
          lo.push_back(o);
        }
        while ((lo.size() - first) % zkgpu::kLdsPacketOps) {
          const unsigned short lane = (unsigned short)((lo.size() - first) % zkgpu::kLdsPacketOps);
          lo.push_back(LdsOp{(unsigned short)(scratch + lane % zkgpu::kLdsScratchSlots), zero, zero, (unsigned short)TK_XOR});
        }
      }
      ln.insert(ln.end(), {first, ((uint32_t)lo.size() - first) / zkgpu::kLdsPacketOps,
                           zkgpu::kLdsChunkSequential | zkgpu::kLdsChunkWave | zkgpu::kLdsChunkBarrier, 0u});
      continue;
    }
    // The ops of a level arrive sorted by kind, the row kinds last: and, xor, not, copy (schedule.cpp order_by_level).
    // First the other kinds (inputs, constants, asserts), one generic chunk each -- they and the rows are independent,
    // the barrier goes behind whatever comes last.
    uint32_t k = 0;
    uint32_t row_ops = 0;
    for (uint32_t q = 0; q < L.count; ++q) row_ops += row_kind(s.ops[L.first + q].kind) >= 0;
    while (k < L.count && row_kind(s.ops[L.first + k].kind) < 0) {
      const uint32_t kind = s.ops[L.first + k].kind;
      uint32_t e = k;
      while (e < L.count && s.ops[L.first + e].kind == kind) ++e;
      const bool last = e >= L.count;
      const LdsOp pad{scratch, 0, 0, (unsigned short)TK_NOP};
      if (lo.size() & 1) lo.push_back(pad);  // 16-B aligned rows
      const uint32_t first = (uint32_t)lo.size();
      for (uint32_t q = k; q < e; ++q) lo.push_back(encode(s.ops[L.first + q]));
      while ((lo.size() - first) % zkgpu::kLdsRowOps) lo.push_back(pad);
      const uint32_t rows = ((uint32_t)lo.size() - first) / zkgpu::kLdsRowOps;
      ln.insert(ln.end(), {first, rows, kind | (last ? zkgpu::kLdsChunkBarrier : 0u), 0u});
      k = e;
    }
    if (!row_ops) continue;
    if (k + row_ops != L.count) throw std::runtime_error("Engine: the row kinds of a level are not its last ops (scheduler / LDS program mismatch)");
    if (lo6.size() / 6 > 0xFFFFFFFFull) throw std::runtime_error("Engine: the row stream of the LDS-resident GF(2) program is too long");
    const uint64_t first_record = lo6.size() / 6;   // in 12-byte thread records
    const size_t seq_start = lo6.size();
    size_t n = 0;
    uint64_t n_and = 0;
    auto put6 = [&](unsigned short dst, unsigned short a, unsigned short b) {
      lo6.push_back(dst);
      lo6.push_back(a);
      lo6.push_back(b);
      ++n;
    };
    for (uint32_t q = k; q < L.count; ++q) {
      const uint32_t kind = s.ops[L.first + q].kind;
      if (kind == TK_AND) {
        if (n_and != n) throw std::runtime_error("Engine: the and ops of a level do not come first in its row sequence");
        ++n_and;
      }
      const LdsOp o = encode(s.ops[L.first + q]);
      put6(o.dst, o.a, kind == TK_NOT ? ones : kind == TK_COPY ? zero : o.b);
    }
    // padding ops (xor ZERO ZERO) write the scratch slots -- 16 pairs, one per pair of banks: the op pair of thread t
    // (positions 2t, 2t + 1) goes to pair (t % 16), so the 16 lanes one ds_write_b64 cycle serves hit 16 different bank pairs
    while (n % zkgpu::kLdsRowOps) {
      const unsigned short pair_slot = (unsigned short)(scratch + 2 * ((n / 2) % (zkgpu::kLdsScratchSlots / 2)) + (n & 1));
      // (an odd sequence: the partner of its last op is the unused half of that op's own pair)
      put6((n & 1) ? (unsigned short)(lo6[lo6.size() - 3] + 1) : pair_slot, zero, zero);
    }
    // the even op of every thread names the slot PAIR: its result and its neighbour's are the two halves
    for (size_t q = 0; q < n; q += 2) {
      unsigned short& d0 = lo6[seq_start + 3 * q];
      const unsigned short d1 = lo6[seq_start + 3 * (q + 1)];
      if ((d0 & 1) || d1 != d0 + 1)
        throw std::runtime_error("Engine: the results of a row are not allocated as aligned slot pairs (scheduler / LDS program mismatch)");
      d0 = (unsigned short)(d0 >> 1);
    }
    emit_blocks(first_record, (uint32_t)(n / zkgpu::kLdsRowOps), n_and);
  }
  // every block fetches block_rows rows whatever it holds: keep the rows past the last one inside the allocation
  lo6.insert(lo6.end(), (size_t)block_rows * zkgpu::kLdsRowOps * 3, scratch);
  return P;
}

}  // namespace zki
