// Host side of the LDS-resident GF(2) kernel: the schedule of a Boolean relation -> the program the kernel walks
// (device/lds_layout.hpp).  Plain C++: the CPU test tier builds and simulates these programs without a GPU
// (tests/test_lds_program.py).
#pragma once
#include <stdint.h>

#include <vector>

#include "device/lds_layout.hpp"
#include "schedule.hpp"

namespace zki {

struct LdsProgram {
  std::vector<zkgpu::LdsOp> ops;        // generic chunks
  std::vector<uint16_t> rows;           // ops6: {dst, a, b} per op, rows of kLdsRowOps ops; kLdsMaxBlockRows rows of slack
  std::vector<uint32_t> blocks;         // two u32 per block
  std::vector<uint32_t> chunks;         // four u32 per chunk
  uint32_t block_rows = 0;              // rows every block fetches = the kernel instantiation
  uint32_t n_slots = 0;                 // words of the LDS table: the schedule's slots + kLdsExtraSlots
};

// true if the relation's wire table and program fit the kernel (160 KiB of LDS, 16-bit slot numbers, 32-bit offsets)
bool lds_program_fits(const Schedule& s);

// block_sizes: the kernel instantiations that exist (ascending); forced_block_rows: 0 = pick the size that fetches least
LdsProgram build_lds_program(const Schedule& s, const std::vector<uint32_t>& block_sizes, uint32_t forced_block_rows = 0);

}  // namespace zki
