// GF(2) kernels and the width-independent verdict reduction: launchers declared in device/args.hpp.
#include "device/bool_kernels.hpp"

namespace zkgpu {

void launch_verdict(dim3 grid, hipStream_t st, const u32* first_fail, const u32* lane_flags, u32 batch,
                    unsigned long long* counts) {
  verdict_kernel<<<grid, 256, 0, st>>>(first_fail, lane_flags, batch, counts);
}

void launch_pack_inputs(dim3 grid, hipStream_t st, const uint8_t* raw, u32 n_vals, u32 batch, u32 total_words,
                        u64* packed, u32* lane_flags, const uint8_t* strict) {
  pack_inputs_kernel<<<grid, 256, 0, st>>>(raw, n_vals, batch, total_words, packed, lane_flags, strict);
}

void launch_bool_replay(dim3 grid, hipStream_t st, const BoolReplayArgs& a) {
  bool_replay_kernel<<<grid, 256, 0, st>>>(a);
}

hipError_t bool_lds_set_max_shared(int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&bool_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

void launch_bool_lds(u32 n_cols, size_t lds_bytes, hipStream_t st, const BoolLdsArgs& a) {
  bool_lds_kernel<<<n_cols, 1024, lds_bytes, st>>>(a);
}

void launch_bool_dump(dim3 grid, hipStream_t st, const u64* table, u32 n_slots, const u32* slots, u32 n_dump, u32 batch,
                      uint8_t* out) {
  bool_dump_slots_kernel<<<grid, 64, 0, st>>>(table, n_slots, slots, n_dump, batch, out);
}

}  // namespace zkgpu
