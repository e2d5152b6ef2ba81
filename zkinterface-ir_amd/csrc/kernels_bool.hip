// GF(2) kernels and the width-independent verdict reduction: launchers declared in device/args.hpp.
#include "device/bool_kernels.hpp"

namespace zkgpu {

void launch_verdict(dim3 grid, hipStream_t st, const u32* first_fail, const u32* lane_flags, u32 batch,
                    unsigned long long* counts) {
  verdict_kernel<<<grid, 256, 0, st>>>(first_fail, lane_flags, batch, counts);
}

void launch_pack_inputs(hipStream_t st, const uint8_t* inst, u32 n_inst, const uint8_t* strict_inst, u64* packed_inst,
                        const uint8_t* wit, u32 n_wit, const uint8_t* strict_wit, u64* packed_wit, u32 batch, u32 total_words,
                        u32* lane_flags) {
  PackArgs pa;
  pa.raw[0] = inst;   pa.strict[0] = strict_inst;   pa.packed[0] = packed_inst;   pa.n_vals[0] = n_inst;
  pa.raw[1] = wit;    pa.strict[1] = strict_wit;    pa.packed[1] = packed_wit;    pa.n_vals[1] = n_wit;
  const u32 longest = n_inst > n_wit ? n_inst : n_wit;
  if (!longest) return;
  pack_inputs_kernel<<<dim3((total_words + 3) / 4, (longest + 255) / 256, 2), 256, 0, st>>>(pa, batch, total_words, lane_flags);
}

void launch_bool_replay(dim3 grid, hipStream_t st, const BoolReplayArgs& a) {
  bool_replay_kernel<<<grid, 256, 0, st>>>(a);
}

#define ZKGPU_LDS_BLOCK_ROWS(X) X(4) X(6) X(8) X(9) X(10) X(12)

hipError_t bool_lds_set_max_shared(int bytes) {
#define X(BR)                                                                                                          \
  if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bool_lds_kernel<BR>),                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, bytes))                           \
    return e;
  ZKGPU_LDS_BLOCK_ROWS(X)
#undef X
  return hipSuccess;
}

bool bool_lds_has_block_rows(u32 rows) {
#define X(BR) if (rows == BR) return true;
  ZKGPU_LDS_BLOCK_ROWS(X)
#undef X
  return false;
}

void launch_bool_lds(u32 n_cols, size_t lds_bytes, hipStream_t st, const BoolLdsArgs& a) {
  switch (a.block_rows) {
#define X(BR) case BR: bool_lds_kernel<BR><<<n_cols, 1024, lds_bytes, st>>>(a); break;
    ZKGPU_LDS_BLOCK_ROWS(X)
#undef X
    default: break;   // Engine::load_program only picks sizes bool_lds_has_block_rows() accepts
  }
}

void launch_bool_dump(dim3 grid, hipStream_t st, const u64* table, u32 n_slots, const u32* slots, u32 n_dump, u32 batch,
                      uint8_t* out) {
  bool_dump_slots_kernel<<<grid, 64, 0, st>>>(table, n_slots, slots, n_dump, batch, out);
}

}  // namespace zkgpu
