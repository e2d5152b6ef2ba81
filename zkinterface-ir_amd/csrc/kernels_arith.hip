// GF(p) kernels for ONE field width: compiled once per width with -DZKGPU_W=<32-bit words> (2, 4, ..., 16), so
// the six widths build in parallel and the engine's host code instantiates no kernel.  Defines the launchers
// declared in device/args.hpp.
#include "device/r1cs_kernels.hpp"
#include "device/replay_kernels.hpp"

#ifndef ZKGPU_W
#error "compile with -DZKGPU_W=<words per field element>"
#endif
#define ZKGPU_CAT2(a, b) a##b
#define ZKGPU_CAT(a, b) ZKGPU_CAT2(a, b)
#define ZKGPU_FN(name) ZKGPU_CAT(name, ZKGPU_W)

namespace zkgpu {

// lds_pad: dynamic LDS the kernel never touches -- the engine's handle on how many workgroups (= waves per SIMD) a CU
// holds at once, since an unused allocation is the only occupancy limit that can be chosen per launch.
void ZKGPU_FN(launch_replay_fused_w)(int cls, dim3 grid, size_t lds_pad, hipStream_t st, const ReplayArgs2& a,
                                     const FieldParams& fp) {
  if (cls == kFusedHot) replay_fused_kernel<ZKGPU_W, kFusedHot><<<grid, 256, lds_pad, st>>>(a, fp);
  else if (cls == kFusedMisc) replay_fused_kernel<ZKGPU_W, kFusedMisc><<<grid, 256, 0, st>>>(a, fp);
  else replay_fused_kernel<ZKGPU_W, kFusedAll><<<grid, 256, 0, st>>>(a, fp);
}

void ZKGPU_FN(launch_replay_strand_w)(int cls, dim3 grid, hipStream_t st, const ReplayArgs2& a, const u32* level_ptr, u32 n_levels,
                                      size_t lds_bytes, const FieldParams& fp) {
  // (up to 128 KiB of dynamic LDS for the strand's own values: above the 64 KiB a kernel gets without asking)
  if (lds_bytes > 32 * 1024) {   // (per device; a strand is a handful of launches per replay)
    const void* k = cls == kFusedAll ? reinterpret_cast<const void*>(&replay_strand_kernel<ZKGPU_W, kFusedAll>)
                                     : reinterpret_cast<const void*>(&replay_strand_kernel<ZKGPU_W, kFusedMisc>);
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  }
  if (cls == kFusedAll) replay_strand_kernel<ZKGPU_W, kFusedAll><<<grid, 256, lds_bytes, st>>>(a, level_ptr, n_levels, fp);
  else replay_strand_kernel<ZKGPU_W, kFusedMisc><<<grid, 256, lds_bytes, st>>>(a, level_ptr, n_levels, fp);
}

// The scheduler always emits ops_per_wave = 1 for a level (measured fastest) and one wave per lane block
// for sequential segments; neither wants the in-wave operand prefetch variant (PIPE, kept for tools/kbench).
void ZKGPU_FN(launch_replay_w)(bool bitops, dim3 grid, hipStream_t st, const ReplayArgs& a, const FieldParams& fp) {
  if (bitops) replay_kernel<ZKGPU_W, false, true><<<grid, 256, 0, st>>>(a, fp);
  else replay_kernel<ZKGPU_W, false, false><<<grid, 256, 0, st>>>(a, fp);
}

// classes: some row of the launch has a combination of class unit / small (args.hpp) -- the instantiation without them
// is the one the rows of random coefficients run on, with the registers of that path alone
void ZKGPU_FN(launch_r1cs_w)(bool assign, bool classes, dim3 grid, hipStream_t st, const R1csArgs& a, const FieldParams& fp) {
  if (classes) {
    if (assign) r1cs_row_kernel<ZKGPU_W, true, true><<<grid, 256, 0, st>>>(a, fp);
    else r1cs_row_kernel<ZKGPU_W, false, true><<<grid, 256, 0, st>>>(a, fp);
  } else {
    if (assign) r1cs_row_kernel<ZKGPU_W, true, false><<<grid, 256, 0, st>>>(a, fp);
    else r1cs_row_kernel<ZKGPU_W, false, false><<<grid, 256, 0, st>>>(a, fp);
  }
}

void ZKGPU_FN(launch_dump_w)(dim3 grid, hipStream_t st, const uint4* table, u32 n_slots, const u32* slots, u32 n_dump,
                             u32 batch, u32* out, const FieldParams& fp) {
  dump_slots_kernel<ZKGPU_W><<<grid, 64, 0, st>>>(table, n_slots, slots, n_dump, batch, out, fp);
}

void ZKGPU_FN(launch_r1cs_corr_w)(dim3 grid, hipStream_t st, const R1csCorrArgs& a, const FieldParams& fp) {
  r1cs_correction_kernel<ZKGPU_W><<<grid, 256, 0, st>>>(a, fp);
}

}  // namespace zkgpu
