// The any-modulus kernels (device/generic_kernels.hpp): one instantiation per capacity class, so that a 520-bit
// modulus does not pay the scratch memory of a 4096-bit one.
#include "device/generic_kernels.hpp"

namespace zkgpu {

// k_words: the words of the characteristic (GenericParams::k): up to eight of them run the instantiation with the word counts
// at compile time (generic_kernels.hpp SmallParams)
void launch_replay_generic(dim3 grid, hipStream_t st, const ReplayArgs& a, const GenericParams* gp, u32 nwords, u32 k_words) {
  switch (k_words <= 8 ? k_words : 0) {
    case 1: replay_generic_kernel<8, 1><<<grid, 256, 0, st>>>(a, gp); return;
    case 2: replay_generic_kernel<8, 2><<<grid, 256, 0, st>>>(a, gp); return;
    case 3: replay_generic_kernel<8, 3><<<grid, 256, 0, st>>>(a, gp); return;
    case 4: replay_generic_kernel<8, 4><<<grid, 256, 0, st>>>(a, gp); return;
    case 5: replay_generic_kernel<8, 5><<<grid, 256, 0, st>>>(a, gp); return;
    case 6: replay_generic_kernel<8, 6><<<grid, 256, 0, st>>>(a, gp); return;
    case 7: replay_generic_kernel<8, 7><<<grid, 256, 0, st>>>(a, gp); return;
    case 8: replay_generic_kernel<8, 8><<<grid, 256, 0, st>>>(a, gp); return;
    default: break;
  }
  if (nwords <= 16) replay_generic_kernel<16, 0><<<grid, 256, 0, st>>>(a, gp);
  else if (nwords <= 32) replay_generic_kernel<32, 0><<<grid, 256, 0, st>>>(a, gp);
  else if (nwords <= 64) replay_generic_kernel<64, 0><<<grid, 256, 0, st>>>(a, gp);
  else replay_generic_kernel<kGenericMaxWords, 0><<<grid, 256, 0, st>>>(a, gp);
}

void launch_dump_generic(dim3 grid, hipStream_t st, const uint4* table, u32 n_slots, const u32* slots, u32 n_dump, u32 batch,
                         u32* out, u32 nwords) {
  dump_generic_kernel<<<grid, 64, 0, st>>>(table, n_slots, slots, n_dump, batch, out, nwords);
}

int generic_selftest(const GenericParams* gp, int op, const u32* a, const u32* b, u32* out) {
  constexpr int CAP = kGenericMaxWords;
  if (gp->k == 0 || gp->k > (u32)CAP || gp->nwords > (u32)CAP || gp->nwords < gp->k) return 1;
  switch (op) {
    case 0: g_add<CAP>(a, b, out, gp); return 0;
    case 1: g_mul<CAP>(a, b, out, gp); return 0;
    case 2: g_reduce<CAP>(a, out, gp); return 0;
    case 3: g_and(a, b, out, gp); return 0;
    case 4: g_xor(a, b, out, gp); return 0;
    default: return 1;
  }
}

}  // namespace zkgpu
