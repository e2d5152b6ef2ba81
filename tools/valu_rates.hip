// Developer microbenchmark (not part of the product path): sustained issue rate of the integer and fp64
// instructions a Montgomery multiplication can be built from, all CUs busy, 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 4096;
constexpr int kUnroll = 16;  // independent chains per thread

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
  uint64_t acc[kUnroll];
  double facc[kUnroll];
  uint32_t lo[kUnroll];
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) { acc[k] = a + k; facc[k] = (double)(a & 1023) + k; lo[k] = a ^ k; }
  const double fa = (double)(a & 0xFFFFF) * 1.0000001, fb = (double)(b & 0xFFFFF);
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      if (OP == 0) {  // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
      } else if (OP == 1) {  // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[k]) : "v"(a));
      } else if (OP == 2) {  // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[k]) : "v"(a));
      } else if (OP == 3) {  // v_fma_f64
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(facc[k]) : "v"(fa), "v"(fb));
      } else if (OP == 4) {  // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo[k]) : "v"(a), "v"(b));
      } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pair (a 64-bit add)
        uint32_t l = (uint32_t)acc[k], h = (uint32_t)(acc[k] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(l), "+v"(h) : "v"(a), "v"(b) : "vcc");
        acc[k] = ((uint64_t)h << 32) | l;
      } else if (OP == 6) {  // v_fma_f32
        float f = __uint_as_float(lo[k]);
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(__uint_as_float(a)), "v"(__uint_as_float(b)));
        lo[k] = __float_as_uint(f);
      } else if (OP == 7) {  // v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(facc[k]) : "v"(fa));
      } else if (OP == 8) {  // v_mul_u32_u24
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo[k]) : "v"(a));
      } else if (OP == 9) {  // v_mul_hi_u32_u24
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[k]) : "v"(a));
      } else if (OP == 10) {  // v_add_f64
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(facc[k]) : "v"(fa));
      } else if (OP == 11) {  // v_lshlrev_b64 (64-bit shift, used by fp64 limb splitting)
        asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc[k]));
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) s += acc[k] + (uint64_t)facc[k] + lo[k];
  if (s == 0x1234567) out[threadIdx.x] = (uint32_t)s;
}

template <int OP>
int run(const char* name, uint32_t* d_out, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 256>>>(d_out, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 256>>>(d_out, 2);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double lane_ops = (double)blocks * 256 * kIters * kUnroll * (OP == 5 ? 2 : 1);
  printf("%-28s %8.3f ms  %8.2f T lane-ops/s\n", name, ms, lane_ops / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 8;  // 8 blocks x 4 waves = 32 waves per CU
  printf("%s: %d CUs, clock %d MHz, %d blocks of 256\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, blocks);
  uint32_t* d_out;
  CK(hipMalloc(&d_out, 4096));
  run<6>("v_fma_f32", d_out, blocks);
  run<0>("v_mad_u64_u32", d_out, blocks);
  run<1>("v_mul_lo_u32", d_out, blocks);
  run<2>("v_mul_hi_u32", d_out, blocks);
  run<4>("v_mad_u32_u24", d_out, blocks);
  run<8>("v_mul_u32_u24", d_out, blocks);
  run<9>("v_mul_hi_u32_u24", d_out, blocks);
  run<5>("v_add_co+v_addc_co", d_out, blocks);
  run<3>("v_fma_f64", d_out, blocks);
  run<7>("v_mul_f64", d_out, blocks);
  run<10>("v_add_f64", d_out, blocks);
  run<11>("v_lshlrev_b64", d_out, blocks);
  return 0;
}
