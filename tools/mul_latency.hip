// Developer microbenchmark: the LATENCY of a chain of dependent Montgomery products in ONE wave (what a strand of a
// structured relation is bound by: one workgroup per lane block, at most one wave per SIMD), against their throughput with
// many waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mul_latency.hip -o /tmp/mul_latency && /tmp/mul_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../zkinterface-ir_amd/csrc/device/fp_mont.hpp"

using namespace zkgpu;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void chain_kernel(u32* out, long long* cycles, int iters, const FieldParams fp) {
  Fp<8> x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = 0x1234567u * (threadIdx.x + 1) + i; y.w[i] = 0x7654321u * (threadIdx.x + 3) + 5 * i; }
  x.w[7] &= 0x0FFFFFFF; y.w[7] &= 0x0FFFFFFF;
  const long long t0 = clock64();
  for (int k = 0; k < iters; ++k) {
    if (VARIANT == 0) x = fp_mul<8>(x, y, fp);
    else x = fp_mul_wide<8>(x, y, fp);
  }
  const long long t1 = clock64();
  for (int i = 0; i < 8; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + i] = x.w[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  // BN254 r
  const u32 p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  FieldParams fp;
  memset(&fp, 0, sizeof fp);
  memcpy(fp.p, p, sizeof p);
  u32 inv = 1;   // -p^{-1} mod 2^32 by Newton
  for (int i = 0; i < 5; ++i) inv *= 2 - p[0] * inv;
  fp.n0inv = 0u - inv;
  fp.nwords = 8;
  u32* d_out; long long* d_cyc;
  CK(hipMalloc(&d_out, 1024 * 256 * 8 * 4));
  CK(hipMalloc(&d_cyc, 8));
  const int iters = 2000;
  u32 ref[8];
  for (int variant = 0; variant < 2; ++variant) {
    for (int waves = 1; waves <= 4; waves *= 4) {
      for (int blocks : {1, 1024}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (variant == 0) chain_kernel<0><<<blocks, 64 * waves>>>(d_out, d_cyc, iters, fp);
        else chain_kernel<1><<<blocks, 64 * waves>>>(d_out, d_cyc, iters, fp);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        u32 got[8]; CK(hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost));
        if (variant == 0 && waves == 1 && blocks == 1) memcpy(ref, got, 32);
        printf("variant %d  blocks %4d x %d wave(s): %.3f ms, %.0f clock64 ticks per product (wall: %.1f ns per product per wave)  %s\n", variant, blocks, waves, ms,
               (double)cyc / iters, ms * 1e6 / iters, memcmp(ref, got, 32) ? "RESULT DIFFERS" : "same result");
      }
    }
  }
  // a wave with fewer active lanes: does a pass over an empty half cost nothing?
  for (int lanes : {64, 32, 16, 8}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    chain_kernel<0><<<1, lanes>>>(d_out, d_cyc, iters, fp);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    chain_kernel<0><<<1, lanes>>>(d_out, d_cyc, iters, fp);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("one wave, %2d active lanes: %.3f ms, %.0f ticks per product, %.1f ns per product\n", lanes, ms, (double)cyc / iters, ms * 1e6 / iters);
  }
  return 0;
}
