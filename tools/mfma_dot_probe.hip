// Exact-integer check of the matrix-core linear combination (tools/dev/r1cs_mfma.hpp) against fp_dot:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zkinterface-ir_amd/csrc/device -I tools/dev -o /tmp/mfma_dot_probe tools/mfma_dot_probe.hip && /tmp/mfma_dot_probe
// Random values below p for 64 lanes and 1..3 random coefficients below p (BN254), many rounds; every lane's eight
// result words must agree.  Also prints cycles per call of the two forms for one wave alone on its SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "args.hpp"
#include "r1cs_mfma.hpp"

using namespace zkgpu;

__global__ __launch_bounds__(64) void probe(const u32* values, const u32* coefs, const u32* strings, u32 n_terms, u32 reps, FieldParams fp,
                                            u32* out_ref, u32* out_mfma, unsigned long long* cycles) {
  const u32 lane = threadIdx.x;
  Fp<8> v[3];
  FpS<8> c[3];
  u32 cs[3];
  for (int t = 0; t < 3; ++t) {
    for (int i = 0; i < 8; ++i) v[t].w[i] = values[(t * 64 + lane) * 8 + i];
    c[t] = fp_load_uniform<8>(coefs, t);
    cs[t] = strings[t * kCoefStringWords + min(lane, kCoefStringWords - 1)];
  }
  const MfmaLane lc = mfma_lane_constants(lane);
  Fp<8> r, s;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (u32 k = 0; k < reps; ++k) {
    if (n_terms == 3) r = fp_dot<8, 3>(v, c, fp, fp.dot_rounds[2]);
    else if (n_terms == 2) { const Fp<8> vv[2] = {v[0], v[1]}; const FpS<8> cc[2] = {c[0], c[1]}; r = fp_dot<8, 2>(vv, cc, fp, fp.dot_rounds[1]); }
    else { const Fp<8> vv[1] = {v[0]}; const FpS<8> cc[1] = {c[0]}; r = fp_dot<8, 1>(vv, cc, fp, fp.dot_rounds[0]); }
    if (reps > 1) v[0].w[0] ^= r.w[0] & 1;   // (a dependency, so that the repetitions are not folded)
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  for (int t = 0; t < 3; ++t)
    for (int i = 0; i < 8; ++i) v[t].w[i] = values[(t * 64 + lane) * 8 + i];
  unsigned long long t2 = __builtin_readcyclecounter();
  for (u32 k = 0; k < reps; ++k) {
    Fp<8> w[3] = {v[0], v[1], v[2]};
    s = mfma_dot3(w, cs, n_terms, lc, fp, fp.dot_rounds[n_terms - 1]);
    if (reps > 1) v[0].w[0] ^= s.w[0] & 1;
  }
  unsigned long long t3 = __builtin_readcyclecounter();
  for (int i = 0; i < 8; ++i) {
    out_ref[lane * 8 + i] = r.w[i];
    out_mfma[lane * 8 + i] = s.w[i];
  }
  if (lane == 0) {
    cycles[0] = t1 - t0;
    cycles[1] = t3 - t2;
  }
}

// throughput: every SIMD of the chip holds `waves` waves (blocks of 256 threads), each repeating the call
template <bool MFMA>
__global__ __launch_bounds__(256) void throughput(const u32* values, const u32* coefs, const u32* strings, u32 reps, FieldParams fp, u32* sink) {
  const u32 lane = threadIdx.x & 63;
  Fp<8> v[3];
  FpS<8> c[3];
  u32 cs[3];
  for (int t = 0; t < 3; ++t) {
    for (int i = 0; i < 8; ++i) v[t].w[i] = values[(t * 64 + lane) * 8 + i];
    c[t] = fp_load_uniform<8>(coefs, t);
    cs[t] = strings[t * kCoefStringWords + min(lane, kCoefStringWords - 1)];
  }
  const MfmaLane lc = mfma_lane_constants(lane);
  u32 x = 0;
  for (u32 k = 0; k < reps; ++k) {
    Fp<8> r;
    if (MFMA) {
      Fp<8> w[3] = {v[0], v[1], v[2]};
      r = mfma_dot3(w, cs, 3, lc, fp, 1);
    } else {
      r = fp_dot<8, 3>(v, c, fp, 1);
    }
    v[0].w[0] ^= r.w[0] & 1;
    x ^= r.w[3];
  }
  if (x == 0x12345u) sink[threadIdx.x] = x;
}

static const uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static void below_p(uint32_t* w) {
  for (;;) {
    for (int i = 0; i < 8; ++i) w[i] = (uint32_t)rnd();
    w[7] &= 0x3FFFFFFFu;
    for (int i = 7; i >= 0; --i) {
      if (w[i] < P[i]) return;
      if (w[i] > P[i]) break;
    }
  }
}
// the coefficient string of a canonical value (host side of R1csArgs::coef_strings)
static void coef_string(const uint32_t* c, uint32_t* out24) {
  uint8_t d[32];
  uint64_t carry = 0;
  for (int i = 0; i < 8; ++i) {
    const uint64_t y = (uint64_t)c[i] + 0x80808080u + carry;
    const uint32_t w = (uint32_t)y ^ 0x80808080u;
    carry = y >> 32;
    for (int b = 0; b < 4; ++b) d[4 * i + b] = (uint8_t)(w >> (8 * b));
  }
  if (carry) { fprintf(stderr, "coefficient too large for the signed digits\n"); exit(2); }
  uint8_t s[96];
  memset(s, 0, sizeof s);
  for (int x = 32; x < 64; ++x) s[x] = d[63 - x];
  memcpy(out24, s, 96);
}

#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  FieldParams fp;
  memset(&fp, 0, sizeof fp);
  memcpy(fp.p, P, sizeof P);
  uint32_t inv = 1;
  for (int i = 0; i < 5; ++i) inv *= 2 - P[0] * inv;
  fp.n0inv = 0u - inv;
  fp.nwords = 8;
  fp.dot_rounds[0] = fp.dot_rounds[1] = fp.dot_rounds[2] = fp.dot_rounds[3] = 1;   // BN254: K p / R < 1 for K <= 4
  u32 *d_values, *d_coefs, *d_strings, *d_ref, *d_mfma;
  unsigned long long* d_cycles;
  OK(hipMalloc(&d_values, 3 * 64 * 8 * 4));
  OK(hipMalloc(&d_coefs, 3 * 8 * 4));
  OK(hipMalloc(&d_strings, 3 * 24 * 4));
  OK(hipMalloc(&d_ref, 64 * 8 * 4));
  OK(hipMalloc(&d_mfma, 64 * 8 * 4));
  OK(hipMalloc(&d_cycles, 16));
  std::vector<uint32_t> values(3 * 64 * 8), coefs(3 * 8), strings(3 * 24), ref(64 * 8), got(64 * 8);
  int bad = 0;
  for (int round = 0; round < 200 && !bad; ++round) {
    for (int k = 0; k < 3 * 64; ++k) below_p(&values[k * 8]);
    for (int t = 0; t < 3; ++t) below_p(&coefs[t * 8]);
    if (round == 0) {   // extremes: p - 1 and 0 and 1
      for (int i = 0; i < 8; ++i) { values[i] = P[i]; values[8 + i] = 0; values[16 + i] = i == 0; coefs[i] = P[i]; }
      values[0] -= 1;
      coefs[0] -= 1;
    }
    for (int t = 0; t < 3; ++t) coef_string(&coefs[t * 8], &strings[t * 24]);
    const u32 n_terms = 1 + round % 3;
    OK(hipMemcpy(d_values, values.data(), values.size() * 4, hipMemcpyHostToDevice));
    OK(hipMemcpy(d_coefs, coefs.data(), coefs.size() * 4, hipMemcpyHostToDevice));
    OK(hipMemcpy(d_strings, strings.data(), strings.size() * 4, hipMemcpyHostToDevice));
    probe<<<1, 64>>>(d_values, d_coefs, d_strings, n_terms, 1, fp, d_ref, d_mfma, d_cycles);
    OK(hipDeviceSynchronize());
    OK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
    OK(hipMemcpy(got.data(), d_mfma, got.size() * 4, hipMemcpyDeviceToHost));
    for (int lane = 0; lane < 64 && bad < 4; ++lane)
      if (memcmp(&ref[lane * 8], &got[lane * 8], 32)) {
        ++bad;
        printf("round %d (%u terms) lane %d differs:\n  fp_dot ", round, n_terms, lane);
        for (int i = 7; i >= 0; --i) printf("%08x", ref[lane * 8 + i]);
        printf("\n  mfma   ");
        for (int i = 7; i >= 0; --i) printf("%08x", got[lane * 8 + i]);
        printf("\n");
      }
  }
  printf(bad ? "MISMATCH\n" : "200 rounds x 64 lanes: the matrix-core combination equals fp_dot\n");
  unsigned long long cyc[2];
  probe<<<1, 64>>>(d_values, d_coefs, d_strings, 3, 200, fp, d_ref, d_mfma, d_cycles);
  OK(hipDeviceSynchronize());
  OK(hipMemcpy(cyc, d_cycles, 16, hipMemcpyDeviceToHost));
  printf("one wave alone, 3 terms: fp_dot %.0f cycles per call, matrix-core form %.0f\n", cyc[0] / 200.0, cyc[1] / 200.0);
  // whole-chip throughput at 1, 2, 3, 4 and 6 waves per SIMD
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0));
  OK(hipEventCreate(&e1));
  for (int waves : {1, 2, 3, 4, 6}) {
    float ms[2];
    for (int form = 0; form < 2; ++form) {
      const dim3 grid(256 * waves);
      const u32 reps = 400;
      for (int pass = 0; pass < 2; ++pass) {
        OK(hipEventRecord(e0));
        if (form) throughput<true><<<grid, 256>>>(d_values, d_coefs, d_strings, reps, fp, d_ref);
        else throughput<false><<<grid, 256>>>(d_values, d_coefs, d_strings, reps, fp, d_ref);
        OK(hipEventRecord(e1));
        OK(hipEventSynchronize(e1));
        OK(hipEventElapsedTime(&ms[form], e0, e1));
      }
    }
    const double calls = 256.0 * waves * 4 * 400;
    printf("%d waves per SIMD: fp_dot %.3f ms (%.0f SIMD-cycles per call at 2.4 GHz), matrix-core form %.3f ms (%.0f)\n", waves, ms[0],
           ms[0] * 1e-3 * 2.4e9 * 1024 / calls, ms[1], ms[1] * 1e-3 * 2.4e9 * 1024 / calls);
  }
  return bad ? 1 : 0;
}
