// Developer microbenchmark for the replay kernel (not part of the product
// path, not used by bench.py): random layered Add/Mul tape over BN254 r,
// checked against a host re-computation, timed with HIP events.
//
//   hipcc -O3 --offload-arch=gfx950 tools/kbench.hip -o gpurun_out/kbench
//   ./kbench [W] [D] [lanes] [mul_percent]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../zkinterface-ir_amd/csrc/device/replay_kernels.hpp"

using namespace zkgpu;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

static uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- tiny host bigint on 8x32 words ----
struct H8 { u32 w[8]; };
static bool h_geq(const H8& a, const H8& b) {
  for (int i = 7; i >= 0; --i) { if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; }
  return true;
}
static void h_sub(H8& a, const H8& b) {
  u64 br = 0;
  for (int i = 0; i < 8; ++i) { u64 d = (u64)a.w[i] - b.w[i] - br; a.w[i] = (u32)d; br = (d >> 63) & 1; }
}
static H8 h_addmod(const H8& a, const H8& b, const H8& p) {
  H8 r; u64 c = 0;
  for (int i = 0; i < 8; ++i) { c += (u64)a.w[i] + b.w[i]; r.w[i] = (u32)c; c >>= 32; }
  if (c || h_geq(r, p)) h_sub(r, p);
  return r;
}
// slow, independent: a*b mod p by double-and-add
static H8 h_mulmod_slow(const H8& a, const H8& b, const H8& p) {
  H8 r; memset(&r, 0, sizeof r);
  for (int i = 255; i >= 0; --i) {
    r = h_addmod(r, r, p);
    if ((b.w[i / 32] >> (i % 32)) & 1) r = h_addmod(r, a, p);
  }
  return r;
}
static H8 h_montmul(const H8& a, const H8& b, const FieldParams& fp) {
  u32 t[10] = {0};
  for (int i = 0; i < 8; ++i) {
    u64 c = 0;
    for (int j = 0; j < 8; ++j) { c += (u64)a.w[j] * b.w[i] + t[j]; t[j] = (u32)c; c >>= 32; }
    c += t[8]; t[8] = (u32)c; t[9] = (u32)(c >> 32);
    u32 m = t[0] * fp.n0inv;
    c = (u64)m * fp.p[0] + t[0]; c >>= 32;
    for (int j = 1; j < 8; ++j) { c += (u64)m * fp.p[j] + t[j]; t[j - 1] = (u32)c; c >>= 32; }
    c += t[8]; t[7] = (u32)c; t[8] = t[9] + (u32)(c >> 32);
  }
  H8 r, P; memcpy(r.w, t, 32); memcpy(P.w, fp.p, 32);
  if (t[8] || h_geq(r, P)) h_sub(r, P);
  return r;
}

int main(int argc, char** argv) {
  const u32 W = argc > 1 ? atoi(argv[1]) : 4096;
  const u32 D = argc > 2 ? atoi(argv[2]) : 32;
  const u32 lanes = argc > 3 ? atoi(argv[3]) : 1024;
  const u32 mulpct = argc > 4 ? atoi(argv[4]) : 50;
  const u32 LB = (lanes + 63) / 64;

  // BN254 scalar field
  static const u32 P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                           0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  FieldParams fp; memset(&fp, 0, sizeof fp);
  memcpy(fp.p, P, 32); fp.nwords = 8;
  H8 Pm; memcpy(Pm.w, P, 32);
  H8 x; memset(&x, 0, sizeof x); x.w[0] = 1;
  for (int i = 0; i < 256; ++i) x = h_addmod(x, x, Pm);
  memcpy(fp.one, x.w, 32);
  for (int i = 0; i < 256; ++i) x = h_addmod(x, x, Pm);
  memcpy(fp.r2, x.w, 32);
  u32 inv = 1; for (int i = 0; i < 5; ++i) inv *= 2 - P[0] * inv;
  fp.n0inv = 0u - inv;

  // self-check of the host Montgomery against the slow independent modmul
  {
    uint64_t s = 42; H8 a, b;
    for (int i = 0; i < 8; ++i) { a.w[i] = (u32)splitmix64(s); b.w[i] = (u32)splitmix64(s); }
    a.w[7] &= 0x0fffffff; b.w[7] &= 0x0fffffff;
    H8 r2; memcpy(r2.w, fp.r2, 32);
    H8 one; memset(&one, 0, sizeof one); one.w[0] = 1;
    H8 am = h_montmul(a, r2, fp), bm = h_montmul(b, r2, fp);
    H8 cm = h_montmul(am, bm, fp), c = h_montmul(cm, one, fp);
    H8 ref = h_mulmod_slow(a, b, Pm);
    if (memcmp(&c, &ref, 32)) { fprintf(stderr, "host montmul self-check FAILED\n"); return 2; }
  }

  // tape: layer 0 = W witness loads; layers 1..D = W gates each; ping-pong slots
  const u32 n_slots = 2 * W;
  std::vector<std::vector<TapeOp>> layers(D + 1);
  uint64_t s = 0x5EED0001ull;
  layers[0].resize(W);
  for (u32 j = 0; j < W; ++j) layers[0][j] = TapeOp{j, j, 0, OP_WITNESS};
  for (u32 l = 1; l <= D; ++l) {
    layers[l].resize(W);
    const u32 src = ((l - 1) & 1) * W, dst = (l & 1) * W;
    for (u32 j = 0; j < W; ++j) {
      uint64_t r = splitmix64(s);
      TapeOp op;
      op.kind = ((r >> 40) % 100 < mulpct) ? OP_MUL : OP_ADD;
      op.a = src + (u32)(r % W);
      op.b = src + (u32)((r >> 20) % W);
      op.dst = dst + j;
      layers[l][j] = op;
    }
    // sort by kind so that a wave's consecutive ops are uniform (scheduler does the same)
    std::stable_sort(layers[l].begin(), layers[l].end(),
                     [](const TapeOp& x, const TapeOp& y) { return x.kind < y.kind; });
  }
  std::vector<TapeOp> flat;
  std::vector<u32> off(D + 2, 0);
  for (u32 l = 0; l <= D; ++l) { off[l] = flat.size(); flat.insert(flat.end(), layers[l].begin(), layers[l].end()); }
  off[D + 1] = flat.size();

  // witnesses: canonical random values [lane][W][32B]
  std::vector<u32> wit((size_t)lanes * W * 8);
  uint64_t sw = 7;
  for (size_t i = 0; i < (size_t)lanes * W; ++i) {
    for (int k = 0; k < 8; ++k) wit[i * 8 + k] = (u32)splitmix64(sw);
    wit[i * 8 + 7] &= 0x1fffffff;  // < 2^253 < p
  }

  TapeOp* d_ops; uint4* d_table; u32 *d_wit, *d_ff, *d_flags; unsigned long long* d_counts;
  const size_t table_bytes = (size_t)LB * n_slots * Layout<8>::kRecord * sizeof(uint4);
  CK(hipMalloc(&d_ops, flat.size() * sizeof(TapeOp)));
  CK(hipMalloc(&d_table, table_bytes));
  CK(hipMalloc(&d_wit, wit.size() * 4));
  CK(hipMalloc(&d_ff, LB * 64 * 4));
  CK(hipMalloc(&d_flags, LB * 64 * 4));
  CK(hipMalloc(&d_counts, 16));
  CK(hipMemcpy(d_ops, flat.data(), flat.size() * sizeof(TapeOp), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wit, wit.data(), wit.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_ff, 0xff, LB * 64 * 4));
  CK(hipMemset(d_flags, 0, LB * 64 * 4));
  CK(hipMemset(d_table, 0, table_bytes));
  printf("W=%u D=%u lanes=%u mul%%=%u table=%.1f MB\n", W, D, lanes, mulpct, table_bytes / 1e6);

  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto run = [&](u32 opw, bool pipe, bool time_it) -> float {
    ReplayArgs A; memset(&A, 0, sizeof A);
    A.table = d_table; A.n_slots = n_slots; A.batch = lanes; A.wit = (const uint8_t*)d_wit; A.n_wit = W;
    A.first_fail = d_ff; A.lane_flags = d_flags; A.ops_per_wave = opw;
    // layer 0 (inputs), untimed
    A.ops = d_ops + off[0]; A.n_ops = W;
    dim3 g((W + 4 * opw - 1) / (4 * opw), LB);
    replay_kernel<8, true><<<g, 256, 0, st>>>(A, fp);
    if (time_it) CK(hipEventRecord(e0, st));
    for (u32 l = 1; l <= D; ++l) {
      A.ops = d_ops + off[l]; A.n_ops = off[l + 1] - off[l];
      if (pipe) replay_kernel<8, true><<<g, 256, 0, st>>>(A, fp);
      else replay_kernel<8, false><<<g, 256, 0, st>>>(A, fp);
    }
    if (time_it) CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; if (time_it) CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };

  // correctness: recompute 2 lanes on host
  run(4, true, false);
  {
    const u32 final_base = (D & 1) * W;
    std::vector<u32> slots(W); for (u32 j = 0; j < W; ++j) slots[j] = final_base + j;
    u32* d_slots; u32* d_out;
    CK(hipMalloc(&d_slots, W * 4)); CK(hipMalloc(&d_out, (size_t)lanes * W * 32));
    CK(hipMemcpy(d_slots, slots.data(), W * 4, hipMemcpyHostToDevice));
    dump_slots_kernel<8><<<dim3(W, LB), 64, 0, st>>>(d_table, n_slots, d_slots, W, lanes, d_out, fp);
    CK(hipStreamSynchronize(st));
    std::vector<u32> out((size_t)lanes * W * 8);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    H8 one; memset(&one, 0, sizeof one); one.w[0] = 1; H8 r2; memcpy(r2.w, fp.r2, 32);
    int bad = 0;
    const u32 check_lanes[3] = {0, lanes / 2 + 1, lanes - 1};
    for (u32 cl : check_lanes) {
      std::vector<H8> tbl(n_slots);
      for (u32 l = 0; l <= D; ++l)
        for (const TapeOp& op : layers[l]) {
          if (op.kind == OP_WITNESS) { H8 v; memcpy(v.w, &wit[((size_t)cl * W + op.a) * 8], 32); tbl[op.dst] = h_montmul(v, r2, fp); }
          else if (op.kind == OP_ADD) tbl[op.dst] = h_addmod(tbl[op.a], tbl[op.b], Pm);
          else tbl[op.dst] = h_montmul(tbl[op.a], tbl[op.b], fp);
        }
      for (u32 j = 0; j < W; ++j) {
        H8 v = h_montmul(tbl[final_base + j], one, fp);
        if (memcmp(v.w, &out[((size_t)cl * W + j) * 8], 32)) { if (bad < 5) fprintf(stderr, "MISMATCH lane %u wire %u\n", cl, j); ++bad; }
      }
    }
    printf("check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    if (bad) return 3;
  }

  const double gate_lanes = (double)W * D * lanes;
  for (int pipe = 1; pipe >= 0; --pipe)
    for (u32 opw : {1u, 2u, 4u, 8u, 16u}) {
      run(opw, pipe, false);
      float best = 1e30f, sum = 0; const int R = 5;
      for (int r = 0; r < R; ++r) { float ms = run(opw, pipe, true); best = std::min(best, ms); sum += ms; }
      printf("opw=%2u pipe=%d : best %.3f ms avg %.3f ms | %.2f us/layer | %.2f Ggate-ops/s | %.2f TB/s algorithmic (96B)\n",
             opw, pipe, best, sum / R, best * 1e3 / D, gate_lanes / best / 1e6, gate_lanes * 96 / best / 1e9);
    }
  return 0;
}
