// TEST INFRASTRUCTURE ONLY (oracle/).  `cpu_opt` of BASELINE.md section 3: an *optimised* CPU evaluator
// -- flat value array, 4x64-bit Montgomery arithmetic, the already flattened tape -- reported next to
// the reference-style `port` so that the GPU speed-up is not flattered by the reference's BigUint +
// hash-map design.  It evaluates the product's recorded tape (kinds / operands as exported by
// zkgpu_tape_dump), so it is also an independent second checker of the device arithmetic:
// __int128 CIOS here, 32-bit product scanning on the GPU, long division in the literal oracle.
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace {
typedef unsigned __int128 u128;
struct F4 { uint64_t l[4]; };

struct Field {
  F4 p, r2, one;
  uint64_t n0inv;
  bool geq_p(const F4& a) const {
    for (int i = 3; i >= 0; --i) if (a.l[i] != p.l[i]) return a.l[i] > p.l[i];
    return true;
  }
  F4 add(const F4& a, const F4& b) const {
    F4 r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_p(r)) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 t = (u128)r.l[i] - p.l[i] - br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; } }
    return r;
  }
  F4 mul(const F4& a, const F4& b) const {  // CIOS, 64-bit limbs
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      u128 c = 0;
      for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * n0inv;
      c = (u128)m * p.l[0] + t[0]; c >>= 64;
      for (int j = 1; j < 4; ++j) { c += (u128)m * p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    F4 r; memcpy(r.l, t, 32);
    if (t[4] || geq_p(r)) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 x = (u128)r.l[i] - p.l[i] - br; r.l[i] = (uint64_t)x; br = (x >> 64) & 1; } }
    return r;
  }
  bool is_zero(const F4& a) const { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
  void init(const uint8_t* mod_le, uint32_t len) {
    memset(&p, 0, sizeof p);
    for (uint32_t i = 0; i < len && i < 32; ++i) p.l[i / 8] |= (uint64_t)mod_le[i] << (8 * (i % 8));
    F4 x; memset(&x, 0, sizeof x); x.l[0] = 1;
    for (int i = 0; i < 256; ++i) x = add(x, x);
    one = x;
    for (int i = 0; i < 256; ++i) x = add(x, x);
    r2 = x;
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - p.l[0] * inv;
    n0inv = 0 - inv;
  }
  F4 from_bytes(const uint8_t* b, uint32_t len) const {  // arbitrary length -> reduced -> Montgomery
    F4 r; memset(&r, 0, sizeof r);
    F4 o; memset(&o, 0, sizeof o); o.l[0] = 1;
    for (uint32_t bit = len * 8; bit-- > 0;) { r = add(r, r); if ((b[bit / 8] >> (bit % 8)) & 1) r = add(r, o); }
    return mul(r, r2);
  }
};
}  // namespace

extern "C" {

// kinds/a/b: the tape (1 add 2 mul 3 addc 4 mulc 5 copy 6 constant 7 instance 8 witness 9 assert_zero);
// consts: n_consts byte strings of const_width bytes; inputs [lane][n][width] little-endian, canonical.
// first_fail[lane] = sequence number of the first failing assert or 0xFFFFFFFF.
// out_values (may be NULL): canonical value of every op for lane `dump_lane` ([n_ops][32] bytes).
// Returns wall-clock seconds of the evaluation.
double zko_opt_eval_dump(const uint8_t* kinds, const uint32_t* a, const uint32_t* b, uint64_t n_ops, const uint8_t* consts,
                         uint32_t const_width, uint32_t n_consts, const uint8_t* mod_le, uint32_t mod_len,
                         const uint8_t* inst, uint32_t n_inst, const uint8_t* wit, uint32_t n_wit, uint32_t width,
                         uint32_t batch, uint32_t threads, uint32_t* first_fail, uint8_t* out_values, uint32_t dump_lane,
                         const uint64_t* dump_ops, uint32_t n_dump, uint8_t* out_dump);

double zko_opt_eval(const uint8_t* kinds, const uint32_t* a, const uint32_t* b, uint64_t n_ops, const uint8_t* consts,
                    uint32_t const_width, uint32_t n_consts, const uint8_t* mod_le, uint32_t mod_len,
                    const uint8_t* inst, uint32_t n_inst, const uint8_t* wit, uint32_t n_wit, uint32_t width,
                    uint32_t batch, uint32_t threads, uint32_t* first_fail, uint8_t* out_values, uint32_t dump_lane) {
  return zko_opt_eval_dump(kinds, a, b, n_ops, consts, const_width, n_consts, mod_le, mod_len, inst, n_inst, wit, n_wit, width,
                           batch, threads, first_fail, out_values, dump_lane, nullptr, 0, nullptr);
}

// The same, and for EVERY lane the canonical values of the n_dump listed tape ops: out_dump[lane][k][32] (the output
// wires of a full-size workload: what tests/golden/make_c2_digests.py hashes and the GPU tier compares against).
double zko_opt_eval_dump(const uint8_t* kinds, const uint32_t* a, const uint32_t* b, uint64_t n_ops, const uint8_t* consts,
                    uint32_t const_width, uint32_t n_consts, const uint8_t* mod_le, uint32_t mod_len,
                    const uint8_t* inst, uint32_t n_inst, const uint8_t* wit, uint32_t n_wit, uint32_t width,
                    uint32_t batch, uint32_t threads, uint32_t* first_fail, uint8_t* out_values, uint32_t dump_lane,
                    const uint64_t* dump_ops, uint32_t n_dump, uint8_t* out_dump) {
  Field f;
  f.init(mod_le, mod_len);
  std::vector<F4> cm(n_consts);
  for (uint32_t i = 0; i < n_consts; ++i) cm[i] = f.from_bytes(consts + (size_t)i * const_width, const_width);
  F4 lit1; memset(&lit1, 0, sizeof lit1); lit1.l[0] = 1;
  std::atomic<uint32_t> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    std::vector<F4> v(n_ops);
    for (;;) {
      const uint32_t lane = next.fetch_add(1);
      if (lane >= batch) break;
      uint32_t ff = 0xFFFFFFFFu;
      for (uint64_t i = 0; i < n_ops; ++i) {
        switch (kinds[i]) {
          case 1: v[i] = f.add(v[a[i]], v[b[i]]); break;
          case 2: v[i] = f.mul(v[a[i]], v[b[i]]); break;
          case 3: v[i] = f.add(v[a[i]], cm[b[i]]); break;
          case 4: v[i] = f.mul(v[a[i]], cm[b[i]]); break;
          case 5: v[i] = v[a[i]]; break;
          case 6: v[i] = cm[a[i]]; break;
          case 7: v[i] = f.from_bytes(inst + ((size_t)lane * n_inst + a[i]) * width, width); break;
          case 8: v[i] = f.from_bytes(wit + ((size_t)lane * n_wit + a[i]) * width, width); break;
          case 9: if (!f.is_zero(v[a[i]]) && b[i] < ff) ff = b[i]; break;
          default: break;
        }
      }
      first_fail[lane] = ff;
      for (uint32_t k = 0; k < n_dump && out_dump; ++k) {
        const F4 c = f.mul(v[dump_ops[k]], lit1);
        memcpy(out_dump + ((size_t)lane * n_dump + k) * 32, c.l, 32);
      }
      if (out_values && lane == dump_lane)
        for (uint64_t i = 0; i < n_ops; ++i) {
          F4 c = kinds[i] == 9 ? F4{{0, 0, 0, 0}} : f.mul(v[i], lit1);
          memcpy(out_values + i * 32, c.l, 32);
        }
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// CPU statement of the R1CS row check <a,w> * <b,w> = <c,w> (the zkinterface `Simulator` the reference's tests run on
// ToR1CSConverter output, rust/src/consumers/to_r1cs.rs:583-589; that crate is not under /root/reference, so this is the
// mathematical definition) -- the `cpu_baseline` of bench.py --workload c5 and a second checker for the row kernel.
// CSR as zkgpu_r1cs_load_csr takes it: row_ptr[3*rows+1] (A, B, C segments), term_var (variable id, or 2^64-1 = the
// constant one), term_coef (index into coefs, coef_width bytes each, little-endian).  base[lane][n_base][width] are the
// variables 0..n_base-1; variables n_base.. are assigned first, in row order, by every row r < n_assign whose C is the
// single term 1*z (z = <a,w>*<b,w>), untimed.  first_fail[lane] = first failing row or 0xFFFFFFFF.
// Returns the wall-clock seconds of the check alone.
double zko_r1cs_check(const uint32_t* row_ptr, const uint64_t* term_var, const uint32_t* term_coef, uint32_t n_rows,
                      const uint8_t* coefs, uint32_t coef_width, uint32_t n_coefs, const uint8_t* mod_le, uint32_t mod_len,
                      const uint8_t* base, uint32_t n_base, uint32_t n_vars, uint32_t width, uint32_t n_assign,
                      uint32_t batch, uint32_t threads, uint32_t* first_fail) {
  Field f;
  f.init(mod_le, mod_len);
  std::vector<F4> cm(n_coefs);
  for (uint32_t i = 0; i < n_coefs; ++i) cm[i] = f.from_bytes(coefs + (size_t)i * coef_width, coef_width);
  std::vector<std::vector<F4>> vars(batch);
  auto lincomb = [&](const std::vector<F4>& v, uint32_t t0, uint32_t t1) {
    F4 acc;
    memset(&acc, 0, sizeof acc);
    for (uint32_t t = t0; t < t1; ++t) {
      const F4& x = term_var[t] == ~0ull ? f.one : v[term_var[t]];
      acc = f.add(acc, f.mul(x, cm[term_coef[t]]));
    }
    return acc;
  };
  auto run = [&](bool check) {
    std::atomic<uint32_t> next(0);
    auto worker = [&]() {
      for (;;) {
        const uint32_t lane = next.fetch_add(1);
        if (lane >= batch) break;
        std::vector<F4>& v = vars[lane];
        if (!check) {
          v.resize(n_vars);
          for (uint32_t k = 0; k < n_base; ++k) v[k] = f.from_bytes(base + ((size_t)lane * n_base + k) * width, width);
          for (uint32_t r = 0; r < n_assign; ++r) {
            const uint32_t* rp = row_ptr + 3 * (size_t)r;
            v[term_var[rp[2]]] = f.mul(lincomb(v, rp[0], rp[1]), lincomb(v, rp[1], rp[2]));
          }
        } else {
          uint32_t ff = 0xFFFFFFFFu;
          for (uint32_t r = 0; r < n_rows; ++r) {
            const uint32_t* rp = row_ptr + 3 * (size_t)r;
            const F4 ab = f.mul(lincomb(v, rp[0], rp[1]), lincomb(v, rp[1], rp[2]));
            const F4 c = lincomb(v, rp[2], rp[3]);
            if (memcmp(ab.l, c.l, 32) != 0 && ff == 0xFFFFFFFFu) ff = r;
          }
          first_fail[lane] = ff;
        }
      }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  };
  run(false);
  auto t0 = std::chrono::steady_clock::now();
  run(true);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
