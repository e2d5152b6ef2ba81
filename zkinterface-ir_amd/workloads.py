"""Synthetic SIEVE IR workloads of BASELINE.json / SURVEY.md 8(d), emitted as
real `.sieve` message streams (the reference publishes no benchmark inputs).

C2 `R_arith`: BN254 scalar field, layered PRNG circuit of W x D Add/Mul gates,
256 instance + (W-256) witness inputs, 64 output comparisons
{Instance, MulConstant(p-1), Add, AssertZero}, `Free` after every layer,
relation split into <= 100,000-gate messages like the reference's GateBuilder
(rust/src/producers/builder.rs:46-49,72).
"""
import numpy as np

from .sieve_writer import int_to_le, write_relation_segments

BN254_R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MAX_GATES_PER_MESSAGE = 100_000
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Stateless splitmix64 finaliser on a numpy uint64 array (wrapping arithmetic)."""
    with np.errstate(over='ignore'):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _limbs_of(v, n=4):
    return [np.uint64((v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(n)]


def _cond_sub(limbs, m):
    """limbs: list of 4 uint64 arrays (little-endian); subtract m where limbs >= m."""
    ml = _limbs_of(m)
    # compare from the top limb down
    ge = np.ones(limbs[0].shape, dtype=bool)
    decided = np.zeros(limbs[0].shape, dtype=bool)
    for i in (3, 2, 1, 0):
        gt = limbs[i] > ml[i]
        lt = limbs[i] < ml[i]
        ge = np.where(~decided & gt, True, ge)
        ge = np.where(~decided & lt, False, ge)
        decided |= gt | lt
    out = []
    borrow = np.zeros(limbs[0].shape, dtype=np.uint64)
    with np.errstate(over='ignore'):
        for i in range(4):
            d = limbs[i] - ml[i]
            b1 = (limbs[i] < ml[i]).astype(np.uint64)
            d2 = d - borrow
            b2 = (d < borrow).astype(np.uint64)
            out.append(np.where(ge, d2, limbs[i]))
            borrow = b1 | b2
    return out


def random_field_elements(seed, shape, p=BN254_R):
    """[shape..., 32] uint8: 4 x splitmix64 words assembled little-endian, reduced mod p."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    limbs = []
    with np.errstate(over='ignore'):
        for k in range(4):
            limbs.append(splitmix64((np.uint64(seed) + idx * np.uint64(4) + np.uint64(k)) & _M64))
    bits = p.bit_length()
    if bits < 256:  # p > 2^(bits-1): at most 2^(256-bits+1) multiples to remove
        for mult in [1 << s for s in range(256 - bits + 1, -1, -1)]:
            if mult * p < (1 << 256):
                limbs = _cond_sub(limbs, mult * p)
    else:
        limbs = _cond_sub(limbs, p)
    arr = np.stack(limbs, axis=-1).astype('<u8')  # [n, 4]
    return arr.view(np.uint8).reshape(tuple(shape) + (32,))


class ArithLayered:
    """The C2 / C3 workload.  Wire ids: layer l occupies [l*W, (l+1)*W); the
    epilogue uses ids from (D+1)*W upwards."""

    def __init__(self, W=4096, D=256, n_instance0=256, n_out=64, seed=0x5EED0001, p=BN254_R, mul_percent=None):
        assert n_instance0 <= W and n_out <= W
        self.W, self.D, self.n_instance0, self.n_out, self.seed, self.p = W, D, n_instance0, n_out, seed, p
        self.default_mix = mul_percent is None
        self.mod_le = int_to_le(p)
        self.width = 8 * ((p.bit_length() + 63) // 64)
        self.n_instance = n_instance0 + n_out
        self.n_witness = W - n_instance0
        self.n_gates = W * D                      # the Add/Mul gates the metric counts
        layer = np.arange(1, D + 1, dtype=np.uint64)[:, None]
        j = np.arange(W, dtype=np.uint64)[None, :]
        with np.errstate(over='ignore'):
            h = splitmix64(np.uint64(seed) ^ ((layer << np.uint64(32)) + j))
            h2 = splitmix64(h)
        if mul_percent is None:
            self.is_mul = (h & np.uint64(1)).astype(bool)
        else:
            self.is_mul = ((h >> np.uint64(8)) % np.uint64(100)) < np.uint64(mul_percent)
        self.src_a = (h2 % np.uint64(W)).astype(np.uint32)                        # index inside layer l-1
        self.src_b = ((h2 >> np.uint64(32)) % np.uint64(W)).astype(np.uint32)

    # ---- relation ---------------------------------------------------------
    def _segments(self, with_epilogue=True, free_last=True):
        W, D = self.W, self.D
        segs = []
        inputs = [('instance', k) for k in range(self.n_instance0)] + [('witness', k) for k in range(self.n_instance0, W)]
        segs.append(('gates', inputs, len(inputs)))
        for l in range(1, D + 1):
            base_prev, base = (l - 1) * W, l * W
            tags = np.where(self.is_mul[l - 1], 5, 4).astype(np.uint8)
            outs = np.arange(base, base + W, dtype=np.uint64)
            lefts = self.src_a[l - 1].astype(np.uint64) + np.uint64(base_prev)
            rights = self.src_b[l - 1].astype(np.uint64) + np.uint64(base_prev)
            segs.append(('bulk', tags, outs, lefts, rights, W))
            segs.append(('gates', [('free', base_prev, base_prev + W - 1)], 1))
        if with_epilogue:
            e = (D + 1) * W
            neg_one = int_to_le(self.p - 1)
            ep = []
            for t in range(self.n_out):
                w0, w1, w2 = e + 3 * t, e + 3 * t + 1, e + 3 * t + 2
                ep += [('instance', w0), ('mulc', w1, w0, neg_one), ('add', w2, D * W + t, w1), ('assert_zero', w2)]
            ep.append(('free', e, e + 3 * self.n_out - 1))
            segs.append(('gates', ep, len(ep)))
        if free_last:
            segs.append(('gates', [('free', D * W, D * W + W - 1)], 1))
        return segs

    def relation_messages(self, with_epilogue=True, free_last=True):
        """List of size-prefixed Relation messages, each <= 100,000 directives."""
        msgs, cur, cur_n = [], [], 0
        for seg in self._segments(with_epilogue, free_last):
            n = seg[-1]
            if cur and cur_n + n > MAX_GATES_PER_MESSAGE:
                msgs.append(cur)
                cur, cur_n = [], 0
            cur.append(seg[:-1])
            cur_n += n
        if cur:
            msgs.append(cur)
        return [write_relation_segments(self.mod_le, 'arithmetic', 'simple', m) for m in msgs]

    # ---- inputs -----------------------------------------------------------
    def inputs(self, batch, lane_offset=0):
        """(instances [batch][n_instance][width], witnesses [batch][n_witness][width]) uint8 arrays.
        The last n_out instances (expected outputs) are left zero: see set_expected_outputs()."""
        W = self.W
        vals = random_field_elements(self.seed + 0x1000 + lane_offset * W * 4, (batch, W), self.p)[..., :self.width]
        inst = np.zeros((batch, self.n_instance, self.width), dtype=np.uint8)
        inst[:, :self.n_instance0] = vals[:, :self.n_instance0]
        wit = np.ascontiguousarray(vals[:, self.n_instance0:])
        return inst, wit

    def set_expected_outputs(self, inst, outputs, lane_offset=0, corrupt_every=97):
        """outputs: [batch][n_out] python ints or [batch][n_out][width] uint8.  Lanes whose global index
        is a multiple of `corrupt_every` get output 0 off by one => those statements are FALSE."""
        batch = inst.shape[0]
        out = np.asarray(outputs)
        if out.dtype != np.uint8:
            raise TypeError('outputs must be a uint8 array of little-endian values')
        inst[:, self.n_instance0:] = out.reshape(batch, self.n_out, self.width)
        bad = [i for i in range(batch) if corrupt_every and (i + lane_offset) % corrupt_every == 0]
        for i in bad:
            v = (int.from_bytes(inst[i, self.n_instance0].tobytes(), 'little') + 1) % self.p
            inst[i, self.n_instance0] = np.frombuffer(v.to_bytes(self.width, 'little'), dtype=np.uint8)
        return len(bad)

    def output_wire_ids(self):
        return [self.D * self.W + t for t in range(self.n_out)]


class BoolLayered:
    """The C4 workload: GF(2), gateset `boolean`, W x D layered And/Xor/Not circuit
    (45 / 45 / 10 %), n_instance0 instance + (W - n_instance0) witness bits, n_out outputs compared by
    {Instance, Xor, AssertZero}.  One byte per input value."""

    def __init__(self, W=16384, D=640, n_instance0=1024, n_out=64, seed=0xB001C4, wiring='random', mix=(45, 45)):
        self.W, self.D, self.n_instance0, self.n_out, self.seed, self.p = W, D, n_instance0, n_out, seed, 2
        self.wiring = wiring
        self.mod_le = bytes([2])
        self.width = 1
        self.n_instance = n_instance0 + n_out
        self.n_witness = W - n_instance0
        self.n_gates = W * D
        layer = np.arange(1, D + 1, dtype=np.uint64)[:, None]
        j = np.arange(W, dtype=np.uint64)[None, :]
        with np.errstate(over='ignore'):
            h = splitmix64(np.uint64(seed) ^ ((layer << np.uint64(32)) + j))
            h2 = splitmix64(h)
        r = (h >> np.uint64(8)) % np.uint64(100)
        # and / xor / not: `mix` = per cent of and, of xor (the C4 definition: 45 / 45 / 10; other mixes for the tests of the
        # row layout -- all `and`, no `and`, ...)
        self.kind = np.where(r < mix[0], 8, np.where(r < mix[0] + mix[1], 9, 10)).astype(np.uint8)
        self.src_a = (h2 % np.uint64(W)).astype(np.uint32)
        self.src_b = ((h2 >> np.uint64(32)) % np.uint64(W)).astype(np.uint32)
        if wiring == 'identity':  # experiment only: gate j reads wires j and j+1 of the previous layer
            self.src_a = np.broadcast_to(np.arange(W, dtype=np.uint32), (D, W)).copy()
            self.src_b = (self.src_a + 1) % W

    def _segments(self, with_epilogue=True, free_last=True):
        W, D = self.W, self.D
        inputs = [('instance', k) for k in range(self.n_instance0)] + [('witness', k) for k in range(self.n_instance0, W)]
        segs = [('gates', inputs, len(inputs))]
        for l in range(1, D + 1):
            base_prev, base = (l - 1) * W, l * W
            k = self.kind[l - 1]
            outs = np.arange(base, base + W, dtype=np.uint64)
            a = self.src_a[l - 1].astype(np.uint64) + np.uint64(base_prev)
            b = self.src_b[l - 1].astype(np.uint64) + np.uint64(base_prev)
            binary = k != 10
            if binary.any():
                segs.append(('bulk', k[binary], outs[binary], a[binary], b[binary], int(binary.sum())))
            if (~binary).any():
                segs.append(('bulk1', k[~binary], outs[~binary], a[~binary], int((~binary).sum())))
            segs.append(('gates', [('free', base_prev, base_prev + W - 1)], 1))
        if with_epilogue:
            e = (D + 1) * W
            ep = []
            for t in range(self.n_out):
                w0, w1 = e + 2 * t, e + 2 * t + 1
                ep += [('instance', w0), ('xor', w1, D * W + t, w0), ('assert_zero', w1)]
            ep.append(('free', e, e + 2 * self.n_out - 1))
            segs.append(('gates', ep, len(ep)))
        if free_last:
            segs.append(('gates', [('free', D * W, D * W + W - 1)], 1))
        return segs

    def relation_messages(self, with_epilogue=True, free_last=True):
        msgs, cur, cur_n = [], [], 0
        for seg in self._segments(with_epilogue, free_last):
            n = seg[-1]
            if cur and cur_n + n > MAX_GATES_PER_MESSAGE:
                msgs.append(cur)
                cur, cur_n = [], 0
            cur.append(seg[:-1])
            cur_n += n
        if cur:
            msgs.append(cur)
        return [write_relation_segments(self.mod_le, 'boolean', 'simple', m) for m in msgs]

    def inputs(self, batch, lane_offset=0):
        idx = (np.arange(batch * self.W, dtype=np.uint64) + np.uint64(lane_offset * self.W)).reshape(batch, self.W)
        with np.errstate(over='ignore'):
            bits = (splitmix64(np.uint64(self.seed + 77) + idx) & np.uint64(1)).astype(np.uint8)
        inst = np.zeros((batch, self.n_instance, 1), dtype=np.uint8)
        inst[:, :self.n_instance0, 0] = bits[:, :self.n_instance0]
        wit = np.ascontiguousarray(bits[:, self.n_instance0:, None])
        return inst, wit

    def set_expected_outputs(self, inst, outputs, lane_offset=0, corrupt_every=97):
        """outputs: [batch][n_out] uint8 bits"""
        batch = inst.shape[0]
        inst[:, self.n_instance0:, 0] = np.asarray(outputs, dtype=np.uint8).reshape(batch, self.n_out)
        bad = [i for i in range(batch) if corrupt_every and (i + lane_offset) % corrupt_every == 0]
        for i in bad:
            inst[i, self.n_instance0, 0] ^= 1
        return len(bad)

    def output_wire_ids(self):
        return [self.D * self.W + t for t in range(self.n_out)]


class Sha256Compress:
    """A real Boolean circuit instead of a layered PRNG one: the SHA-256 compression of ONE padded message block (FIPS 180-4)
    over GF(2), gateset `boolean` -- 64 rounds and the 48-word message schedule out of 32-bit ripple-carry adders (3 xor +
    2 and per bit), Ch, Maj and the sigma functions (rotations are wiring): about 1.2 * 10^5 And / Xor / Not gates in a few
    thousand dependency levels of a few dozen gates each, the shape the narrow-level path of the GF(2) kernel is for.
    Witness = the 512 bits of the block (word by word, least significant bit first inside a word), instance = the 256 digest
    bits a statement claims; the epilogue compares them {Instance, Xor, AssertZero}.  The expected digests are hashlib's."""

    K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
         0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
         0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
         0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
         0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
         0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
    IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]

    def __init__(self, rounds=64, seed=0x5A256):
        assert 1 <= rounds <= 64
        self.closed_form = True   # (bench.py: the expected values come with the inputs, no probe pass)
        self.rounds, self.seed, self.p = rounds, seed, 2
        self.mod_le = bytes([2])
        self.width = 1
        self.n_witness, self.n_instance, self.n_out = 512, 256, 256
        g = []                                   # gates, in wire order: wire k is defined by gate k (inputs and constants included)
        self._g = g

        def new(t):
            g.append(t)
            return len(g) - 1
        xor = lambda a, b: new(('xor', len(g), a, b))
        and_ = lambda a, b: new(('and', len(g), a, b))
        msg = [[new(('witness', len(g))) for _ in range(32)] for _ in range(16)]
        zero = new(('constant', len(g), bytes([0])))
        one = new(('constant', len(g), bytes([1])))
        const = lambda v: [one if (v >> i) & 1 else zero for i in range(32)]
        rotr = lambda x, r: [x[(i + r) % 32] for i in range(32)]
        shr = lambda x, r: [x[i + r] if i + r < 32 else zero for i in range(32)]
        xor3 = lambda x, y, z: [xor(xor(a, b), c) for a, b, c in zip(x, y, z)]

        def add(x, y):                           # ripple carry, bit 0 first; the carry out of bit 31 is dropped (mod 2^32)
            out, c = [], None
            for i in range(32):
                t = xor(x[i], y[i])
                out.append(t if c is None else xor(t, c))
                if i < 31:
                    u = and_(x[i], y[i])
                    c = u if c is None else xor(u, and_(c, t))
            return out
        W = list(msg)
        for t in range(16, max(16, rounds)):
            s0 = xor3(rotr(W[t - 15], 7), rotr(W[t - 15], 18), shr(W[t - 15], 3))
            s1 = xor3(rotr(W[t - 2], 17), rotr(W[t - 2], 19), shr(W[t - 2], 10))
            W.append(add(add(add(W[t - 16], s0), W[t - 7]), s1))
        st = [const(v) for v in self.IV]
        a, b, c, d, e, f, h_g, h = st
        gg = h_g
        for t in range(rounds):
            S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25))
            ch = [xor(gi, and_(ei, xor(fi, gi))) for ei, fi, gi in zip(e, f, gg)]          # (e & f) ^ (~e & g)
            t1 = add(add(add(add(h, S1), ch), const(self.K[t])), W[t])
            S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22))
            maj = [xor(and_(ai, bi), and_(ci, xor(ai, bi))) for ai, bi, ci in zip(a, b, c)]
            t2 = add(S0, maj)
            h, gg, f, e, d, c, b, a = gg, f, e, add(d, t1), c, b, a, add(t1, t2)
        out = [add(x, y) for x, y in zip(st, [a, b, c, d, e, f, gg, h])]
        self.digest_wires = [w for word in out for w in word]       # word by word, least significant bit first
        self.n_gates = sum(1 for t in g if t[0] in ('xor', 'and', 'not'))
        self.n_wires = len(g)

    def relation_messages(self, with_epilogue=True, free_last=True):
        from .sieve_writer import write_relation
        gates = list(self._g)
        n = self.n_wires
        if with_epilogue:
            for k, w in enumerate(self.digest_wires):
                gates += [('instance', n + 2 * k), ('xor', n + 2 * k + 1, w, n + 2 * k), ('assert_zero', n + 2 * k + 1)]
            gates.append(('free', n, n + 2 * len(self.digest_wires) - 1))
        if free_last:
            gates.append(('free', 0, n - 1))
        msgs = []
        for at in range(0, len(gates), MAX_GATES_PER_MESSAGE):
            msgs.append(write_relation(self.mod_le, 'boolean', 'simple', [], gates[at:at + MAX_GATES_PER_MESSAGE]))
        return msgs

    def output_wire_ids(self):
        return list(self.digest_wires)

    @staticmethod
    def _block_bits(block):
        """64 bytes -> 512 witness bits: big-endian words, least significant bit first inside a word"""
        words = np.frombuffer(block, dtype='>u4').astype(np.uint32)
        return ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(512)

    def inputs(self, batch, lane_offset=0, corrupt_every=97):
        """(instances [batch][256][1], witnesses [batch][512][1], lanes made false): every lane hashes its own 55-byte message
        (one padded block); the claimed digest is hashlib's, with one bit flipped on every `corrupt_every`-th lane"""
        import hashlib
        assert self.rounds == 64, 'the expected digests are those of the full compression function'
        inst = np.zeros((batch, 256, 1), dtype=np.uint8)
        wit = np.zeros((batch, 512, 1), dtype=np.uint8)
        bad = 0
        for lane in range(batch):
            rng = np.random.default_rng(self.seed + lane + lane_offset)
            msg = rng.integers(0, 256, size=55, dtype=np.uint8).tobytes()
            block = msg + b'\x80' + (8 * len(msg)).to_bytes(8, 'big')
            wit[lane, :, 0] = self._block_bits(block)
            inst[lane, :, 0] = self._block_bits(hashlib.sha256(msg).digest() + bytes(32))[:256]
            if corrupt_every and (lane + lane_offset) % corrupt_every == 0:
                inst[lane, (lane + lane_offset) % 256, 0] ^= 1
                bad += 1
        return inst, wit, bad


class R1csSynthetic:
    """The C5 workload: M rows over n_base + M variables over BN254.  Row i is
    (sum of 3 coef*var) * (sum of 3 coef*var) = z_i with the six variables drawn from the variables that
    exist before z_i (base variables and earlier z) and coefficients from a pool of random field
    elements; z_i is fresh and is assigned the product, so the system is satisfiable.  One last row
    compares z_{M-1} with a per-lane expected value held in an extra base variable (the handle used to
    make every 97th lane unsatisfied).  Rows are emitted sorted by dependency level so that the
    witness generation of one level is a single independent launch."""

    def __init__(self, M=1 << 20, n_base=4096, n_coefs=1 << 16, seed=0xC5, p=BN254_R, coef_kind='random'):
        """coef_kind: 'random' -- coefficients are random field elements (BASELINE configs[4]); 'small' -- what
        FromR1CSConverter expansions and hand-written systems mostly hold (from_r1cs.rs:110-125): 1 (35 %), -1 (25 %) and
        signed integers of up to 16 bits (40 %), same rows and variables"""
        self.M, self.n_base, self.p, self.seed, self.coef_kind = M, n_base, p, seed, coef_kind
        self.width = 8 * ((p.bit_length() + 63) // 64)
        self.mod_le = int_to_le(p)
        self.n_witness = n_base + 1  # + the expected-output variable E
        rng = np.random.default_rng(seed)
        # variable ids (caller space of zkgpu_r1cs_load_csr): base k -> k, E -> n_base, z_i -> n_base + 1 + i
        hi = (np.arange(M, dtype=np.int64) + n_base)[:, None]           # row i may use vars < n_base + i
        picks = (rng.random((M, 6)) * hi).astype(np.int64)               # in [0, n_base + i)
        picks = np.where(picks >= n_base, picks + 1, picks)              # skip E's id
        self.coef_idx = rng.integers(0, n_coefs, size=(M, 6), dtype=np.int64)
        self.coefs = random_field_elements(seed + 17, (n_coefs,), p)[:, :self.width]
        if coef_kind == 'small':
            crng = np.random.default_rng(seed + 18)
            u = crng.random(n_coefs)
            mag = crng.integers(2, 1 << 16, size=n_coefs)
            sign = crng.integers(0, 2, size=n_coefs)
            vals = [1 if u[i] < 0.35 else (p - 1) if u[i] < 0.60 else (int(mag[i]) % p if sign[i] else (p - int(mag[i])) % p) or 1
                    for i in range(n_coefs)]
            self.coefs = np.frombuffer(b''.join(v.to_bytes(self.width, 'little') for v in vals), dtype=np.uint8).reshape(n_coefs, self.width).copy()
        elif coef_kind != 'random':
            raise ValueError('coef_kind: random or small')
        # dependency levels (sequential by construction: row i only sees earlier z)
        level = [0] * (n_base + 1 + M)
        pl = picks.tolist()
        for i in range(M):
            r = pl[i]
            level[n_base + 1 + i] = 1 + max(level[r[0]], level[r[1]], level[r[2]], level[r[3]], level[r[4]], level[r[5]])
        lv = np.array(level[n_base + 1:], dtype=np.int64)
        order = np.argsort(lv, kind='stable')
        self.row_level = lv[order]
        self.n_levels = int(lv.max())
        # renumber z so that the emitted row r defines extra variable r
        new_id = np.empty(M, dtype=np.int64)
        new_id[order] = np.arange(M, dtype=np.int64)
        zmask = picks > n_base
        picks = np.where(zmask, n_base + 1 + new_id[np.clip(picks - n_base - 1, 0, M - 1)], picks)
        self.picks = picks[order]
        self.coef_idx = self.coef_idx[order]
        self.last_z = n_base + 1 + int(new_id[M - 1])   # any z would do; keep the original last row
        self.level_bounds = np.searchsorted(self.row_level, np.arange(1, self.n_levels + 2))

    def csr(self):
        """(row_ptr, term_var, term_coef, coef_bytes) for M product rows + the final comparison row"""
        M = self.M
        one = 1 << 16  # coefficient index of the literal 1 appended to the pool below
        coef_bytes = np.concatenate([self.coefs, np.frombuffer((1).to_bytes(self.width, 'little'), dtype=np.uint8)[None, :]])
        one = len(coef_bytes) - 1
        tv = np.empty((M, 7), dtype=np.uint64)
        tc = np.empty((M, 7), dtype=np.uint32)
        tv[:, :6] = self.picks
        tv[:, 6] = np.arange(M, dtype=np.uint64) + np.uint64(self.n_base + 1)
        tc[:, :6] = self.coef_idx
        tc[:, 6] = one
        row_ptr = np.empty(3 * (M + 1) + 1, dtype=np.uint32)
        base = (np.arange(M, dtype=np.uint32) * 7)
        row_ptr[0:3 * M:3] = base
        row_ptr[1:3 * M:3] = base + 3
        row_ptr[2:3 * M:3] = base + 6
        # final row: (z_last * 1) * (one) = (E * 1)
        t0 = 7 * M
        row_ptr[3 * M:3 * M + 4] = [t0, t0 + 1, t0 + 2, t0 + 3]
        term_var = np.concatenate([tv.reshape(-1), np.array([self.last_z, 0xFFFFFFFFFFFFFFFF, self.n_base], dtype=np.uint64)])
        term_coef = np.concatenate([tc.reshape(-1), np.array([one, one, one], dtype=np.uint32)])
        return row_ptr, term_var, term_coef, coef_bytes

    def base_relation(self):
        """a relation that only loads the n_base + 1 base variables (witness gates)"""
        from .sieve_writer import write_relation
        gates = [('witness', k) for k in range(self.n_witness)]
        return write_relation(self.mod_le, 'arithmetic', 'simple', [], gates)

    def witnesses(self, batch, lane_offset=0):
        w = random_field_elements(self.seed + 0x2000 + lane_offset * self.n_witness * 4, (batch, self.n_witness), self.p)
        if self.width > w.shape[-1]:   # (a field wider than the 256 bits the generator makes: values below 2^256)
            w = np.concatenate([w, np.zeros(w.shape[:-1] + (self.width - w.shape[-1],), dtype=np.uint8)], axis=-1)
        w = np.ascontiguousarray(w[..., :self.width])
        w[:, self.n_base] = 0
        return w


class StructuredArith:
    """A structured relation of about a million backend calls (the shape of the reference's own example,
    rust/src/producers/examples.rs:72-212, scaled up): a `For` loop of N iterations over a named function that calls
    another function and multiplexes two anonymous branches with a `Switch`, followed by a second `For` loop (anonymous
    body) that compares every result with an expected instance value.  Nothing here is flat: the host has to inline
    the calls, unroll the loops, copy wires in and out of every scope (evaluator.rs:698-746) and build the 352-product
    exponent ladder of each of the 2N case weights (evaluator.rs:801-839) -- about 745 backend calls per iteration
    over BN254.

      step(o; a, b, c):  t = mul(a, b);  switch c { 0: o = t + a;  1: o = t * b }

    Wire ids: a_i = i, b_i = N + i, c_i = 2N + i (witness), o_i = 3N + i, e_i = 4N + i (instance)."""

    def __init__(self, N=1408, seed=0x57C7, p=BN254_R, chained=False):
        """chained: iteration i takes iteration i-1's result as its first input (acc = step(acc, b_i, c_i)) and only the
        last result is compared -- the same calls, but a dependency chain N iterations deep instead of N independent ones:
        the shape that leaves a GPU nothing but the witnesses to run in parallel."""
        self.N, self.seed, self.p, self.chained = N, seed, p, chained
        self.closed_form = True
        self.mod_le = int_to_le(p)
        self.width = 8 * ((p.bit_length() + 63) // 64)
        self.n_instance = 1 if chained else N
        self.n_witness = 2 * N + 1 if chained else 3 * N
        self.n_out = 1 if chained else N

    def relation_messages(self):
        from .sieve_writer import write_relation
        N = self.N
        neg_one = int_to_le(self.p - 1)
        mul = 'wl::mul'
        step = 'wl::step'
        functions = [
            (mul, 1, 2, 0, 0, [('mul', 0, 1, 2)]),
            (step, 1, 3, 0, 0, [
                ('call', mul, [4], [1, 2]),                                   # t = a * b
                ('switch', 3, [0], [bytes([0]), bytes([1])], [
                    ('anon', [4, 1], 0, 0, [('add', 0, 1, 2)]),              # case 0: t + a
                    ('anon', [4, 2], 0, 0, [('mul', 0, 1, 2)]),              # case 1: t * b
                ]),
                ('free', 4, None),
            ]),
        ]
        if self.chained:
            # b_i = i, c_i = N + i, acc_0 = 2N (witness), acc_{i+1} = 2N + 1 + i, expected = 3N + 1 (instance)
            gates = [('witness', k) for k in range(2 * N + 1)]
            gates.append(('for', 'i', 0, N - 1, [(2 * N + 1, 3 * N)],
                          ('call', step, [('add', ('name', 'i'), ('const', 2 * N + 1))],
                           [('add', ('name', 'i'), ('const', 2 * N)), ('name', 'i'), ('add', ('name', 'i'), ('const', N))])))
            gates += [('instance', 3 * N + 1), ('mulc', 3 * N + 2, 3 * N + 1, neg_one), ('add', 3 * N + 3, 3 * N, 3 * N + 2),
                      ('assert_zero', 3 * N + 3), ('free', 0, 3 * N + 3)]
            return [write_relation(self.mod_le, '@add,@mul,@mulc,', '@for,@switch,@function,', functions, gates)]
        gates = [('witness', k) for k in range(3 * N)]
        gates.append(('for', 'i', 0, N - 1, [(3 * N, 4 * N - 1)],
                      ('call', step, [('add', ('name', 'i'), ('const', 3 * N))],
                       [('name', 'i'), ('add', ('name', 'i'), ('const', N)), ('add', ('name', 'i'), ('const', 2 * N))])))
        gates.append(('free', 0, 3 * N - 1))
        gates += [('instance', 4 * N + k) for k in range(N)]
        gates.append(('for', 'j', 0, N - 1, [],
                      ('anon', [], [('add', ('name', 'j'), ('const', 3 * N)), ('add', ('name', 'j'), ('const', 4 * N))], 0, 0,
                       [('mulc', 2, 1, neg_one), ('add', 3, 0, 2), ('assert_zero', 3)])))
        gates.append(('free', 3 * N, 5 * N - 1))
        return [write_relation(self.mod_le, '@add,@mul,@mulc,', '@for,@switch,@function,', functions, gates)]

    def inputs(self, batch, lane_offset=0, corrupt_every=97):
        """(instances [batch][N][width], witnesses [batch][3N][width], lanes made false); the expected values are the
        closed form of `step` in Python integers, off by one on every `corrupt_every`-th lane (global index)."""
        N, p = self.N, self.p
        if self.chained:
            return self._chained_inputs(batch, lane_offset, corrupt_every)
        ab = random_field_elements(self.seed + 0x3000 + lane_offset * 2 * N * 4, (batch, 2 * N), p)[..., :self.width]
        idx = (np.arange(batch * N, dtype=np.uint64) + np.uint64(lane_offset * N)).reshape(batch, N)
        with np.errstate(over='ignore'):
            c = (splitmix64(np.uint64(self.seed + 99) + idx) & np.uint64(1)).astype(np.uint8)
        wit = np.zeros((batch, 3 * N, self.width), dtype=np.uint8)
        wit[:, :2 * N] = ab
        wit[:, 2 * N:, 0] = c
        inst = np.zeros((batch, N, self.width), dtype=np.uint8)
        bad = 0
        for lane in range(batch):
            row = ab[lane].reshape(2 * N, self.width).tobytes()
            vals = [int.from_bytes(row[k * self.width:(k + 1) * self.width], 'little') for k in range(2 * N)]
            out = bytearray()
            for i in range(N):
                a, b = vals[i], vals[N + i]
                t = a * b % p
                o = t * b % p if c[lane, i] else (t + a) % p
                if i == 0 and corrupt_every and (lane + lane_offset) % corrupt_every == 0:
                    o = (o + 1) % p
                out += o.to_bytes(self.width, 'little')
            bad += int(bool(corrupt_every) and (lane + lane_offset) % corrupt_every == 0)
            inst[lane] = np.frombuffer(bytes(out), dtype=np.uint8).reshape(N, self.width)
        return inst, wit, bad


def _chained_inputs(self, batch, lane_offset=0, corrupt_every=97):
    N, p = self.N, self.p
    vals = random_field_elements(self.seed + 0x5000 + lane_offset * (N + 1) * 4, (batch, N + 1), p)[..., :self.width]
    idx = (np.arange(batch * N, dtype=np.uint64) + np.uint64(lane_offset * N)).reshape(batch, N)
    with np.errstate(over='ignore'):
        c = (splitmix64(np.uint64(self.seed + 99) + idx) & np.uint64(1)).astype(np.uint8)
    wit = np.zeros((batch, 2 * N + 1, self.width), dtype=np.uint8)
    wit[:, :N] = vals[:, :N]            # b_i
    wit[:, N:2 * N, 0] = c              # c_i
    wit[:, 2 * N] = vals[:, N]          # acc_0
    inst = np.zeros((batch, 1, self.width), dtype=np.uint8)
    bad = 0
    for lane in range(batch):
        row = vals[lane].reshape(N + 1, self.width).tobytes()
        v = [int.from_bytes(row[k * self.width:(k + 1) * self.width], 'little') for k in range(N + 1)]
        acc = v[N]
        for i in range(N):
            t = acc * v[i] % p
            acc = t * v[i] % p if c[lane, i] else (t + acc) % p
        if corrupt_every and (lane + lane_offset) % corrupt_every == 0:
            acc = (acc + 1) % p
            bad += 1
        inst[lane, 0] = np.frombuffer(acc.to_bytes(self.width, 'little'), dtype=np.uint8)
    return inst, wit, bad


StructuredArith._chained_inputs = _chained_inputs


def expected_satisfied(batch, lane_offset=0, corrupt_every=97):
    return batch - sum(1 for i in range(batch) if (i + lane_offset) % corrupt_every == 0)
