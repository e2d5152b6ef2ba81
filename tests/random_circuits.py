"""Random structured SIEVE IR relations for fuzzing the product against the oracle:
functions, anonymous calls, for loops (named and anonymous bodies, iterator
expressions), switches (also nested in functions and in other switches), frees,
weighted and unweighted asserts.  Pure data generation (gate tuples for
zkinterface_ir_amd.sieve_writer)."""
import random

from zkinterface_ir_amd import sieve_writer as sw


class Gen:
    def __init__(self, seed, p, boolean=False, switches=True):
        self.r = random.Random(seed)
        self.p = p
        self.boolean = boolean
        self.switches = switches   # False: no Switch gates (each case costs a ladder of bits(p) products)
        self.functions = []  # (name, n_out, n_in, n_inst, n_wit, body)
        self.n_inst = 0      # top-level consumption counters
        self.n_wit = 0

    # ---- helpers -------------------------------------------------------------
    def const(self):
        if self.boolean:
            return bytes([self.r.randrange(2)])
        v = self.r.choice([0, 1, 2, self.p - 1, self.r.randrange(self.p)]) % self.p  # canonical constants only
        width = self.r.choice([None, 4, 32]) if v < 2 ** 32 else None  # trailing zeros are legal (sieve_ir.fbs:55-59)
        return sw.int_to_le(v, width)

    def simple_ops(self, live, next_id, n, counters, allow_inputs=True, assert_prob=0.15):
        """n random simple gates over the wires in `live` (list, extended in place).
        counters = [n_inst, n_wit] consumed by this body.  Returns (gates, next_id)."""
        gates = []
        for _ in range(n):
            kinds = ['bin', 'bin', 'bin', 'un', 'const']
            if allow_inputs:
                kinds += ['inst', 'wit']
            if live:
                kinds += ['assert'] if self.r.random() < assert_prob * 4 else []
            k = self.r.choice(kinds if live else ['const'] + (['inst', 'wit'] if allow_inputs else []))
            out = next_id
            if k == 'bin':
                op = self.r.choice(['and', 'xor'] if self.boolean else ['add', 'mul'])
                gates.append((op, out, self.r.choice(live), self.r.choice(live)))
            elif k == 'un':
                if self.boolean:
                    gates.append((self.r.choice(['not', 'copy']), out, self.r.choice(live)))
                else:
                    op = self.r.choice(['addc', 'mulc', 'copy'])
                    if op == 'copy':
                        gates.append(('copy', out, self.r.choice(live)))
                    else:
                        gates.append((op, out, self.r.choice(live), self.const()))
            elif k == 'const':
                gates.append(('constant', out, self.const()))
            elif k == 'inst':
                gates.append(('instance', out))
                counters[0] += 1
            elif k == 'wit':
                gates.append(('witness', out))
                counters[1] += 1
            elif k == 'assert':
                # mostly true assertions: x - x == 0 (arith) / x ^ x == 0 (bool); sometimes a raw wire
                x = self.r.choice(live)
                if self.r.random() < 0.7:
                    if self.boolean:
                        gates.append(('xor', out, x, x))
                        gates.append(('assert_zero', out))
                    else:
                        gates.append(('mulc', out, x, sw.int_to_le(self.p - 1)))
                        gates.append(('add', out + 1, x, out))
                        gates.append(('assert_zero', out + 1))
                        live.append(out + 1)
                        next_id += 1
                else:
                    gates.append(('assert_zero', x))
                    continue
            live.append(out)
            next_id += 1
        return gates, next_id

    def make_function(self, depth=0):
        n_out, n_in = self.r.randrange(1, 3), self.r.randrange(0, 3)
        live = list(range(n_out, n_out + n_in))
        counters = [0, 0]
        body, nid = self.simple_ops(live, n_out + n_in, self.r.randrange(2, 7), counters, assert_prob=0.1)
        if depth == 0 and self.functions and self.r.random() < 0.4:
            g, nid = self.call_gate(live, nid, counters)
            body += g
        if depth == 0 and self.r.random() < 0.3 and live and self.switches:
            g, nid = self.switch_gate(live, nid, counters, depth=1)
            body += g
        for o in range(n_out):  # outputs must be assigned exactly once
            if live:
                body.append(('copy', o, self.r.choice(live)))
            else:
                body.append(('constant', o, self.const()))
        name = 'f%d' % len(self.functions)
        self.functions.append((name, n_out, n_in, counters[0], counters[1], body))
        return name

    def call_gate(self, live, nid, counters):
        name, n_out, n_in, fi, fw, _ = self.r.choice(self.functions)
        if n_in and not live:
            return [], nid
        outs = list(range(nid, nid + n_out))
        ins = [self.r.choice(live) for _ in range(n_in)]
        counters[0] += fi
        counters[1] += fw
        out_list = [(outs[0], outs[-1])] if n_out > 1 and self.r.random() < 0.5 else outs
        live.extend(outs)
        return [('call', name, out_list, ins)], nid + n_out

    def anoncall_gate(self, live, nid, counters):
        n_out = self.r.randrange(1, 3)
        n_in = self.r.randrange(0, min(3, len(live) + 1))
        ins = [self.r.choice(live) for _ in range(n_in)]
        inner_live = list(range(n_out, n_out + n_in))
        c = [0, 0]
        body, inner_n = self.simple_ops(inner_live, n_out + n_in, self.r.randrange(1, 5), c)
        for o in range(n_out):
            body.append(('copy', o, self.r.choice(inner_live)) if inner_live else ('constant', o, self.const()))
        outs = list(range(nid, nid + n_out))
        counters[0] += c[0]
        counters[1] += c[1]
        live.extend(outs)
        return [('anoncall', outs, ins, c[0], c[1], body)], nid + n_out

    def for_gate(self, live, nid, counters):
        iters = self.r.randrange(1, 5)
        first = self.r.randrange(0, 3)
        base = nid
        if self.functions and self.r.random() < 0.5:
            cands = [f for f in self.functions if f[1] == 1]
            if not cands or not live:
                return [], nid
            name, n_out, n_in, fi, fw, _ = self.r.choice(cands)
            src = self.r.choice(live)
            # out = base + (i - first); inputs: constant wire src
            outs = [('sub', ('add', ('name', 'it'), ('const', base)), ('const', first))]
            ins = [('const', src) for _ in range(n_in)]
            g = ('for', 'it', first, first + iters - 1, [(base, base + iters - 1)] if iters > 1 else [base],
                 ('call', name, outs, ins))
            counters[0] += fi * iters
            counters[1] += fw * iters
        else:
            c = [0, 0]
            n_in = 1 if live else 0
            inner_live = [1] if n_in else []
            body, _ = self.simple_ops(inner_live, 2, self.r.randrange(1, 4), c)
            body.append(('copy', 0, self.r.choice(inner_live)) if inner_live else ('constant', 0, self.const()))
            outs = [('add', ('mul', ('name', 'it'), ('const', 1)), ('const', base - first))]
            # chain: iteration k reads the previous output (or a live wire for the first one)
            if n_in:
                src0 = self.r.choice(live)
                if iters > 1 and base - first - 1 >= 0:
                    # input wire = base + (it - first) - 1 for it > first is not expressible for the first
                    # iteration, so feed a fixed live wire instead
                    ins = [('const', src0)]
                else:
                    ins = [('const', src0)]
            else:
                ins = []
            g = ('for', 'it', first, first + iters - 1, [(base, base + iters - 1)] if iters > 1 else [base],
                 ('anon', outs, ins, c[0], c[1], body))
            counters[0] += c[0] * iters
            counters[1] += c[1] * iters
        live.extend(range(base, base + iters))
        return [g], nid + iters

    def switch_gate(self, live, nid, counters, depth=0):
        if not live:
            return [], nid
        cond = self.r.choice(live)
        n_out = self.r.randrange(1, 3)
        n_cases = self.r.randrange(1, 4)
        if self.boolean:
            n_cases = min(n_cases, 2)
            cases = [bytes([c]) for c in self.r.sample([0, 1], n_cases)]
        else:
            cases = [sw.int_to_le(c) for c in self.r.sample(range(0, min(6, self.p)), min(n_cases, self.p))]
            n_cases = len(cases)
        outs = list(range(nid, nid + n_out))
        branches = []
        max_i = max_w = 0
        for _ in range(n_cases):
            fcands = [f for f in self.functions if f[1] == n_out]
            if fcands and self.r.random() < 0.4 and (live or all(f[2] == 0 for f in fcands)):
                name, _, n_in, fi, fw, _ = self.r.choice(fcands)
                ins = [self.r.choice(live) for _ in range(n_in)]
                branches.append(('call', name, ins))
                max_i, max_w = max(max_i, fi), max(max_w, fw)
            else:
                n_in = self.r.randrange(0, min(3, len(live) + 1))
                ins = self.r.sample(live, n_in) if n_in <= len(set(live)) else []
                ins = list(dict.fromkeys(ins))
                n_in = len(ins)
                inner_live = list(range(n_out, n_out + n_in))
                c = [0, 0]
                body, inner_n = self.simple_ops(inner_live, n_out + n_in, self.r.randrange(1, 5), c, assert_prob=0.3)
                if depth == 0 and inner_live and self.r.random() < 0.25:
                    g, inner_n = self.switch_gate(inner_live, inner_n, c, depth=1)
                    body += g
                for o in range(n_out):
                    body.append(('copy', o, self.r.choice(inner_live)) if inner_live
                                else ('constant', o, self.const()))
                branches.append(('anon', ins, c[0], c[1], body))
                max_i, max_w = max(max_i, c[0]), max(max_w, c[1])
        counters[0] += max_i
        counters[1] += max_w
        live.extend(outs)
        return [('switch', cond, outs, cases, branches)], nid + n_out

    # ---- whole relation --------------------------------------------------------
    def relation(self, n_top=12):
        for _ in range(self.r.randrange(0, 3)):
            self.make_function()
        live, nid = [], 0
        counters = [0, 0]
        gates = []
        g, nid = self.simple_ops(live, nid, 4, counters, assert_prob=0.0)
        gates += g
        for _ in range(n_top):
            k = self.r.choice(['simple', 'simple', 'call', 'anon', 'for', 'switch', 'free'])
            if k == 'simple':
                g, nid = self.simple_ops(live, nid, self.r.randrange(1, 4), counters)
            elif k == 'call' and self.functions:
                g, nid = self.call_gate(live, nid, counters)
            elif k == 'anon':
                g, nid = self.anoncall_gate(live, nid, counters)
            elif k == 'for':
                g, nid = self.for_gate(live, nid, counters)
            elif k == 'switch' and self.switches:
                g, nid = self.switch_gate(live, nid, counters)
            elif k == 'free' and len(live) > 3:
                w = self.r.choice(live)
                live[:] = [x for x in live if x != w]
                g = [('free', w, None)]
            else:
                g = []
            gates += g
        self.n_inst, self.n_wit = counters
        mod_le = sw.int_to_le(self.p)
        gateset = 'boolean' if self.boolean else 'arithmetic'
        self.spec = {'type': 'relation', 'mod': mod_le, 'gateset': gateset, 'features': '@function,@for,@switch',
                     'functions': self.functions, 'gates': gates}
        return sw.write_relation(mod_le, gateset, '@function,@for,@switch', self.functions, gates), mod_le

    def lane_inputs(self, lanes, seed):
        r = random.Random(seed)
        small = [0, 1, 2, 3, 4, 5]
        rows_i, rows_w = [], []
        for _ in range(lanes):
            def val():
                if self.boolean and self.p == 2:
                    return r.randrange(2)
                if self.boolean:  # boolean gateset over an odd field: the gates are integer bit operations
                    return r.choice(small) % self.p if r.random() < 0.3 else r.randrange(self.p)
                return r.choice(small) % self.p if r.random() < 0.6 else r.randrange(self.p)
            rows_i.append([val() for _ in range(self.n_inst)])
            rows_w.append([val() for _ in range(self.n_wit)])
        return rows_i, rows_w
