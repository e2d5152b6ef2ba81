"""R1CS -> SIEVE IR: the gate expansion of FromR1CSConverter (rust/src/producers/from_r1cs.rs:16-156) on top
of builder.GateBuilder.  The zkinterface crate that defines the reference's input types is not part of the
reference tree, so the R1CS side is stated with plain Python data:

  field_maximum : int, p - 1  (zkiCircuitHeader.field_maximum; the IR characteristic is field_maximum + 1, :146-156)
  instance_variables : [(id, value bytes)]   id 0 is the constant one and must carry value 1 (:49-59)
  witness_ids   : [id]                        zki_header.list_witness_ids() (:62-65)
  constraints   : [(A, B, C)] with each linear combination = [(id, coefficient bytes)]  (:115-129)
  witness values: [(id, value bytes)] in the order they are to be pushed (:131-140)

Wire numbering follows the reference exactly: wire 0 = constant 1, wire 1 = constant -1, then one wire per
instance variable, one per witness id, then per constraint the terms of A, B, C followed by
Mul(a,b), Mul(-1,c), Add and an AssertZero."""
from .builder import ARITH, SIMPLE, BuilderError, GateBuilder, Header


def _le(v):
    return v.to_bytes(max(1, (v.bit_length() + 7) // 8), 'little')


class FromR1CSConverter:
    def __init__(self, sink, field_maximum, instance_variables, witness_ids):
        if field_maximum is None:
            raise BuilderError('field_maximum must be provided')
        header = Header(_le(field_maximum + 1))
        self.b = GateBuilder(sink, header, ARITH, SIMPLE)
        self.r1cs_to_ir_wire = {}
        one = self.b.create_gate(('constant', bytes([1])))
        assert one == 0
        self.r1cs_to_ir_wire[0] = one
        self.minus_one = self.b.create_gate(('constant', _le(field_maximum)))
        for var, value in instance_variables:
            if var == 0:
                assert int.from_bytes(value, 'little') == 1, 'value for instance id:0 should be a constant 1'
            else:
                self.r1cs_to_ir_wire[var] = self.b.create_gate(('instance', bytes(value)))
        for var in witness_ids:
            self.r1cs_to_ir_wire[var] = self.b.create_gate(('witness', None))

    def _build_term(self, var, value):  # :70-93
        value = bytes(value) if len(value) else bytes([0])
        if var == 0:
            return self.b.create_gate(('constant', value))
        val_id = self.b.create_gate(('constant', value))
        if var not in self.r1cs_to_ir_wire:
            raise BuilderError('The WireId %d has not been defined yet.' % var)
        return self.b.create_gate(('mul', self.r1cs_to_ir_wire[var], val_id))

    def _add_lc(self, lc):  # :95-110
        if not lc:
            return self.b.create_gate(('constant', bytes([0])))
        total = self._build_term(*lc[0])
        for term in lc[1:]:
            t = self._build_term(*term)
            total = self.b.create_gate(('add', total, t))
        return total

    def ingest_constraints(self, constraints):  # :112-129
        for a, b, c in constraints:
            sa, sb, sc = self._add_lc(a), self._add_lc(b), self._add_lc(c)
            prod = self.b.create_gate(('mul', sa, sb))
            neg_c = self.b.create_gate(('mul', self.minus_one, sc))
            claim = self.b.create_gate(('add', prod, neg_c))
            self.b.create_gate(('assert_zero', claim))

    def ingest_witness(self, assigned_variables):  # :131-140
        for var, value in assigned_variables:
            if var not in self.r1cs_to_ir_wire:
                raise BuilderError('The ZKI witness id %d does not exist.' % var)
            self.b.push_witness_value(bytes(value))

    def finish(self):
        return self.b.finish()
