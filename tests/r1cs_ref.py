"""Test-side restatement of the reference's IR -> R1CS rules, used to check the product's
zkgpu_r1cs_from_tape.  Source: /root/reference/rust/src/consumers/to_r1cs.rs
  :117      variable 0 is the constant one
  :143-145  copy      -> same variable, no constraint
  :147-153  constant  -> new (instance) variable
  :155-161  assert_zero(w): (w)*(one) = (0*one)
  :163-211  add:       (out [+ corr*p])*(one) = (a + b)
  :213-260  multiply:  (a)*(b) = (out [+ corr*p])
  :262-312  add_constant: (out [+ corr*p])*(one) = (a + c*one)
  :314-359  mul_constant: (c*a)*(one) = (out [+ corr*p])
  :361-371  and = multiply, xor = add, not = add_constant(a, 1)
  :373-392  instance / witness -> new variable
Variable ids: zkinterface's StatementBuilder hands them out sequentially from 1 (crate not under
/root/reference: the numbering is an assumption stated in DESIGN.md, "parity unpinned")."""

K = {'add': 1, 'mul': 2, 'addc': 3, 'mulc': 4, 'copy': 5, 'constant': 6, 'instance': 7, 'witness': 8,
     'assert_zero': 9, 'and': 10, 'xor': 11, 'not': 12}


def rows_from_tape(kinds, a, b, constants, p, use_correction=False):
    """returns (rows, var_of_op): rows = [(A, B, C)], each a list of (variable, coefficient)"""
    ONE = 0
    next_var = [1]

    def new_var():
        v = next_var[0]
        next_var[0] += 1
        return v
    var_of = [None] * len(kinds)
    rows = []
    for i, k in enumerate(kinds):
        k, x, y = int(k), int(a[i]), int(b[i])
        if k == K['copy']:
            var_of[i] = var_of[x]
        elif k in (K['constant'], K['instance'], K['witness']):
            var_of[i] = new_var()
        elif k == K['assert_zero']:
            rows.append(([(var_of[x], 1)], [(ONE, 1)], [(ONE, 0)]))
        else:
            out = new_var()
            corr = new_var() if use_correction else None
            var_of[i] = out
            outs = [(out, 1)] + ([(corr, p)] if use_correction else [])
            if k in (K['add'], K['xor']):
                rows.append((outs, [(ONE, 1)], [(var_of[x], 1), (var_of[y], 1)]))
            elif k == K['addc']:
                rows.append((outs, [(ONE, 1)], [(var_of[x], 1), (ONE, int.from_bytes(constants[y], 'little'))]))
            elif k == K['not']:
                rows.append((outs, [(ONE, 1)], [(var_of[x], 1), (ONE, 1)]))
            elif k in (K['mul'], K['and']):
                rows.append(([(var_of[x], 1)], [(var_of[y], 1)], outs))
            elif k == K['mulc']:
                rows.append(([(var_of[x], int.from_bytes(constants[y], 'little'))], [(ONE, 1)], outs))
            else:
                raise ValueError(k)
    return rows, var_of


def lincomb(terms, w):
    return sum(c * w[v] for v, c in terms)
