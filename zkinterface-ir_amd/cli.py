"""`zki_sieve evaluate | validate | metrics | valid-eval-metrics <paths...>` on the GPU path
(rust/src/cli.rs:130,302-363,557-571): same file discovery and ordering, same verdict text on stderr,
Stats JSON on stdout, non-zero exit on violations.

  python -m zkinterface_ir_amd.cli valid-eval-metrics <workspace dir or .sieve files ...>
(run through `python zkinterface-ir_amd/cli.py ...` or after __graft_entry__.load_package())."""
import sys


def print_violations(errors, which_statement='The statement', what='TRUE', err=sys.stderr):
    """cli.rs:557-571"""
    print(file=err)
    if errors:
        print('%s is NOT %s!' % (which_statement, what), file=err)
        print('Violations:\n- %s\n' % '\n- '.join(errors), file=err)
        return 'Found %d violations.' % len(errors)
    print('%s is %s!' % (which_statement, what), file=err)
    return None


def synth(kind, out_dir, lane=0, corrupt=False, err=sys.stderr):
    """Write one statement of a benchmark workload as a workspace (000_instance / 001_witness /
    002_relation .sieve, the naming of producers/sink.rs:84-100) that the original `zki_sieve evaluate`
    can consume elsewhere -- the external cross-check SURVEY.md 7 H8 asks for.  The expected outputs of
    the statement come from a GPU replay of the same relation."""
    import os
    import numpy as np
    import zkinterface_ir_amd as zk
    from zkinterface_ir_amd import workloads
    from zkinterface_ir_amd.sieve_writer import write_instance, write_witness
    wl = workloads.ArithLayered() if kind == 'c2' else workloads.BoolLayered()
    inst, wit = wl.inputs(1, lane)
    probe = zk.Evaluator()
    probe.declare_inputs(wl.n_instance0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        probe.ingest_message(m)
    probe.finalize()
    probe.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), 1)
    probe.replay()
    probe.synchronize()
    outs = [probe.get(w, 1)[0] for w in wl.output_wire_ids()]
    if kind == 'c2':
        o = np.frombuffer(b''.join(v.to_bytes(wl.width, 'little') for v in outs), dtype=np.uint8).reshape(1, wl.n_out, wl.width)
    else:
        o = np.array(outs, dtype=np.uint8).reshape(1, wl.n_out)
    wl.set_expected_outputs(inst, o, lane_offset=0, corrupt_every=1 if corrupt else 0)
    os.makedirs(out_dir, exist_ok=True)
    w = wl.width
    with open(os.path.join(out_dir, '000_instance.sieve'), 'wb') as f:
        f.write(write_instance(wl.mod_le, [inst[0, k].tobytes()[:w] for k in range(wl.n_instance)]))
    with open(os.path.join(out_dir, '001_witness.sieve'), 'wb') as f:
        f.write(write_witness(wl.mod_le, [wit[0, k].tobytes()[:w] for k in range(wl.n_witness)]))
    with open(os.path.join(out_dir, '002_relation.sieve'), 'wb') as f:
        for m in wl.relation_messages():
            f.write(m)
    print('wrote %s workspace (lane %d, %s) to %s' % (kind, lane, 'FALSE' if corrupt else 'TRUE', out_dir), file=err)
    return 0


def main(argv=None, err=sys.stderr, out=sys.stdout):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) >= 3 and argv[0] == 'synth' and argv[1] in ('c2', 'c4'):
        return synth(argv[1], argv[2], lane=int(argv[3]) if len(argv) > 3 else 0, corrupt='--incorrect' in argv, err=err)
    if len(argv) >= 4 and argv[0] in ('flatten', 'expand-definable') and '--out' in argv:  # cli.rs:442-472,515-555
        rest = argv[1:]
        gate_mask = None
        if '--gate-set' in rest:
            g = rest.index('--gate-set')
            from zkinterface_ir_amd.flatten import parse_gate_set
            try:
                gate_mask = parse_gate_set(rest[g + 1])
            except ValueError as e:
                print('Error: %s' % e, file=err)
                return 1
            rest = rest[:g] + rest[g + 2:]
        if argv[0] == 'expand-definable' and gate_mask is None:
            return 0  # cli.rs:520: without --gate-set the tool does nothing
        k = rest.index('--out')
        out_dir, paths = rest[k + 1], rest[:k] + rest[k + 2:]
        if out_dir.endswith('.sieve'):
            print('Error: IR flattening requires a directory as output value', file=err)
            return 1
        from zkinterface_ir_amd.flatten import flatten_workspace
        try:
            flatten_workspace(paths, out_dir, gate_mask if argv[0] == 'expand-definable' else None)
        except (ValueError, Exception) as e:  # noqa: B014 -- any recording error is reported like the reference's Err
            print('Error: %s' % e, file=err)
            return 1
        return 0
    tools = ('evaluate', 'validate', 'metrics', 'valid-eval-metrics')
    if len(argv) < 2 or argv[0] not in tools:
        print('usage: cli.py evaluate|validate|metrics|valid-eval-metrics <workspace dir | file.sieve ...>\n'
              '       cli.py flatten <workspace dir | file.sieve ...> --out <dir>\n'
              '       cli.py expand-definable <workspace dir | file.sieve ...> --gate-set "@add,@mul,..." --out <dir>\n'
              '       cli.py synth c2|c4 <out dir> [lane] [--incorrect]', file=err)
        return 2
    import zkinterface_ir_amd as zk
    tool, paths = argv[0], argv[1:]
    failures = []
    try:
        if tool == 'evaluate':
            failures.append(print_violations(zk.evaluate(paths), err=err))
        elif tool == 'validate':  # cli.rs:302-313
            failures.append(print_violations(zk.validate(paths), what='COMPLIANT with the specification', err=err))
        elif tool == 'metrics':   # cli.rs:322-330
            print(zk.metrics(paths), file=out)
        else:                     # cli.rs:333-363: three reports, then the first failure decides the exit
            valid, evald, stats = zk.valid_eval_metrics(paths)
            failures.append(print_violations(valid, what='COMPLIANT with the specification', err=err))
            failures.append(print_violations(evald, err=err))
            print(stats, file=out)
    except zk.ZkGpuError as e:
        failures.append(str(e))
    for msg in failures:
        if msg:
            print('Error: %s' % msg, file=err)
            return 1
    return 0


if __name__ == '__main__':
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__
    __graft_entry__.load_package()
    sys.exit(main())
