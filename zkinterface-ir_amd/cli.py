"""`zki_sieve evaluate <paths...>` on the GPU path (rust/src/cli.rs:130,315-320,557-571): same file
discovery and ordering, same verdict text on stderr, non-zero exit on violations.

  python -m zkinterface_ir_amd.cli evaluate <workspace dir or .sieve files ...>
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


def main(argv=None, err=sys.stderr):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2 or argv[0] != 'evaluate':
        print('usage: cli.py evaluate <workspace dir | file.sieve ...>', file=err)
        return 2
    import zkinterface_ir_amd as zk
    violations = zk.evaluate(argv[1:])
    msg = print_violations(violations, err=err)
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
