"""bench.py's ONE JSON line stays inside what the driver keeps (6 KB) while it carries every workload of the default run:
checked on the committed detail file of the round (profiles/r04_bench_detail.json = everything that run measured)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_compact_line_of_the_committed_run_fits_and_keeps_every_workload():
    import bench
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_detail.json')))
    line = bench.compact_line(full, 'profiles/bench_detail_latest.json')
    text = json.dumps(line)
    assert len(text) <= 6144, len(text)
    assert set(line['secondary']) == {'c4', 'c5', 'c5_small', 'structured', 'structured_chained', 'sha256'}
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in line['roofline'], k
    assert 0 < line['roofline']['frac'] <= 1.0
    for name, w in line['secondary'].items():
        assert w['ms_per_step'] > 0 and w['value'] > 0 and w['roofline']['bound'], name
        if name != 'c5_small':      # (a variant of c5: no CPU sample and no first-verdict session of its own)
            assert w['cpu_baseline']['value'] > 0 and w['first_verdict_s'] and w['break_even_batch'], name
    # the committed line is what compact_line makes of the detail file
    committed = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench_default.json')).read())
    assert committed['secondary'].keys() == line['secondary'].keys() and committed['value'] == line['value']
