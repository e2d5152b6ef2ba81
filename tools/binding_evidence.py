#!/usr/bin/env python3
"""Writes profiles/binding_<workload>.json (c2, c2_hbm, c4, c5) -- which resource binds the dominant kernel of a
workload and the COUNTER-DERIVED, TIME-INDEPENDENT facts that say so -- from the files tools/collect_profiles.sh
collected (copied into profiles/ first):

  python tools/binding_evidence.py [profiles dir] [tag]

bench.py reads these files (make_roofline): it takes byte and instruction counts from here and every time from its own
run, so a rate or fraction is never copied from a profile.  `program` = the program the counters were collected on;
bench.py only uses a file whose program is the one that ran."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# sustained issue rates on this chip, T lane-ops/s (profiles/r01_valu_rates.txt, tools/valu_rates.hip)
RATE_MAD, RATE_ADDC, RATE_OTHER = 26.6, 67.9, 35.0


def load(d, name):
    try:
        return json.load(open(os.path.join(d, name)))
    except (OSError, ValueError):
        return None


def dump(d, name, obj):
    # the kernel / scheduler sources the counters were collected on: run this script on the tree that was profiled
    # (tools/collect_profiles.sh does, on the GPU box, right behind the --pmc passes)
    obj['device_source_digest'] = entry.device_source_digest()
    json.dump(obj, open(os.path.join(d, name), 'w'), indent=1)


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
    tag = sys.argv[2] if len(sys.argv) > 2 else 'r04'
    # ---- C2 (headline batch: the wire table sits in the Infinity Cache) and its 4096-lane variant (HBM)
    bench = load(d, '%s_bench_c2.json' % tag)
    tr = load(d, 'pmc_traffic_latest.json')
    sq = load(d, '%s_pmc_c2_sq_counters.json' % tag)
    tcc = load(d, '%s_pmc_c2_tcc_counters.json' % tag)
    hv = load(d, 'pmc_traffic_c2_hbm_variant.json')
    if bench and tr:
        cfg = bench['config']
        entries, launches, batch = cfg['program_entries'], cfg['launches_per_step'], cfg.get('batch_per_gpu', 1024)
        wide = bench['roofline']['launches_per_step']
        c = {'memory_side': 'fabric+infinity-cache', 'traffic_bytes_per_launch': tr['traffic_bytes_per_launch'],
             'traffic_launches_per_step': wide, 'traffic_source': 'profiles/pmc_traffic_latest.json',
             'wire_table_MB': cfg['wire_table_MB'], 'infinity_cache_MiB': 256}
        sources = ['profiles/pmc_traffic_latest.json']
        if tcc and tcc.get('TCC_HIT_sum') is not None:
            c['L2_hit_rate'] = tcc['TCC_HIT_sum'] / max(tcc['TCC_HIT_sum'] + tcc['TCC_MISS_sum'], 1)
            sources.append('profiles/%s_pmc_c2_tcc_counters.json' % tag)
        valu_per_wave = None
        if sq:
            wc = sq.get('SQ_WAVE_CYCLES') or 1
            c['waves_parked_on_memory_frac'] = sq.get('SQ_WAIT_ANY', 0) / wc
            valu_per_wave = sq.get('SQ_INSTS_VALU', 0) / max(sq.get('SQ_WAVES', 1), 1)
            c['valu_insts_per_wave'] = valu_per_wave
            # the second resource: the VALU pipe at this chip's measured rates.  A wave = one program entry x 64 witnesses;
            # half of the relation's 2^20 gates are multiplications of 136 v_mad_u64_u32 + v_addc_co_u32 pairs
            mads = 136.0 * 0.5 * (1 << 20) / entries
            other = max(valu_per_wave - 2 * mads, 0)
            c['valu_pipe_ms_per_step'] = (mads / RATE_MAD + mads / RATE_ADDC + other / RATE_OTHER) * entries * batch / 1e12 * 1e3
            c['valu_insts_per_wave_outside_the_multiplier'] = other
            sources.append('profiles/%s_pmc_c2_sq_counters.json' % tag)
        dump(d, 'binding_c2.json', {
            'workload': 'c2', 'kernel': tr['kernel'], 'binding': 'fabric+infinity-cache',
            'program': {'entries': entries, 'launches': launches, 'batch': batch, 'lane_group': 0}, 'constants': c,
            'reading': 'FETCH_SIZE x2 + WRITE_SIZE count L2<->fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md, '
                       'HBM section): with a 263 MB table these bytes are served on-die, not by HBM; the guide puts random '
                       'gathers from a 151 MB table at 7.4-7.9 TB/s.  Fusion and pair entries keep about 38 % of the gate values '
                       'in registers and L2 serves the stated share of the requests, so fewer bytes move than SURVEY 8(d) counts',
            'sources': sources, 'collected': {'tag': tag, 'ms_per_step_at_collection': bench['ms_per_step']}})
        if hv:
            ch = {'memory_side': 'hbm', 'traffic_bytes_per_launch': hv['traffic_bytes_per_launch'], 'traffic_launches_per_step': wide,
                  'traffic_source': 'profiles/pmc_traffic_c2_hbm_variant.json'}
            if 'valu_pipe_ms_per_step' in c:
                ch['valu_pipe_ms_per_step'] = c['valu_pipe_ms_per_step'] * 4096 / batch
                ch['valu_insts_per_wave'] = valu_per_wave
            dump(d, 'binding_c2_hbm.json', {
                'workload': 'c2_hbm', 'kernel': tr['kernel'], 'binding': 'hbm',
                'program': {'entries': entries, 'launches': launches, 'batch': 4096, 'lane_group': 4096}, 'constants': ch,
                'reading': 'the same program with 4096 witnesses in flight: a 1.05 GB wire table, 4x the Infinity Cache, so the '
                           'L2<->fabric bytes are HBM bytes (a float4 copy reaches 6.3 TB/s on this chip)',
                'sources': ['profiles/pmc_traffic_c2_hbm_variant.json'], 'collected': {'tag': tag}})
    # ---- C4
    bench = load(d, '%s_bench_c4.json' % tag)
    lds = load(d, '%s_pmc_c4_lds_counters.json' % tag)
    lds0 = load(d, '%s_pmc_c4_lds_counters_bank_unaware.json' % tag)
    sq4 = load(d, '%s_pmc_c4_sq_counters.json' % tag)
    tr4 = load(d, 'pmc_traffic_c4.json')
    if bench and lds:
        cfg = bench['config']

        def ratio(x):
            return x['SQ_LDS_BANK_CONFLICT'] / max(x['SQ_LDS_IDX_ACTIVE'], 1)
        c = {'lds_bank_conflict_cycles_over_lds_active_cycles': ratio(lds), 'lds_insts_per_replay': lds.get('SQ_INSTS_LDS'),
             'program_bytes_per_workgroup': 6.0 * cfg['backend_ops_per_witness'],   # rows: three u16 per op
             'workgroups': cfg.get('batch_per_gpu', 4096) // 32,
             'waves_parked_frac': lds.get('SQ_WAIT_ANY', 0) / max(lds.get('SQ_WAVE_CYCLES', 1), 1)}
        # the second candidate: the CU's LDS array (SQ_LDS_IDX_ACTIVE = its busy cycles, summed over the CUs that run a workgroup)
        c['lds_array_cycles_per_step'] = lds.get('SQ_LDS_IDX_ACTIVE', 0)   # all LDS-array cycles of one replay, summed over the CUs
        sources = ['profiles/%s_pmc_c4_lds_counters.json' % tag]
        if tr4:
            c.update({'memory_side': 'fabric', 'traffic_bytes_per_launch': tr4['traffic_bytes_per_launch'], 'traffic_launches_per_step': 1,
                      'traffic_source': 'profiles/pmc_traffic_c4.json'})
            sources.append('profiles/pmc_traffic_c4.json')
        if sq4 and sq4.get('SQ_WAVES'):
            keys = ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SMEM')
            per_wave = sum(sq4.get(k, 0) for k in keys) / sq4['SQ_WAVES']
            c['insts_per_wave'] = per_wave
            c['instructions_per_wave'] = {k: sq4.get(k, 0) / sq4['SQ_WAVES'] for k in keys}
            sources.append('profiles/%s_pmc_c4_sq_counters.json' % tag)
        if lds0:
            c['lds_bank_conflict_ratio_before_the_bank_aware_schedule'] = ratio(lds0)
        dump(d, 'binding_c4.json', {
            'workload': 'c4', 'kernel': 'bool_lds_kernel', 'binding': 'lds_array',
            'program': {'entries': cfg['program_entries'], 'launches': cfg['launches_per_step'], 'batch': cfg.get('batch_per_gpu', 4096)},
            'constants': c,
            'reading': 'one workgroup = one CU walks the whole program with its 32-witness slice of the wire table in LDS (batch / 32 of '
                       'the 256 CUs).  No single resource is saturated (profiles/r03_tuning_sweeps.txt, "C4 decomposition": without '
                       'the LDS write, with half the reads or a third of the program bytes the replay takes the same time): the '
                       'hardware figure reported is the LDS array -- SQ_LDS_IDX_ACTIVE cycles against 256 CUs x 2.4 GHz -- and the '
                       'CUs the batch occupies.  The wire traffic never leaves the LDS: the fabric bytes are the program, once per '
                       'XCD, + inputs',
            'sources': sources, 'collected': {'tag': tag, 'ms_per_step_at_collection': bench['ms_per_step']}})
    # ---- C5
    bench = load(d, '%s_bench_c5.json' % tag)
    sq = load(d, '%s_pmc_c5_sq_counters.json' % tag)
    tr = load(d, 'pmc_traffic_c5.json')
    if bench and sq:
        cfg = bench['config']
        rows, batch = cfg.get('rows', (1 << 20) + 1), cfg.get('batch_per_gpu', 1024)
        waves = max(sq.get('SQ_WAVES', 1), 1)
        per_wave = sq['SQ_INSTS_VALU'] / waves
        other = max(per_wave - 1328, 0)
        c = {'valu_insts_per_wave': per_wave, 'salu_insts_per_row_and_wave': sq.get('SQ_INSTS_SALU', 0) / waves,
             'word_products_per_row': 664, 'v_mad_u64_u32_plus_addc_per_row': 1328,
             'waves_parked_on_memory_frac': sq.get('SQ_WAIT_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1),
             'valu_pipe_ms_per_step': (664 / RATE_MAD + 664 / RATE_ADDC + other / RATE_OTHER) * rows * batch / 1e12 * 1e3}
        sources = ['profiles/%s_pmc_c5_sq_counters.json' % tag]
        if tr:
            c.update({'memory_side': 'hbm', 'traffic_bytes_per_launch': tr['traffic_bytes_per_launch'], 'traffic_launches_per_step': 1,
                      'traffic_source': 'profiles/pmc_traffic_c5.json'})
            sources.append('profiles/pmc_traffic_c5.json')
        dump(d, 'binding_c5.json', {
            'workload': 'c5', 'kernel': 'r1cs_row_kernel<8, false, false>', 'binding': 'valu',
            'program': {'entries': rows, 'launches': 1, 'batch': batch}, 'constants': c,
            'reading': 'a row = 664 word products (v_mad_u64_u32 + v_addc_co_u32 each) + the rest; at the instruction rates measured '
                       'on this chip (v_mad_u64_u32 26.6 T lane-ops/s, add-with-carry 68 T, other 32-bit integer ops about 35 T) that '
                       'is valu_pipe_ms_per_step of VALU pipe per check; the gathers (7 x 32 B per row and witness) come from HBM',
            'sources': sources, 'collected': {'tag': tag, 'ms_per_step_at_collection': bench['ms_per_step']}})
    # ---- C5 rows with small coefficients (bench.py --coefs small; the default run's secondary.c5_small)
    bench = load(d, '%s_bench_c5_small.json' % tag)
    sq = load(d, '%s_pmc_c5_small_sq_counters.json' % tag)
    tr = load(d, 'pmc_traffic_c5_small.json')
    if bench and tr:
        cfg = bench['config']
        rows, batch = cfg.get('rows', (1 << 20) + 1), cfg.get('batch_per_gpu', 1024)
        c = {'memory_side': 'hbm', 'traffic_bytes_per_launch': tr['traffic_bytes_per_launch'], 'traffic_launches_per_step': 1,
             'traffic_source': 'profiles/pmc_traffic_c5_small.json'}
        sources = ['profiles/pmc_traffic_c5_small.json']
        if sq:
            waves = max(sq.get('SQ_WAVES', 1), 1)
            c.update({'valu_insts_per_wave': sq['SQ_INSTS_VALU'] / waves,
                      'waves_parked_on_memory_frac': sq.get('SQ_WAIT_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1)})
            sources.append('profiles/%s_pmc_c5_small_sq_counters.json' % tag)
        dump(d, 'binding_c5_small.json', {
            'workload': 'c5_small', 'kernel': 'r1cs_row_kernel<8, false, true>', 'binding': 'hbm',
            'program': {'entries': rows, 'launches': 1, 'batch': batch}, 'constants': c,
            'reading': 'the rows of C5 with coefficients 1 / -1 / 16-bit signed integers: additions, or N word products per term and '
                       'two word rounds per combination, and one product per row -- what is left is the gathers (7 x 32 B per row '
                       'and witness)',
            'sources': sources, 'collected': {'tag': tag, 'ms_per_step_at_collection': bench['ms_per_step']}})
    print('written:', [f for f in sorted(os.listdir(d)) if f.startswith('binding_')])


if __name__ == '__main__':
    main()
