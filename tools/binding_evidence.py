#!/usr/bin/env python3
"""Writes profiles/binding_<workload>.json -- which resource binds the dominant kernel of a workload and the counters
that say so -- from the files tools/collect_profiles.sh collected (copied into profiles/ first):

  python tools/binding_evidence.py [profiles dir] [tag]

bench.py reads these files into `roofline.binding` / `roofline.binding_evidence`."""
import json
import os
import sys

HBM_PEAK = 8.0e12


def load(d, name):
    try:
        return json.load(open(os.path.join(d, name)))
    except (OSError, ValueError):
        return None


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
    tag = sys.argv[2] if len(sys.argv) > 2 else 'r02'
    # ---- C2
    bench = load(d, '%s_bench_c2.json' % tag)
    tr = load(d, 'pmc_traffic_latest.json')
    sq = load(d, '%s_pmc_c2_sq_counters.json' % tag)
    tcc = load(d, '%s_pmc_c2_tcc_counters.json' % tag)
    hv = load(d, 'pmc_traffic_c2_hbm_variant.json')
    if bench and tr:
        r = bench['roofline']
        launches = r['launches_per_step']
        step_ms = r['avg_launch_ms'] * launches
        per_step = tr['traffic_bytes_per_launch'] * launches
        ev = {'L2_fabric_bytes_per_step': per_step, 'algorithmic_bytes_per_step': r['algorithmic_bytes_per_launch'] * launches,
              'traffic_over_algorithmic': per_step / (r['algorithmic_bytes_per_launch'] * launches),
              'L2_fabric_TB_per_s': per_step / (step_ms * 1e-3) / 1e12, 'wire_table_MB': bench['config']['wire_table_MB'],
              'infinity_cache_MiB': 256,
              'reading': 'FETCH_SIZE x2 + WRITE_SIZE count L2<->fabric requests, Infinity-Cache hits included '
                         '(MI355X_MICROARCH.md, HBM section): with a 263 MB table these bytes are served on-die, not by HBM',
              'sources': ['profiles/pmc_traffic_latest.json']}
        if tcc and tcc.get('TCC_HIT_sum') is not None:
            ev['L2_hit_rate'] = tcc['TCC_HIT_sum'] / max(tcc['TCC_HIT_sum'] + tcc['TCC_MISS_sum'], 1)
            ev['sources'].append('profiles/%s_pmc_c2_tcc_counters.json' % tag)
        if sq:
            wc = sq.get('SQ_WAVE_CYCLES') or 1
            ev['waves_parked_on_memory_frac'] = sq.get('SQ_WAIT_ANY', 0) / wc
            ev['valu_insts_per_wave'] = sq.get('SQ_INSTS_VALU', 0) / max(sq.get('SQ_WAVES', 1), 1)
            # the second resource: the VALU pipe at this chip's measured rates (r01_valu_rates.txt).  A wave = one program
            # entry x 64 witnesses; half of the relation's gates are multiplications of 136 v_mad_u64_u32 + v_addc pairs
            cfg = bench['config']
            entries, gates = cfg.get('program_entries'), cfg.get('backend_ops_per_witness')
            if entries and gates:
                mads = 136.0 * 0.5 * (1 << 20) / entries
                other = max(ev['valu_insts_per_wave'] - 2 * mads, 0)
                ev['valu_pipe_ms'] = (mads / 26.6 + mads / 67.9 + other / 35.0) * entries * 1024 / 1e12 * 1e3
                ev['valu_pipe_busy_frac'] = ev['valu_pipe_ms'] / bench['ms_per_step']
            ev['sources'].append('profiles/%s_pmc_c2_sq_counters.json' % tag)
        if hv and r.get('hbm_variant'):
            h = r['hbm_variant']
            hl = launches
            ev['hbm_variant'] = {'batch': h['batch'], 'wire_table_MB': h['wire_table_MB'], 'ms_per_step': h['ms_per_step'],
                                 'HBM_bytes_per_step': hv['traffic_bytes_per_launch'] * hl,
                                 'HBM_TB_per_s': hv['traffic_bytes_per_launch'] * hl / (h['ms_per_step'] * 1e-3) / 1e12,
                                 'frac_of_8TBs': hv['traffic_bytes_per_launch'] * hl / (h['ms_per_step'] * 1e-3) / HBM_PEAK,
                                 'source': 'profiles/pmc_traffic_c2_hbm_variant.json'}
        json.dump({'workload': 'c2', 'kernel': tr['kernel'], 'binding': 'fabric+infinity-cache', 'evidence': ev},
                  open(os.path.join(d, 'binding_c2.json'), 'w'), indent=1)
    # ---- C4
    bench = load(d, '%s_bench_c4.json' % tag)
    lds = load(d, '%s_pmc_c4_lds_counters.json' % tag)
    lds0 = load(d, '%s_pmc_c4_lds_counters_bank_unaware.json' % tag)
    ident = load(d, '%s_bench_c4_identity_wiring.json' % tag)
    if bench and lds:
        prog_bytes = 6.0 * bench['config']['backend_ops_per_witness']   # rows: three u16 per op (generic entries, 8 B, are a sliver)
        wgs = 4096 // 32
        ms = bench['ms_per_step']

        def ratio(c):
            return c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1)
        ev = {'lds_bank_conflict_cycles_over_lds_active_cycles': ratio(lds),
              'lds_insts_per_replay': lds.get('SQ_INSTS_LDS'), 'lds_active_cycles': lds.get('SQ_LDS_IDX_ACTIVE'),
              'lds_bank_conflict_cycles': lds.get('SQ_LDS_BANK_CONFLICT'),
              'program_stream_GB_per_s_per_CU': prog_bytes / (ms * 1e-3) / 1e9,
              'program_bytes_per_workgroup': prog_bytes, 'workgroups': wgs,
              'waves_parked_frac': lds.get('SQ_WAIT_ANY', 0) / max(lds.get('SQ_WAVE_CYCLES', 1), 1),
              'lds_issue_stall_frac': lds.get('SQ_WAIT_INST_LDS', 0) / max(lds.get('SQ_WAVE_CYCLES', 1), 1),
              'valu_insts_per_gate_and_wave': lds.get('SQ_INSTS_VALU', 0) / max(lds.get('SQ_INSTS_LDS', 1) / 3.0, 1),
              'reading': 'one workgroup = one CU walks all 10.5 M ops (128 of 256 CUs at batch 4096).  The kernel is bound by '
                         'instruction issue: a SIMD hands out one issue slot every four cycles, the 16 waves of the workgroup '
                         '(4 per SIMD) fill them, and a wave spends ~17 instructions per row of 2 gates per lane (8 VALU: six '
                         'address shifts + two gates; 6 LDS; 1 VMEM; the waits).  issue_bound_ms = counted instructions per wave '
                         '(waits, branches and barriers are not in these counters: about a tenth more) x 4 waves x 4 cycles at '
                         '2.4 GHz: three quarters of the kernel time, the rest is the drain -> barrier -> refill of the 644 '
                         'levels.  At every step of the rebuild the kernel time followed the instruction count, not the LDS '
                         'traffic (unchanged), the conflicts (0.13 of the LDS cycles, 0.67 before the bank-aware schedule) or the '
                         'program stream (63 MB per workgroup from L2)',
              'sources': ['profiles/%s_pmc_c4_lds_counters.json' % tag]}
        sq4 = load(d, '%s_pmc_c4_sq_counters.json' % tag)
        if sq4 and sq4.get('SQ_WAVES'):
            per_wave = sum(sq4.get(k, 0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SMEM')) / sq4['SQ_WAVES']
            ev['instructions_per_wave'] = {k: sq4.get(k, 0) / sq4['SQ_WAVES'] for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SMEM')}
            ev['issue_bound_ms'] = per_wave * 4 * 4 / 2.4e9 * 1e3   # 4 waves per SIMD, one slot per SIMD every 4 cycles, 2.4 GHz
            ev['sources'].append('profiles/%s_pmc_c4_sq_counters.json' % tag)
        if lds0:
            ev['before_bank_aware_schedule'] = {'lds_bank_conflict_cycles_over_lds_active_cycles': ratio(lds0),
                                                'lds_active_cycles': lds0.get('SQ_LDS_IDX_ACTIVE'),
                                                'lds_bank_conflict_cycles': lds0.get('SQ_LDS_BANK_CONFLICT'),
                                                'kernel_ms_under_pmc': lds0.get('avg_ns', 0) / 1e6,
                                                'source': 'profiles/%s_pmc_c4_lds_counters_bank_unaware.json' % tag}
            ev['kernel_ms_under_pmc'] = lds.get('avg_ns', 0) / 1e6
        if ident:
            ev['conflict_free_wiring_same_program_size_ms'] = ident['ms_per_step']
        json.dump({'workload': 'c4', 'kernel': 'bool_lds_kernel', 'binding': 'instruction issue (one slot per SIMD every four cycles)', 'evidence': ev},
                  open(os.path.join(d, 'binding_c4.json'), 'w'), indent=1)
    # ---- C5
    bench = load(d, '%s_bench_c5.json' % tag)
    sq = load(d, '%s_pmc_c5_sq_counters.json' % tag)
    tr = load(d, 'pmc_traffic_c5.json')
    if bench and sq:
        waves = max(sq.get('SQ_WAVES', 1), 1)
        ev = {'valu_insts_per_row_and_wave': sq['SQ_INSTS_VALU'] / waves, 'salu_insts_per_row_and_wave': sq.get('SQ_INSTS_SALU', 0) / waves,
              'word_products_per_row': 664, 'v_mad_u64_u32_plus_addc_per_row': 1328,
              'waves_parked_on_memory_frac': sq.get('SQ_WAIT_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1),
              'round1_valu_insts_per_row_and_wave': 2270,
              'reading': 'at the instruction rates measured on this chip (profiles/r01_valu_rates.txt: v_mad_u64_u32 26.6 T lane-ops/s, '
                         'add-with-carry 68 T, other 32-bit integer ops about 35 T) the 664 + 664 + ~190 VALU instructions of a row '
                         'are valu_pipe_ms of VALU pipe per check over 1024 witnesses: the kernel time is that pipe busy '
                         'valu_pipe_busy_frac of the time.  The HBM floor of the 240.5 GB of gathers is 38 ms at 6.3 TB/s',
              'sources': ['profiles/%s_pmc_c5_sq_counters.json' % tag]}
        other = max(ev['valu_insts_per_row_and_wave'] - 1328, 0)
        units = (664 / 26.6 + 664 / 67.9 + other / 35.0) * (bench['config'].get('rows', 1 << 20) * 1024 / 1e12)   # seconds
        ev['valu_pipe_ms'] = units * 1e3
        ev['valu_pipe_busy_frac'] = units * 1e3 / bench['ms_per_step']
        if tr:
            ev['traffic_over_algorithmic'] = tr['traffic_bytes_per_launch'] / bench['roofline']['algorithmic_bytes_per_launch']
            ev['sources'].append('profiles/pmc_traffic_c5.json')
        json.dump({'workload': 'c5', 'kernel': 'r1cs_row_kernel<8, false>', 'binding': 'valu', 'evidence': ev},
                  open(os.path.join(d, 'binding_c5.json'), 'w'), indent=1)
    print('written:', [f for f in sorted(os.listdir(d)) if f.startswith('binding_')])


if __name__ == '__main__':
    main()
