#!/usr/bin/env python3
"""Turn the rocprofv3 outputs `tools/profile_round.sh <tag>` left under gpurun_out/prof<tag>/ into the committed
summaries in profiles/.

    python tools/collect_profiles.py <tag> <round>      e.g.  tag "2", round "r02"

Writes, per schedule (fused / materialised) and for the C3 / C5 run (configs):
  profiles/<round>_<what>_kernel_stats.csv         rocprofv3 --kernel-trace --stats summary
  profiles/<round>_<what>_bench_under_rocprof.json the JSON line the profiled run printed
  profiles/<round>_<what>_pmc_<COUNTER>.csv        rows of our kernels from the separate --pmc passes
  profiles/<round>_<what>_sq_counters.json         per-kernel averages of the SQ / GRBM passes + derived ratios
  profiles/traffic.json                            HBM bytes per launch (read by bench.py into roofline.traffic)
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM); counters are
KiB; the first launch of each kernel is dropped.
"""
import collections
import csv
import glob
import json
import pathlib
import re
import shutil
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
# one sig_fused_voice_bus call = steady_prep_kernel (when the constants change) + fused_steady_bus_kernel (or
# fused_walk_kernel) + partials_kernel: their bytes are summed into 'fused_voice_bus'; likewise for the cascade
FAMILY = {'fused_walk_kernel': 'fused_voice_bus', 'fused_steady_bus_kernel': 'fused_voice_bus', 'steady_prep_kernel': 'fused_voice_bus',
          'partials_kernel': 'fused_voice_bus', 'sum_bus_fast_kernel': 'sum_bus', 'sum_bus_kernel': 'sum_bus',
          'osc_bank_kernel': 'osc_bank', 'biquad_coldstart_kernel': 'biquad_coldstart', 'biquad_walk_kernel': 'biquad_coldstart',
          'ew_fast_kernel': 'elementwise', 'fused_scan_kernel': 'fused_scan', 'fused_cascade_kernel': 'fused_cascade_bus',
          'mix_matrix_kernel': 'mix_matrix', 'biquad_bus_kernel': 'biquad_bus', 'fused_steady_mix_kernel': 'fused_osc_biquad_mix',
          'control_program_kernel': 'control_program', 'voice_program_kernel': 'voice_program', 'sig_vp_specialised': 'voice_program_specialised',
          'sig_ctl_specialised': 'control_program_specialised'}
SUMMED = ('fused_voice_bus', 'fused_cascade_bus')
FIELDS = ['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count', 'Counter_Name', 'Counter_Value',
          'Start_Timestamp', 'End_Timestamp']


KERNEL = re.compile(r'(?:' + '|'.join(sorted(FAMILY, key=len, reverse=True)) + r')\b')


def short(name: str) -> str:
    """`void (anonymous namespace)::fused_steady_bus_kernel<8, 2>(...)` -> `fused_steady_bus_kernel`"""
    m = KERNEL.search(name)
    return m.group(0) if m else name


def ours(name: str) -> bool:
    return KERNEL.search(name) is not None and 'at::native' not in name


def load(src: pathlib.Path):
    files = glob.glob(str(src / '*' / '*_counter_collection.csv'))
    if not files:
        return {}, []
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = []
    for row in csv.DictReader(open(files[0])):
        if not ours(row['Kernel_Name']):
            continue
        rows.append(row)
        acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    return acc, rows


def mean(vals):
    vals = vals[1:] if len(vals) > 1 else vals
    return sum(vals) / len(vals)


def main(tag, rnd):
    src = ROOT / 'gpurun_out' / f'prof{tag}'
    prof = ROOT / 'profiles'
    traffic, raw = {}, {}
    for what in ('materialised', 'fused', 'configs', 'k256'):
        stats = glob.glob(str(src / f'stats_{what}' / '*' / '*_kernel_stats.csv'))
        if stats:
            shutil.copy(stats[0], prof / f'{rnd}_{what}_kernel_stats.csv')
        if (src / f'stats_{what}.json').exists():
            shutil.copy(src / f'stats_{what}.json', prof / f'{rnd}_{what}_bench_under_rocprof.json')
        fe, fe_rows = load(src / f'pmc_{what}_FETCH_SIZE')
        wr, wr_rows = load(src / f'pmc_{what}_WRITE_SIZE')
        prefix = '' if what != 'configs' else 'configs/'
        first = collections.defaultdict(bool)
        for k in fe:
            fam = FAMILY.get(k)
            if fam is None or len(fe[k]['FETCH_SIZE']) < 2:
                continue
            f, w = mean(fe[k]['FETCH_SIZE']), mean(wr[k]['WRITE_SIZE'])
            raw[f'{what}/{k}'] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'launches': len(fe[k]['FETCH_SIZE'])}
            key = prefix + fam
            if what == 'configs':            # tools/measure_configs.py: C3 = cascade kernel + its tile sum, C5 = the walker with the MFMA sink
                key = {'fused_cascade_kernel': 'C3/fused_cascade_bus', 'partials_kernel': 'C3/fused_cascade_bus',
                       'fused_walk_kernel': 'C2_modulated/fused_voice_bus', 'control_program_kernel': 'C2_modulated/control_program',
                       'fused_steady_bus_kernel': 'C2_sine_sweep/fused_voice_bus', 'voice_program_kernel': 'programs/voice_program', 'sig_vp_specialised': 'programs/voice_program_specialised', 'sig_ctl_specialised': 'C2_modulated/control_program_specialised',
                       'fused_steady_mix_kernel': 'C5/fused_osc_biquad_mix'}.get(k, 'configs/' + fam)
            if fam in SUMMED and what != 'configs':
                traffic[key] = (traffic.get(key, 0) if first[key] else 0) + int((2 * f + w) * 1024)
                first[key] = True
            elif what == 'configs':
                traffic[key] = traffic.get(key, 0) + int((2 * f + w) * 1024)    # (C3 and C5 launch different kernels)
            else:
                traffic[key] = int((2 * f + w) * 1024)
        for ctr, rows in (('FETCH_SIZE', fe_rows), ('WRITE_SIZE', wr_rows)):
            if rows:
                with open(prof / f'{rnd}_{what}_pmc_{ctr}.csv', 'w', newline='') as fh:
                    w_ = csv.DictWriter(fh, fieldnames=FIELDS, extrasaction='ignore')
                    w_.writeheader()
                    w_.writerows(rows)
        # issue-side counters
        sq = {}
        for group in ('SQ_issue', 'SQ_lds'):
            acc, rows = load(src / f'pmc_{what}_{group}')
            for k, ctrs in acc.items():
                sq.setdefault(k, {}).update({c: mean(v) for c, v in ctrs.items()})
                sq[k]['launches'] = max(sq[k].get('launches', 0), max(len(v) for v in ctrs.values()))
            if rows:
                with open(prof / f'{rnd}_{what}_pmc_{group}.csv', 'w', newline='') as fh:
                    w_ = csv.DictWriter(fh, fieldnames=FIELDS, extrasaction='ignore')
                    w_.writeheader()
                    w_.writerows(rows)
        for k, c in sq.items():
            if c.get('SQ_WAVE_CYCLES') and c.get('SQ_INSTS_VALU'):
                c['derived'] = {
                    'valu_busy_share_of_wave_cycles': c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES'],
                    'waiting_share_of_wave_cycles': c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES'],
                    'cycles_per_valu_instruction': 4.0 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU'],
                    'valu_instructions_per_wave': c['SQ_INSTS_VALU'] / max(c.get('SQ_WAVES', 1.0), 1.0),
                    'note': 'SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md); '
                            'GRBM_GUI_ACTIVE is summed over the 8 XCDs'}
        if sq:
            (prof / f'{rnd}_{what}_sq_counters.json').write_text(json.dumps(sq, indent=1) + '\n')
    bench = json.loads((src / 'stats_fused.json').read_text())
    cfg = bench['config']
    traffic['_meta'] = {
        'note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); counters are KiB; '
                'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); '
                'first (warm-up) launch of each kernel dropped; configs/* = tools/measure_configs.py (C3: fused_cascade_bus, '
                'C5: fused_osc_biquad_mix)',
        'workload': cfg['workload'], 'voice_samples_per_launch': cfg['voices_per_gpu'] * cfg['block_frames'] * cfg['blocks_per_step'],
        'raw': raw}
    (prof / 'traffic.json').write_text(json.dumps(traffic, indent=1) + '\n')
    print(json.dumps({k: v for k, v in traffic.items() if k != '_meta'}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
