#!/usr/bin/env python3
"""Turn the rocprofv3 outputs a gpurun call left under gpurun_out/ into the committed summaries in profiles/.

    python tools/collect_profiles.py <tag> <round>      e.g.  pmc2/stats2 -> tag "2", round "r01"

Expects gpurun_out/pmc<tag>_{fused,materialised}_{FETCH_SIZE,WRITE_SIZE}/ and gpurun_out/stats<tag>_{fused,materialised}/
(see profiles/README.md for the exact commands).  FETCH_SIZE is doubled (gfx950 reports half the bytes of wide
coalesced reads, MI355X_MICROARCH.md §HBM); counters are KiB; the first launch of each kernel is dropped.
"""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
# one sig_fused_voice_bus call = steady_prep_kernel + fused_steady_bus_kernel + fused_walk_kernel (the waves the closed
# form does not take; template argument C = 0 is the other entry point's f32 store, not launched by the C2 bench)
# + bus_partials_kernel: their bytes are summed into 'fused_voice_bus'
FUSED_CALL = ('fused_steady_bus_kernel', 'fused_walk_kernel', 'steady_prep_kernel', 'bus_partials_kernel', 'partials_kernel')
FAMILY = {'fused_walk_kernel': 'fused_voice_bus', 'fused_steady_bus_kernel': 'fused_voice_bus', 'steady_prep_kernel': 'fused_voice_bus',
          'bus_partials_kernel': 'fused_voice_bus', 'partials_kernel': 'fused_voice_bus', 'sum_bus_fast_kernel': 'sum_bus', 'sum_bus_kernel': 'sum_bus',
          'osc_bank_kernel': 'osc_bank', 'biquad_coldstart_kernel': 'biquad_coldstart',
          'biquad_walk_kernel': 'biquad_coldstart', 'ew_fast_kernel': 'elementwise', 'fused_scan_kernel': 'fused_scan'}


def load(tag, mode, ctr):
    f = glob.glob(str(ROOT / f'gpurun_out/pmc{tag}_{mode}_{ctr}/*/*_counter_collection.csv'))[0]
    acc = collections.defaultdict(list)
    rows = []
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        if 'anonymous' not in name and 'sig_bus::' not in name:          # our kernels only
            continue
        rows.append(row)
        acc[name.split('::')[1].split('<')[0]].append(float(row['Counter_Value']))
    return acc, rows


def main(tag, rnd):
    out, raw = {}, {}
    for mode in ('materialised', 'fused'):
        fe, fe_rows = load(tag, mode, 'FETCH_SIZE')
        wr, wr_rows = load(tag, mode, 'WRITE_SIZE')
        for k in fe:
            if k not in FAMILY or len(fe[k]) < 2:
                continue
            f = sum(fe[k][1:]) / len(fe[k][1:])
            w = sum(wr[k][1:]) / len(wr[k][1:])
            raw[f'{mode}/{k}'] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'launches': len(fe[k])}
            if k in FUSED_CALL:
                out[FAMILY[k]] = out.get(FAMILY[k], 0) + int((2 * f + w) * 1024)
            else:
                out[FAMILY[k]] = int((2 * f + w) * 1024)
        for ctr, rows in (('FETCH_SIZE', fe_rows), ('WRITE_SIZE', wr_rows)):
            with open(ROOT / f'profiles/{rnd}_{mode}_pmc_{ctr}.csv', 'w', newline='') as fh:
                w_ = csv.DictWriter(fh, fieldnames=['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count',
                                                    'Counter_Name', 'Counter_Value', 'Start_Timestamp', 'End_Timestamp'],
                                    extrasaction='ignore')
                w_.writeheader()
                w_.writerows(rows)
        src = glob.glob(str(ROOT / f'gpurun_out/stats{tag}_{mode}/*/*_kernel_stats.csv'))[0]
        shutil.copy(src, ROOT / f'profiles/{rnd}_{mode}_kernel_stats.csv')
        shutil.copy(ROOT / f'gpurun_out/stats{tag}_{mode}.json', ROOT / f'profiles/{rnd}_{mode}_bench_under_rocprof.json')
    bench = json.loads((ROOT / f'gpurun_out/stats{tag}_fused.json').read_text())
    cfg = bench['config']
    out['_meta'] = {
        'note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); counters are KiB; '
                'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); '
                'first (warm-up) launch of each kernel dropped',
        'workload': cfg['workload'], 'voice_samples_per_launch': cfg['voices_per_gpu'] * cfg['block_frames'] * cfg['blocks_per_step'],
        'raw': raw}
    (ROOT / 'profiles/traffic.json').write_text(json.dumps(out, indent=1) + '\n')
    print(json.dumps({k: v for k, v in out.items() if k != '_meta'}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
