#!/usr/bin/env python3
"""Summarise tools/gpu_roofline2.sh output: per conv-math mode and kernel, average launch time (kernel trace),
HBM bytes per launch (FETCH_SIZE x2 gfx950 correction for 16 B/lane streams + WRITE_SIZE, both in KiB units of
rocprofv3) and MFMA-busy fraction.  Writes <dir>/summary.txt and <dir>/roofline_traffic.json."""
import csv, glob, json, os, re, sys, collections

root = sys.argv[1]
KEYS = {'fprop': ('conv_patch_kernel', 'conv_patch_pers_kernel', 'fast_conv_dma_kernel'), 'wgrad': ('fast_wgrad_dma_kernel', 'wgrad_rowtap_kernel', 'wgrad_flat8_kernel', 'wgrad_flat_kernel', 'pp_from_f32_kernel', 'fast_wgrad_reduce_kernel', 'fast_wgrad_reduce4_kernel')}
out, lines = {}, []


def short(n):
    return re.sub(r'\(.*', '', n.replace('void ', ''))[:70]


for mode in ('fp32', 'bf16x3', 'half'):
    d = os.path.join(root, mode)
    if not os.path.isdir(d):
        continue
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ('fetch', 'write', 'mfma'):
        for f in glob.glob(os.path.join(d, sub, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                pmc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    stats = {}
    for f in glob.glob(os.path.join(d, 'kt', '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']))
    lines.append('## conv_math = %s' % mode)
    out[mode] = {}
    for key, names in KEYS.items():
        total = 0.0
        # a grouped weight-gradient launch is ONE main kernel + one reduce per convolution: weigh every kernel by its number of
        # dispatches relative to the main kernel's, so the figure stays "HBM bytes per launch of the dominant kernel"
        aux = lambda k: 'reduce' in k or 'pp_from_f32' in k          # kernels that ride beside the main launch (reduces, operand conversion passes)
        main_calls = max([len(pmc[k].get('FETCH_SIZE', [])) for k in pmc if any(n in k for n in names) and not aux(k)] or [0])
        for k in sorted(pmc):
            if not any(n in k for n in names):
                continue
            c = pmc[k]
            weight = (len(c.get('FETCH_SIZE', [])) / main_calls) if (main_calls and aux(k)) else 1.0
            if not aux(k) and main_calls and len(c.get('FETCH_SIZE', [])) < main_calls:
                weight = 0.0                                         # a kernel timed only for comparison beside the main one
            avg = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else 0.0
            fetch, write = avg('FETCH_SIZE') * 1024 * 2, avg('WRITE_SIZE') * 1024
            calls, ns = stats.get(k, (0, 0.0))
            busy = avg('SQ_VALU_MFMA_BUSY_CYCLES') / (avg('GRBM_GUI_ACTIVE') / 8 * 1024) if avg('GRBM_GUI_ACTIVE') else 0.0
            lines.append('%-72s calls %-4d avg %8.1f us  HBM read %7.1f MB (FETCH_SIZE x2) write %7.1f MB  MFMA busy %4.1f %% of SIMD-cycles  VALU/MFMA instr %.1f'
                         % (k, calls, ns / 1e3, fetch / 1e6, write / 1e6, 100 * busy, avg('SQ_INSTS_VALU') / max(avg('SQ_INSTS_MFMA'), 1)))
            total += (fetch + write) * weight
        out[mode][key] = int(total)
open(os.path.join(root, 'summary.txt'), 'w').write('\n'.join(lines) + '\n')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out['source_sha16'] = bench.kernel_source_sha16()      # bench.py refuses the file once the kernel sources have changed
json.dump(out, open(os.path.join(root, 'roofline_traffic.json'), 'w'), indent=1)
print('\n'.join(lines))
print(json.dumps(out))
