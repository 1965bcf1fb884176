#!/bin/bash
# Round 4: where the waves of the two roofline kernels spend their cycles (SQ wait / issue / active buckets), bf16x3.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sqwait
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $O/a -o p --output-format csv -- python3 $R/bench.py --roofline-only --no-sustained --conv-math bf16x3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $O/b -o p --output-format csv -- python3 $R/bench.py --roofline-only --no-sustained --conv-math bf16x3 > $O/b.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, re
for d in ('a', 'b'):
    for path in glob.glob('gpurun_out/sqwait/%s/**/*counter_collection.csv' % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name']))[:64]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in agg.items():
            if 'rowtap' not in k and 'patch_pers' not in k and 'reduce4' not in k: continue
            wc = sum(cs['SQ_WAVE_CYCLES']) / len(cs['SQ_WAVE_CYCLES']) if 'SQ_WAVE_CYCLES' in cs else 0
            print(k)
            for c, v in sorted(cs.items()):
                a = sum(v) / len(v)
                print('    %-28s n=%-4d avg=%.4g   %s' % (c, len(v), a, ('%.1f %% of wave cycles' % (100 * a / wc)) if wc and c.startswith('SQ_') and c != 'SQ_WAVE_CYCLES' else ''))
PY
