R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_conv2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/a -o p --output-format csv -- python3 $R/tools/_run_conv2.py > $O/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/b -o p --output-format csv -- python3 $R/tools/_run_conv2.py > $O/b.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum -d $O/c -o p --output-format csv -- python3 $R/tools/_run_conv2.py > $O/c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM -d $O/d -o p --output-format csv -- python3 $R/tools/_run_conv2.py > $O/d.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, re
for d in 'abcd':
    for path in glob.glob('gpurun_out/pmc_conv2/%s/**/*counter_collection.csv' % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = re.sub(r'^void ', '', r['Kernel_Name'])[:90]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in sorted(agg.items()):
            if 'patch_pers' not in k: continue
            print(k)
            for c, v in sorted(cs.items()):
                print('    %-30s n=%-4d avg=%.5g' % (c, len(v), sum(v) / len(v)))
PY
tail -3 $O/c.log
