#!/bin/bash
# rocprofv3 kernel stats of one python command: tools/gpu_kstats.sh <tag> <script> [args...]
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/kstats_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/p -o st --output-format csv -- python3 "$@" > $O/run.log 2>&1
cd $R
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:12]:
    print('%-100s calls %6s avg %9.1f us  total %9.1f us' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
rm -rf $O/p
