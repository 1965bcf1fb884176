#!/bin/bash
# per-kernel profile of one training tail (forward + backward, B = 32) with and without the dz-free backward
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tailprof
cd /tmp && export TMPDIR=/tmp
for k in 1 0; do
  export SRHIP_TAIL_NODZ=$k
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tailprof/nodz$k -o t --output-format csv -- python3 $R/tools/time_tail_train.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/tailprof/nodz$k -name "*kernel_stats.csv" | head -1)
  echo "== SRHIP_TAIL_NODZ=$k"
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
tot = 0.0
for r in rows:
    if int(r['Calls']) >= 60:
        tot += float(r['AverageNs']) / 1e3 * int(r['Calls']) / 60
        print('%-84s calls %5s avg %8.1f us' % (r['Name'][:84], r['Calls'], float(r['AverageNs']) / 1e3))
print('kernel time per tail (forward + backward): %.1f us' % tot)
PY
done
