#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --trace-losses 2>&1 | grep -E "losses per step|Error|error" | head -5; }
run AMD_SERIALIZE_KERNEL=3
run SRHIP_GRAPH_DUMP=gpurun_out/graph.dot
ls -la gpurun_out/graph.dot 2>/dev/null; 
python - <<'PY'
import re,collections
try:
    txt=open('gpurun_out/graph.dot').read()
except Exception as e:
    print('no dot', e); raise SystemExit
edges=re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
out=collections.Counter(a for a,b in edges); inn=collections.Counter(b for a,b in edges)
nodes=set(out)|set(inn)
print('nodes', len(nodes), 'edges', len(edges), 'fan-out>1', sum(1 for n in nodes if out[n]>1), 'fan-in>1', sum(1 for n in nodes if inn[n]>1), 'roots', sum(1 for n in nodes if inn[n]==0))
PY
head -c 1500 gpurun_out/graph.dot 2>/dev/null
