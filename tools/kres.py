"""Compile one .hip translation unit for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, SGPRs, spills, LDS, occupancy).  usage: python tools/kres.py sradsgan_amd/csrc/conv_patch_pers.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
obj = os.path.splitext(src)[0] + '.o'
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-ffp-contract=off',
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', obj]
if os.environ.get('SAVE_TEMPS'):
    cmd.insert(1, '-save-temps=obj')
p = subprocess.run(cmd, capture_output=True, text=True)
cur = None
rows = {}
for line in p.stderr.splitlines():
    if 'error' in line or 'warning' in line:
        print(line)
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r'remark:(?: \S+:\d+:\d+:)?\s+([A-Za-z][A-Za-z \[\]/]*): (\S+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip().split('(')[0]
    if flt and flt not in name:
        continue
    print('%-70s VGPR %s AGPR %s SGPR %s spillV %s spillS %s scratch %s LDS %s occ %s' % (
        name[-70:], v.get('VGPRs'), v.get('AGPRs'), v.get('TotalSGPRs'), v.get('VGPRs Spill'), v.get('SGPRs Spill'),
        v.get('ScratchSize [bytes/lane]'), v.get('LDS Size [bytes/block]'), v.get('Occupancy [waves/SIMD]')))
sys.exit(p.returncode)
