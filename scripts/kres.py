"""Registers, scratch, occupancy and LDS of every kernel of the engine (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kres.py [filter-substring ...]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fno-slp-vectorize', '--cuda-device-only',
       '-Rpass-analysis=kernel-resource-usage', '-c', '-o', '/dev/null', os.path.join(ROOT, 'fluidlab_amd', 'csrc', 'fe_engine.hip')] + [a for a in sys.argv[1:] if a.startswith('-D')]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
filt = [a for a in sys.argv[1:] if not a.startswith('-D')]
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['/usr/bin/c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]
        rows[cur] = {}
        continue
    m = re.search(r'remark: +([A-Za-z \[\]/]+): (\S+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
    if 'error' in line:
        print(line)
print(f'{"kernel":44s} {"VGPR":>5s} {"AGPR":>5s} {"SGPR":>5s} {"scratch":>8s} {"occ":>4s} {"LDS":>7s}')
for k, r in rows.items():
    if filt and not any(f in k for f in filt):
        continue
    print(f'{k[:44]:44s} {r.get("VGPRs","?"):>5s} {r.get("AGPRs","?"):>5s} {r.get("TotalSGPRs","?"):>5s} {r.get("ScratchSize [bytes/lane]","?"):>8s} {r.get("Occupancy [waves/SIMD]","?"):>4s} {r.get("LDS Size [bytes/block]","?"):>7s}')
