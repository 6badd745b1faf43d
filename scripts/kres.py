"""Registers, scratch, occupancy and LDS of every kernel of the engine.
usage: python scripts/kres.py [--lib] [-DNAME=VALUE ...] [filter-substring ...]
  default: recompile fe_engine.hip with -Rpass-analysis=kernel-resource-usage (honours -D flags; ~40 s)
  --lib:   read the AMDGPU metadata notes of the code object inside the BUILT fluidlab_amd/csrc/libfluidengine_hip.so (instant; what ships)
`kernel_resources()` is what tests/test_build_resources.py asserts on."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
LIB = os.path.join(ROOT, 'fluidlab_amd', 'csrc', 'libfluidengine_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fno-slp-vectorize']


def _demangle(names):
    out = subprocess.run(['/usr/bin/c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return [o.strip().split('(')[0].replace('void ', '') for o in out]


def kernel_resources(lib=LIB):
    """{kernel name (demangled, no argument list): {'vgpr', 'sgpr', 'scratch' (bytes per lane), 'vgpr_spills', 'sgpr_spills', 'lds'}} of the
    gfx950 code object bundled into `lib`."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'code.co')
        subprocess.check_call([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', lib, os.path.join(d, 'x.so')])
        subprocess.check_call([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'])
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True, check=True).stdout
    rows, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r'\s*(?:- )?\.(\w+):\s+(\S+)', line)
        if not m:
            continue
        k, v = m.groups()
        if k == 'agpr_count' and cur.get('name'):           # (the first key of the next kernel's record)
            rows.append(cur); cur = {}
        if k in ('name', 'private_segment_fixed_size', 'sgpr_count', 'sgpr_spill_count', 'vgpr_count', 'vgpr_spill_count', 'group_segment_fixed_size'):
            if k != 'name' or v.startswith('_Z') or v.startswith('k_'):
                cur[k] = v
    if cur.get('name'):
        rows.append(cur)
    names = _demangle([r['name'] for r in rows])
    return {n: dict(vgpr=int(r.get('vgpr_count', -1)), sgpr=int(r.get('sgpr_count', -1)), scratch=int(r.get('private_segment_fixed_size', -1)),
                    vgpr_spills=int(r.get('vgpr_spill_count', -1)), sgpr_spills=int(r.get('sgpr_spill_count', -1)), lds=int(r.get('group_segment_fixed_size', -1)))
            for n, r in zip(names, rows)}


def kernel_addresses(lib=LIB):
    """{kernel name (demangled, no argument list): (address, size)} of the gfx950 code object bundled into `lib` (where the kernels sit:
    the substep kernels are aligned in the code object, fe_engine.hip FE_KALIGN)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'code.co')
        subprocess.check_call([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', lib, os.path.join(d, 'x.so')])
        subprocess.check_call([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'])
        syms = subprocess.run([f'{LLVM}/llvm-readelf', '-s', '-W', co], capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in syms.splitlines() if ' FUNC ' in l]
    rows = [r for r in rows if len(r) >= 8]
    names = _demangle([r[7] for r in rows])
    return {n: (int(r[1], 16), int(r[2])) for n, r in zip(names, rows)}


def compile_resources(defs):
    cmd = ['/opt/rocm/bin/hipcc'] + FLAGS + ['--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-c', '-o', '/dev/null',
                                              os.path.join(ROOT, 'fluidlab_amd', 'csrc', 'fe_engine.hip')] + defs
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r'remark: +Function Name: (\S+)', line)
        if m:
            cur = _demangle([m.group(1)])[0]
            rows[cur] = {}
            continue
        m = re.search(r'remark: +([A-Za-z \[\]/]+): (\S+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
        if 'error' in line:
            print(line)
    return {k: dict(vgpr=r.get('VGPRs', '?'), agpr=r.get('AGPRs', '?'), sgpr=r.get('TotalSGPRs', '?'), scratch=r.get('ScratchSize [bytes/lane]', '?'),
                    occ=r.get('Occupancy [waves/SIMD]', '?'), lds=r.get('LDS Size [bytes/block]', '?')) for k, r in rows.items()}


if __name__ == '__main__':
    args = sys.argv[1:]
    filt = [a for a in args if not a.startswith('-')]
    if '--lib' in args:
        rows = kernel_resources()
        print(f'{"kernel":44s} {"VGPR":>5s} {"SGPR":>5s} {"scratch":>8s} {"vspill":>7s} {"sspill":>7s} {"LDS":>7s}')
        for k, r in rows.items():
            if not filt or any(f in k for f in filt):
                print(f'{k[:44]:44s} {r["vgpr"]:>5d} {r["sgpr"]:>5d} {r["scratch"]:>8d} {r["vgpr_spills"]:>7d} {r["sgpr_spills"]:>7d} {r["lds"]:>7d}')
    else:
        rows = compile_resources([a for a in args if a.startswith('-D')])
        print(f'{"kernel":44s} {"VGPR":>5s} {"AGPR":>5s} {"SGPR":>5s} {"scratch":>8s} {"occ":>4s} {"LDS":>7s}')
        for k, r in rows.items():
            if not filt or any(f in k for f in filt):
                print(f'{k[:44]:44s} {r["vgpr"]:>5s} {r["agpr"]:>5s} {r["sgpr"]:>5s} {r["scratch"]:>8s} {r["occ"]:>4s} {r["lds"]:>7s}')
