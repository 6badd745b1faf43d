"""Static counts of VALU / v_readlane / v_writelane / s_load / scratch instructions per kernel.  usage: lane_ops.py <lib.so> "<kernel>" ... (a trailing * matches a prefix)"""
import re,subprocess,sys,os,tempfile
lib=sys.argv[1]; pats=sys.argv[2:]
d=tempfile.mkdtemp()
L='/opt/rocm/lib/llvm/bin'
subprocess.check_call([f'{L}/llvm-objcopy','--dump-section',f'.hip_fatbin={d}/fat.bin',lib,f'{d}/x.so'])
subprocess.check_call([f'{L}/clang-offload-bundler','--unbundle','--type=o',f'--input={d}/fat.bin','--targets=hipv4-amdgcn-amd-amdhsa--gfx950',f'--output={d}/code.co'])
txt=subprocess.run([f'{L}/llvm-objdump','-d',f'{d}/code.co'],capture_output=True,text=True).stdout
for b in re.split(r'\n(?=[0-9a-f]{16} <)',txt):
    m=re.match(r'[0-9a-f]{16} <(\S+)>:',b)
    if not m: continue
    name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().split('(')[0].replace('void ','')
    if not any(name==p or (p.endswith('*') and name.startswith(p[:-1])) for p in pats): continue
    lines=[l for l in b.splitlines() if re.match(r'\s+[a-z]',l)]
    c=lambda f: sum(1 for l in lines if f(l.strip()))
    print(f"{name:36s} instr {len(lines):6d} VALU {c(lambda l: l.startswith('v_')):5d} readlane {c(lambda l: 'v_readlane' in l):5d} writelane {c(lambda l: 'v_writelane' in l):4d} s_load {c(lambda l: l.startswith('s_load')):3d} scratch {c(lambda l: 'scratch_' in l):3d}")
