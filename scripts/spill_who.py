"""Which spilled scalar registers a kernel reads back most (v_readlane per spill slot) and what defined them.  usage: spill_who.py <lib.so> "<kernel name as kres.py prints it>" """
import re,subprocess,sys,collections,tempfile
lib=sys.argv[1]; kern=sys.argv[2]
d=tempfile.mkdtemp(); L='/opt/rocm/lib/llvm/bin'
subprocess.check_call([f'{L}/llvm-objcopy','--dump-section',f'.hip_fatbin={d}/fat.bin',lib,f'{d}/x.so'])
subprocess.check_call([f'{L}/clang-offload-bundler','--unbundle','--type=o',f'--input={d}/fat.bin','--targets=hipv4-amdgcn-amd-amdhsa--gfx950',f'--output={d}/code.co'])
txt=subprocess.run([f'{L}/llvm-objdump','-d',f'{d}/code.co'],capture_output=True,text=True).stdout
for b in re.split(r'\n(?=[0-9a-f]{16} <)',txt):
    m=re.match(r'[0-9a-f]{16} <(\S+)>:',b)
    if not m: continue
    name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().split('(')[0].replace('void ','')
    if name!=kern: continue
    lines=[l.split('//')[0].strip() for l in b.splitlines() if re.match(r'\s+[a-z]',l)]
    reads=collections.Counter(); wr={}
    def sregs(tok):
        m=re.match(r's\[(\d+):(\d+)\]',tok)
        if m: return list(range(int(m.group(1)),int(m.group(2))+1))
        m=re.match(r's(\d+)$',tok)
        return [int(m.group(1))] if m else []
    for i,l in enumerate(lines):
        t=re.split(r'[ ,]+',l)
        if t[0]=='v_readlane_b32': reads[(t[2],t[3])]+=1
        if t[0]=='v_writelane_b32':
            slot=(t[1],t[3]); src=t[2]
            # find def of src
            sr=sregs(src); df='?'
            for j in range(i-1,max(0,i-400),-1):
                tt=re.split(r'[ ,]+',lines[j])
                if tt[0].startswith('v_writelane') or tt[0].startswith('s_cbranch') or tt[0].startswith('s_waitcnt') or tt[0].startswith('s_nop'): continue
                if len(tt)>1 and any(r in sregs(tt[1]) for r in sr) and not tt[0].startswith('s_cmp') and not tt[0].startswith('v_cmp_') :
                    df=lines[j]; break
            wr.setdefault(slot,[]).append(df)
    tot=sum(reads.values()); print(name,'readlanes',tot,'slots',len(reads))
    for slot,c in reads.most_common(70):
        print(f'{slot[0]}:{slot[1]:>3s} reads {c:4d}  defs: {" | ".join(sorted(set(wr.get(slot,["-"])))[:3])[:150]}')
