"""Static instruction profile of a kernel by source statement: where the VALU instructions of the particle kernels come from.

usage: python scripts/valu_profile.py [--asm FILE] [-DNAME=VALUE ...] <kernel substring> [<body function>]
  Compiles fe_engine.hip for the device with line tables (`-gline-tables-only -S`: debug line info does not change the generated
  code) unless --asm names an assembly file of an earlier call, and attributes every instruction of the kernel to
    * the STATEMENT of the kernel's body function it was inlined under (the outermost frame of its .loc chain), and
    * the innermost device function it belongs to.
  Counts are static (one per instruction in the code object); the particle kernels' hot paths are straight-line code (the 27-node
  loops are unrolled), so the count of a statement on the hot path is what one wave issues for one unit.  VALU = v_* (DPP forms
  included), LDS = ds_*, VMEM = global_/buffer_/flat_/scratch_, SALU = s_* without s_waitcnt / s_nop / branches.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'fluidlab_amd', 'csrc', 'fe_engine.hip')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fno-slp-vectorize']


def compile_asm(defs, out):
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['--cuda-device-only', '-gline-tables-only', '-S', '-o', out, SRC] + defs,
                   check=True, stderr=subprocess.DEVNULL)


def function_extents(path):
    """[(first line, last line, name)] of the function definitions of a source file (brace matching from lines that look like a
    definition; good enough for this code base's one-definition-per-line-start style)."""
    out, lines = [], open(path).read().split('\n')
    pat = re.compile(r'^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+|inline\s+|__host__\s+|__device__\s+|__forceinline__\s+|__global__\s+|FE_KALIGN\s+|__launch_bounds__\([^)]*\)\s+)+'
                     r'[\w:<>\*&\s]+?[\s\*&](\w+)\s*\(')
    i = 0
    while i < len(lines):
        m = pat.match(lines[i])
        if m:
            depth, j, seen = 0, i, False
            while j < len(lines):
                for ch in lines[j]:
                    if ch == '{':
                        depth += 1; seen = True
                    elif ch == '}':
                        depth -= 1
                if seen and depth <= 0:
                    break
                if not seen and lines[j].rstrip().endswith(';'):
                    break
                j += 1
            if seen:
                out.append((i + 1, j + 1, m.group(1)))
                i = j
        i += 1
    return out


def classify(op):
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'VMEM'
    if op.startswith('s_'):
        if op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_barrier', 's_endpgm', 's_sleep', 's_setprio', 's_load', 's_buffer_load')):
            return 'other'
        return 'SALU'
    return 'other'


def profile(asm, kernel_sub, body=None):
    ext = {}
    def fn_of(path, line):
        if path not in ext:
            ext[path] = function_extents(path) if os.path.exists(path) else []
        for a, b, n in ext[path]:
            if a <= line <= b:
                return n
        return os.path.basename(path) + ':?'
    loc_re = re.compile(r'; (\S+):(\d+):\d+((?: @\[ \S+:\d+:\d+)*)')
    chain_re = re.compile(r'@\[ (\S+):(\d+):\d+')
    by_stmt = collections.defaultdict(collections.Counter)
    by_fn = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    inside, cur = False, (None, 0, 'prologue')
    name = None
    for line in open(asm):
        if not inside:
            if line.startswith('_Z') and ':' in line:
                head = line.split(':')[0]
                dem = subprocess.run(['/usr/bin/c++filt', head], capture_output=True, text=True).stdout.strip().replace('void ', '')
                if dem.split('(')[0].startswith(kernel_sub):
                    inside, name = True, dem.split('(')[0]
            continue
        if line.startswith('.Lfunc_end'):
            break
        s = line.strip()
        if s.startswith('.loc'):
            m = loc_re.search(s)
            if m:
                path, ln = m.group(1), int(m.group(2))
                chain = [(p, int(l)) for p, l in chain_re.findall(m.group(3))]
                frames = [(path, ln)] + chain                    # innermost first
                inner_fn = fn_of(path, ln) if ln else '(line 0)'
                # the statement of the body function: the outermost frame that lies inside `body` (default: the frame below the kernel's own line)
                stmt = None
                for p, l in reversed(frames):
                    f = fn_of(p, l)
                    if body is None:
                        if f != name.split('<')[0]:
                            stmt = (f, l); break
                    elif f == body:
                        stmt = (f, l); break
                if stmt is None:
                    stmt = ('(kernel entry)', frames[-1][1])
                cur = (stmt, inner_fn)
            continue
        if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
            continue
        op = s.split()[0]
        c = classify(op)
        total[c] += 1
        if 'dpp' in s and c == 'VALU':
            total['VALU_dpp'] += 1
        stmt, inner = cur if isinstance(cur[0], tuple) else (('(prologue)', 0), 'prologue')
        by_stmt[stmt][c] += 1
        by_fn[inner][c] += 1
    return name, total, by_stmt, by_fn


def main():
    args = sys.argv[1:]
    asm, defs, rest = None, [], []
    while args:
        a = args.pop(0)
        if a == '--asm':
            asm = args.pop(0)
        elif a.startswith('-D'):
            defs.append(a)
        else:
            rest.append(a)
    if not rest:
        sys.exit(__doc__)
    if asm is None or not os.path.exists(asm):
        asm = asm or os.path.join(tempfile.gettempdir(), 'fe_engine_lines.s')
        compile_asm(defs, asm)
    src_lines = open(SRC).read().split('\n')
    name, total, by_stmt, by_fn = profile(asm, rest[0], rest[1] if len(rest) > 1 else None)
    print(f'== {name}: static instruction counts  VALU {total["VALU"]} (DPP {total["VALU_dpp"]})  SALU {total["SALU"]}  LDS {total["LDS"]}  VMEM {total["VMEM"]}  other {total["other"]}')
    print('-- by statement of the body function (VALU >= 8), source order')
    for (f, l), c in sorted(by_stmt.items(), key=lambda kv: (kv[0][0] != (rest[1] if len(rest) > 1 else kv[0][0]), kv[0][1])):
        if c['VALU'] >= 8:
            txt = src_lines[l - 1].strip()[:110] if 0 < l <= len(src_lines) and f != '(prologue)' else ''
            print(f'{f:>22s}:{l:<5d} VALU {c["VALU"]:5d} SALU {c["SALU"]:4d} LDS {c["LDS"]:4d} VMEM {c["VMEM"]:4d} | {txt}')
    print('-- by innermost function (VALU >= 8), largest first')
    for f, c in sorted(by_fn.items(), key=lambda kv: -kv[1]['VALU']):
        if c['VALU'] >= 8:
            print(f'{f:>34s} VALU {c["VALU"]:5d} SALU {c["SALU"]:4d} LDS {c["LDS"]:4d} VMEM {c["VMEM"]:4d}')


if __name__ == '__main__':
    main()
