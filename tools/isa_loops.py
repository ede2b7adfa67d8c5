"""tools/isa_loops.py <dir of tools/isa_c2.sh output> [kernel symbol substring] — every loop of a kernel: first / last instruction, size, VALU / SALU / LDS / scalar-load / division / lane-copy / scratch counts"""
import re, collections, sys
d = sys.argv[1]; sym = sys.argv[2] if len(sys.argv) > 2 else 'k_fusedILb1ELb1ELb0ELi4ELb0ELb0ELb0ELj15E'
S = open(d + '/' + (sys.argv[4] if len(sys.argv) > 4 else 'mtr_kernels') + '-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
on = False; ins = []; labels = {}
for l in S:
    if not on and l.startswith('_ZN3mtr') and sym in l and (':' in l): on = True; continue
    if not on: continue
    if 's_endpgm' in l: break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    t = l.strip()
    if not t or t[0] in ';.': continue
    ins.append(t.split(';')[0].strip())
loops = []
for i, t in enumerate(ins):
    m = re.match(r'(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)', t)
    if m and m.group(2) in labels and labels[m.group(2)] <= i: loops.append((labels[m.group(2)], i))
def st(a, b):
    seg = ins[a:b + 1]; c = collections.Counter(x.split()[0] for x in seg)
    return dict(n=len(seg), valu=sum(v for k, v in c.items() if k.startswith('v_')), pk=sum(v for k, v in c.items() if k.startswith('v_pk_')), salu=sum(v for k, v in c.items() if k.startswith('s_')), ds=sum(v for k, v in c.items() if k.startswith('ds_')),
                sload=sum(v for k, v in c.items() if k.startswith('s_load')), div=c['v_div_fixup_f32'], lane=c['v_readlane_b32'] + c['v_writelane_b32'], scr=sum(v for k, v in c.items() if k.startswith('scratch_')))
print('kernel', st(0, len(ins) - 1))
for a, b in sorted(loops): print(a, b, st(a, b))
if len(sys.argv) > 3:
    a, b = map(int, sys.argv[3].split(':'))
    print('\n'.join(ins[a:b + 1]))
