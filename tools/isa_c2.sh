#!/bin/bash
# tools/isa_c2.sh <tag> [extra flags] — static ISA statistics of the ONE k_fused instantiation config 2 runs (cross-compiles, ~1 min):
# registers / spills, instruction mix of the persistent loop and of its two inlined tree walks.  Output under /tmp/isa_<tag>/.
tag=$1; shift
src="$(cd "$(dirname "$0")/../mitransient_amd/csrc" && pwd)/mtr_kernels.hip"
d=/tmp/isa_$tag; mkdir -p $d; cd $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function -DMTR_ONLY_C2 "$@" \
  -Rpass-analysis=kernel-resource-usage --save-temps -c "$src" -o k.o 2> log.txt
grep -A12 "k_fusedILb1ELb1ELb0ELi4ELb0ELb0ELb0ELj15E" log.txt | grep -E "VGPRs:|Spill|Scratch" | sed 's/.*remark: *//' | tr '\n' ' '; echo
python3 - <<'PY'
import re, collections
S = open('mtr_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
on = False; ins = []; labels = {}
for l in S:
    if l.startswith('_ZN3mtr7k_fusedILb1ELb1ELb0ELi4ELb0ELb0ELb0ELj15EEEvNS_9FusedArgsE:'): on = True; continue
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
    seg = ins[a:b + 1]
    c = collections.Counter(x.split()[0] for x in seg)
    return dict(n=len(seg), valu=sum(v for k, v in c.items() if k.startswith('v_')), salu=sum(v for k, v in c.items() if k.startswith('s_')),
                mov=c['v_mov_b32_e32'] + c['v_mov_b64_e32'], lane=c['v_readlane_b32'] + c['v_writelane_b32'], ds=sum(v for k, v in c.items() if k.startswith('ds_')),
                scratch=sum(v for k, v in c.items() if k.startswith('scratch_')), div=c['v_div_fixup_f32'], nop=c['s_nop'])
print('kernel', st(0, len(ins) - 1))
big = sorted(loops, key=lambda ab: ab[0] - ab[1])
main = big[0]
print('persistent loop', st(*main))
walks = [ab for ab in loops if 600 < ab[1] - ab[0] < 1500 and ab[0] > main[0]]
seen = set()
for a, b in sorted(walks, key=lambda ab: (ab[0], -ab[1])):
    if a in seen: continue
    seen.add(a); print('walk loop @%d' % a, st(a, b))
PY
