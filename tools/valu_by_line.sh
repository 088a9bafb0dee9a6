#!/bin/bash
# Static VALU instruction count of one compiled pt_kernel variant per source line (runs here: hipcc cross-compiles).
# usage: tools/valu_by_line.sh <mis|nee|mis_cc|nee_cc|generic> <FEAT> [MODE]    e.g.  tools/valu_by_line.sh mis 1 1   (C2's kernel), nee_cc 4 2 (C5's)
# Method: hipcc -gline-tables-only -S, then every v_* instruction is charged to the .loc line in effect (inlined callees are charged
# to their own line: sqrtf / expf / sincosf show up under __clang_hip_math.h).  Static, not dynamic: a guide to what to look at.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TU=$1; FEAT=$2; MODE=${3:-0}
# <TU> = mis | nee | mis_cc | nee_cc | generic: the translation unit that holds the kernel (csrc/Makefile), compiled with that unit's backend options
SRC=$R/toy-cpu-pathtracing_amd/csrc/pt_kernels$([ "$TU" = generic ] && echo "" || echo "_$TU").hip
OUT=$(mktemp -d)
OPT="-mllvm -disable-machine-licm"
case $TU in mis|nee) OPT="$OPT -mllvm -sink-insts-to-avoid-spills=1 -mllvm -amdgpu-use-amdgpu-trackers=1";; mis_cc|nee_cc) OPT="$OPT -mllvm -sink-insts-to-avoid-spills=1";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off $OPT -gline-tables-only -S --cuda-device-only -o $OUT/k.s $SRC 2>/dev/null
python3 - "$OUT/k.s" "_ZN2pt9pt_kernelILb0ELj${FEAT}ELj${MODE}EE" "$R/toy-cpu-pathtracing_amd/csrc/" <<'PY'
import re, collections, sys, os
path, sym, srcdir = sys.argv[1:4]
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
cur = None
div, tot = collections.Counter(), collections.Counter()
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    t = l.strip().split(' ')[0] if l.strip() else ''
    if t.startswith('v_'): tot[cur] += 1
    if t == 'v_div_fixup_f32': div[cur] += 1
src = {}
def text(f, n):
    p = srcdir + f
    if os.path.exists(p):
        if p not in src: src[p] = open(p).read().split('\n')
        return src[p][n - 1].strip()[:100] if 0 < n <= len(src[p]) else ''
    return ''
print(sym, ': VALU instructions', sum(tot.values()), ' correctly rounded divisions', sum(div.values()))
print('--- divisions by line')
for k, v in div.most_common(20): print(f'{v:5d} {k[0]}:{k[1]} | {text(*k)}')
print('--- VALU by line')
for k, v in tot.most_common(30): print(f'{v:5d} {k[0]}:{k[1]} | {text(*k)}')
PY
rm -rf $OUT
