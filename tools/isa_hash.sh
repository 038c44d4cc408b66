#!/bin/bash
# hash of the gfx950 ISA hipcc generates for one .hip unit (device code only; comments, debug lines and file names stripped):
# two sources with the same hash are the same kernels.   tools/isa_hash.sh file.hip [extra flags]      (-k first: one hash per kernel)
PER=0; if [ "$1" = "-k" ]; then PER=1; shift; fi
F=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function --cuda-device-only -S "$@" "$F" -o - 2>/dev/null \
  | grep -v '^\s*;\|^\s*\.file\|^\s*\.ident\|^\s*\.loc\|\.amdgcn_target' | sed 's/;.*$//' > /tmp/isa_hash_$$.s
if [ $PER = 1 ]; then
  python3 - /tmp/isa_hash_$$.s <<'PY'
import sys, re, hashlib
cur, body, out = None, [], {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m and cur is None:
        cur, body = m.group(1), []
    if cur is not None:
        body.append(re.sub(r'\.(LBB|Lfunc_end|Lfunc_begin|Ltmp)\d+', r'.\1', re.sub(r'_Z\w+', 'SYM', line)))
        if line.strip().startswith('.Lfunc_end'):
            out[cur] = hashlib.md5(''.join(body).encode()).hexdigest()[:16]; cur = None
import subprocess
for k, v in out.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    print(v, name[:110])
PY
else
  md5sum < /tmp/isa_hash_$$.s | cut -c1-32
fi
rm -f /tmp/isa_hash_$$.s
