#!/usr/bin/env python3
"""Which kernels WAIT between the loads of one batch?  For every kernel of the given gfx950 .s files (hipcc -S --cuda-device-only): the
sequence of global loads (L), vmcnt waits (W) and runs of VALU instructions (their count), and the number of places where a load or a
pair of loads is followed by its own wait before the next load is issued (the pattern `LW n LW n ...` or `LLWW n LLWW n ...`) -- what the
compiler makes of `x[k] = cond ? f(load(k)) : 0` in an unrolled loop when f() is sunk into the guarded block.  usage: isa_loadwaits.py a.s ..."""
import re, subprocess, sys
for path in sys.argv[1:]:
    src = open(path).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm', src, re.S | re.M):
        ins = [l.split(';')[0].strip() for l in m.group(2).splitlines()]
        ins = [i for i in ins if i and not i.endswith(':') and not i.startswith('.')]
        seq = ''.join('L' if i.startswith('global_load') else ('W' if 's_waitcnt' in i and 'vmcnt' in i else ('.' if i.startswith('v_') else '')) for i in ins)
        tight = len(re.findall(r'L{1,2}W{1,2}\.{0,24}(?=L{1,2}W)', seq))
        if tight >= 4:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()[:110]
            pat = re.sub(r'\.+', lambda q: str(len(q.group(0))) + ' ' if len(q.group(0)) > 3 else '', seq)
            print(f"{tight:3d} serialised loads  {name}\n      {pat[:260]}")
