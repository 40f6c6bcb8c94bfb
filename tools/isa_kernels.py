"""Per-kernel instruction listing and register counts from a gfx950 .s file (hipcc --save-temps); used to check that a
source change leaves a given template instantiation's code untouched:  python tools/isa_kernels.py a.s b.s [filter]"""
import re, sys


def kernels(path):
    src = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm', src, re.S | re.M):
        ins = []
        for l in m.group(2).splitlines():
            t = l.split(';')[0].strip()
            if t and not t.startswith('.') and not t.endswith(':'):
                ins.append(re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r'\s+', ' ', t)))   # labels are numbered per function index
        out[m.group(1)] = ins
    regs = {}
    for m in re.finditer(r'\.name:\s+(_Z\w+)\n(.*?)(?=\n  - |\namdhsa\.target|\Z)', src, re.S):
        v = re.search(r'\.vgpr_count:\s+(\d+)', m.group(2)); s = re.search(r'\.sgpr_count:\s+(\d+)', m.group(2))
        regs[m.group(1)] = (int(v.group(1)) if v else None, int(s.group(1)) if s else None)
    return out, regs


if __name__ == '__main__':
    a, ra = kernels(sys.argv[1]); b, rb = kernels(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ''
    for k in sorted(set(a) | set(b)):
        if flt not in k: continue
        ia, ib = a.get(k), b.get(k)
        nd = sum(x != y for x, y in zip(ia or [], ib or [])) if ia and ib and len(ia) == len(ib) else None
        print('%-70s %s  n=%s/%s  vgpr,sgpr=%s/%s  differing lines=%s' % (
            k[:70], 'IDENTICAL' if ia == ib else 'DIFFERENT', len(ia) if ia else None, len(ib) if ib else None, ra.get(k),
            rb.get(k), nd))
        if nd and nd <= 4:
            for x, y in zip(ia, ib):
                if x != y: print('      %s  ->  %s' % (x, y))
