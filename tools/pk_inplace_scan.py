"""round 6: scan gfx950 assembly (hipcc -S --cuda-device-only) for packed-fp32 instructions whose destination register
pair is also a source AND whose op_sel / op_sel_hi make one half read the OTHER half's register of that pair (e.g.
`v_pk_mul_f32 v[0:1], v[2:3], v[0:1] op_sel_hi:[1,0]`: the high half multiplies by v0, which the low half of the same instruction
overwrites).  Element-wise in-place forms (each half reads its own register) are not counted.

    python tools/pk_inplace_scan.py [--fail] file.s [file.s ...]        (also reads llvm-objdump -d output; --fail: exit 1 on any hit)"""
import re
import sys

PAT = re.compile(r'\s*(v_pk_(?:mul|add|fma)_f32)\s+(v\[\d+:\d+\]),\s*(.*)')


def bits(text, key, n, default):
    m = re.search(key + r':\[([01,]+)\]', text)
    if not m:
        return [default] * n
    b = [int(v) for v in m.group(1).split(',')]
    return b + [default] * (n - len(b))


def scan(path):
    total = inplace = cross = 0
    per = {}
    kern = None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            kern = m.group(1)
        m = PAT.match(line)
        if not m:
            continue
        total += 1
        dst = m.group(2)
        rest = m.group(3)
        ops = [o.strip() for o in re.split(r',\s*(?![^\[]*\])', rest.split(' op_sel')[0].split(' neg_')[0])]
        n = len(ops)
        sel, sel_hi = bits(rest, 'op_sel', n, 0), bits(rest, 'op_sel_hi', n, 1)
        hit = False
        for i, o in enumerate(ops):
            if o == dst:
                inplace += 1
                if sel[i] == 1 or sel_hi[i] == 0:
                    hit = True
                break
        if hit:
            cross += 1
            per[kern] = per.get(kern, 0) + 1
    return total, inplace, cross, per


fail = "--fail" in sys.argv
bad = 0
for p in [a for a in sys.argv[1:] if a != "--fail"]:
    t, i, c, per = scan(p)
    bad += c
    print(f"{p}: packed fp32 {t}, in-place {i}, in-place with a cross-half read of the destination pair {c}")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]:
        print(f"      {v:4d}  {k}")
if fail and bad:
    sys.exit(f"{bad} packed-fp32 instruction(s) read their own destination pair across halves")
