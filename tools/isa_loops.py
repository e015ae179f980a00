"""Static look at the hot loops of a hipcc -save-temps .s file: per function, the backward-branch loop with the most v_add_f64 with its
instruction mix (VALU / SALU / LDS / scratch / SGPR-spill lane moves) and the function's register counts.
usage: python tools/isa_loops.py file.s [name-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r'\n(?=\S+:\s+; @)', s)
for f in funcs:
    m = re.match(r'(\S+):\s+; @', f)
    if not m or pat not in m.group(1): continue
    name = m.group(1)
    lines = f.split('\n')
    labels = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r'^(\.LBB\d+_\d+):', l))}
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            body = lines[labels[mm.group(1)]:i + 1]
            ins = [b.split()[0] for b in body if b.startswith('\t') and not b.strip().startswith((';', '.'))]
            if best is None or ins.count('v_add_f64') > best.count('v_add_f64'): best = ins
    info = re.search(r'; TotalNumSgprs: (\d+)\n; NumVgprs: (\d+)\n.*?; ScratchSize: (\d+)', f, re.S)
    if best is None: continue
    c = Counter(best)
    cls = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    lanes = sum(v for k, v in c.items() if k in ('v_readlane_b32', 'v_writelane_b32'))
    print(f"{name[:60]:60s} loop {len(best):4d} ins: VALU {cls('v_') - lanes:4d} (add {c['v_add_f64']}, mul {c['v_mul_f64']}, fma {c['v_fma_f64']}) SALU {cls('s_'):3d} "
          f"LDS {cls('ds_'):2d} scratch {cls('scratch_') + cls('buffer_'):2d} lane-moves {lanes:2d} | sgpr {info.group(1) if info else '?'} vgpr {info.group(2) if info else '?'} scratch {info.group(3) if info else '?'}")
