#!/usr/bin/env python
"""Static look at one kernel of a hipcc -save-temps .s file: register / scratch footprint, instruction counts, and the basic blocks
that hold MFMAs (lines, accumulator moves, scratch accesses per block).    python tools/isa_blocks.py <file.s> <kernel substring> [dump block #]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
dump = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    name = m.group(1)
    if want not in name:
        continue
    try:
        body = s[m.end():s.index('.end_amdhsa_kernel', m.end())]
    except ValueError:
        continue
    code = body[:body.index('s_endpgm')] if 's_endpgm' in body else body
    cnt = lambda pat: len(re.findall(pat, code))
    print(name[:70], 'mfma', cnt(r'v_mfma'), 'ds_read', cnt(r'ds_read'), 'v_mov', cnt(r'v_mov_b'), 'accvgpr', cnt(r'v_accvgpr'), 'scratch', cnt(r'scratch_'),
          'branches', cnt(r's_cbranch'), 'lines', code.count('\n'))
    for key in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_private_segment_fixed_size', '.amdhsa_next_free_sgpr'):
        mm = re.search(re.escape(key) + r'\s+(\d+)', body)
        if mm:
            print('   ', key, mm.group(1))
    blocks, cur, lab = [], [], 'entry'
    for l in code.split('\n'):
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append((lab, cur))
            lab, cur = l.split(':')[0], []
        elif l.strip() and not l.strip().startswith(';'):
            cur.append(l)
    blocks.append((lab, cur))
    k = 0
    hist = {}
    for lab, b in blocks:
        nm = sum('v_mfma' in x for x in b)
        if nm:
            hist.setdefault((nm, len(b), sum('v_accvgpr' in x for x in b), sum('scratch_' in x for x in b), sum('s_nop' in x for x in b)), []).append(lab)
            if k == dump:
                print(lab)
                print('\n'.join(b))
            k += 1
    for key, labs in sorted(hist.items()):
        print('   blocks with (mfma, lines, accvgpr, scratch, s_nop) =', key, 'x', len(labs))
