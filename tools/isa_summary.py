"""Compact per-basic-block summary of a kernel's gfx950 ISA (M = MFMA, D = ds_read_b128, G = LDS-DMA, waits and
branches verbatim): python tools/isa_summary.py file.s <mangled-name-substring>"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l.split(':')[0])
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
blk = 'entry'; blocks = collections.OrderedDict({blk: []})
for l in s[start + 1:end]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t): blk = t; blocks[blk] = []; continue
    if t and not t.startswith(';') and not t.startswith('.'): blocks[blk].append(t)
names = {'v_mfma_f32_16x16x32_bf16': 'M', 'v_mfma_f32_16x16x32_f16': 'M', 'ds_read_b128': 'D', 'global_load_lds_dwordx4': 'G', 's_barrier': 'BAR'}
for b, ops in blocks.items():
    out = []
    for o in ops:
        k = o.split()[0]
        if k in names: k = names[k]
        elif k.startswith('s_waitcnt') or k.startswith('s_cbranch') or k.startswith('s_branch'): k = '[' + o.split(';')[0].strip() + ']'
        elif k.startswith('global_store'): k = 'ST'
        elif k.startswith('global_load') or k.startswith('buffer_load'): k = 'LD'
        elif k.startswith('scratch_'): k = 'SCR'
        elif k.startswith('v_'): k = 'v'
        elif k.startswith('s_'): k = 's'
        else: k = '?'
        out.append(k)
    rl = []
    for k in out:
        if rl and rl[-1][0] == k: rl[-1][1] += 1
        else: rl.append([k, 1])
    print(b, len(ops), 'ops:', ' '.join(f"{k}{n if n > 1 else ''}" for k, n in rl))
