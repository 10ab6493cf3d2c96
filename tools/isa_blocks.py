"""Debug aid: instruction mix per basic block of one kernel in a hipcc -S listing.  usage: isa_blocks.py file.s kernel_substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*%s\S*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.start():s.index('.Lfunc_end', m.start())]
blocks = []; cur = ['entry', []]; blocks.append(cur)
for l in body.splitlines():
    t = l.strip()
    if re.match(r'\.LBB\d+_\d+:', t): cur = [t.split(':')[0], []]; blocks.append(cur)
    elif t and not t.startswith((';', '.')) and not t.endswith(':'): cur[1].append(t.split()[0])
for name, ins in blocks:
    c = collections.Counter(ins)
    mf = sum(v for k, v in c.items() if 'mfma' in k)
    if len(ins) > 20:
        print(name, len(ins), 'mfma', mf, 'exp', c.get('v_exp_f32_e32', 0), 'valu', sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k))
    if mf >= 8:
        key = lambda x: 'M' if 'mfma' in x else 'E' if 'exp' in x else 'd' if x.startswith('ds_') else 'w' if 'waitcnt' in x else 'n' if 'nop' in x else 'b' if 'buffer' in x else 's' if x.startswith('s_') else 'v'
        print(''.join(key(x) for x in ins))
        print({k: v for k, v in c.most_common(14)})
