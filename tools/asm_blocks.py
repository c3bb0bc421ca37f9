"""Per-basic-block instruction mix of one kernel in a device assembly file (hipcc -S --cuda-device-only):
python tools/asm_blocks.py file.s <mangled-name-substring> [min_mfma]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 50
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ': ' in l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
blk = None; blocks = []
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blk = dict(name=m.group(1), n=0, mfma=0, dsr=0, dsw=0, sl=0, ss=0, gl=0, gs=0, lane=0, valu=0, bar=0, dpp=0, acc=0)
        blocks.append(blk); continue
    if blk is None: continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    blk['n'] += 1
    if 'v_mfma' in t: blk['mfma'] += 1
    elif t.startswith('ds_read'): blk['dsr'] += 1
    elif t.startswith('ds_write'): blk['dsw'] += 1
    elif t.startswith('scratch_load'): blk['sl'] += 1
    elif t.startswith('scratch_store'): blk['ss'] += 1
    elif t.startswith('global_load'): blk['gl'] += 1
    elif t.startswith('global_store'): blk['gs'] += 1
    elif 'v_readlane' in t or 'v_writelane' in t: blk['lane'] += 1
    elif t.startswith('s_barrier'): blk['bar'] += 1
    elif t.startswith('v_accvgpr'): blk['acc'] += 1
    elif t.startswith('v_'):
        blk['valu'] += 1
        if 'dpp' in t: blk['dpp'] += 1
tot = dict(sl=sum(b['sl'] for b in blocks), ss=sum(b['ss'] for b in blocks), mfma=sum(b['mfma'] for b in blocks))
print("total", tot)
for b in blocks:
    if b['mfma'] >= mn: print(b)
