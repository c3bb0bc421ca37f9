"""Instruction mix of named basic blocks of one kernel in a device assembly file, priced with the measured issue costs of
tools/ubench_coissue.hip (cycles per wave instruction on one SIMD with two waves):
python tools/asm_mix.py file.s <mangled-name-substring> .LBB49_439 .LBB49_441 ..."""
import re, sys, collections
COST = {"mfma": 16.7, "f64_fma": 5.8, "f64_add": 5.8, "f64_mul": 5.2, "dpp": 4.45, "valu_other": 2.6, "ds_bpermute": 4.0, "ds_read": 1.0, "ds_write": 1.0, "salu": 1.0, "wait": 0.0, "other": 1.0}
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
want = set(sys.argv[3:])
cur = None
mix = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1); continue
    if cur not in want: continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if 'v_mfma' in op: k = "mfma"
    elif op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): k = "f64_fma"
    elif op.startswith("v_add_f64"): k = "f64_add"
    elif op.startswith("v_mul_f64"): k = "f64_mul"
    elif op.endswith("_dpp") or "dpp" in t: k = "dpp"
    elif op.startswith("v_"): k = "valu_other"
    elif op.startswith("ds_bpermute"): k = "ds_bpermute"
    elif op.startswith("ds_read"): k = "ds_read"
    elif op.startswith("ds_write"): k = "ds_write"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"): k = "wait"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    mix[cur][k] += 1
    mix[cur]["op:" + op] += 1
tot = collections.Counter()
for b in sys.argv[3:]:
    c = mix[b]
    cyc = sum(COST[k] * v for k, v in c.items() if not k.startswith("op:"))
    print(b, {k: v for k, v in c.items() if not k.startswith("op:")}, f"priced {cyc:.0f} cycles")
    tot.update(c)
cyc = sum(COST[k] * v for k, v in tot.items() if not k.startswith("op:"))
print("sum", {k: v for k, v in tot.items() if not k.startswith("op:")}, f"priced {cyc:.0f} cycles")
print("valu_other ops:", {k[3:]: v for k, v in tot.items() if k.startswith("op:v_") and k[3:] not in ("v_fma_f64", "v_add_f64", "v_mul_f64") and "mfma" not in k and "dpp" not in k})
