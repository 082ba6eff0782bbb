#!/usr/bin/env python
"""Developer tool: instruction census of the hottest loop of a kernel in a hipcc -S listing.
   python tools/isa_loop_census.py <file.s> <kernel-name-substring> [nth-largest-loop]"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def ninstr(a, b):
    return sum(1 for l in body[a:b + 1] if re.match(r"^\s+[vsdgb]\w+_", l))
loops.sort(key=lambda ab: -ninstr(*ab))
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
print("loops (start line, end line, instructions): %s" % [(start + a, start + b, ninstr(a, b)) for a, b in loops[:6]])
a, b = loops[nth]
cls = collections.Counter()
ops = collections.Counter()
for l in body[a:b + 1]:
    m = re.match(r"^\s+(\w+)", l)
    if not m or l.strip().startswith(";") or l.strip().startswith("."):
        continue
    op = m.group(1)
    if not re.match(r"^[vsdgb]\w*_", op):
        continue
    ops[op] += 1
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): c = "global load"
    elif op.startswith("global_store") or op.startswith("buffer_store"): c = "global store"
    elif op.startswith("ds_"): c = "LDS"
    elif op.startswith("s_waitcnt"): c = "s_waitcnt"
    elif op.startswith("s_"): c = "scalar"
    elif op in ("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_mov_b64"): c = "v_mov"
    elif op.startswith("v_cndmask") or op.startswith("v_cmp") or op.startswith("v_cmpx"): c = "compare / select"
    elif op.startswith("v_pk_"): c = "packed f32 (2 per instruction)"
    elif op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_mul_f32") or op.startswith("v_add_f32") or op.startswith("v_sub_f32") or op.startswith("v_mac"): c = "f32 mul/add/fma"
    elif op.startswith("v_rcp") or op.startswith("v_sqrt") or op.startswith("v_rsq") or op.startswith("v_div") or op.startswith("v_frexp") or op.startswith("v_ldexp"): c = "rcp / sqrt / division sequence"
    elif op.startswith("v_cvt") or op.startswith("v_floor") or op.startswith("v_trunc"): c = "convert"
    elif re.match(r"v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|add3|lshl_add|mad_u|mul_lo|mul_hi|mad_i|add_co|addc|sub_co|subb|min_[iu]|max_[iu]|not|bfi|lshlrev|lshrrev|ashrrev)", op) and not op.endswith("_f32"): c = "integer / address"
    else: c = "other vector (%s)" % op
    cls[c] += 1
tot = sum(cls.values())
print("loop at lines %d-%d: %d instructions" % (start + a, start + b, tot))
for c, n in cls.most_common():
    print("  %-40s %4d" % (c, n))
print("top opcodes: " + ", ".join("%s %d" % kv for kv in ops.most_common(24)))
