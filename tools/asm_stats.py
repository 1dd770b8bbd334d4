#!/usr/bin/env python3
"""Per-function statistics of an AMDGPU assembly file (hipcc -S --cuda-device-only): instructions, scratch accesses,
SGPR-spill lane moves, calls, flat accesses, waits, code bytes.  usage: tools/asm_stats.py file.s"""
import re, sys, collections
fn = None
st = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^([A-Za-z_][A-Za-z0-9_]*):", line)
    if m and not m.group(1).startswith("__hip_cuid"):
        fn = m.group(1)
        st[fn] = collections.Counter()
        continue
    if fn is None:
        continue
    t = line.strip()
    m = re.match(r"; codeLenInByte = (\d+)", t)
    if m:
        st[fn]["bytes"] = int(m.group(1))
    m = re.match(r"; NumVgprs: (\d+)", t)
    if m:
        st[fn]["vgpr"] = int(m.group(1))
    m = re.match(r"; ScratchSize: (\d+)", t)
    if m:
        st[fn]["scratch_bytes"] = int(m.group(1))
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    c = st[fn]
    c["instr"] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
    elif op.startswith("v_writelane"): c["writelane"] += 1
    elif op.startswith("v_readlane"): c["readlane"] += 1
    elif op.startswith("s_swappc"): c["calls"] += 1
    elif op.startswith("flat_"): c["flat"] += 1
    elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_"): c["vmem"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer_load"): c["smem"] += 1
cols = ["bytes", "instr", "vgpr", "scratch_bytes", "scratch", "writelane", "readlane", "calls", "flat", "waitcnt", "lds", "vmem", "smem"]
print("%-34s" % "function" + "".join("%10s" % c for c in cols))
for f, c in st.items():
    print("%-34s" % f[:34] + "".join("%10d" % c[k] for k in cols))
