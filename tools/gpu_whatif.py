#!/usr/bin/env python3
"""What-if profile of the encode kernel's roles on the bench workload: a -DFQSX_WHATIF build (tools/libfqsx_whatif.so,
`python tools/gpu_whatif.py build`) sleeps `units` x ~0.2 us per event of ONE role -- resolving wave (1: per chunk iteration),
models wave (2: per 64 queue entries), range-coder wave (3: per 64 entries), scout waves (4: per chunk made), read-head wave
(5: per read) -- and the file's slowdown per microsecond added says how much of that role's time is on the critical path
(1 = all of it, 0 = slack).  usage: python tools/gpu_whatif.py [units=4]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libfqsx_whatif.so")
if sys.argv[1:2] == ["build"]:
    import __graft_entry__ as g
    print(g.build_hip(True, extra_flags=["-DFQSX_WHATIF"], out=LIB))
    sys.exit(0)
import numpy as np, torch
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import read_id, synth_reads
units = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# [reads genome gs seed reps roles]: another file (e.g. 10000000 300000000 300 19 1 1,4,5,6,8 = the bench's large-file row)
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
genome = int(sys.argv[3]) if len(sys.argv) > 3 else 7_500_000
gs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 2
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
roles = [int(x) for x in sys.argv[7].split(",")] if len(sys.argv) > 7 else [1, 2, 3, 4, 5, 6, 7, 8]
reads = synth_reads(n_reads, 150, genome, seed)
rec = hp.Records([read_id(i) for i in range(len(reads))], reads, reads)
header = hp.make_header(64, "se_sorted", gs)
groups = None
if n_reads > 2_000_000:
    from fqsqueezer_amd.codec import sort_order
    groups = sort_order(reads.reshape(-1), np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(150), device=0)
dev = []
for idx in hp.form_blocks(rec, "se_sorted", groups=groups):
    bases, off = hp.block_arrays(rec, idx)
    dev.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off))
nb = [int(o[-1]) for _, _, o in dev]


def one(role):
    os.environ["FQSX_WHATIF"] = f"{role},{units if role else 0}"
    best = None
    for rep in range(reps):
        c = DnaCodec(header, device=0, lib_path=LIB)
        marks = [time.perf_counter()]
        tot = 0
        for g, (d_b, d_o, off) in enumerate(dev):
            tot += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
            marks.append(time.perf_counter())
        c.close()
        r = {"file_s": marks[-1] - marks[0], "warm_s": marks[70] - marks[0], "steady_s": marks[256] - marks[100], "dna_bytes": tot}
        if best is None or r["file_s"] < best["file_s"]:
            best = r
    return best


names = {0: "none", 1: "resolving wave (per chunk iteration)", 2: "models wave (per 64 entries)", 3: "range coder (per 64 entries)",
         4: "scout waves (per chunk made, each of the three)", 5: "read-head wave (per read)", 6: "local-table inserter (per batch)",
         7: "read-head wave, once per launch (calibration: all of it is critical path)", 8: "scout waves (per sweep pass of a chunk)"}
base = one(0)
chunks_per_read = 2.2
n_reads = len(reads)
out = {"file": [n_reads, genome, gs, seed], "units": units, "us_per_event": round(units * 0.213, 3), "baseline": base, "roles": {}}
for role in roles:
    r = one(role)
    assert r["dna_bytes"] == base["dna_bytes"]
    # events per worker over the file (rough): chunk iterations ~ reads/64 workers * 2.2 ...; report the raw slowdowns, and the
    # slowdown per read per worker in us beside the delay a read got (per worker: reads / 64)
    per_read_us = lambda k: (r[k] - base[k]) * 1e6
    rw = n_reads / 64
    out["roles"][names[role]] = {
        "file_s": round(r["file_s"], 4), "slowdown_file_pct": round(100 * (r["file_s"] / base["file_s"] - 1), 2),
        "slowdown_warm_pct": round(100 * (r["warm_s"] / base["warm_s"] - 1), 2), "slowdown_steady_pct": round(100 * (r["steady_s"] / base["steady_s"] - 1), 2),
        "added_wall_us_per_read_of_a_worker": round(per_read_us("file_s") / rw, 3),
        "steady_added_wall_us_per_read_of_a_worker": round(per_read_us("steady_s") / (rw * 156 / 256), 3)}
print(json.dumps(out))
