"""Run on the GPU box: HIP path vs golden .fqs fixtures (and the oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads, synth_quals, read_id

def golden_case(n, L, G, seed, fqs):
    reads = synth_reads(n, L, G, seed)
    rec = hp.Records([read_id(i) for i in range(n)], reads, synth_quals(n, L, seed))
    header, blocks = hp.parse_fqs(open(fqs, "rb").read())
    mode = {0: "se_original", 1: "se_sorted"}[header[5]]
    blks = hp.form_blocks(rec, mode)
    assert len(blks) == len(blocks), (len(blks), len(blocks))
    gpu = DnaCodec(header)
    t0 = time.time(); bad = 0
    for g, (idx, ref) in enumerate(zip(blks, blocks)):
        bases, off = hp.block_arrays(rec, idx)
        streams = gpu.encode_block(bases, off, g)
        for w, s in enumerate(streams):
            r = ref.streams[w][hp.STREAM_DNA]
            if s != r:
                bad += 1
                k = next((i for i in range(min(len(s), len(r))) if s[i] != r[i]), min(len(s), len(r)))
                if bad <= 4: print(f"  MISMATCH block {g} worker {w}: ours {len(s)} ref {len(r)} first diff at {k}")
    dt = time.time() - t0
    print(f"{os.path.basename(fqs)}: {len(blks)} blocks T={header[4]} {'OK' if not bad else 'FAIL(%d)' % bad} {dt:.2f}s {n*L/dt/1e6:.3f} Mbases/s", flush=True)
    return bad == 0

if __name__ == "__main__":
    gd = os.path.join(ROOT, "tests", "golden")
    ok = True
    for f in sorted(os.listdir(gd)):
        if f.startswith("c1_10k") and f.endswith(".fqs"):
            ok &= golden_case(10000, 100, 200000, 1, os.path.join(gd, f))
    sys.exit(0 if ok else 1)
