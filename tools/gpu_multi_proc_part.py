#!/usr/bin/env python3
"""K independent compressions of the bench workload at once on one GPU, one PROCESS per file, every codec on its own partition
of the compute units (fqsx_dna_create_on_partition) -- the process form of bench.py's `concurrent_files` (threads of one process).
usage: python tools/gpu_multi_proc_part.py [K=4] [partitioned=1]"""
import json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(k, K, part, bar, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec
    from fqsqueezer_amd.synth import read_id, synth_reads
    n, L = 1_000_000, 150
    reads = synth_reads(n, L, 7_500_000, 2)
    rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
    header = hp.make_header(64, "se_sorted", 8)
    dev = []
    for idx in hp.form_blocks(rec, "se_sorted"):
        bases, off = hp.block_arrays(rec, idx)
        dev.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off))
    torch.cuda.synchronize()
    res = []
    for rep in range(2):   # (pass 0: warm-up)
        c = DnaCodec(header, device=0, partition=(k, K) if part else None)
        bar.wait()
        t0 = time.time()
        nb = 0
        marks = [t0]
        for g, (d_b, d_o, off) in enumerate(dev):
            nb += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
            marks.append(time.time())
        t1 = time.time()
        c.close()
        res = (t0, t1, nb, marks[100], marks[-1], sum(int(o[-1]) for (_, _, o) in dev[100:]))
        bar.wait()
    q.put((k, res))


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    part = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
    ctx = mp.get_context("spawn")
    bar, q = ctx.Barrier(K), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(k, K, part, bar, q)) for k in range(K)]
    for p in ps:
        p.start()
    out = [q.get() for _ in ps]
    for p in ps:
        p.join()
    t0 = min(r[0] for _, r in out); t1 = max(r[1] for _, r in out)
    n_bases = 150_000_000
    steady = sum(r[5] / (r[4] - r[3]) for _, r in out) / 1e6
    print(json.dumps({"processes": K, "partitioned": part, "value": round(K * n_bases / (t1 - t0) / 1e6, 3), "unit": "Mbases/s",
                      "steady_state_value": round(steady, 3), "identical_output": len({r[2] for _, r in out}) == 1, "dna_bytes": out[0][1][2]}))
