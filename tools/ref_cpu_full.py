#!/usr/bin/env python3
"""Times the unmodified reference (oracle/_ref/fqs-1.1) on the FULL bench workload (1 M x 100 bp, -gs 5)
at several -t values, plus a 1000-read run of the same command line to expose its fixed start-up cost.
CPU only; run on the GPU box so the numbers sit beside bench.py's (tools/, not part of the default bench:
takes minutes)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqsqueezer_amd.synth import synth_quals, synth_reads, write_fastq

REF = os.path.join(ROOT, "oracle", "_ref", "fqs-1.1")


def run(fq, t, gs, td):
    cmd = [REF, "e", "-s", "-om", "s", "-t", str(t), "-gs", str(gs), "-qm", "n", "-im", "n", "-v", "0",
           "-tmp", os.path.join(td, "tmp_"), "-out", os.path.join(td, "o.fqs"), fq]
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t0


def main():
    n, L, G, gs = 1000000, 100, 5000000, 5
    reads = synth_reads(n, L, G, 2)
    out = {"cores": os.cpu_count(), "workload": f"{n}x{L}bp SE G={G} -om s -gs {gs} -qm n -im n", "runs": []}
    with tempfile.TemporaryDirectory(prefix="fqsx_ref_") as td:
        full, tiny = os.path.join(td, "full.fq"), os.path.join(td, "tiny.fq")
        write_fastq(full, reads, synth_quals(n, L, 2))
        write_fastq(tiny, reads[:1000], synth_quals(1000, L, 2))
        for t in [int(x) for x in (sys.argv[1:] or ["64", "8"])]:
            s = run(tiny, t, gs, td)
            d = run(full, t, gs, td)
            out["runs"].append({"t": t, "startup_s": round(s, 2), "full_s": round(d, 2),
                                "mbases_s_whole_process": round(n * L / d / 1e6, 3),
                                "mbases_s_post_startup": round(n * L / max(d - s, 1e-9) / 1e6, 3)})
            print(json.dumps(out["runs"][-1]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
