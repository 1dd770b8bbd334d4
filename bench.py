#!/usr/bin/env python3
"""bench.py -- throughput of the FQSX DNA hot path on MI355X.

Workload = the one BASELINE.json's metric is quoted on (bold row): 1 M x 150 bp synthetic
single-end reads, 20x coverage of a 7.5 Mbp random genome (seed 2), `-om s` (sorted), `-gs 8`, T logical workers
(default 64 = the reference CLI's maximum; T is part of the .fqs bitstream).  `--len 100 --genome 5000000 --gs 5`
gives BASELINE configs[1] (last round's default).  One *step* = one pass of the
DNA path over the whole file: fresh codec state, every reads block encoded in file order, the
per-worker range-coder streams copied back to the host.  Inputs (base bytes + read offsets of
every block) are resident in HBM before the timed region starts.

N > 1 (torch.distributed.run, one rank per GPU): ONE file sharded over the N GPUs with partitioned k-mer tables -- the layout
BASELINE.json's north_star and configs[3] name: logical worker w lives on rank w % N, the mailboxes travel in one RCCL
all-to-all per synchronisation phase, every rank holds 1/N of the s-/b-mer tables and reads the rest over xGMI through peer
mappings (fqsx_shard_attach + fqsx_shard_partition_tables + fqsx_shard_encode_block).  Strong scaling: `value` = the file's
bases / wall time.  The file is configs[3]-shaped (150 bp SE sorted at the reference's default geometry -gs 3100, k = 13/18/21/27,
16 GiB p-mer vector per rank): 2 M reads of a 50 Mbp genome unless --reads / --genome say otherwise.  `--replicas` gives the
other multi-GPU mode instead (N independent files, one per GPU, no collective: weak scaling); its rate also rides along in
the sharded line as `replicas`.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


# slots one cluster scan of the REFERENCE's tables reads, terminator included (measured there: 3.96).  Rounds 1-3
# put this library's own slot counters into the formula (linear probing at <= 50 % load: 1.8 slots per probe); since round 4 the
# tables are two-choice buckets that read eight slots per probe whatever they hold, so the library's counter no longer says
# anything about the algorithm -- the formula takes the reference's figure, as SURVEY 8d states it (39.8 B per probe).
REF_SLOTS_PER_PROBE = 3.96
R03_SLOTS_PER_PROBE, R03_SLOTS_PER_INSERT = 1.805, 1.611   # what this library's linear tables scanned (round 3 final, same file)


def algorithmic_bytes(st: dict, slots_per_probe: float = REF_SLOTS_PER_PROBE, slots_per_insert: float = REF_SLOTS_PER_PROBE) -> float:
    """B = sum_probes(24+4*slots) + sum_global_inserts(24+4*slots+4) + 8*siv_words
    + 24*ctx_slots + 2*40*models_updated   (reference structure sizes; data-dependent counters).  siv words = the words the
    ALGORITHM sweeps (the prefix scan's whole range, SURVEY: "8 B per word swept"): what the kernels read (`siv_words`) plus what
    the count index spares them (`siv_saved`) -- the same definition as rounds 1-2, whose kernels swept the range themselves."""
    probes = st["gprobe"] + st["lprobe"]
    return ((24.0 + 4.0 * slots_per_probe) * probes + (28.0 + 4.0 * slots_per_insert) * st["gins"] + 8.0 * (st["siv_words"] + st.get("siv_saved", 0))
            + 24.0 * st["ctx_slots"] + 80.0 * st["coded"])


def insert_bytes(st: dict, slots_per_insert: float = REF_SLOTS_PER_PROBE) -> float:
    return (28.0 + 4.0 * slots_per_insert) * st["gins"]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=None, help="default: 1 M (one GPU / replicas), 2 M (one file sharded over N > 1 GPUs)")
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=None, help="default: 7.5 Mbp (one GPU / replicas), 50 Mbp (sharded)")
    ap.add_argument("--gs", type=int, default=None, help="default: 8 (one GPU / replicas), 3100 = the reference's default geometry (sharded)")
    ap.add_argument("--threads", type=int, default=64, help="logical workers T (header byte, <=255)")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the extra host-buffer (PCIe-inclusive) pass")
    ap.add_argument("--no-t255", action="store_true", help="skip the extra pass with 255 workers")
    ap.add_argument("--concurrent", type=int, default=4, help="extra pass: this many codec instances at once (0/1 = skip)")
    ap.add_argument("--no-rows", action="store_true", help="skip the extra passes over the other rows (decoder, quality, PE, original order)")
    ap.add_argument("--no-large", action="store_true", help="skip the extra pass over the large file (10 M x 150 bp, G = 300 Mbp, -gs 300: tables of GBs)")
    ap.add_argument("--chunked-tables", action="store_true",
                    help="the capacity mode (fqsx_dna_use_chunked_tables): k-mer tables in per-sub-table chunks, growth without old + new side by side")
    ap.add_argument("--partition", action="store_true",
                    help="with --sharded: the k-mer tables partitioned over the ranks (each GPU holds 1/N of them, look-ups of the "
                         "rest over xGMI through peer mappings; fqsx_shard_partition_tables) instead of a replica on every rank")
    ap.add_argument("--sharded", action="store_true",
                    help="ONE file sharded over the N GPUs (workers w %% N on rank w, RCCL all-to-all of the mailboxes; "
                         "fqsqueezer_amd/sharded.py): strong scaling.  The default for N > 1, with --partition")
    ap.add_argument("--replicas", action="store_true", help="N > 1: N independent files, one per GPU (weak scaling) instead of one sharded file")
    ap.add_argument("--no-partition", action="store_true", help="sharded: table replicas on every rank instead of partitioned tables")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not a.replicas:   # the driver's `--gpus N` line: the north star's layout
        a.sharded = True
        a.partition = not a.no_partition
    if a.sharded and world > 1:
        a.reads = a.reads or 2_000_000; a.genome = a.genome or 50_000_000; a.gs = a.gs or 3100
    else:
        a.reads = a.reads or 1_000_000; a.genome = a.genome or 7_500_000; a.gs = a.gs or 8
    emu = os.environ.get("FQSX_BENCH_EMU")   # tests only: the launcher path of the sharded line on CPUs (gloo + the emulated kernels)
    if emu:
        if not a.sharded:
            raise SystemExit("FQSX_BENCH_EMU only rehearses the sharded multi-rank line (tests/test_multirank_cpu.py)")
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if world > 1 and os.environ.get("FQSX_BENCH_GUARD", "1") == "1":
            return guarded_sharded_main(a, rank, local_rank, world, emu_lib=emu)
        return sharded_main(a, rank, local_rank, world, emu_lib=emu)

    import torch
    import torch.distributed as dist
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec
    from fqsqueezer_amd.synth import read_id, synth_quals, synth_reads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the FQSX DNA path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or a.sharded:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29543")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if a.sharded and world > 1:
        return guarded_sharded_main(a, rank, local_rank, world)
    if a.sharded:
        return sharded_main(a, rank, local_rank, world)

    # ---- synthetic input, host preparation (not timed): binning + sort + block formation
    seed = 2 + rank
    reads = synth_reads(a.reads, a.len, a.genome, seed)
    rec = hp.Records([read_id(i) for i in range(a.reads)], reads, reads if a.no_cpu_baseline else synth_quals(a.reads, a.len, seed))   # (qualities: only the reference CPU leg reads them)
    header = hp.make_header(a.threads, "se_sorted", a.gs)
    groups = None
    if a.reads > 2_000_000:   # large files: the sorted order from the GPU pre-pass (fqsx_sort_order) instead of numpy's lexsort
        from fqsqueezer_amd.codec import sort_order
        groups = sort_order(reads.reshape(-1), np.arange(a.reads + 1, dtype=np.uint64) * np.uint64(a.len), device=local_rank)
    blocks = hp.form_blocks(rec, "se_sorted", groups=groups)
    dev_blocks = []
    for idx in blocks:
        bases, off = hp.block_arrays(rec, idx)
        d_b = torch.from_numpy(np.ascontiguousarray(bases)).cuda()
        d_o = torch.from_numpy(off.view(np.int64)).cuda()
        dev_blocks.append((d_b, d_o, off))
    n_bases = int(a.reads) * int(a.len)
    torch.cuda.synchronize()

    capacity = {}     # table occupancy / device memory at the end of the most recent pass (fqsx_dna_capacity)
    block_done = []   # host clock after every block of the most recent pass (a block call returns when its streams are back)

    def one_step(profile: bool = False):
        codec = DnaCodec(header, device=local_rank, chunked_tables=a.chunked_tables)
        if profile:
            codec.set_profiling(True)
        out_bytes = 0
        block_done.clear()
        block_done.append(time.perf_counter())
        for g, (d_b, d_o, off) in enumerate(dev_blocks):
            out_bytes += codec.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
            block_done.append(time.perf_counter())
        res = (out_bytes, codec.stats() if profile else None, codec.kernel_times() if profile else None)
        capacity.clear()
        capacity.update(codec.capacity())
        codec.close()
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    dna_bytes = 0
    for _ in range(a.steps):
        dna_bytes, _, _ = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = world * n_bases * a.steps / elapsed / 1e6
    # rate over the blocks past the reference's warm-up schedule (calc_no_synchronizations, application.h:85-92: from
    # block 100 on a block is one segment), from the last timed pass -- what a file much longer than this one runs at
    steady = None
    if len(dev_blocks) > 110 and len(block_done) == len(dev_blocks) + 1:
        sb = sum(int(off[-1]) for (_, _, off) in dev_blocks[100:])
        steady = {"value": round(sb / (block_done[-1] - block_done[100]) / 1e6, 4), "unit": "Mbases/s",
                  "blocks": f"100..{len(dev_blocks) - 1}", "warmup_blocks_value": round(sum(int(off[-1]) for (_, _, off) in dev_blocks[:70]) / (block_done[70] - block_done[0]) / 1e6, 4),
                  "note": "this rank, same timed pass: blocks >= 100 (one synchronisation segment per block) vs blocks 0..69 (30 segments per block, two reads per worker and segment)"}

    cap_line = dict(capacity, note="k-mer tables at the end of the file: distinct s-/b-mers stored, slots allocated (8-byte slots, all owners), "
                                   "device bytes held / peak (old + new table alive during a growth), growth events")
    # ---- kernel-level measurement pass (HIP events around every launch on the codec's stream; untimed)
    _, st, kt = one_step(profile=True)
    alg = algorithmic_bytes(st)
    enc_s, ins_s = kt["encode_ms"] / 1e3, kt["insert_ms"] / 1e3
    dominant = "k_encode_se_sorted" if enc_s >= ins_s else "k_insert_phase"
    dom_s = max(enc_s, ins_s)
    dom_launches = kt["encode_launches"] if enc_s >= ins_s else kt["insert_launches"]
    # bytes attributable to the dominant kernel
    ins_bytes = insert_bytes(st)
    dom_bytes = alg - ins_bytes if dominant == "k_encode_se_sorted" else ins_bytes
    r03_bytes = algorithmic_bytes(st, R03_SLOTS_PER_PROBE, R03_SLOTS_PER_INSERT) - insert_bytes(st, R03_SLOTS_PER_INSERT)   # (encode kernel, rounds 1-3's slot counts)
    achieved = dom_bytes / dom_s / 1e9 if dom_s > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": None, "kernel": dominant,
                "launches": dom_launches, "avg_launch_ms": round(dom_s * 1e3 / max(1, dom_launches), 4),
                "algorithmic_bytes_per_launch": round(dom_bytes / max(1, dom_launches), 1),
                "encode_kernel_s": round(enc_s, 4), "insert_kernel_s": round(ins_s, 4),
                "bytes_the_kernel_reads": {"achieved": round((dom_bytes - 8.0 * st.get("siv_saved", 0)) / dom_s / 1e9, 3), "unit": "GB/s",
                                           "frac": round((dom_bytes - 8.0 * st.get("siv_saved", 0)) / dom_s / 1e9 / 8000.0, 6),
                                           "per_launch": round((dom_bytes - 8.0 * st.get("siv_saved", 0)) / max(1, dom_launches), 1),
                                           "note": "the same formula without the p-mer-vector words the count index spares the kernel (siv_saved): what the kernel asks the "
                                                   "memory system for; `achieved` / `frac` above follow SURVEY 8d literally (the words the ALGORITHM sweeps)"},
                "with_round_3_slot_counts": {"frac": round(r03_bytes / max(enc_s, 1e-9) / 1e9 / 8000.0, 6), "per_launch": round(r03_bytes / max(1, kt["encode_launches"]), 1),
                                             "note": "encode kernel; 1.8 slots per probe as this library's linear tables scanned in round 3 (that round's line: 0.006872), for "
                                                     "comparison across rounds; `achieved` / `frac` use the reference's 3.96 slots per probe (SURVEY 8d's 39.8 B per probe)"},
                "probes_per_s": round((st["gprobe"] + st["lprobe"]) / max(enc_s, 1e-9), 1),
                "siv_words_per_base": {"algorithm": round((st["siv_words"] + st.get("siv_saved", 0)) / max(1, st["bases"]), 2),
                                       "read_by_the_kernels": round(st["siv_words"] / max(1, st["bases"]), 2)}}

    # ---- PCIe-inclusive rate: the same pass through the host-buffer entry point (H2D copy per block; untimed run)
    pcie = None
    if world == 1 and not a.no_pcie:
        host_blocks = [hp.block_arrays(rec, idx) for idx in blocks]
        codec = DnaCodec(header, device=local_rank)
        t1 = time.perf_counter()
        for g, (bases, off) in enumerate(host_blocks):
            codec.encode_block(bases, off, g)
        pcie = round(n_bases / (time.perf_counter() - t1) / 1e6, 4)
        codec.close()

    # the same file at the bitstream's maximum worker count (T = 255; reference CLI caps -t at 64): extra, untimed-region info
    t255 = None
    if world == 1 and not a.no_t255 and a.threads != 255:
        h255 = hp.make_header(255, "se_sorted", a.gs)
        codec = DnaCodec(h255, device=local_rank)
        t1 = time.perf_counter()
        nb = 0
        for g, (d_b, d_o, off) in enumerate(dev_blocks):
            nb += codec.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
        dt = time.perf_counter() - t1
        codec.close()
        t255 = {"value": round(n_bases / dt / 1e6, 4), "unit": "Mbases/s", "bits_per_base": round(8.0 * nb / n_bases, 5),
                "note": "same file with header T=255 (valid .fqs, decodable by fqs d; not producible by the reference CLI)"}

    # K independent codec instances at once on this GPU (own tables, own stream, one host thread each): one file with
    # T = 64 occupies 64 of the 256 CUs, so a GPU has room for several files.  Extra information, not `value`.
    conc = None
    if world == 1 and a.concurrent > 1:
        import threading

        def conc_pass(partitioned):
            outs = [0] * a.concurrent
            marks = [None] * a.concurrent

            def run(k):
                c = DnaCodec(header, device=local_rank, partition=(k, a.concurrent) if partitioned else None)
                nb = 0
                tm = [time.perf_counter()]
                for g, (d_b, d_o, off) in enumerate(dev_blocks):
                    nb += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
                    tm.append(time.perf_counter())
                outs[k] = nb
                marks[k] = tm
                c.close()

            th = [threading.Thread(target=run, args=(k,)) for k in range(a.concurrent)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            r = {"value": round(a.concurrent * n_bases / dt / 1e6, 4), "identical_output": len(set(outs)) == 1 and outs[0] == dna_bytes}
            if len(dev_blocks) > 110:   # aggregate rate while every instance is past its block 100
                sb = sum(int(off[-1]) for (_, _, off) in dev_blocks[100:])
                r["steady_state_value"] = round(sum(sb / (m[-1] - m[100]) for m in marks) / 1e6, 4)
            return r

        part, free = conc_pass(True), conc_pass(False)
        # the same with one PROCESS per file (tools/gpu_multi_proc_part.py: own HIP queues and runtime locks per file; the threads
        # above share this process's), on CU partitions as well
        procs = None
        if a.reads == 1_000_000 and a.len == 150 and a.genome == 7_500_000 and a.gs == 8 and a.threads == 64 and a.concurrent <= 4:
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_multi_proc_part.py"), str(a.concurrent), "1"],
                                   capture_output=True, text=True, timeout=600)
                procs = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
                if "value" in procs:
                    procs["speedup_over_one_file"] = round(procs["value"] / value, 3)
            except Exception as e:   # noqa: BLE001 -- extra information only
                procs = {"error": str(e)[:300]}
        conc = {"instances": a.concurrent, "value": part["value"], "unit": "Mbases/s", "identical_output": part["identical_output"] and free["identical_output"],
                "steady_state_value": part.get("steady_state_value"), "speedup_over_one_file": round(part["value"] / value, 3),
                "unpartitioned": free, "one_process_per_file": procs,
                "note": "independent compressions of the workload file running concurrently on one GPU as threads of this process, each codec on "
                        "its own partition of the compute units (fqsx_dna_create_on_partition: CU-masked stream; one file occupies T of the 256 CUs); "
                        "`unpartitioned` = the same with plain streams (the files' kernels queue behind each other for CUs); aggregate rates"}

    # HBM traffic of the dominant kernel from a separate rocprofv3 --pmc pass (profiles/traffic.json), if recorded
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            if tj.get("workers_T") == a.threads and tj.get("reads") == a.reads and tj.get("len") == a.len:
                traffic = tj.get("bytes_per_launch")
                roofline["traffic_source"] = ("profiles/traffic.json: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this "
                                              "workload on the build named there (" + str(tj.get("build", "?"))[:60] + "); NOT measured by this run")
        except Exception:
            traffic = None
    roofline["traffic"] = traffic

    # ---- the rows beside the headline path (other dna_modes): 300 k x 150 bp each, extras of the N=1 line
    rows = None
    if world == 1 and not a.no_rows:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_rows
        rows = bench_rows.measure(300_000, a.len, a.threads, local_rank)
        # HBM traffic of the rows' kernels from separate rocprofv3 --pmc passes of tools/bench_rows.py (profiles/r0N_rows_traffic.json), if recorded
        for tf2 in ("r04_rows_traffic.json", "r03_rows_traffic.json"):
            tp = os.path.join(ROOT, "profiles", tf2)
            if not os.path.exists(tp):
                continue
            try:
                tk = json.load(open(tp))
                for row, kern in (("decode", "k_decode_se_sorted"), ("sorted", "k_encode_se_sorted"), ("original_order", "k_encode_se_orig"),
                                  ("pe_sorted", "k_encode_pe_sorted"), ("quality_o", "k_qual_encode"), ("quality_8", "k_qual_encode")):
                    kk = tk["kernels"].get(kern + (":" + row if kern == "k_qual_encode" else ""), tk["kernels"].get(kern))
                    if kk and isinstance(rows.get(row), dict) and "roofline" in rows[row]:
                        rows[row]["roofline"]["traffic"] = kk["bytes_per_launch"]
                        rows[row]["roofline"]["traffic_source"] = f"profiles/{tf2} ({tk.get('build', '?')}): separate --pmc FETCH_SIZE / WRITE_SIZE passes; NOT measured by this run" + \
                            ("" if kern + ":" + row in tk["kernels"] or kern != "k_qual_encode" else "; -qm o and -qm 8 launches mixed in one pass")
                break
            except Exception:   # noqa: BLE001 -- extra information only
                pass

    # ---- a file whose tables do not fit any cache (the headline file's are 0.5 GB, twice the Infinity Cache): 10 M x 150 bp of a
    # 300 Mbp genome at -gs 300 (k = 12/17/21/26; the configs[3]-shaped c19 file of tests/test_gpu_fullsize.py), one pass, inputs in HBM
    large = None
    if world == 1 and not a.no_large and a.reads == 1_000_000 and a.threads == 64:
        from fqsqueezer_amd.codec import sort_order
        lr = synth_reads(10_000_000, 150, 300_000_000, 19)
        lrec = hp.Records([b""] * 0, lr, lr)
        lgroups = sort_order(lr.reshape(-1), np.arange(len(lr) + 1, dtype=np.uint64) * np.uint64(150), device=local_rank)
        lorder = np.concatenate(lgroups)
        per = -(-len(lr) // 256)
        loff = np.arange(per + 1, dtype=np.uint64) * np.uint64(150)
        lblocks = []
        for i0 in range(0, len(lr), per):
            idx = lorder[i0:i0 + per]
            lblocks.append((torch.from_numpy(np.ascontiguousarray(lr[idx]).reshape(-1)).cuda(), torch.from_numpy(loff[:len(idx) + 1].view(np.int64).copy()).cuda(), loff[:len(idx) + 1]))
        del lgroups, lorder
        lc = DnaCodec(hp.make_header(64, "se_sorted", 300), device=local_rank)
        torch.cuda.synchronize()
        tm = [time.perf_counter()]
        lbytes = 0
        for g, (d_b, d_o, off) in enumerate(lblocks):
            lbytes += lc.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
            tm.append(time.perf_counter())
        lcap = lc.capacity()
        lc.close()
        nb_l = 1.5e9
        large = {"value": round(nb_l / (tm[-1] - tm[0]) / 1e6, 4), "unit": "Mbases/s", "blocks_ge_100_value": round(sum(int(o[-1]) for (_, _, o) in lblocks[100:]) / (tm[-1] - tm[100]) / 1e6, 4),
                 "bits_per_base": round(8.0 * lbytes / nb_l, 5), "table_bytes_held": lcap["table_bytes_held"], "bytes_per_bmer": lcap["bytes_per_bmer"],
                 "device_bytes_peak": lcap["device_bytes_peak"], "growths": lcap["growths"],
                 "workload": "10000000x150bp SE, G=300000000 (seed 19), -om s -gs 300, T=64, 256 blocks of the sorted order; one timed pass, inputs resident in HBM"}
        del lblocks, lr, lrec

    # ---- CPU baseline on a bounded sample of the same workload (rank 0, N=1 only)
    cpu = None
    if not a.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(a, hp, reads, rec)

    line = {
        "metric": "Mbases/s compressed (DNA stream, SE sorted)", "value": round(value, 4), "unit": "Mbases/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{a.reads}x{a.len}bp SE, G={a.genome} (seed 2+rank), -om s -gs {a.gs} -qm n -im n",
                   "workers_T": a.threads, "blocks": len(blocks), "per_gpu": "one independent file per GPU" + (" (--replicas)" if world > 1 else "")},
        "bits_per_base": round(8.0 * dna_bytes / n_bases, 5), "steady_state": steady, "pcie_inclusive_mbases_s": pcie, "workers_255": t255, "concurrent_files": conc,
        "capacity": cap_line, "large_file": large, "other_rows": rows, "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def sharded_main(a, rank, local_rank, world, emu_lib=None, replicas=None, emit=True):
    """ONE file (seed 2) over the N GPUs, strong scaling: logical worker w on rank w % N, per synchronisation phase an all-reduce of
    the count matrix, the three mailboxes in one grouped all-to-all and one all-gather (RCCL on the codec's stream, inside
    libfqsx.so: fqsx_shard_encode_block); with --partition (the default for N > 1) every rank holds 1/N of the k-mer tables and
    reads the rest through peer mappings.  The blocks are resident in HBM on every rank before the timed region.  Streams are
    bit-identical to the one-GPU run's (tests/test_sharded_cpu.py, tests/test_gpu_sharded.py).  emu_lib: tests only (see main)."""
    import torch
    import torch.distributed as dist
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec
    from fqsqueezer_amd.sharded import NativeShardedDnaCodec
    from fqsqueezer_amd.synth import read_id, synth_reads
    gpu = emu_lib is None
    reads = synth_reads(a.reads, a.len, a.genome, 2)
    rec = hp.Records([read_id(i) for i in range(a.reads)], reads, reads)
    header = hp.make_header(a.threads, "se_sorted", a.gs)
    groups = None
    if gpu and a.reads > 2_000_000:   # large files: the sorted order from the GPU pre-pass (fqsx_sort_order)
        from fqsqueezer_amd.codec import sort_order
        groups = sort_order(reads.reshape(-1), np.arange(a.reads + 1, dtype=np.uint64) * np.uint64(a.len), device=local_rank)
    dev_blocks = []   # the blocks resident in the codec's memory space (HBM) before the timed region, as in the unsharded bench
    for idx in hp.form_blocks(rec, "se_sorted", groups=groups):
        bases, off = hp.block_arrays(rec, idx)
        t_b, t_o = torch.from_numpy(np.ascontiguousarray(bases)), torch.from_numpy(off.view(np.int64))
        dev_blocks.append((t_b.cuda(), t_o.cuda(), off) if gpu else (t_b, t_o, off))
    n_bases = int(a.reads) * int(a.len)
    traffic, cap, info = {}, {}, {}
    comm = [None]   # the process's RCCL communicator: made with the first codec, rebound to every later one (ncclCommInitRank takes seconds)

    def sync():
        dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    def one_step(profile=False):
        if not gpu:
            c = NativeShardedDnaCodec(header, rank, world, lib_path=emu_lib, transport="torch", partition=a.partition)
        elif comm[0] is None:
            ids = [NativeShardedDnaCodec.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            c = NativeShardedDnaCodec(header, rank, world, device=local_rank, transport="rccl", id_bytes=ids[0], partition=a.partition)
            comm[0] = c.detach_comm()
            info.update(c.rccl_info())
        else:
            c = NativeShardedDnaCodec(header, rank, world, device=local_rank, transport="rccl", comm=comm[0], partition=a.partition)
        if profile:
            c.codec.set_profiling(True)
        nb = 0
        for g, (d_b, d_o, off) in enumerate(dev_blocks):
            nb += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g)
        traffic.update(c.traffic)
        cap.update(c.codec.capacity())
        info["partitioned"] = bool(c.partitioned)
        info["partition_note"] = c.partition_note
        res = (nb, c.codec.stats() if profile else None, c.codec.kernel_times() if profile else None)
        c.close()
        return res

    for _ in range(a.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    mine = 0
    for _ in range(a.steps):
        mine, _, _ = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    dd = "cuda" if gpu else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=dd)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    tot = torch.tensor([mine], dtype=torch.int64, device=dd)
    dist.all_reduce(tot)
    held = [None] * world
    dist.all_gather_object(held, {"rank": rank, "table_bytes_held": int(cap.get("table_bytes_held", 0)), "device_bytes_peak": int(cap.get("device_bytes_peak", 0)),
                                  "ranks_seen": info.get("ranks_seen")})
    value = n_bases * a.steps / elapsed / 1e6

    # ---- kernel-level pass (HIP events around every launch of this rank's stream; untimed): the dominant kernel's roofline on rank 0
    roofline = None
    if gpu:
        _, st, kt = one_step(profile=True)
        sync()
        if rank == 0:
            alg = algorithmic_bytes(st)
            ins_bytes = insert_bytes(st)
            enc_s = kt["encode_ms"] / 1e3
            read_bytes = alg - ins_bytes - 8.0 * st.get("siv_saved", 0)
            roofline = {"bound": "hbm", "achieved": round((alg - ins_bytes) / max(enc_s, 1e-9) / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                        "frac": round((alg - ins_bytes) / max(enc_s, 1e-9) / 1e9 / 8000.0, 6), "traffic": None, "kernel": "k_encode_se_sorted",
                        "launches": kt["encode_launches"], "avg_launch_ms": round(kt["encode_ms"] / max(1, kt["encode_launches"]), 4),
                        "algorithmic_bytes_per_launch": round((alg - ins_bytes) / max(1, kt["encode_launches"]), 1),
                        "frac_of_bytes_the_kernel_reads": round(read_bytes / max(enc_s, 1e-9) / 1e9 / 8000.0, 6),
                        "note": f"rank 0's launches only: its {len(range(0, a.threads, world))} of the {a.threads} workers (one workgroup each), SURVEY 8d bytes of "
                                "those workers / rank 0's encode-kernel time (HIP events on the codec's stream); look-ups of the other ranks' "
                                "sub-tables are xGMI loads, not HBM reads of this GPU"}

    if rank == 0:
        line = {
            "metric": "Mbases/s compressed (DNA stream, SE sorted)", "value": round(value, 4), "unit": "Mbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic" if gpu else "synthetic (EMULATION BUILD on CPUs: a test of the launcher path, not a measurement)",
            "config": {"workload": f"{a.reads}x{a.len}bp SE, G={a.genome} (seed 2), -om s -gs {a.gs} -qm n -im n: ONE file sharded over {world} GPU(s)",
                       "workers_T": a.threads, "blocks": len(dev_blocks),
                       "per_gpu": f"workers w % {world} == rank ({len(range(0, a.threads, world))} workgroups on rank 0); per phase: all-reduce of the count matrix, "
                                  "one grouped all-to-all of the three mailboxes, one all-gather (RCCL over xGMI, inside libfqsx.so)"
                                  + ("; s-/b-mer tables partitioned over the ranks (1/N each, the rest read through peer mappings)" if info.get("partitioned")
                                     else "; table replicas on every rank" + (" (asked for partitioned tables: " + info.get("partition_note", "") + ")" if a.partition else ""))},
            "bits_per_base": round(8.0 * int(tot.item()) / n_bases, 5),
            "partitioned_tables": bool(info.get("partitioned")), "ranks_seen": info.get("ranks_seen"),
            "per_rank": held, "exchange_rank0_per_file": traffic, "replicas": replicas,
            "roofline": roofline, "cpu_baseline": None}
        if emit:
            print(json.dumps(line), flush=True)
    dist.barrier()
    if gpu and comm[0] is not None:
        from fqsqueezer_amd.sharded import _Comm  # noqa: F401
        import ctypes as C
        from fqsqueezer_amd.codec import load_library
        load_library().fqsx_rccl_comm_destroy(C.byref(comm[0]))
    if emit:
        dist.destroy_process_group()
    return line if rank == 0 else None


def replicas_pass(a, rank, local_rank, world):
    """The other multi-GPU mode: N independent 1 M-read files, one per GPU (weak scaling, no data-path collective); one timed pass."""
    import torch
    import torch.distributed as dist
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec
    from fqsqueezer_amd.synth import read_id, synth_reads
    r_reads = synth_reads(1_000_000, a.len, 7_500_000, 2 + rank)
    r_rec = hp.Records([read_id(i) for i in range(len(r_reads))], r_reads, r_reads)
    r_header = hp.make_header(a.threads, "se_sorted", 8)
    r_blocks = []
    for idx in hp.form_blocks(r_rec, "se_sorted"):
        bases, off = hp.block_arrays(r_rec, idx)
        r_blocks.append((torch.from_numpy(np.ascontiguousarray(bases)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off))

    def r_step():
        c = DnaCodec(r_header, device=local_rank)
        nb = 0
        for g, (d_b, d_o, off) in enumerate(r_blocks):
            nb += c.encode_block_dev(d_b.data_ptr(), d_o.data_ptr(), off, g, collect=False)
        c.close()
        return nb

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    r_step()
    sync()
    t1 = time.perf_counter()
    nb = r_step()
    sync()
    tr = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
    dist.all_reduce(tr, op=dist.ReduceOp.MAX)
    tot = torch.tensor([nb], dtype=torch.int64, device="cuda")
    dist.all_reduce(tot)
    return {"value": round(world * 150.0 / float(tr.item()), 4), "unit": "Mbases/s", "scaling": "weak", "ms_per_step": round(float(tr.item()) * 1e3, 2),
            "bits_per_base": round(8.0 * int(tot.item()) / (world * 150e6), 5),
            "workload": "1000000x150bp SE, G=7500000 (seed 2+rank), -om s -gs 8, one independent file per GPU, no data-path collective; one timed pass"}


def guarded_sharded_main(a, rank, local_rank, world, emu_lib=None):
    """The driver's `--gpus N` line must come out whatever the node does to the sharded mode (RCCL inside libfqsx.so, descriptor passing
    and peer mappings between N real GPUs).  So: the replicas mode is measured first (it needs nothing but N GPUs), the sharded file
    runs on a side thread under a deadline, the ranks agree over a gloo group (CPU sockets: it still answers when a GPU stream is
    stuck in a collective) whether it finished everywhere, and if it did not rank 0 prints the replicas line -- `scaling: weak`, with
    the failure spelled out in `sharded_failed` -- and the processes leave without waiting for the stuck stream."""
    import datetime
    import threading
    import torch.distributed as dist
    gpu = emu_lib is None
    deadline_s = float(os.environ.get("FQSX_BENCH_SHARD_DEADLINE_S", "420"))
    # (a rank that failed early waits in the agreement below until the slowest rank's deadline has passed: longer than the deadline)
    side = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=deadline_s + 180))
    replicas = replicas_pass(a, rank, local_rank, world) if gpu else None
    res = {}
    store = None
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        pass

    def body():
        try:
            if gpu:
                import torch
                torch.cuda.set_device(local_rank)   # (the current device is per thread)
            if os.environ.get("FQSX_BENCH_TEST_RAISE") == str(rank):   # tests: one rank fails alone, before any collective
                raise RuntimeError("injected failure")
            res["line"] = sharded_main(a, rank, local_rank, world, emu_lib=emu_lib, replicas=replicas, emit=False)
            res["ok"] = True
        except BaseException as e:   # noqa: BLE001 -- whatever it is, the other ranks have to hear of it
            res["err"] = f"rank {rank}: {type(e).__name__}: {e}"[:600]
            try:
                store.set("fqsx_shard_failed", res["err"])
            except Exception:
                pass

    th = threading.Thread(target=body, daemon=True)
    th.start()
    deadline = time.time() + deadline_s
    while th.is_alive() and time.time() < deadline:
        th.join(1.0)
        try:
            if th.is_alive() and store is not None and store.check(["fqsx_shard_failed"]):
                th.join(5.0)
                break
        except Exception:
            store = None
    if th.is_alive() and "err" not in res:
        res["err"] = f"rank {rank}: the sharded pass had not finished" + (" when another rank reported a failure" if time.time() < deadline else " at its deadline")
    import torch
    ok = torch.tensor([1 if res.get("ok") else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=side)
    if int(ok.item()):
        if rank == 0:
            print(json.dumps(res["line"]), flush=True)
        dist.destroy_process_group()
        return
    errs = [None] * world
    dist.all_gather_object(errs, res.get("err"), group=side)
    if rank == 0:
        r = replicas or {"value": 0.0, "ms_per_step": None, "bits_per_base": None, "workload": "(no replicas pass in the CPU rehearsal)"}
        print(json.dumps({
            "metric": "Mbases/s compressed (DNA stream, SE sorted)", "value": r["value"], "unit": "Mbases/s", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": r["workload"], "workers_T": a.threads, "per_gpu": "one independent file per GPU"},
            "bits_per_base": r["bits_per_base"],
            "sharded_failed": {"what": "the sharded-partitioned pass (one file over the N GPUs) did not finish on every rank; this line is the replicas mode instead",
                               "ranks": [e for e in errs if e]},
            "roofline": None, "cpu_baseline": None}), flush=True)
    sys.stdout.flush()
    os._exit(0)   # (a rank stuck in a collective never returns: no destructors, no process-group teardown)


def cpu_baseline(a, hp, reads, rec):
    """Times the unmodified reference (oracle/_ref/fqs-1.1, built by oracle/Makefile) on the first `cpu_sample_reads`
    reads of the same synthetic file (default: the whole workload) in two legs: `-t min(64, cores)` -- the same
    bitstream as the GPU run at T = 64 -- and `-t 8`, the reference's fastest setting on a many-core box (three thread
    barriers per synchronisation point make it slow down beyond ~8 threads).  Its fixed start-up (table allocation:
    ~0.45 GiB per thread, ~18 s at -t 64) is measured per leg on a 1000-read file with the same command line.  `value` is
    the better whole-process leg.  Falls back to the oracle restatement when the binary is absent."""
    from fqsqueezer_amd.synth import write_fastq
    n = min(a.cpu_sample_reads, a.reads)
    ref = os.path.join(ROOT, "oracle", "_ref", "fqs-1.1")
    cores = os.cpu_count() or 1
    if os.path.exists(ref):
        def run(fq, td, t):
            cmd = [ref, "e", "-s", "-om", "s", "-t", str(t), "-gs", str(a.gs), "-qm", "n", "-im", "n", "-v", "0",
                   "-tmp", os.path.join(td, "tmp_"), "-out", os.path.join(td, "o.fqs"), fq]
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return time.perf_counter() - t0, os.path.getsize(os.path.join(td, "o.fqs"))

        legs = []
        with tempfile.TemporaryDirectory(prefix="fqsx_bench_") as td:
            fq, tiny = os.path.join(td, "s.fq"), os.path.join(td, "t.fq")
            write_fastq(fq, reads[:n], rec.qual[:n])
            write_fastq(tiny, reads[:1000], rec.qual[:1000])
            for t in sorted({max(1, min(a.threads, 64, cores)), max(1, min(8, cores))}, reverse=True):
                st, _ = run(tiny, td, t)
                dt, nbytes = run(fq, td, t)
                legs.append({"threads": t, "value": round(n * a.len / dt / 1e6, 4), "wall_s": round(dt, 2), "startup_s": round(st, 2),
                             "post_startup_value": round(n * a.len / max(dt - st, 1e-9) / 1e6, 4),
                             "fqs_bits_per_base": round(8.0 * nbytes / (n * a.len), 5),
                             "same_bitstream_as_gpu_run": t == a.threads})
        best = max(legs, key=lambda x: x["value"])
        return {"value": best["value"], "unit": "Mbases/s", "cores": best["threads"], "kind": "reference",
                "post_startup_value": best["post_startup_value"], "startup_s": best["startup_s"], "legs": legs,
                "sample": f"first {n} reads of the workload file, `fqs-1.1 e -s -om s -t <T> -gs {a.gs} -qm n -im n` for T in "
                          f"{[x['threads'] for x in legs]}; whole-process wall (includes its binning/sort pre-pass and its start-up, measured "
                          f"per leg on a 1000-read file; post_startup_value excludes the latter); value = the faster leg (-t {best['threads']}); "
                          f"host has {cores} logical CPUs"}
    from oracle.pyoracle import OracleCodec
    sub = hp.Records(rec.ids[:n], reads[:n], rec.qual[:n])
    oc = OracleCodec(hp.make_header(a.threads, "se_sorted", a.gs))
    t0 = time.perf_counter()
    for g, idx in enumerate(hp.form_blocks(sub, "se_sorted")):
        bases, off = hp.block_arrays(sub, idx)
        oc.encode_block(bases, off, g)
    dt = time.perf_counter() - t0
    return {"value": round(n * a.len / dt / 1e6, 4), "unit": "Mbases/s", "cores": 1, "kind": "port",
            "sample": f"first {n} reads, oracle restatement single-threaded, {dt:.1f}s"}


if __name__ == "__main__":
    main()
