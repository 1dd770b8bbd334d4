"""Partitioned tables of the sharded mode on real hardware (SURVEY.md 8e option (i); fqsx_shard_partition_tables).

A gpurun box has ONE GPU and RCCL refuses two ranks on one device, so the multi-rank run here is two processes on device 0
whose collectives are staged through the host and done by gloo (`transport="staged"`).  What the GPU run adds to the gloo /
emulation tests of tests/test_sharded_cpu.py: the HIP virtual-memory path itself -- hipMemCreate, the dma-buf export, the
descriptor hand-over between the processes, hipMemImportFromShareableHandle, hipMemMap into one address range -- look-ups of
the encode kernels through imported mappings, owner-only inserts, collective growth with re-export, and the occupancy
counters riding on the all-gather.  Every worker's stream must equal the one-GPU run's (which is pinned to the reference).
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.sharded import NativeShardedDnaCodec
from fqsqueezer_amd.synth import synth_pairs, synth_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
T, mode = int(os.environ["FQSX_T"]), os.environ["FQSX_MODE"]
header = hp.make_header(T, mode, 1)
if mode.startswith("pe"):
    r1, r2 = synth_pairs(6000, 100, 120000, 33)
    rec1 = hp.Records([b"@a%d" % i for i in range(len(r1))], r1, r1)
    rec2 = hp.Records([b"@b%d" % i for i in range(len(r2))], r2, r2)
    blocks = [hp.block_arrays_pe(rec1, rec2, idx) for idx in hp.form_blocks_pe(rec1, rec2, mode)[:24]]
else:
    reads = synth_reads(20000, 100, 150000, 37)
    rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
    blocks = [hp.block_arrays(rec, idx) for idx in hp.form_blocks(rec, mode)[:48]]
sh = NativeShardedDnaCodec(header, rank, world, device=0, transport="staged", partition=True)
one = DnaCodec(header, device=0)                    # the one-GPU run, for comparison (every rank runs it)
for g, (bases, off) in enumerate(blocks):
    mine = sh.encode_block(bases, off, g)
    ref = one.encode_block(bases, off, g)
    assert sorted(mine) == list(range(rank, T, world))
    for w, s in mine.items():
        assert s == ref[w], f"rank {rank}: block {g} worker {w} differs from the one-GPU run"
cap, cap1 = sh.codec.capacity(), one.capacity()
assert (cap["smers"], cap["bmers"]) == (cap1["smers"], cap1["bmers"]), (cap, cap1)
assert cap["growths"] >= 2, cap
lst = [None] * world
dist.all_gather_object(lst, (cap["table_bytes_held"], 8 * (cap["smer_slots"] + cap["bmer_slots"]), sh.traffic, cap["pair_bytes_held"], 16 * cap["pair_slots"]))
if rank == 0:
    held, whole = [x[0] for x in lst], lst[0][1]   # (held counts whole chunks, and a chunk is at least 2 MiB: >= for these small tables)
    assert sum(held) >= whole and max(held) * T <= sum(held) * ((T + world - 1) // world), (held, whole)
    if mode.startswith("pe"):   # the pair table is partitioned with them: a rank holds (and writes) its owners' sub-tables only
        pheld, pwhole = [x[3] for x in lst], lst[0][4]
        assert (cap["pairs"], cap["pair_slots"]) == (cap1["pairs"], cap1["pair_slots"]) and cap["pair_slots"] > 256 * T, (cap, cap1)
        assert sum(pheld) >= pwhole and max(pheld) * T <= sum(pheld) * ((T + world - 1) // world), (pheld, pwhole)
    print("PARTITIONED_GPU_OK", world, T, mode, held, whole, lst[0][2])
sh.close(); one.close()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,T,mode,port", [(2, 16, "se_sorted", 29561), (3, 8, "pe_sorted", 29562)])
def test_partitioned_tables_ranks_share_one_gpu(tmp_path, world, T, mode, port):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_T=str(T), FQSX_MODE=mode, FQSX_GTAB_INIT="1024", FQSX_PTAB_INIT="256")   # (small tables: several collective growths)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    i = r.stderr.find("Fatal Python error")   # (faulthandler of a rank: the interesting part of a native crash)
    assert r.returncode == 0, (r.stdout[-1500:] + (r.stderr[max(0, i - 500):i + 2500] if i >= 0 else r.stderr[-3000:]))
    assert "PARTITIONED_GPU_OK" in r.stdout


def test_partitioned_tables_world_of_one_over_rccl():
    """the same entry points with the RCCL transport inside the library (one rank: no peer mappings, but the tables live in
    hipMemCreate chunks inside one reserved range, growth re-creates them, the all-gather carries the occupancy words)"""
    import numpy as np
    from fqsqueezer_amd import hostpipe as hp
    from fqsqueezer_amd.codec import DnaCodec
    from fqsqueezer_amd.sharded import NativeShardedDnaCodec
    from fqsqueezer_amd.synth import synth_reads
    os.environ["FQSX_GTAB_INIT"] = "1024"
    try:
        reads = synth_reads(20000, 100, 150000, 37)
        rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
        header = hp.make_header(16, "se_sorted", 1)
        sh = NativeShardedDnaCodec(header, 0, 1, device=0, transport="rccl", id_bytes=NativeShardedDnaCodec.rccl_unique_id(), partition=True)
        one = DnaCodec(header, device=0)
        for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")[:60]):
            bases, off = hp.block_arrays(rec, idx)
            mine, ref = sh.encode_block(bases, off, g), one.encode_block(bases, off, g)
            assert [mine[w] for w in range(16)] == ref, f"block {g}"
        cap = sh.codec.capacity()
        assert cap["growths"] >= 2 and cap["table_bytes_held"] >= 8 * (cap["smer_slots"] + cap["bmer_slots"])
        sh.close()
        one.close()
    finally:
        os.environ.pop("FQSX_GTAB_INIT", None)


# ---- configs[3]-shaped input (c19: 10 M x 150 bp, G = 300 Mbp, -gs 300, T = 64) with the k-mer tables partitioned over two ranks:
# every block's digest against the REFERENCE's (tests/golden/c19_10M150_gs300_s_t64.json), tables of several GB in chunks of tens
# of MB, half of them on either rank.  Both ranks share the box's one GPU (collectives staged through the host).
C19_WORKER = r'''
import faulthandler, hashlib, json, os, sys
faulthandler.enable()
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import sort_order
from fqsqueezer_amd.sharded import NativeShardedDnaCodec
from fqsqueezer_amd.synth import read_id, synth_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = json.load(open(os.path.join(os.environ["FQSX_ROOT"], "tests", "golden", os.environ["FQSX_GOLDEN"])))
limit = int(os.environ.get("FQSX_FULLSIZE_BLOCKS", "0")) or d["n_blocks"]
n = d["reads"]
reads = synth_reads(n, d["len"], d["genome"], d["seed"])
rec = hp.Records([read_id(i) for i in range(n)], reads, reads)
groups = sort_order(rec.seq.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(d["len"]))
blks = hp.form_blocks(rec, "se_sorted", groups=groups)
assert len(blks) == d["n_blocks"]
header = bytes.fromhex(d["header"])
T = header[4]
sh = NativeShardedDnaCodec(header, rank, world, device=0, transport="staged", partition=True)
for g, (idx, ref) in enumerate(zip(blks[:limit], d["blocks"])):
    bases, off = hp.block_arrays(rec, idx)
    mine = sh.encode_block(bases, off, g)
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        streams = {}
        for p in parts:
            streams.update(p)
        h = hashlib.sha256()
        for w in range(T):
            h.update(streams[w])
        assert sum(len(streams[w]) for w in range(T)) == ref["bytes"] and h.hexdigest() == ref["sha256"], f"block {g} differs from the reference"
cap = sh.codec.capacity()
lst = [None] * world
dist.all_gather_object(lst, (cap["table_bytes_held"], 8 * (cap["smer_slots"] + cap["bmer_slots"]), cap["device_bytes_peak"], sh.traffic))
if rank == 0:
    held, whole = [x[0] for x in lst], lst[0][1]
    assert sum(held) >= whole and max(held) * world <= sum(held) + world * (4 << 20), (held, whole)   # (a chunk is at least 2 MiB)
    print("PARTITIONED_C19_OK", os.environ["FQSX_GOLDEN"], limit, "blocks;", "table bytes per rank", held, "of", whole, "peak device bytes per rank", [x[2] for x in lst], lst[0][3])
sh.close()
dist.destroy_process_group()
'''


# c17 = the reference's DEFAULT geometry (-gs 3100: k = 13 / 18 / 21 / 27, 16 GiB p-mer vector per rank) on 1 M reads, T = 8, over four ranks
# c21 = SURVEY 8d-4's prefix of configs[3]'s own file (5 M reads of a G = 3.1 Gbp genome, -gs 3100, T = 8) over four ranks: the first 32
# blocks by default (64 ran in round 4: profiles/r04_c21_full_tests.txt) (eight workers spread over four ranks sharing one GPU, collectives staged through gloo), all 256 with FQSX_SLOW=1
@pytest.mark.parametrize("golden,world,port", [("c19_10M150_gs300_s_t64.json", 2, 29563), ("c17_1M150_gs3100_s_t8.json", 4, 29564),
                                               ("c21_5M150_G3100_gs3100_s_t8.json", 4, 29566)])
def test_partitioned_tables_c19_two_ranks_against_the_reference(tmp_path, golden, world, port):
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", golden)):
        pytest.skip(f"{golden} has not been generated (tools/make_golden.py)")
    script = tmp_path / "w.py"
    script.write_text(C19_WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_GOLDEN=golden)
    slow = os.environ.get("FQSX_SLOW") == "1"
    if golden.startswith("c21") and not slow and "FQSX_FULLSIZE_BLOCKS" not in env:
        env["FQSX_FULLSIZE_BLOCKS"] = "32"
    if not golden.startswith("c21") and not slow:   # (both ran green in rounds 3 and 4; c21 over four ranks covers the path in the driver's suite)
        pytest.skip("c19 over two ranks (60 s) and c17 over four (28 s) run with FQSX_SLOW=1: the suite's time goes to c21")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100)
    i = r.stderr.find("Fatal Python error")
    assert r.returncode == 0, (r.stdout[-1500:] + (r.stderr[max(0, i - 500):i + 2500] if i >= 0 else r.stderr[-3000:]))
    assert "PARTITIONED_C19_OK" in r.stdout
    print(r.stdout[r.stdout.find("PARTITIONED_C19_OK"):][:600])


# ---- BASELINE configs[2] at its full size (c18: 5 M pairs x 150 bp, -p -om s, T = 8) over two ranks with partitioned tables: the
# DNA stream digests of every block against the reference's.  Minutes (eight workers): run with FQSX_SLOW=1.
C18_WORKER = r'''
import faulthandler, hashlib, json, os, sys
faulthandler.enable()
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import sort_order
from fqsqueezer_amd.sharded import NativeShardedDnaCodec
from fqsqueezer_amd.synth import read_id, synth_pairs, synth_quals
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = json.load(open(os.path.join(os.environ["FQSX_ROOT"], "tests", "golden", "c18_pe5M_s_q8_t8.json")))
limit = int(os.environ.get("FQSX_FULLSIZE_BLOCKS", "0")) or d["n_blocks"]
n, L, seed = d["pairs"], d["len"], d["seed"]
r1, r2 = synth_pairs(n, L, d["genome"], seed)
rec1 = hp.Records([read_id(i, 1) for i in range(n)], r1, synth_quals(n, L, seed))       # (ids and qualities decide the block boundaries)
rec2 = hp.Records([read_id(i, 2) for i in range(n)], r2, synth_quals(n, L, seed + 1))
groups = sort_order(r1.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
blks = hp.form_blocks_pe(rec1, rec2, "pe_sorted", groups=groups)
assert len(blks) == d["n_blocks"]
print(f"rank {rank}: input ready", flush=True)
header = hp.make_header(d["threads"], "pe_sorted", d["gs"])
T = header[4]
sh = NativeShardedDnaCodec(header, rank, world, device=0, transport="staged", partition=True)
for g, (idx, ref) in enumerate(zip(blks[:limit], d["blocks"])):
    bases, off = hp.block_arrays_pe(rec1, rec2, idx)
    assert len(off) - 1 == ref["n_reads"]
    mine = sh.encode_block(bases, off, g)
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        streams = {}
        for p in parts:
            streams.update(p)
        h = hashlib.sha256()
        for w in range(T):
            h.update(streams[w])
        assert h.hexdigest() == ref["2"], f"block {g}: DNA streams differ from the reference"
        if g % 16 == 15:
            print(f"block {g} ok", flush=True)
cap = sh.codec.capacity()
if rank == 0:
    print("PARTITIONED_C18_OK", limit, "blocks;", {k: cap[k] for k in ("smers", "bmers", "pairs", "table_bytes_held", "device_bytes_peak", "growths")}, sh.traffic)
sh.close()
dist.destroy_process_group()
'''


def test_partitioned_tables_c18_paired_end_full_size_against_the_reference(tmp_path):
    """BASELINE configs[2]'s file over two ranks with partitioned k-mer tables AND a partitioned pair table, against the reference's
    per-block DNA digests: the file's first 16 blocks by default (about a minute: most of it is generating and sorting the 5 M pairs),
    all 256 with FQSX_SLOW=1 (220 s; run in full in round 4: profiles/r04_partitioned_pair_table_c18.txt)."""
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "c18_pe5M_s_q8_t8.json")):
        pytest.skip("c18 golden has not been generated (tools/make_golden.py)")
    script = tmp_path / "w.py"
    script.write_text(C18_WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT)
    if os.environ.get("FQSX_SLOW") != "1":
        env.setdefault("FQSX_FULLSIZE_BLOCKS", "16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29565", str(script)]
    r = subprocess.run(cmd, env=env, text=True, timeout=1150, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    i = r.stderr.find("Fatal Python error")
    assert r.returncode == 0, (r.stdout[-1500:] + (r.stderr[max(0, i - 500):i + 2500] if i >= 0 else r.stderr[-3000:]))
    assert "PARTITIONED_C18_OK" in r.stdout
    print(r.stdout[r.stdout.find("PARTITIONED_C18_OK"):][:700])


# ---- the real thing: one rank per PHYSICAL GPU, RCCL inside the library, partitioned tables read over xGMI.  Needs a node with at
# least two GPUs (a gpurun box has one: skipped there; the driver's multi-GPU node runs bench.py --gpus N, which takes the same path).
XGMI_WORKER = r'''
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.sharded import NativeShardedDnaCodec
from fqsqueezer_amd.synth import synth_reads
rank, world, dev = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
T = 16
header = hp.make_header(T, "se_sorted", 1)
reads = synth_reads(40000, 100, 300000, 41)
rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
blocks = [hp.block_arrays(rec, idx) for idx in hp.form_blocks(rec, "se_sorted")[:80]]
ids = [NativeShardedDnaCodec.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, src=0)
sh = NativeShardedDnaCodec(header, rank, world, device=dev, transport="rccl", id_bytes=ids[0], partition=True)
assert sh.partitioned, sh.partition_note            # (a node whose GPUs cannot reach each other would fall back to replicas)
assert sh.rccl_info()["ranks_seen"] == world
one = DnaCodec(header, device=dev)                   # the one-GPU run on this rank's own device, for comparison
for g, (bases, off) in enumerate(blocks):
    mine = sh.encode_block(bases, off, g)
    ref = one.encode_block(bases, off, g)
    assert sorted(mine) == list(range(rank, T, world))
    for w, s in mine.items():
        assert s == ref[w], f"rank {rank}: block {g} worker {w} differs from the one-GPU run"
cap, cap1 = sh.codec.capacity(), one.capacity()
assert (cap["smers"], cap["bmers"]) == (cap1["smers"], cap1["bmers"]), (cap, cap1)
lst = [None] * world
dist.all_gather_object(lst, (cap["table_bytes_held"], 8 * (cap["smer_slots"] + cap["bmer_slots"]), sh.traffic))
if rank == 0:
    print("PARTITIONED_XGMI_OK", world, [x[0] for x in lst], lst[0][1], lst[0][2])
sh.close(); one.close()
dist.barrier()
dist.destroy_process_group()
'''


def test_partitioned_tables_one_rank_per_physical_gpu_over_rccl(tmp_path):
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one GPU on this box: the xGMI path needs a node with at least two")
    world = min(n_dev, 4)
    script = tmp_path / "x.py"
    script.write_text(XGMI_WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_GTAB_INIT="4096", HSA_ENABLE_IPC_MODE_LEGACY="0")   # (small tables: collective growths with re-export)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", "29567", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    i = r.stderr.find("Fatal Python error")
    assert r.returncode == 0, (r.stdout[-1500:] + (r.stderr[max(0, i - 500):i + 2500] if i >= 0 else r.stderr[-3000:]))
    assert "PARTITIONED_XGMI_OK" in r.stdout
