"""Sharded mode: workers w % world on rank w, mailboxes routed to their owners' rank by all-to-all,
replicas refreshed from the owners -- world_size 2 / 3 on the gloo backend with the emulated kernels.  Every worker's
DNA stream must be bit-identical to the one-process run's (and hence to the reference's, to which that run is pinned)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.sharded import ShardedDnaCodec
from fqsqueezer_amd.synth import synth_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = os.environ["FQSX_EMU_LIB"]
T, mode = int(os.environ["FQSX_T"]), os.environ["FQSX_MODE"]
reads = synth_reads(6000, 90, 40000, 31)           # ~13x coverage: misses, hits, corrections, table growth
rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
header = hp.make_header(T, mode, 1)
order = np.concatenate(hp.sorted_order(rec)) if mode == "se_sorted" else np.arange(len(reads))
blocks = [order[lo:lo + 600] for lo in range(0, len(order), 600)]        # ten blocks, several segments each
sh = ShardedDnaCodec(header, rank, world, lib_path=lib, tensor_device=torch.device("cpu"))
one = DnaCodec(header, lib_path=lib)                # the one-process run, for comparison (every rank runs it)
h = hashlib.sha256()
for g, idx in enumerate(blocks):
    bases, off = hp.block_arrays(rec, idx)
    mine = sh.encode_block(bases, off, g)
    ref = one.encode_block(bases, off, g)
    assert sorted(mine) == list(range(rank, T, world))
    for w, s in mine.items():
        assert s == ref[w], f"rank {rank}: block {g} worker {w} differs from the one-process run"
        h.update(s)
lst = [None] * world
dist.all_gather_object(lst, (h.hexdigest(), sh.traffic))
if rank == 0:
    assert sh.traffic["phases"] > len(blocks) and sh.traffic["all_to_all_bytes"] > 0
    print("SHARDED_OK", world, T, mode, lst[0][1])
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,T,mode,port", [(2, 5, "se_sorted", 29531), (3, 4, "se_original", 29532)])
def test_sharded_streams_identical_to_one_process_run(tmp_path, built, world, T, mode, port):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_EMU_LIB=os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"), FQSX_T=str(T), FQSX_MODE=mode)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "SHARDED_OK" in r.stdout


# ---- the native driver (fqsx_shard_encode_block: phase loop inside the library, three collectives per phase) over a
# callback transport into torch.distributed / gloo, single- and paired-end
NATIVE_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.sharded import NativeShardedDnaCodec
from fqsqueezer_amd.synth import synth_pairs, synth_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = os.environ["FQSX_EMU_LIB"]
T, mode = int(os.environ["FQSX_T"]), os.environ["FQSX_MODE"]
header = hp.make_header(T, mode, 1)
if mode.startswith("pe"):
    r1, r2 = synth_pairs(2500, 90, 40000, 33)
    rec1 = hp.Records([b"@a%d" % i for i in range(len(r1))], r1, r1)
    rec2 = hp.Records([b"@b%d" % i for i in range(len(r2))], r2, r2)
    order = np.concatenate(hp.sorted_order_exact(rec1)) if mode == "pe_sorted" else np.arange(len(r1))
    blocks = [hp.block_arrays_pe(rec1, rec2, order[lo:lo + 300]) for lo in range(0, len(order), 300)]
else:
    reads = synth_reads(6000, 90, 40000, 31)
    rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
    order = np.concatenate(hp.sorted_order(rec)) if mode == "se_sorted" else np.arange(len(reads))
    blocks = [hp.block_arrays(rec, order[lo:lo + 600]) for lo in range(0, len(order), 600)]
part = os.environ.get("FQSX_PARTITION") == "1"
sh = NativeShardedDnaCodec(header, rank, world, lib_path=lib, transport="torch", partition=part)
one = DnaCodec(header, lib_path=lib)                # the one-process run, for comparison (every rank runs it)
for g, (bases, off) in enumerate(blocks):
    mine = sh.encode_block(bases, off, g)
    ref = one.encode_block(bases, off, g)
    assert sorted(mine) == list(range(rank, T, world))
    for w, s in mine.items():
        assert s == ref[w], f"rank {rank}: block {g} worker {w} differs from the one-process run"
tr = sh.traffic
# three collectives per phase + one status vote in each phase in which buffers or tables grow (every rank alike): a few per file
# (a partitioned pair table: one more per phase, the barrier behind its inserts)
votes = tr["collectives"] - (4 if part and mode.startswith("pe") and world > 1 else 3) * tr["phases"]
assert 0 <= votes <= tr["phases"] // 2 and tr["phases"] > len(blocks), tr
cap, cap1 = sh.codec.capacity(), one.capacity()
assert (cap["smers"], cap["bmers"]) == (cap1["smers"], cap1["bmers"])     # every rank knows every sub-table's occupancy
assert cap["growths"] >= 2, cap                                              # the tables grew on the way (from 64-slot sub-tables)
lst = [None] * world
dist.all_gather_object(lst, (tr, cap["table_bytes_held"], 8 * (cap["smer_slots"] + cap["bmer_slots"]), cap["pair_bytes_held"], 16 * cap["pair_slots"], cap["siv_bytes_held"], cap["siv_bytes"]))
if rank == 0:
    assert tr["all_to_all_bytes"] > 0 and tr["all_gather_bytes"] > 0
    held, whole = [x[1] for x in lst], lst[0][2]
    if part:   # a rank holds the memory of its owners' sub-tables only: ceil(T / world) or floor(T / world) of the T
        # (a sub-table is one chunk: its capacity rounded up to the allocation granule, a page in this build)
        slack = 2 * T * 4096
        assert whole <= sum(held) <= whole + slack and max(held) <= (whole + slack) * ((T + world - 1) // world) // T, (held, whole)
        # the p-mer vector: its 4096 owner ranges (16 KiB each at this geometry) live on their owners' ranks -- (range % T) % world
        sheld, swhole = [x[5] for x in lst], lst[0][6]
        share = [sum(1 for r in range(4096) if (r % T) % world == q) for q in range(world)]
        assert sum(sheld) == swhole and sheld == [swhole // 4096 * n for n in share], (sheld, swhole, share)
        # the pair table of a paired-end file with them (key and value array: two chunks per sub-table); it grew on the way
        pheld, pwhole = [x[3] for x in lst], lst[0][4]
        if mode.startswith("pe"):
            assert pwhole > 16 * 64 * T and pwhole <= sum(pheld) <= pwhole + slack and max(pheld) <= (pwhole + slack) * ((T + world - 1) // world) // T, (pheld, pwhole)
            assert (cap["pairs"], cap["pair_slots"]) == (cap1["pairs"], cap1["pair_slots"])   # (the owners' occupancies travel with the all-gather)
    else:
        assert all(h == whole for h in held), (held, whole)
        assert all(x[3] == x[4] and x[5] == x[6] for x in lst), lst
    print("NATIVE_SHARDED_OK", world, T, mode, "partitioned" if part else "replicas", held, whole, lst[0][0])
dist.destroy_process_group()
'''


# partition = 1: the k-mer tables partitioned over the ranks (fqsx_shard_partition_tables) -- each rank holds the memory of its
# owners' sub-tables (memfd chunks in the emulation build, handed over as descriptors like the HIP build's dma-buf exports)
# and looks the others' up through the mapping; the tables start at 64 slots per sub-table so that they grow several times
@pytest.mark.parametrize("world,T,mode,port,partition", [(2, 5, "se_sorted", 29533, 0), (3, 4, "se_original", 29534, 0), (2, 4, "pe_sorted", 29535, 0),
                                                         (3, 5, "pe_original", 29536, 0), (2, 5, "se_sorted", 29537, 1), (3, 7, "se_original", 29538, 1),
                                                         (3, 4, "pe_sorted", 29539, 1), (2, 6, "pe_original", 29540, 1)])
def test_native_sharded_driver_streams_identical_to_one_process_run(tmp_path, built, world, T, mode, port, partition):
    script = tmp_path / "w.py"
    script.write_text(NATIVE_WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_EMU_LIB=os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"), FQSX_T=str(T), FQSX_MODE=mode,
               FQSX_PARTITION=str(partition), FQSX_GTAB_INIT="64", FQSX_PTAB_INIT="64",
               FQSX_SHARD_APPLY_OWN="1" if world == 2 else "0")   # (world 2 also applies the rank's own items to its replica: must change nothing)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "NATIVE_SHARDED_OK" in r.stdout


# ---- a rank that fails between two collectives must take the whole world out of the phase (RCCL has no timeout: a rank that
# simply returned would leave the others waiting for ever).  One rank reports an injected allocation failure in phase 7: it comes
# back with its own error, the other rank with FQSX_E_PEER (-6), and nobody hangs.
FAIL_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import FqsxError
from fqsqueezer_amd.sharded import NativeShardedDnaCodec, ShardedDnaCodec
from fqsqueezer_amd.synth import synth_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = os.environ["FQSX_EMU_LIB"]
reads = synth_reads(3000, 90, 40000, 35)
rec = hp.Records([b"@r%d" % i for i in range(len(reads))], reads, reads)
header = hp.make_header(4, "se_sorted", 1)
order = np.concatenate(hp.sorted_order(rec))
blocks = [order[lo:lo + 600] for lo in range(0, len(order), 600)]
sh = NativeShardedDnaCodec(header, rank, world, lib_path=lib, transport="torch", partition=os.environ["FQSX_PARTITION"] == "1")
msg = None
try:
    for g, idx in enumerate(blocks):
        bases, off = hp.block_arrays(rec, idx)
        sh.encode_block(bases, off, g)
except FqsxError as e:
    msg = str(e)
assert msg is not None, "the injected failure went unnoticed"
if rank == 1:
    assert ": -4:" in msg and "injected" in msg, msg            # FQSX_E_NOMEM, this rank's own failure
else:
    assert ": -6:" in msg and "another rank" in msg, msg        # FQSX_E_PEER
assert sh.traffic["phases"] == 7, sh.traffic                    # both left in the same phase
# the step-wise driver refuses paired-end worlds (it does not exchange the pair-table triples)
try:
    ShardedDnaCodec(hp.make_header(4, "pe_sorted", 1), rank, world, lib_path=lib, tensor_device=torch.device("cpu"))
    raise SystemExit("paired-end step-wise world accepted")
except ValueError:
    pass
dist.barrier()
if rank == 0:
    print("FAIL_VOTE_OK")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("partition,port", [(0, 29561), (1, 29562)])
def test_a_failing_rank_takes_the_world_out_of_the_phase(tmp_path, built, partition, port):
    script = tmp_path / "f.py"
    script.write_text(FAIL_WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_EMU_LIB=os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"), FQSX_PARTITION=str(partition),
               FQSX_TEST_FAIL="1,7")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "FAIL_VOTE_OK" in r.stdout


def test_stepwise_library_entry_refuses_paired_end_worlds(built):
    import ctypes as C
    from fqsqueezer_amd import hostpipe as hp
    lib = C.CDLL(os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"))
    lib.fqsx_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert lib.fqsx_dna_create(hp.make_header(4, "pe_sorted", 1), 0, C.byref(h)) == 0
    assert lib.fqsx_shard_config(h, C.c_uint32(1), C.c_uint32(2)) == 0
    off = (C.c_uint64 * 3)(0, 50, 100)
    bases = C.create_string_buffer(b"A" * 100)
    nseg = C.c_uint32()
    rc = lib.fqsx_shard_begin_block(h, bases, off, off, C.c_uint32(2), C.c_uint32(0), C.byref(nseg))
    assert rc == -1 and b"pair-table triples" in lib.fqsx_last_error()
    lib.fqsx_dna_destroy(h)
