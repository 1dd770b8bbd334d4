"""N>1 path of bench.py (one independent file per rank, barrier, max-over-ranks timing) with
world_size 2 on the gloo backend; the ranks run the emulated kernels."""
import os
import subprocess
import sys
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys, time
sys.path.insert(0, os.environ["FQSX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
from fqsqueezer_amd.synth import synth_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
reads = synth_reads(600, 80, 20000, 2 + rank)
rec = hp.Records([b"@r%d" % i for i in range(600)], reads, reads)
codec = DnaCodec(hp.make_header(3, "se_sorted", 1), lib_path=os.environ["FQSX_EMU_LIB"])
dist.barrier(); t0 = time.perf_counter(); n = 0
for g, idx in enumerate(hp.form_blocks(rec, "se_sorted")[:10]):
    bases, off = hp.block_arrays(rec, idx)
    n += sum(len(s) for s in codec.encode_block(bases, off, g))
dist.barrier()
t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
tot = torch.tensor([n], dtype=torch.int64); dist.all_reduce(tot)
lst = [None] * world; dist.all_gather_object(lst, n)
if rank == 0:
    assert lst[0] != lst[1], "ranks must compress different files"
    assert int(tot.item()) == sum(lst) and t.item() > 0
    print("MULTIRANK_OK", lst)
dist.destroy_process_group()
'''


def test_two_ranks_gloo(tmp_path, built):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, FQSX_ROOT=ROOT, FQSX_EMU_LIB=os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "MULTIRANK_OK" in r.stdout


def test_bench_py_sharded_line_two_ranks_gloo(built):
    """`torchrun --nproc-per-node=2 bench.py --gpus 2` -- the command line the driver uses for N > 1 -- takes the sharded, partitioned
    path (one file over the ranks, strong scaling).  Rehearsed here on CPUs: FQSX_BENCH_EMU swaps RCCL for gloo and the HIP
    kernels for their emulation build; everything else (argument defaults, the phase loop inside the library with its three
    collectives, partitioned tables with descriptor passing, the JSON line) is the code the GPU run executes."""
    import json
    env = dict(os.environ, FQSX_BENCH_EMU=os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29518", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--reads", "3000", "--len", "90", "--genome", "40000", "--gs", "1", "--threads", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:] + r.stderr[-3000:])
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["partitioned_tables"] is True
    assert "ONE file sharded over 2" in line["config"]["workload"] and "EMULATION" in line["data"]
    held = [x["table_bytes_held"] for x in line["per_rank"]]
    assert len(held) == 2 and min(held) > 0 and abs(held[0] - held[1]) <= max(held) // 2, held   # each rank holds its owners' share
    assert line["exchange_rank0_per_file"]["all_to_all_bytes"] > 0 and 0 < line["bits_per_base"] < 2.5


@pytest.mark.parametrize("inject", [{"FQSX_TEST_FAIL": "1,3"}, {"FQSX_BENCH_TEST_RAISE": "1"}], ids=["voted-in-the-library", "one-rank-alone"])
def test_bench_py_prints_a_line_when_the_sharded_pass_fails(built, inject):
    """The driver's multi-GPU line has one shot on a node nobody rehearsed on: when the sharded pass fails on any rank (here: an
    injected allocation failure inside rank 1's phase loop, which the library's vote turns into an error on every rank) the
    ranks agree over a gloo side group and rank 0 still prints a line -- the replicas mode on GPUs, marked `sharded_failed` --
    and all processes leave with exit code 0.  Second case: one rank raises alone while the other already waits in a collective
    (it hears of it through the rendezvous store and stops waiting)."""
    import json
    env = dict(os.environ, FQSX_BENCH_EMU=os.path.join(ROOT, "tests", "emu", "libfqsx_emu.so"), FQSX_BENCH_SHARD_DEADLINE_S="120", **inject)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--reads", "3000", "--len", "90", "--genome", "40000", "--gs", "1", "--threads", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:] + r.stderr[-3000:])
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["sharded_failed"]["ranks"], line
