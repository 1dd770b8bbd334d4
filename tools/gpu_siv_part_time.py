#!/usr/bin/env python3
"""How long does a sharded codec with partitioned tables take to set up at the default geometry (-gs 3100: the 16 GiB p-mer vector
in 4096 chunks)?  torchrun --nproc-per-node=N tools/gpu_siv_part_time.py  (ranks share GPU 0; host-staged transport)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.sharded import NativeShardedDnaCodec
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
header = hp.make_header(64, "se_sorted", 3100)
for rep in range(2):
    dist.barrier()
    t0 = time.perf_counter()
    sh = NativeShardedDnaCodec(header, rank, world, device=0, transport="staged", partition=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    cap = sh.codec.capacity()
    sh.close()
    t2 = time.perf_counter()
    print(f"rank {rank} rep {rep}: create+partition {t1 - t0:.3f}s close {t2 - t1:.3f}s siv_bytes_held {cap['siv_bytes_held']} of {cap['siv_bytes']} partitioned {sh.partitioned}", flush=True)
dist.destroy_process_group()
