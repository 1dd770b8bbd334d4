"""One .fqs file's DNA path sharded over the GPUs of a node (BASELINE configs[3]).

Logical worker w of the T the bitstream fixes -- its coder state, RNG streams, local tables and the k-mer sub-tables it
owns (owner = function of the k-mer's prefix, fqs/dna.cpp:825; p-mers: index range, :658) -- lives on rank w % world,
one process per GPU.  Every rank holds a replica of all sub-tables for the look-ups.  One synchronisation phase of the
reference (the T x T mailboxes of fqs/application.h:56-59, applied by their owners in fqs/dna.cpp:2393-2472) becomes:

    encode (own workers)                                     fqsx_shard_encode
    all-reduce  per-(source, owner) entry counts, 3 T^2 u32  -> every rank knows every transfer size
    pack own workers' entries by (rank, owner, source, push)  fqsx_shard_pack
    all-to-all  the three mailboxes (u64 keys)                -> keys reach the rank of their owner, in (source, push) order
    merge into the owners' groups                             fqsx_shard_merge
    all-reduce(max) table demand                              -> all replicas grow alike
    insert phase of own owners (their RNG streams stay here)  fqsx_shard_insert
    all-gather  one (k-mer, final count) item per applied key -> the other replicas take the owners' new slot values over
    apply to the replicas                                     fqsx_shard_apply
    all-reduce  p-mer vector statistics; clear local tables   fqsx_shard_end_phase

The collectives are torch.distributed's: backend "nccl" is RCCL over xGMI on MI355X (tensors in HBM, no host staging);
the world_size-2 CPU test runs the same driver over gloo with the emulation build of the kernels.  The per-worker
streams are bit-identical to the one-GPU run's (tests/test_sharded_cpu.py), so the file a sharded run writes is the
same file.  With T <= 255 workers already concurrent on one GPU this mode cannot be faster than one GPU for one file
(DESIGN.md section 4 has the measured / counted overhead); replicas (bench.py --gpus N) remain the throughput mode.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from .codec import DnaCodec, FqsxError


class ShardedDnaCodec:
    def __init__(self, header: bytes, rank: int, world: int, device: int = 0, lib_path: Optional[str] = None,
                 tensor_device: Optional[torch.device] = None, group=None, apply_own: bool = False):
        if header[5] >= 2 and world > 1:   # dna_mode 2 / 3 (the library refuses the same in fqsx_shard_begin_block)
            raise ValueError("the step-wise sharded driver does not exchange the pair-table triples: shard a paired-end file with "
                             "NativeShardedDnaCodec (fqsx_shard_attach + fqsx_shard_encode_block)")
        self.codec = DnaCodec(header, device=device, lib_path=lib_path)
        self._lib, self._h, self.T = self.codec._lib, self.codec._h, self.codec.T
        self.rank, self.world, self.group = rank, world, group
        self.apply_own = apply_own   # tests: also run the replica update on this rank's own items (must change nothing)
        # exchange buffers live in the codec's memory space: HBM for the HIP build, host memory for the emulation build
        self.dev = tensor_device if tensor_device is not None else torch.device("cuda", device)
        L = self._lib
        L.fqsx_shard_config.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.fqsx_shard_begin_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.fqsx_shard_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.fqsx_shard_pack.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.fqsx_shard_merge.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.fqsx_shard_insert.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.fqsx_shard_apply.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
        L.fqsx_shard_end_phase.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.fqsx_shard_finish_block.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        self._ck(L.fqsx_shard_config(self._h, rank, world), "fqsx_shard_config")
        self.own_workers = list(range(rank, self.T, world))
        # accounting of what crosses the links (bytes this rank sends, collectives issued)
        self.traffic = {"phases": 0, "collectives": 0, "all_to_all_bytes": 0, "all_gather_bytes": 0, "all_reduce_bytes": 0}

    def _ck(self, rc: int, what: str) -> None:
        if rc:
            raise FqsxError(f"{what}: {rc}: {self._lib.fqsx_last_error().decode()}")

    def _ptrs(self, tensors):
        return (C.c_void_p * 3)(*[t.data_ptr() for t in tensors])

    def encode_block(self, bases: np.ndarray, read_off: np.ndarray, generation: int) -> Dict[int, bytes]:
        """All ranks call this with the same block; returns {worker: DNA stream} for this rank's workers."""
        T, G, me, L, dev = self.T, self.world, self.rank, self._lib, self.dev
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        t_bases = torch.from_numpy(bases).to(dev)
        t_off = torch.from_numpy(read_off.view(np.int64)).to(dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        nseg = C.c_uint32()
        self._ck(L.fqsx_shard_begin_block(self._h, t_bases.data_ptr(), t_off.data_ptr(), read_off.ctypes.data, len(read_off) - 1,
                                          generation, C.byref(nseg)), "fqsx_shard_begin_block")
        for seg in range(nseg.value):
            self._phase(seg)
        streams = (C.c_void_p * T)()
        lens = (C.c_uint64 * T)()
        self._ck(L.fqsx_shard_finish_block(self._h, read_off.ctypes.data, streams, lens), "fqsx_shard_finish_block")
        return {w: (C.string_at(streams[w], lens[w]) if lens[w] else b"") for w in self.own_workers}

    def _sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def _phase(self, seg: int) -> None:
        T, G, me, L, dev, tr = self.T, self.world, self.rank, self._lib, self.dev, self.traffic
        counts = torch.zeros(3 * T * T, dtype=torch.int32, device=dev)
        self._sync()
        self._ck(L.fqsx_shard_encode(self._h, seg, counts.data_ptr()), "fqsx_shard_encode")
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        self._sync()
        Cm = counts.cpu().numpy().reshape(3, T, T).astype(np.int64)   # [kind][source][owner]
        # transfer sizes: what rank q's workers pushed for rank r's owners
        vol = np.stack([[[int(Cm[k][q::G][:, r::G].sum()) for r in range(G)] for q in range(G)] for k in range(3)])   # [kind][from][to]
        send = [torch.empty(max(1, int(vol[k][me].sum())), dtype=torch.int64, device=dev) for k in range(3)]
        self._sync()
        self._ck(L.fqsx_shard_pack(self._h, counts.data_ptr(), self._ptrs(send)), "fqsx_shard_pack")
        recv = []
        for k in range(3):
            n_out, n_in = [int(x) for x in vol[k][me]], [int(x) for x in vol[k][:, me]]
            r = torch.empty(max(1, sum(n_in)), dtype=torch.int64, device=dev)
            dist.all_to_all_single(r[:sum(n_in)], send[k][:sum(n_out)], n_in, n_out, group=self.group)
            recv.append(r)
            tr["all_to_all_bytes"] += 8 * (sum(n_out) - n_out[me])
        self._sync()
        need = (C.c_uint64 * 2)()
        self._ck(L.fqsx_shard_merge(self._h, self._ptrs(recv), need), "fqsx_shard_merge")
        t_need = torch.tensor([need[0], need[1]], dtype=torch.int64, device=dev)
        dist.all_reduce(t_need, op=dist.ReduceOp.MAX, group=self.group)
        self._sync()
        n_items = [[int(vol[k][:, q].sum()) for q in range(G)] for k in range(3)]   # entries applied by rank q's owners
        items = [torch.zeros(max(1, max(n_items[k])), dtype=torch.int64, device=dev) for k in range(3)]   # padded to the largest
        delta = (C.c_uint64 * 2)()
        self._ck(L.fqsx_shard_insert(self._h, int(t_need[0].item()), int(t_need[1].item()), self._ptrs(items), delta), "fqsx_shard_insert")
        for k in range(3):
            parts = [torch.empty_like(items[k]) for _ in range(G)]
            dist.all_gather(parts, items[k], group=self.group)
            self._sync()
            for q in range(G):
                if (q != me or self.apply_own) and n_items[k][q]:
                    self._ck(L.fqsx_shard_apply(self._h, k, parts[q].data_ptr(), n_items[k][q]), "fqsx_shard_apply")
            self._sync()
            tr["all_gather_bytes"] += 8 * n_items[k][me] * (G - 1)
        t_delta = torch.tensor([delta[0], delta[1]], dtype=torch.int64, device=dev)
        dist.all_reduce(t_delta, op=dist.ReduceOp.SUM, group=self.group)
        self._sync()
        tot = (C.c_uint64 * 2)(int(t_delta[0].item()), int(t_delta[1].item()))
        self._ck(L.fqsx_shard_end_phase(self._h, tot), "fqsx_shard_end_phase")
        tr["phases"] += 1
        tr["collectives"] += 1 + 3 + 1 + 3 + 1
        tr["all_reduce_bytes"] += 4 * 3 * T * T + 16 + 16

    def close(self) -> None:
        self.codec.close()


# ---------------------------------------------------------------------------------------------------------------------
# The native driver: the phase loop lives in libfqsx.so (fqsx_shard_encode_block), three collectives and one host round
# trip per phase, single- and paired-end.  The transport is a C struct of three functions (include/fqsx.h: fqsx_comm):
#   * RCCL inside the library, on the codec's own stream (fqsx_rccl_comm_create) -- the product path on MI355X;
#   * callbacks into torch.distributed on buffers in host memory -- what the gloo tests run with the emulation build.
class _Comm(C.Structure):
    _AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)
    _A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64))
    _AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
    _AB = C.CFUNCTYPE(None, C.c_void_p)
    _fields_ = [("ctx", C.c_void_p), ("allreduce_sum_u32", _AR), ("alltoallv_u64", _A2A), ("allgather_u64", _AG), ("abort", _AB)]


def _host_tensor(addr: int, n: int, ctype, dtype) -> torch.Tensor:
    """n elements at a host address as a tensor that shares the memory"""
    return torch.from_numpy(np.ctypeslib.as_array((ctype * max(n, 1)).from_address(addr))[:n]).view(dtype)


def _torch_comm(world: int, group=None) -> _Comm:
    """fqsx_comm over torch.distributed for a codec whose memory space is host memory (the emulation build)."""
    def allreduce(_ctx, buf, n):
        try:
            dist.all_reduce(_host_tensor(buf, n, C.c_uint32, torch.int32), op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:   # noqa: BLE001 -- reported through the C ABI
            print("fqsx transport:", e)
            return 1

    def alltoallv(_ctx, n_buf, send, scnt, recv, rcnt):
        try:
            for b in range(n_buf):
                n_out = [int(scnt[b * world + r]) for r in range(world)]
                n_in = [int(rcnt[b * world + r]) for r in range(world)]
                s = _host_tensor(send[b], sum(n_out), C.c_uint64, torch.int64)
                r = _host_tensor(recv[b], sum(n_in), C.c_uint64, torch.int64)
                dist.all_to_all_single(r, s, n_in, n_out, group=group)
            return 0
        except Exception as e:   # noqa: BLE001
            print("fqsx transport:", e)
            return 1

    def allgather(_ctx, send, n, recv):
        try:
            out = _host_tensor(recv, n * world, C.c_uint64, torch.int64)
            dist.all_gather_into_tensor(out, _host_tensor(send, n, C.c_uint64, torch.int64).clone(), group=group)
            return 0
        except Exception as e:   # noqa: BLE001
            print("fqsx transport:", e)
            return 1

    return _Comm(None, _Comm._AR(allreduce), _Comm._A2A(alltoallv), _Comm._AG(allgather))


def _staged_comm(world: int, group=None) -> _Comm:
    """fqsx_comm for a codec whose memory space is HBM when RCCL cannot carry the world -- several ranks on ONE device, which
    RCCL refuses (the two-process test of the partitioned tables on a one-GPU box): every collective is staged through host
    memory (hipMemcpy) and done by torch.distributed on the CPU (gloo).  A test and bring-up transport, not a fast one."""
    # (device buffers are wrapped as torch tensors in place: loading libamdhip64 a second time beside torch's own copy
    # would put two HIP runtimes into the process)
    class _Dev:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    def dev(ptr, n, dtype):
        return torch.as_tensor(_Dev(ptr, n, "<i4" if dtype == torch.int32 else "<i8"), device="cuda")

    def down(ptr, n, dtype):
        return dev(ptr, n, dtype).cpu() if n else torch.empty(0, dtype=dtype)

    def up(ptr, t):
        if t.numel():
            dev(ptr, t.numel(), t.dtype).copy_(t)
            torch.cuda.synchronize()

    def guarded(f):
        def g(*a):
            try:
                torch.cuda.synchronize()   # (the codec's stream included: a device-wide wait)
                f(*a)
                return 0
            except Exception as e:   # noqa: BLE001 -- reported through the C ABI
                print("fqsx transport:", e)
                return 1
        return g

    @guarded
    def allreduce(_ctx, buf, n):
        t = down(buf, n, torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        up(buf, t)

    @guarded
    def alltoallv(_ctx, n_buf, send, scnt, recv, rcnt):
        for b in range(n_buf):
            n_out = [int(scnt[b * world + r]) for r in range(world)]
            n_in = [int(rcnt[b * world + r]) for r in range(world)]
            r = torch.empty(max(sum(n_in), 1), dtype=torch.int64)[:sum(n_in)]
            dist.all_to_all_single(r, down(send[b], sum(n_out), torch.int64), n_in, n_out, group=group)
            up(recv[b], r)

    @guarded
    def allgather(_ctx, send, n, recv):
        out = torch.empty(n * world, dtype=torch.int64)
        dist.all_gather_into_tensor(out, down(send, n, torch.int64).clone(), group=group)
        up(recv, out)

    return _Comm(None, _Comm._AR(allreduce), _Comm._A2A(alltoallv), _Comm._AG(allgather))


class NativeShardedDnaCodec:
    """One file's DNA path over `world` ranks with the phase loop inside the library (fqsx_shard_encode_block).
    partition = True: the k-mer tables are partitioned over the ranks (fqsx_shard_partition_tables; one node).
    transport = "rccl": RCCL on the codec's stream; `id_bytes` = the 128-byte id rank 0 made with rccl_unique_id() and
    every rank received (e.g. through torch.distributed's store).  transport = "torch": torch.distributed collectives on
    host buffers (gloo; emulation build).  transport = "staged": HBM buffers staged through the host (see _staged_comm)."""

    def __init__(self, header: bytes, rank: int, world: int, device: int = 0, lib_path: Optional[str] = None, transport: str = "rccl",
                 id_bytes: Optional[bytes] = None, group=None, partition: bool = False, comm: Optional[_Comm] = None):
        self.codec = DnaCodec(header, device=device, lib_path=lib_path)
        self._lib, self._h, self.T = self.codec._lib, self.codec._h, self.codec.T
        self.rank, self.world, self.device = rank, world, device
        L = self._lib
        L.fqsx_shard_attach.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(_Comm)]
        L.fqsx_shard_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.fqsx_shard_traffic.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.fqsx_rccl_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(_Comm)]
        L.fqsx_rccl_comm_destroy.argtypes = [C.POINTER(_Comm)]
        self._rccl = transport == "rccl"
        self._staged = transport == "staged"
        self._own_comm = True
        if self._rccl and comm is not None:   # the process's communicator, made with an earlier codec: goes on with this one's stream
            L.fqsx_rccl_comm_rebind.argtypes = [C.POINTER(_Comm), C.c_void_p]
            self._comm, self._own_comm = comm, False
            self._ck(L.fqsx_rccl_comm_rebind(C.byref(self._comm), self._h), "fqsx_rccl_comm_rebind")
        elif self._rccl:
            if id_bytes is None or len(id_bytes) != 128:
                raise ValueError("the RCCL transport needs the 128-byte unique id (rccl_unique_id on rank 0)")
            self._comm = _Comm()
            self._ck(L.fqsx_rccl_comm_create(self._h, bytes(id_bytes), rank, world, C.byref(self._comm)), "fqsx_rccl_comm_create")
        elif transport == "staged":
            self._comm = _staged_comm(world, group)
        else:
            self._comm = comm if comm is not None else _torch_comm(world, group)
        self._ck(L.fqsx_shard_attach(self._h, rank, world, C.byref(self._comm)), "fqsx_shard_attach")
        self.partitioned = False
        self.partition_note = ""
        if partition:   # (collective) each rank keeps 1/world of the k-mer tables and maps the rest from its peers
            L.fqsx_shard_partition_tables.argtypes = [C.c_void_p]
            L.fqsx_shard_is_partitioned.argtypes = [C.c_void_p]
            self._ck(L.fqsx_shard_partition_tables(self._h), "fqsx_shard_partition_tables")
            self.partitioned = bool(L.fqsx_shard_is_partitioned(self._h))
            if not self.partitioned:   # GPUs without peer access: the whole world stays with replicas (same streams)
                self.partition_note = self._lib.fqsx_last_error().decode()
                if rank == 0:
                    import warnings
                    warnings.warn("fqsx: " + self.partition_note)
        self.own_workers = list(range(rank, self.T, world))
        self._streams = (C.c_void_p * self.T)()
        self._lens = (C.c_uint64 * self.T)()

    @staticmethod
    def rccl_unique_id(lib_path: Optional[str] = None) -> bytes:
        from .codec import load_library
        lib = load_library(lib_path)
        buf = C.create_string_buffer(128)
        lib.fqsx_rccl_unique_id.argtypes = [C.c_char_p]
        if lib.fqsx_rccl_unique_id(buf):
            raise FqsxError(f"fqsx_rccl_unique_id: {lib.fqsx_last_error().decode()}")
        return buf.raw

    def _ck(self, rc: int, what: str) -> None:
        if rc:
            raise FqsxError(f"{what}: {rc}: {self._lib.fqsx_last_error().decode()}")

    def encode_block(self, bases: np.ndarray, read_off: np.ndarray, generation: int) -> Dict[int, bytes]:
        """All ranks call this with the same block; returns {worker: DNA stream} for this rank's workers."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        if self._rccl or self._staged:   # the codec's memory space is HBM
            dev = torch.device("cuda", self.device)
            t_b, t_o = torch.from_numpy(bases).to(dev), torch.from_numpy(read_off.view(np.int64)).to(dev)
            torch.cuda.synchronize(dev)
            pb, po = t_b.data_ptr(), t_o.data_ptr()
        else:
            pb, po = bases.ctypes.data, read_off.ctypes.data
        self._ck(self._lib.fqsx_shard_encode_block(self._h, pb, po, read_off.ctypes.data, len(read_off) - 1, generation, self._streams, self._lens),
                 "fqsx_shard_encode_block")
        return {w: (C.string_at(self._streams[w], self._lens[w]) if self._lens[w] else b"") for w in self.own_workers}

    def encode_block_dev(self, d_bases_ptr: int, d_off_ptr: int, read_off: np.ndarray, generation: int) -> int:
        """block already in HBM; returns the bytes of this rank's streams"""
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        self._ck(self._lib.fqsx_shard_encode_block(self._h, d_bases_ptr, d_off_ptr, read_off.ctypes.data, len(read_off) - 1, generation, self._streams, self._lens),
                 "fqsx_shard_encode_block")
        return sum(self._lens[w] for w in self.own_workers)

    @property
    def traffic(self) -> dict:
        a = (C.c_uint64 * 4)()
        self._lib.fqsx_shard_traffic(self._h, a)
        return {"phases": a[0], "collectives": a[1], "all_to_all_bytes": 8 * a[2], "all_gather_bytes": 8 * a[3]}

    def rccl_info(self) -> dict:
        """what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank)"""
        a = (C.c_uint32 * 2)()
        self._lib.fqsx_rccl_comm_info.argtypes = [C.POINTER(_Comm), C.POINTER(C.c_uint32)]
        self._ck(self._lib.fqsx_rccl_comm_info(C.byref(self._comm), a), "fqsx_rccl_comm_info")
        return {"ranks_seen": int(a[0]), "rank": int(a[1])}

    def detach_comm(self) -> "_Comm":
        """keep the RCCL communicator beyond this codec (hand it to the next NativeShardedDnaCodec as `comm`)"""
        self._own_comm = False
        return self._comm

    def destroy_comm(self) -> None:
        if self._rccl and self._comm.ctx:
            self._lib.fqsx_rccl_comm_destroy(C.byref(self._comm))

    def close(self) -> None:
        if self._rccl and self._own_comm and self._comm.ctx:
            self._lib.fqsx_rccl_comm_destroy(C.byref(self._comm))
        self.codec.close()
