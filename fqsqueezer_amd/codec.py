"""ctypes binding of the C ABI in include/fqsx.h (libfqsx.so, HIP/gfx950).

The product path has no CPU implementation: constructing a DnaCodec without the
built HIP library or without a GPU raises.  (tests/emu passes the path of the
host-emulation build of the *same kernels* explicitly; nothing in this package
ever selects it.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_PKG, "libfqsx.so")

STAT_NAMES = ["gprobe", "gslot", "lprobe", "lslot", "gins", "gins_slot", "siv_words", "ctx_slots",
              "coded", "lins", "mail", "bases", "siv_saved"]


class FqsxError(RuntimeError):
    pass


def _load(path: str):
    if not os.path.exists(path):
        raise FqsxError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:  # PyTorch ships its own ROCm runtime: load it first so the process has ONE HIP runtime
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.fqsx_dna_create.restype = C.c_int
    lib.fqsx_dna_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    if hasattr(lib, "fqsx_dna_create_on_partition"):   # (tools/ab_bench.py also loads builds that predate these entry points)
        lib.fqsx_dna_create_on_partition.restype = C.c_int
        lib.fqsx_dna_create_on_partition.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.fqsx_dna_destroy.argtypes = [C.c_void_p]
    lib.fqsx_dna_encode_block.restype = C.c_int
    lib.fqsx_dna_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.fqsx_dna_encode_block_dev.restype = C.c_int
    lib.fqsx_dna_encode_block_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.fqsx_dna_decode_block.restype = C.c_int
    lib.fqsx_dna_decode_block.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.fqsx_dna_stats.restype = C.c_int
    lib.fqsx_dna_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    if hasattr(lib, "fqsx_dna_capacity"):
        lib.fqsx_dna_capacity.restype = C.c_int
        lib.fqsx_dna_capacity.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.fqsx_dna_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.fqsx_dna_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.fqsx_qual_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.fqsx_qual_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.fqsx_qual_destroy.argtypes = [C.c_void_p]
    lib.fqsx_meta_create.argtypes = [C.c_uint32, C.POINTER(C.c_void_p)]
    lib.fqsx_meta_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.fqsx_meta_destroy.argtypes = [C.c_void_p]
    lib.fqsx_meta_encode_block_pe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.fqsx_id_create.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.fqsx_id_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.fqsx_id_destroy.argtypes = [C.c_void_p]
    lib.fqsx_sort_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    lib.fqsx_last_error.restype = C.c_char_p
    lib.fqsx_version.restype = C.c_char_p
    return lib


_libs = {}


def load_library(path: Optional[str] = None):
    path = path or DEFAULT_LIB
    if path not in _libs:
        _libs[path] = _load(path)
    return _libs[path]


class DnaCodec:
    """One .fqs file's DNA-stream encoder state on one GPU (fqsx_dna_*)."""

    def __init__(self, header: bytes, device: int = 0, lib_path: Optional[str] = None, partition: Optional[tuple] = None,
                 chunked_tables: bool = False):
        """partition = (k, n): confine the codec's kernels to the k-th of n equal sets of compute units (several files at once on one GPU).
        chunked_tables: the capacity mode (fqsx_dna_use_chunked_tables): a growth never holds the old and the new table side by side."""
        if len(header) != 17:
            raise ValueError("header must be the 17 .fqs parameter bytes")
        self._lib = load_library(lib_path)
        self.T = header[4]
        self._h = C.c_void_p()
        if partition is None:
            rc = self._lib.fqsx_dna_create(bytes(header), device, C.byref(self._h))
        else:
            rc = self._lib.fqsx_dna_create_on_partition(bytes(header), device, partition[0], partition[1], C.byref(self._h))
        if rc:
            raise FqsxError(f"fqsx_dna_create: {rc}: {self._lib.fqsx_last_error().decode()}")
        if chunked_tables:
            self._lib.fqsx_dna_use_chunked_tables.argtypes = [C.c_void_p]
            if self._lib.fqsx_dna_use_chunked_tables(self._h):
                raise FqsxError(f"fqsx_dna_use_chunked_tables: {self._lib.fqsx_last_error().decode()}")
        self._streams = (C.c_void_p * self.T)()
        self._lens = (C.c_uint64 * self.T)()

    def _collect(self) -> List[bytes]:
        return [C.string_at(self._streams[w], self._lens[w]) if self._lens[w] else b"" for w in range(self.T)]

    def encode_block(self, bases: np.ndarray, read_off: np.ndarray, generation: int) -> List[bytes]:
        """Host-buffer entry point: returns the T per-worker DNA streams of the block."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        rc = self._lib.fqsx_dna_encode_block(self._h, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                                             generation, self._streams, self._lens)
        if rc:
            raise FqsxError(f"fqsx_dna_encode_block: {rc}: {self._lib.fqsx_last_error().decode()}")
        return self._collect()

    def encode_block_dev(self, d_bases_ptr: int, d_off_ptr: int, read_off: np.ndarray, generation: int,
                         collect: bool = True):
        """Device-resident entry point (inputs already in HBM)."""
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        rc = self._lib.fqsx_dna_encode_block_dev(self._h, d_bases_ptr, d_off_ptr, read_off.ctypes.data,
                                                 len(read_off) - 1, generation, self._streams, self._lens)
        if rc:
            raise FqsxError(f"fqsx_dna_encode_block_dev: {rc}: {self._lib.fqsx_last_error().decode()}")
        if collect:
            return self._collect()
        return sum(self._lens[w] for w in range(self.T))

    def decode_block(self, streams, read_off: np.ndarray, generation: int) -> np.ndarray:
        """Inverse of encode_block: the T DNA streams of a block + read offsets -> concatenated base bytes."""
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        arr = (C.c_char_p * self.T)(*[bytes(x) for x in streams])
        lens = np.array([len(x) for x in streams], dtype=np.uint64)
        out = np.zeros(max(1, int(read_off[-1])), dtype=np.uint8)
        rc = self._lib.fqsx_dna_decode_block(self._h, arr, lens.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                                             generation, out.ctypes.data)
        if rc:
            raise FqsxError(f"fqsx_dna_decode_block: {rc}: {self._lib.fqsx_last_error().decode()}")
        return out[:int(read_off[-1])]

    def stats(self) -> dict:
        a = (C.c_uint64 * 64)()
        rc = self._lib.fqsx_dna_stats(self._h, a)
        if rc:
            raise FqsxError(f"fqsx_dna_stats: {rc}: {self._lib.fqsx_last_error().decode()}")
        d = dict(zip(STAT_NAMES, list(a)))
        d["timers"] = list(a)[16:64]
        return d

    def capacity(self) -> dict:
        """Table occupancy and device memory (fqsx_dna_capacity)."""
        a = (C.c_uint64 * 16)()
        rc = self._lib.fqsx_dna_capacity(self._h, a)
        if rc:
            raise FqsxError(f"fqsx_dna_capacity: {rc}: {self._lib.fqsx_last_error().decode()}")
        d = {"smers": a[0], "bmers": a[1], "smer_slots": a[2], "bmer_slots": a[3], "siv_bytes": a[4], "ctx_slots": a[5],
             "contexts": a[6], "device_bytes": a[7], "device_bytes_peak": a[8], "growths": a[9], "pairs": a[10], "pair_slots": a[11],
             "table_bytes_held": a[13], "pair_bytes_held": a[14], "siv_bytes_held": a[15]}
        d["bytes_per_bmer"] = round(a[3] * a[12] / a[1], 2) if a[1] else 0.0
        d["bytes_per_smer"] = round(a[2] * a[12] / a[0], 2) if a[0] else 0.0
        return d

    def set_profiling(self, on: bool) -> None:
        self._lib.fqsx_dna_set_profiling(self._h, int(on))

    def kernel_times(self) -> dict:
        a = (C.c_double * 6)()
        self._lib.fqsx_dna_kernel_times(self._h, a)
        return {"encode_ms": a[0], "insert_ms": a[1], "other_ms": a[2],
                "encode_launches": int(a[3]), "insert_launches": int(a[4]), "other_launches": int(a[5])}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.fqsx_dna_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MetaCodec:
    """Host-side read-length stream of the container (fqsx_meta_*)."""

    def __init__(self, threads: int, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        self.T = threads
        self._h = C.c_void_p()
        if self._lib.fqsx_meta_create(threads, C.byref(self._h)):
            raise FqsxError("fqsx_meta_create failed")
        self._streams = (C.c_void_p * threads)()
        self._lens = (C.c_uint64 * threads)()

    def encode_block(self, read_len: np.ndarray, paired: bool = False) -> List[bytes]:
        read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
        if self._lib.fqsx_meta_encode_block_pe(self._h, read_len.ctypes.data, len(read_len), int(paired), self._streams, self._lens):
            raise FqsxError("fqsx_meta_encode_block failed")
        return [C.string_at(self._streams[w], self._lens[w]) for w in range(self.T)]

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.fqsx_meta_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sort_order(bases: np.ndarray, read_off: np.ndarray, device: int = 0, lib_path: Optional[str] = None,
               max_batch_bases: int = 0, stats: Optional[dict] = None) -> List[np.ndarray]:
    """Read order of `fqs e -om s` (fqsx_sort_order: GPU radix sort + ranks, host replay of std::sort per bin).
    Returns one index array per non-empty bin, in bin order -- the same shape hostpipe.sorted_order_exact returns.
    max_batch_bases > 0: bounded device memory (fqsx_sort_order_batched: bins packed into batches of at most that many bases)."""
    lib = load_library(lib_path)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    n = len(read_off) - 1
    order = np.empty(max(n, 1), dtype=np.uint32)
    bins = np.zeros(257, dtype=np.uint32)
    nb = C.c_uint32(0)
    lib.fqsx_sort_order_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    rc = lib.fqsx_sort_order_batched(bases.ctypes.data, read_off.ctypes.data, n, device, max_batch_bases, order.ctypes.data, bins.ctypes.data, C.byref(nb))
    if rc:
        raise FqsxError(f"fqsx_sort_order: {rc}: {lib.fqsx_last_error().decode()}")
    if stats is not None:
        stats["batches"] = nb.value
    return [order[bins[b]:bins[b + 1]].astype(np.int64) for b in range(256) if bins[b + 1] > bins[b]]


class IdCodec:
    """Read-id stream of the container (header byte 7 = id mode).  device = None: the host coder (fqsx_id_*, one host thread per
    worker); device = ordinal: the GPU coder (fqsx_idg_*, one wavefront per worker) -- the same bytes."""

    def __init__(self, header: bytes, lib_path: Optional[str] = None, device: Optional[int] = None):
        self._lib = load_library(lib_path)
        self.T = header[4]
        self._h = C.c_void_p()
        self._gpu = device is not None
        L = self._lib
        if self._gpu:
            L.fqsx_idg_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
            L.fqsx_idg_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
            L.fqsx_idg_destroy.argtypes = [C.c_void_p]
            L.fqsx_idg_destroy.restype = None
            if L.fqsx_idg_create(bytes(header), device, C.byref(self._h)):
                raise FqsxError(f"fqsx_idg_create: {L.fqsx_last_error().decode()}")
        elif L.fqsx_id_create(bytes(header), C.byref(self._h)):
            raise FqsxError("fqsx_id_create failed")
        self._streams = (C.c_void_p * self.T)()
        self._lens = (C.c_uint64 * self.T)()

    def encode_block(self, ids: np.ndarray, id_off: np.ndarray, paired: bool = False) -> List[bytes]:
        ids = np.ascontiguousarray(ids, dtype=np.uint8)
        id_off = np.ascontiguousarray(id_off, dtype=np.uint64)
        f = self._lib.fqsx_idg_encode_block if self._gpu else self._lib.fqsx_id_encode_block
        rc = f(self._h, ids.ctypes.data, id_off.ctypes.data, len(id_off) - 1, int(paired), self._streams, self._lens)
        if rc:
            raise FqsxError(f"fqsx_id{'g' if self._gpu else ''}_encode_block: {rc}: {self._lib.fqsx_last_error().decode()}")
        return [C.string_at(self._streams[w], self._lens[w]) for w in range(self.T)]

    def close(self) -> None:
        if getattr(self, "_h", None):
            (self._lib.fqsx_idg_destroy if self._gpu else self._lib.fqsx_id_destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QualCodec:
    """Quality-stream encoder on the GPU (fqsx_qual_*)."""

    def __init__(self, header: bytes, device: int = 0, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        self.T = header[4]
        self._h = C.c_void_p()
        rc = self._lib.fqsx_qual_create(bytes(header), device, C.byref(self._h))
        if rc:
            raise FqsxError(f"fqsx_qual_create: {rc}: {self._lib.fqsx_last_error().decode()}")
        self._streams = (C.c_void_p * self.T)()
        self._lens = (C.c_uint64 * self.T)()

    def encode_block(self, quals: np.ndarray, read_off: np.ndarray) -> List[bytes]:
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        rc = self._lib.fqsx_qual_encode_block(self._h, quals.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                                              self._streams, self._lens)
        if rc:
            raise FqsxError(f"fqsx_qual_encode_block: {rc}: {self._lib.fqsx_last_error().decode()}")
        return [C.string_at(self._streams[w], self._lens[w]) if self._lens[w] else b"" for w in range(self.T)]

    def encode_block_dev(self, d_quals_ptr: int, d_off_ptr: int, read_off: np.ndarray) -> int:
        """Device-resident entry point; returns the total stream bytes of the block."""
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        self._lib.fqsx_qual_encode_block_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        rc = self._lib.fqsx_qual_encode_block_dev(self._h, d_quals_ptr, d_off_ptr, read_off.ctypes.data, len(read_off) - 1, self._streams, self._lens)
        if rc:
            raise FqsxError(f"fqsx_qual_encode_block_dev: {rc}: {self._lib.fqsx_last_error().decode()}")
        return sum(self._lens[w] for w in range(self.T))

    def set_profiling(self, on: bool) -> None:
        self._lib.fqsx_qual_set_profiling.argtypes = [C.c_void_p, C.c_int]
        self._lib.fqsx_qual_set_profiling(self._h, int(on))

    def kernel_times(self) -> dict:
        a = (C.c_double * 2)()
        self._lib.fqsx_qual_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._lib.fqsx_qual_kernel_times(self._h, a)
        return {"encode_ms": a[0], "encode_launches": int(a[1])}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.fqsx_qual_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
