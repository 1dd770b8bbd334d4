"""Host-side logic around the DNA hot path: .fqs header, read binning/sorting,
reads-block formation and the .fqs container (parse + write).

Everything here mirrors host code of the reference that decides *what the DNA
path is fed* (block boundaries, read order) or *where its output goes*
(container).  Reference citations are relative to /root/reference/fqs.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

READS_BLOCK_SIZE = 16 << 20          # application.h:33
BLOCK_SIZE_MARGIN = 102400           # reads_block.h:25
NO_BINS = 256                        # defs.h:22-23
STREAM_META, STREAM_ID, STREAM_DNA, STREAM_QUALITY = 0, 1, 2, 3   # defs.h:33-37

DNA_MODES = {"se_original": 0, "se_sorted": 1, "pe_original": 2, "pe_sorted": 3}   # params.h:18
QUALITY_MODES = {"lossless": 0, "illumina_8": 1, "illumina_4": 2, "binary": 3, "none": 4}
ID_MODES = {"lossless": 0, "instrument": 1, "none": 2}

# params.h:133-143: genome size (Mbp) -> prefix, pmer, smer, bmer, imer
_KMER_THR = [
    (1, 9, 14, 17, 19, 18), (4, 9, 15, 18, 20, 19), (16, 10, 15, 18, 21, 20),
    (64, 11, 16, 18, 23, 21), (256, 12, 17, 20, 24, 22), (1024, 12, 17, 21, 26, 24),
    (4096, 13, 18, 21, 27, 24), (16384, 14, 18, 22, 27, 25), (65536, 15, 18, 22, 27, 25),
]


def kmer_lengths(genome_size_mbp: int) -> Tuple[int, int, int, int, int]:
    """adjust_kmer_sizes(), params.h:131-155."""
    for row in _KMER_THR:
        if genome_size_mbp <= row[0]:
            return row[1:]
    return (14, 13, 15, 26, 26)      # CParams() defaults, params.h:69-75 (order: prefix,pmer,smer,bmer,imer)


def make_header(threads: int, dna_mode: str = "se_sorted", genome_size_mbp: int = 3100,
                quality_mode: str = "none", id_mode: str = "none", quality_thr: int = 20) -> bytes:
    """The 17 parameter bytes of a .fqs file, store_params(), params.h:80-100."""
    if not 1 <= threads <= 255:
        raise ValueError("threads must be in 1..255 (one header byte)")
    prefix, pmer, smer, bmer, imer = kmer_lengths(genome_size_mbp)
    return bytes([ord("K"), ord("C"), ord("S"), ord("D"), threads, DNA_MODES[dna_mode],
                  QUALITY_MODES[quality_mode], ID_MODES[id_mode], quality_thr, 1,
                  prefix, pmer, smer, bmer, imer, 31, 11])


# ----------------------------------------------------------------------------------------
# FASTQ records
@dataclass
class Records:
    """FASTQ records held column-wise.  `seq`/`qual` are (n, L) uint8 arrays for
    fixed-length data (the synthetic workloads) or lists of bytes."""
    ids: Sequence[bytes]
    seq: "np.ndarray | List[bytes]"
    qual: "np.ndarray | List[bytes]"
    plus: bytes = b"+"
    plus_len: Optional[np.ndarray] = None   # per-record length of the separator line when the file has `+id` lines (read_fastq)

    def __len__(self) -> int:
        return len(self.ids)

    def seq_bytes(self, i: int) -> bytes:
        s = self.seq[i]
        return s.tobytes() if isinstance(s, np.ndarray) else s

    def qual_bytes(self, i: int) -> bytes:
        q = self.qual[i]
        return q.tobytes() if isinstance(q, np.ndarray) else q

    def record_sizes(self) -> np.ndarray:
        """read_size(), defs.h:79-81: all four lines incl. their EOLs."""
        n = len(self)
        idl = np.fromiter((len(x) for x in self.ids), dtype=np.int64, count=n)
        if isinstance(self.seq, np.ndarray):
            sl = np.full(n, self.seq.shape[1], dtype=np.int64)
        else:
            sl = np.fromiter((len(x) for x in self.seq), dtype=np.int64, count=n)
        pl = len(self.plus) if self.plus_len is None else np.asarray(self.plus_len, dtype=np.int64)
        return idl + 1 + sl + 1 + pl + 1 + sl + 1


def read_fastq(path: str) -> Records:
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    n = len(lines) // 4
    ids = lines[0:4 * n:4]
    seq = lines[1:4 * n:4]
    qual = lines[3:4 * n:4]
    # The separator line's text is not stored in a .fqs, but its length is part of read_size() (defs.h:79-81), which
    # places the block boundaries (reads_block.h:119-139).  Lines are split at 0x0A only, as the reference does
    # (io.h:447-478): in a CRLF file the 0x0D stays part of every field, the DNA included.
    pl = np.fromiter((len(x) for x in lines[2:4 * n:4]), dtype=np.int64, count=n)
    plus_len = None if n == 0 or bool((pl == 1).all()) else pl
    L = len(seq[0]) if n else 0
    if n and all(len(s) == L for s in seq):
        seq_a = np.frombuffer(b"".join(seq), dtype=np.uint8).reshape(n, L)
        qual_a = np.frombuffer(b"".join(qual), dtype=np.uint8).reshape(n, L)
        return Records(ids, seq_a, qual_a, plus_len=plus_len)
    return Records(ids, seq, qual, plus_len=plus_len)


_NT = bytes(i if i in b"ACG" else ord("T") for i in range(256))      # io.h:563-571 (N and everything else -> T)
_NT_ARR = np.frombuffer(_NT, dtype=np.uint8)
_CODE_NT = np.full(256, 3, dtype=np.int64)
_CODE_NT[ord("A")], _CODE_NT[ord("C")], _CODE_NT[ord("G")] = 0, 1, 2


def sorted_order(rec: Records) -> List[np.ndarray]:
    """Read order of `fqs e -om s`: 256 bins by the first 4 bases with N->T
    (preprocess_se, application.cpp:349-412), each bin sorted by
    (N->T sequence, length, raw sequence) (sort_reads, io.h:499-528).  Reads
    that compare equal have identical DNA, so the DNA stream does not depend on
    how std::sort orders them; ties are kept in input order here.
    Returns one index array per non-empty bin, in bin order."""
    n = len(rec)
    if isinstance(rec.seq, np.ndarray):
        L = rec.seq.shape[1]
        raw = np.ascontiguousarray(rec.seq).view(f"S{L}").reshape(n)
        nt = np.ascontiguousarray(_NT_ARR[rec.seq]).view(f"S{L}").reshape(n)
        order = np.lexsort((np.arange(n), raw, nt))
        first4 = _NT_ARR[rec.seq[:, :4]]
        bins = ((_CODE_NT[first4[:, 0]] * 4 + _CODE_NT[first4[:, 1]]) * 4 + _CODE_NT[first4[:, 2]]) * 4 + _CODE_NT[first4[:, 3]]
        sb = bins[order]
        # the global (nt, raw) order is consistent with bin order because bins are the nt 4-prefix
        cuts = np.flatnonzero(np.diff(sb)) + 1
        return [a for a in np.split(order, cuts) if len(a)]
    keyed = sorted(range(n), key=lambda i: (rec.seq[i].translate(_NT), len(rec.seq[i]), rec.seq[i], i))
    out: List[List[int]] = []
    last = None
    for i in keyed:
        b = rec.seq[i][:4].translate(_NT)
        if b != last:
            out.append([])
            last = b
        out[-1].append(i)
    return [np.asarray(x, dtype=np.int64) for x in out]


def form_blocks(rec: Records, dna_mode: str = "se_sorted", exact_ties: bool = False,
                groups: "List[np.ndarray] | None" = None) -> List[np.ndarray]:
    """Partition the input into reads blocks exactly as the reference's reader
    does (CReadsBlock::Read, reads_block.h:119-139): records are appended until
    fewer than 102400 bytes of the 16 MiB buffer remain.  In sorted mode every
    non-empty bin is a separate input file (compress_se_files, application.cpp:538-569).
    exact_ties: order reads with identical DNA exactly as the reference's std::sort does (matters only for
    the id / quality / meta streams, whose records follow the DNA order).
    groups: precomputed per-bin read order (e.g. from codec.sort_order, the GPU pre-pass).
    Returns index arrays (into `rec`), one per block, in file order."""
    sizes = rec.record_sizes()
    if groups is not None:
        pass
    elif dna_mode == "se_sorted":
        groups = sorted_order_exact(rec) if exact_ties else sorted_order(rec)
    else:
        groups = [np.arange(len(rec), dtype=np.int64)]
    blocks: List[np.ndarray] = []
    for g in groups:
        cs = np.cumsum(sizes[g])
        start, base = 0, 0
        while start < len(g):
            # first j with READS_BLOCK_SIZE - (cs[j]-base) < margin
            j = int(np.searchsorted(cs, base + READS_BLOCK_SIZE - BLOCK_SIZE_MARGIN, side="right"))
            end = min(j + 1, len(g))
            blocks.append(g[start:end])
            start = end
            if end < len(g) + 1 and end > 0:
                base = int(cs[end - 1])
    return blocks


def block_arrays(rec: Records, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenated base bytes + n+1 offsets for one block (the encode_block ABI)."""
    if isinstance(rec.seq, np.ndarray):
        L = rec.seq.shape[1]
        bases = np.ascontiguousarray(rec.seq[idx]).reshape(-1)
        off = np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(L)
        return bases, off
    parts = [rec.seq[int(i)] for i in idx]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    return np.frombuffer(b"".join(parts), dtype=np.uint8), off


def partition_for_workers(n_reads: int, workers: int) -> List[Tuple[int, int]]:
    """PartitionForWorkers, reads_block.h:197-214."""
    out, lower = [], 0
    for i in range(workers):
        upper = (i + 1) * n_reads // workers
        if i < workers - 1:
            upper &= ~1
        out.append((lower, upper))
        lower = upper
    return out


def no_synchronizations(generation: int, n_reads: int, workers: int) -> int:
    """calc_no_synchronizations, application.h:85-92."""
    r = 100 - generation if generation < 100 else 0
    r = max(0, min(r, n_reads // workers // 2))
    return r - 1 if r else 0


# ----------------------------------------------------------------------------------------
# .fqs container (application.cpp:674-728, io.h:131-156,300-322)
def put_varint(x: int) -> bytes:
    if x < 0x8000:
        return bytes([x >> 8, x & 0xFF])
    if x < 0x400000:
        return bytes([0x80 + (x >> 16), (x >> 8) & 0xFF, x & 0xFF])
    if x < 0x40000000:
        return bytes([0xC0 + (x >> 24), (x >> 16) & 0xFF, (x >> 8) & 0xFF, x & 0xFF])
    raise ValueError("value too large for the .fqs varint")


def get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    c = buf[pos]
    if c >> 7 == 0:
        return (c << 8) + buf[pos + 1], pos + 2
    if c >> 6 == 0b10:
        return ((c & 0x3F) << 16) + (buf[pos + 1] << 8) + buf[pos + 2], pos + 3
    return ((c & 0x3F) << 24) + (buf[pos + 1] << 16) + (buf[pos + 2] << 8) + buf[pos + 3], pos + 4


@dataclass
class FqsBlock:
    n_reads: int
    offsets: List[int] = field(default_factory=list)              # per worker
    streams: List[dict] = field(default_factory=list)             # per worker: {stream_id: bytes}


def stored_streams(header: bytes) -> List[int]:
    """a_store_stream, application.cpp:533."""
    s = [STREAM_META]
    if header[7] != ID_MODES["none"]:
        s.append(STREAM_ID)
    s.append(STREAM_DNA)
    if header[6] != QUALITY_MODES["none"]:
        s.append(STREAM_QUALITY)
    return s


def parse_fqs(data: bytes) -> Tuple[bytes, List[FqsBlock]]:
    if data[0] != 17:
        raise ValueError("not a .fqs file (header length byte)")
    header = bytes(data[1:18])
    if header[:4] != b"KCSD":
        raise ValueError("not a .fqs file (magic)")
    T = header[4]
    sids = stored_streams(header)
    pos, blocks = 18, []
    while pos < len(data):
        n, pos = get_varint(data, pos)
        blk = FqsBlock(n)
        for _ in range(T):
            off, pos = get_varint(data, pos)
            blk.offsets.append(off)
            st = {}
            for sid in sids:
                size, pos = get_varint(data, pos)
                st[sid] = bytes(data[pos:pos + size])
                pos += size
            blk.streams.append(st)
        blocks.append(blk)
    return header, blocks


def fqs_chunks(header: bytes, blocks: Iterable[FqsBlock]):
    """The file as a sequence of byte chunks: the header, then one chunk per container block."""
    yield bytes([17]) + header
    sids = stored_streams(header)
    for b in blocks:
        out = [put_varint(b.n_reads)]
        for off, st in zip(b.offsets, b.streams):
            out.append(put_varint(off))
            for sid in sids:
                out.append(put_varint(len(st[sid])))
                out.append(st[sid])
        yield b"".join(out)


def write_fqs(header: bytes, blocks: Iterable[FqsBlock]) -> bytes:
    return b"".join(fqs_chunks(header, blocks))


# ----------------------------------------------------------------------------------------
# paired-end host logic
def _host_lib():
    """libfqsx_host.so: the CPU-only helpers of csrc/fqsx_host.cpp (exact std::sort bin order, meta coder)."""
    import ctypes as C
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "libfqsx_host.so")
    src = os.path.join(here, "csrc", "fqsx_host.cpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        # (several ranks / processes may come here at once: build under a private name, then rename atomically)
        tmp = "%s.tmp%d" % (lib, os.getpid())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", tmp])
        os.replace(tmp, lib)
    h = C.CDLL(lib)
    h.fqsx_sort_bin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    return h


def sorted_order_exact(rec: Records) -> List[np.ndarray]:
    """Like sorted_order(), but reads that compare equal are ordered exactly as the reference's
    std::sort leaves them (needed for paired-end data, where the mates follow this order)."""
    n = len(rec)
    all_idx = np.arange(n, dtype=np.int64)
    bases, off = block_arrays(rec, all_idx)
    bases = np.ascontiguousarray(bases)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    if isinstance(rec.seq, np.ndarray):
        first4 = _NT_ARR[rec.seq[:, :4]]
        bins = ((_CODE_NT[first4[:, 0]] * 4 + _CODE_NT[first4[:, 1]]) * 4 + _CODE_NT[first4[:, 2]]) * 4 + _CODE_NT[first4[:, 3]]
    else:
        bins = np.array([sum(_CODE_NT[c] << (2 * (3 - k)) for k, c in enumerate(s[:4])) for s in rec.seq], dtype=np.int64)
    h = _host_lib()
    out = []
    for b in range(NO_BINS):
        members = np.flatnonzero(bins == b).astype(np.uint32)     # input order inside the bin file
        if not len(members):
            continue
        res = np.empty_like(members)
        rc = h.fqsx_sort_bin(bases.ctypes.data, off.ctypes.data, members.ctypes.data, len(members), res.ctypes.data)
        assert rc == 0
        out.append(res.astype(np.int64))
    return out


def form_blocks_pe(rec1: Records, rec2: Records, dna_mode: str = "pe_sorted", groups: "List[np.ndarray] | None" = None) -> List[np.ndarray]:
    """Blocks of *pair* indices (CReadsBlock::Read(f1,f2), reads_block.h:141-166: pairs are appended
    until fewer than 2*102400 bytes remain; sorted mode: one input file pair per non-empty bin, mates
    follow mate 1's order, io.h:541-550)."""
    sizes = rec1.record_sizes() + rec2.record_sizes()
    if groups is None:
        groups = sorted_order_exact(rec1) if dna_mode == "pe_sorted" else [np.arange(len(rec1), dtype=np.int64)]
    blocks: List[np.ndarray] = []
    for g in groups:
        cs = np.cumsum(sizes[g])
        start, base = 0, 0
        while start < len(g):
            j = int(np.searchsorted(cs, base + READS_BLOCK_SIZE - 2 * BLOCK_SIZE_MARGIN, side="right"))
            end = min(j + 1, len(g))
            blocks.append(g[start:end])
            start = end
            base = int(cs[end - 1])
    return blocks


def block_arrays_pe(rec1: Records, rec2: Records, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Interleaved mates (r1_0, r2_0, r1_1, r2_1, ...) of one block: bases + 2*n_pairs+1 offsets."""
    if isinstance(rec1.seq, np.ndarray) and isinstance(rec2.seq, np.ndarray) and rec1.seq.shape[1] == rec2.seq.shape[1]:
        L = rec1.seq.shape[1]
        both = np.stack([rec1.seq[idx], rec2.seq[idx]], axis=1)      # (n, 2, L)
        return both.reshape(-1), np.arange(2 * len(idx) + 1, dtype=np.uint64) * np.uint64(L)
    parts = []
    for i in idx:
        parts.append(rec1.seq_bytes(int(i)))
        parts.append(rec2.seq_bytes(int(i)))
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts])
    return np.frombuffer(b"".join(parts), dtype=np.uint8), off


def qual_arrays(rec: Records, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenated quality bytes + n+1 offsets of one block (same shape as block_arrays)."""
    if isinstance(rec.qual, np.ndarray):
        L = rec.qual.shape[1]
        return np.ascontiguousarray(rec.qual[idx]).reshape(-1), np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(L)
    parts = [rec.qual[int(i)] for i in idx]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    return np.frombuffer(b"".join(parts), dtype=np.uint8), off


def id_arrays(rec: Records, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenated id lines (each with its EOL, as read_desc_t::id_len counts it) + n+1 offsets of one block."""
    parts = [rec.ids[int(i)] + b"\n" for i in idx]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    return np.frombuffer(b"".join(parts), dtype=np.uint8), off


def id_arrays_pe(rec1: Records, rec2: Records, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    parts = []
    for i in idx:
        parts.append(rec1.ids[int(i)] + b"\n")
        parts.append(rec2.ids[int(i)] + b"\n")
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts])
    return np.frombuffer(b"".join(parts), dtype=np.uint8), off


def qual_arrays_pe(rec1: Records, rec2: Records, idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    if isinstance(rec1.qual, np.ndarray) and isinstance(rec2.qual, np.ndarray) and rec1.qual.shape[1] == rec2.qual.shape[1]:
        L = rec1.qual.shape[1]
        both = np.stack([rec1.qual[idx], rec2.qual[idx]], axis=1)
        return both.reshape(-1), np.arange(2 * len(idx) + 1, dtype=np.uint64) * np.uint64(L)
    parts = []
    for i in idx:
        parts.append(rec1.qual_bytes(int(i)))
        parts.append(rec2.qual_bytes(int(i)))
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts])
    return np.frombuffer(b"".join(parts), dtype=np.uint8), off
