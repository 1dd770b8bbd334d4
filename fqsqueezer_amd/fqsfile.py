"""Write a complete `.fqs` file around the GPU streams, readable by the reference decompressor `fqs d` and
byte-identical to what `fqs e -t <threads>` writes for the same input and modes.  Host plumbing: block
formation, worker offsets, the container, the meta and id streams (host C++ helpers,
csrc/fqsx_host.cpp); the DNA and quality streams come from the GPU (codec.DnaCodec / codec.QualCodec)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import hostpipe as hp
from .codec import DnaCodec, IdCodec, MetaCodec, QualCodec, sort_order


def _gpu_groups(rec: hp.Records, device: int, lib_path: Optional[str]):
    """Sorted-mode read order from the GPU pre-pass (bins + per-bin order incl. the reference's order of equal reads)."""
    bases, off = hp.block_arrays(rec, np.arange(len(rec), dtype=np.int64))
    return sort_order(bases, off, device=device, lib_path=lib_path)


def encode_blocks(header: bytes, blocks, arrays, sizes_of, paired: bool, device: int, lib_path: Optional[str], stats: Optional[dict] = None,
                  gpu_ids: bool = True):
    """Generator of the file's container blocks.  Per block the four coders run side by side, as the reference's worker
    codes meta, id, DNA and quality of a read in one loop (application.cpp:633-641): the DNA kernels and the quality kernel
    and the id kernel on their own HIP streams (host threads inside the C ABI, which releases the GIL), the meta coder on a host
    thread meanwhile."""
    from concurrent.futures import ThreadPoolExecutor
    threads = header[4]
    stored = hp.stored_streams(header)
    dna = DnaCodec(header, device=device, lib_path=lib_path)
    meta = MetaCodec(threads, lib_path=lib_path)
    # (the id stream comes from the GPU coder too: fqsx_idg_*, one wavefront per worker on a stream of its own; gpu_ids = False:
    # the host coder, one host thread per worker -- the same bytes)
    idc = IdCodec(header, lib_path=lib_path, device=device if gpu_ids else None) if hp.STREAM_ID in stored else None
    qual = QualCodec(header, device=device, lib_path=lib_path) if hp.STREAM_QUALITY in stored else None
    pool = ThreadPoolExecutor(max_workers=3)
    try:
        for g, idx in enumerate(blocks):
            bases, off, ids, id_off, quals = arrays(idx)
            n = len(off) - 1
            jobs = {hp.STREAM_DNA: pool.submit(dna.encode_block, bases, off, g)}
            if qual is not None:
                jobs[hp.STREAM_QUALITY] = pool.submit(qual.encode_block, quals, off)
            if idc is not None:
                jobs[hp.STREAM_ID] = pool.submit(idc.encode_block, ids, id_off, paired)
            st = {hp.STREAM_META: meta.encode_block(np.diff(off.astype(np.int64)).astype(np.uint32), paired)}
            for k, f in jobs.items():
                st[k] = f.result()
            cs = np.concatenate([[0], np.cumsum(sizes_of(idx))])
            blk = hp.FqsBlock(n)
            for w, (first, _) in enumerate(hp.partition_for_workers(n, threads)):
                blk.offsets.append(int(cs[first]) if n else 0)   # application.cpp:716
                blk.streams.append({k: v[w] for k, v in st.items()})
            yield blk
        if stats is not None:
            stats["dna"] = dna.stats()
            stats["dna_capacity"] = dna.capacity()
            try:   # everything this process holds on the device at the end of the file: DNA tables + quality models + id models + block buffers
                import torch
                free, total = torch.cuda.mem_get_info(device)
                stats["device_bytes_in_use"] = int(total - free)
            except Exception:   # noqa: BLE001 -- extra information only
                pass
    finally:
        pool.shutdown(wait=True)
        dna.close()
        meta.close()
        if idc is not None:
            idc.close()
        if qual is not None:
            qual.close()


def _encode(header: bytes, blocks, arrays, sizes_of, paired: bool, device: int, lib_path: Optional[str], gpu_ids: bool = True) -> bytes:
    return hp.write_fqs(header, encode_blocks(header, blocks, arrays, sizes_of, paired, device, lib_path, gpu_ids=gpu_ids))


# the id kernel stages an id line in LDS: lines up to 1024 bytes, 128 tokens, instrument names up to 63 bytes (csrc/fqsx_idk.h:15-17)
_ID_LINE_MAX, _ID_TOKENS_MAX = 1024, 128


def _ids_fit_the_kernel(*recs) -> bool:
    """False if an id of the file is beyond what the GPU id coder stages (the host coder, like the reference, has no limits):
    a cheap pre-scan so that the file falls back to the host coder as a whole instead of failing in mid-file."""
    for rec in recs:
        for x in rec.ids:
            if len(x) > _ID_LINE_MAX or len(x) > 64 and sum(ch in b" :._/-|=#" for ch in x) >= _ID_TOKENS_MAX // 2:
                return False
    return True


def compress_records(rec: hp.Records, threads: int, order: str = "s", genome_size_mbp: int = 3100, device: int = 0,
                     lib_path: Optional[str] = None, quality_mode: str = "none", id_mode: str = "none",
                     quality_thr: int = 20, as_blocks: bool = False, gpu_ids: Optional[bool] = None):
    """`fqs e -s -om <order> -t <threads> -gs <g> -qm <..> -im <..>` on single-end records.  Returns the file's bytes, or
    -- as_blocks -- (header, generator of container blocks) for files too large to hold (hostpipe.fqs_chunks serialises them).
    gpu_ids: the id stream from the GPU kernel (True), from the host coder (False: one thread per worker, no limits on the id
    lines), or -- None -- the kernel unless an id of the file is beyond its staging limits."""
    if gpu_ids is None:
        gpu_ids = id_mode == "none" or _ids_fit_the_kernel(rec)
    mode = "se_sorted" if order == "s" else "se_original"
    header = hp.make_header(threads, mode, genome_size_mbp, quality_mode, id_mode, quality_thr)
    sizes = rec.record_sizes()

    def arrays(idx):
        bases, off = hp.block_arrays(rec, idx)
        ids, id_off = hp.id_arrays(rec, idx) if id_mode != "none" else (None, None)
        quals = hp.qual_arrays(rec, idx)[0] if quality_mode != "none" else None
        return bases, off, ids, id_off, quals

    groups = _gpu_groups(rec, device, lib_path) if mode == "se_sorted" else None
    if as_blocks:
        return header, encode_blocks(header, hp.form_blocks(rec, mode, groups=groups), arrays, lambda idx: sizes[idx], False, device, lib_path, gpu_ids=gpu_ids)
    return _encode(header, hp.form_blocks(rec, mode, groups=groups), arrays, lambda idx: sizes[idx], False, device, lib_path, gpu_ids)


def compress_records_pe(rec1: hp.Records, rec2: hp.Records, threads: int, order: str = "s", genome_size_mbp: int = 3100,
                        device: int = 0, lib_path: Optional[str] = None, quality_mode: str = "none", id_mode: str = "none",
                        quality_thr: int = 20, as_blocks: bool = False, gpu_ids: Optional[bool] = None, stats: Optional[dict] = None):
    """`fqs e -p ...` on two mate files (records interleaved mate 1 / mate 2 inside a block).  gpu_ids: see compress_records."""
    if gpu_ids is None:
        gpu_ids = id_mode == "none" or _ids_fit_the_kernel(rec1, rec2)
    mode = "pe_sorted" if order == "s" else "pe_original"
    header = hp.make_header(threads, mode, genome_size_mbp, quality_mode, id_mode, quality_thr)
    s1, s2 = rec1.record_sizes(), rec2.record_sizes()

    def arrays(idx):
        bases, off = hp.block_arrays_pe(rec1, rec2, idx)
        ids, id_off = hp.id_arrays_pe(rec1, rec2, idx) if id_mode != "none" else (None, None)
        quals = hp.qual_arrays_pe(rec1, rec2, idx)[0] if quality_mode != "none" else None
        return bases, off, ids, id_off, quals

    def sizes_of(idx):
        z = np.empty(2 * len(idx), dtype=np.int64)
        z[0::2], z[1::2] = s1[idx], s2[idx]
        return z

    groups = _gpu_groups(rec1, device, lib_path) if mode == "pe_sorted" else None   # mates follow mate 1's order, io.h:541-550
    if as_blocks:
        return header, encode_blocks(header, hp.form_blocks_pe(rec1, rec2, mode, groups=groups), arrays, sizes_of, True, device, lib_path, stats=stats, gpu_ids=gpu_ids)
    return _encode(header, hp.form_blocks_pe(rec1, rec2, mode, groups=groups), arrays, sizes_of, True, device, lib_path, gpu_ids)
