"""Write a complete `.fqs` file (DNA-only modes: `-qm n -im n`) around the GPU DNA streams, readable by
the reference decompressor `fqs d`.  Host plumbing: block formation, worker offsets, the meta stream
and the container; the DNA streams come from fqsqueezer_amd.codec.DnaCodec."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import hostpipe as hp
from .codec import DnaCodec, MetaCodec


def compress_records(rec: hp.Records, threads: int, order: str = "s", genome_size_mbp: int = 3100, device: int = 0,
                     lib_path: Optional[str] = None) -> bytes:
    mode = "se_sorted" if order == "s" else "se_original"
    header = hp.make_header(threads, mode, genome_size_mbp, "none", "none")
    dna = DnaCodec(header, device=device, lib_path=lib_path)
    meta = MetaCodec(threads, lib_path=lib_path)
    sizes = rec.record_sizes()
    out = []
    for g, idx in enumerate(hp.form_blocks(rec, mode)):
        bases, off = hp.block_arrays(rec, idx)
        d = dna.encode_block(bases, off, g)
        m = meta.encode_block(np.diff(off.astype(np.int64)).astype(np.uint32))
        cs = np.concatenate([[0], np.cumsum(sizes[idx])])
        blk = hp.FqsBlock(len(idx))
        for w, (first, _) in enumerate(hp.partition_for_workers(len(idx), threads)):
            blk.offsets.append(int(cs[first]) if len(idx) else 0)
            blk.streams.append({hp.STREAM_META: m[w], hp.STREAM_DNA: d[w]})
        out.append(blk)
    dna.close()
    meta.close()
    return hp.write_fqs(header, out)
