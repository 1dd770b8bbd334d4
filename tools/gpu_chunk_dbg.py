#!/usr/bin/env python3
"""debug: chunked tables vs plain tables on the GPU (c1 10k reads, sorted, T = 4), codecs run one after the other"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import c1_records
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import DnaCodec
rec = c1_records()
header = hp.make_header(4, "se_sorted", 1)
blks = hp.form_blocks(rec, "se_sorted")
for rep in range(3):
  for init in ("256", "1024"):
    os.environ["FQSX_GTAB_INIT"] = init
    out = []
    for chunked in (True, False):
        c = DnaCodec(header, device=0, chunked_tables=chunked, lib_path=os.environ.get("FQSX_LIB"))
        h = []
        for g, idx in enumerate(blks):
            bases, off = hp.block_arrays(rec, idx)
            h.append(hashlib.md5(b"".join(c.encode_block(bases, off, g))).hexdigest()[:6])
        out.append(h); c.close()
    bad = [g for g in range(len(blks)) if out[0][g] != out[1][g]]
    print("rep", rep, "init", init, "first differing blocks:", bad[:5], "of", len(blks), flush=True)
