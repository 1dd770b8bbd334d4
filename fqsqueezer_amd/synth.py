"""Seeded synthetic FASTQ generator (SURVEY.md §8d workload definition).

Uniform-random genome of length G, reads sampled uniformly, strand 50/50,
per-base substitution 0.5 %, N 0.1 %, ids ``@SRR000001.<i> <i>/1``, qualities
iid from {2,14,22,27,33,37,40}+33.  Deterministic for a given
(n_reads, read_len, genome_len, seed) and numpy version-independent
(uses only PCG64 integer/uniform draws).
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b
_QUALS = np.array([2, 14, 22, 27, 33, 37, 40], dtype=np.uint8) + 33


def synth_reads(n_reads: int, read_len: int, genome_len: int, seed: int,
                sub_rate: float = 0.005, n_rate: float = 0.001) -> np.ndarray:
    """Return an (n_reads, read_len) uint8 array of ASCII bases (ACGTN)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    genome = _ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    chunk = 1 << 18
    ar = np.arange(read_len, dtype=np.int64)
    for s in range(0, n_reads, chunk):
        e = min(n_reads, s + chunk)
        m = e - s
        pos = rng.integers(0, genome_len - read_len + 1, size=m, dtype=np.int64)
        strand = rng.integers(0, 2, size=m, dtype=np.uint8)
        r = genome[pos[:, None] + ar[None, :]]
        rc = _COMP[r[:, ::-1]]
        r = np.where(strand[:, None] == 1, rc, r)
        u = rng.random(size=(m, read_len))
        sub = u < sub_rate
        alt = _ACGT[rng.integers(0, 4, size=(m, read_len), dtype=np.uint8)]
        r = np.where(sub, alt, r)
        isn = (u >= sub_rate) & (u < sub_rate + n_rate)
        r = np.where(isn, np.uint8(ord("N")), r)
        out[s:e] = r
    return out


def synth_two_haplotypes(n_reads: int, read_len: int, genome_len: int, seed: int, snp_every: int = 150,
                         sub_rate: float = 0.005, n_rate: float = 0.001) -> np.ndarray:
    """Very deep coverage of a small genome with two haplotypes (a substitution every `snp_every` bases): k-mer counters
    run into saturation, both alleles of a site reach the counter maximum (counts_level_t::mixed), the probabilistic
    counters work above their thresholds.  Same read model as synth_reads."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    b = a.copy()
    sites = np.arange(snp_every // 2, genome_len, snp_every)
    b[sites] = (b[sites] + 1 + rng.integers(0, 3, size=len(sites), dtype=np.uint8)) & 3
    hap = np.stack([_ACGT[a], _ACGT[b]])
    pos = rng.integers(0, genome_len - read_len + 1, size=n_reads, dtype=np.int64)
    which = rng.integers(0, 2, size=n_reads, dtype=np.int64)
    strand = rng.integers(0, 2, size=n_reads, dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    r = hap[which[:, None], pos[:, None] + ar[None, :]]
    r = np.where(strand[:, None] == 1, _COMP[r[:, ::-1]], r)
    u = rng.random(size=(n_reads, read_len))
    alt = _ACGT[rng.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)]
    r = np.where(u < sub_rate, alt, r)
    r = np.where((u >= sub_rate) & (u < sub_rate + n_rate), np.uint8(ord("N")), r)
    return np.ascontiguousarray(r)


def synth_quals(n_reads: int, read_len: int, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    return _QUALS[rng.integers(0, len(_QUALS), size=(n_reads, read_len), dtype=np.uint8)]


def read_id(i: int, mate: int = 1) -> bytes:
    return b"@SRR000001.%d %d/%d" % (i + 1, i + 1, mate)


def synth_ids_varied(n: int, seed: int, mate: int = 0) -> list:
    """Illumina-style ids that exercise the id coder's branches: numeric fields with small / 2- / 3- / 4- / 8-byte
    positive and negative deltas, literal fields that change with and without a length change, 11+-digit digit
    runs (literal by rule), occasional format changes (token structure differs -> plain coding), several
    instrument names.  mate = 0: no mate suffix; 1 / 2: ' <mate>:N:0:<index>' (not 'typical' PE ids) -- every 7th
    pair instead ends in '/1' or '/2' (typical)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x1D5))
    instr = [b"@HWI-ST1234", b"@M00123", b"@NB501234", b"@A00987"]
    cells = [b"C0ABCACXX", b"H7LKJBGXY", b"000000000-AB12C", b"HW2YFDSXX"]
    out = []
    x = y = 1000
    big = 5_000_000_000
    for i in range(n):
        r = rng.integers(0, 1000)
        if r < 30:
            x = int(rng.integers(1, 30000)); y = int(rng.integers(1, 200000))
        elif r < 400:
            y += int(rng.integers(-1, 2))
        elif r < 800:
            y += int(rng.integers(-150, 300))
        else:
            y = int(rng.integers(1, 20_000_000))
        if r % 97 == 0:
            big = int(rng.integers(1, 9_999_999_999))
        elif r % 13 == 0:
            big += int(rng.integers(-70000, 70000))
        big = max(1, big)
        ins = instr[(i // 700) % 4 if r > 5 else int(rng.integers(0, 4))]
        cell = cells[(i // 1900) % 4]
        lane = 1 + (i // 500) % 8
        if r in (7, 8, 9):      # a different format altogether
            body = b"@read_%d length=%d" % (i, 100 + r)
        elif r in (10, 11):     # 11+ digits: literal by rule; separators only
            body = b"%s:%012d::%d" % (ins, big * 7, y)
        else:
            body = b"%s:%d:%s:%d:%d:%d:%d %d" % (ins, 100 + (i // 3000), cell, lane, 1101 + (i // 250) % 16, x, y, big)
        if mate:
            if i % 7 == 3:
                body += b"/%d" % mate
            else:
                body += b" %d:N:0:%s" % (mate, b"ATCACG" if (i // 1000) % 2 == 0 else b"TTAGGCA")
        out.append(body)
    return out


def write_fastq(path: str, reads: np.ndarray, quals: np.ndarray | None = None,
                seed: int = 0, mate: int = 1) -> None:
    n, L = reads.shape
    if quals is None:
        quals = synth_quals(n, L, seed)
    with open(path, "wb") as f:
        buf = []
        for i in range(n):
            buf.append(read_id(i, mate))
            buf.append(b"\n")
            buf.append(reads[i].tobytes())
            buf.append(b"\n+\n")
            buf.append(quals[i].tobytes())
            buf.append(b"\n")
            if len(buf) >= 1 << 16:
                f.write(b"".join(buf))
                buf = []
        f.write(b"".join(buf))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("out")
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--genome", type=int, default=200000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    write_fastq(a.out, synth_reads(a.reads, a.len, a.genome, a.seed), seed=a.seed)


def synth_ragged(n_reads: int, genome_len: int, seed: int, min_len: int = 30, max_len: int = 160,
                 n_rate: float = 0.02, dup_rate: float = 0.05):
    """Variable-length reads with N runs and exact duplicates (edge cases of the DNA path).
    Returns (ids, seqs, quals) as lists of bytes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    genome = _ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
    seqs = []
    for i in range(n_reads):
        if seqs and rng.random() < dup_rate:
            seqs.append(seqs[int(rng.integers(0, len(seqs)))])
            continue
        L = int(rng.integers(min_len, max_len + 1))
        pos = int(rng.integers(0, genome_len - L + 1))
        r = genome[pos:pos + L].copy()
        if rng.random() < 0.5:
            r = _COMP[r[::-1]]
        u = rng.random(L)
        r[u < 0.01] = _ACGT[rng.integers(0, 4, size=int((u < 0.01).sum()), dtype=np.uint8)]
        if rng.random() < 0.3:                      # an N run somewhere (possibly in the prefix)
            a = int(rng.integers(0, L))
            r[a:a + int(rng.integers(1, 6))] = ord("N")
        r[(u > 1.0 - n_rate)] = ord("N")
        seqs.append(r.tobytes())
    ids = [b"@rag.%d" % (i + 1) for i in range(n_reads)]
    quals = [bytes([33 + int(x) for x in rng.integers(2, 41, size=len(s))]) for s in seqs]
    return ids, seqs, quals


def synth_pairs(n_pairs: int, read_len: int, genome_len: int, seed: int, frag_min: int = 300, frag_max: int = 600,
                sub_rate: float = 0.005, n_rate: float = 0.001, chunk: int = 1 << 16):
    """Paired-end reads: fragment of random length, mate 1 from its start (forward), mate 2 the reverse
    complement of its end; fragment strand 50/50.  Returns two (n_pairs, read_len) uint8 arrays.
    The draws are, in this order: genome, fragment lengths, positions, strands, then per mate the (n, L) uniforms
    followed by the (n, L) substitution letters.  They are made in row chunks (memory: 5 M pairs need a few GB instead
    of tens) WITHOUT changing the stream: the uniforms of a mate come from a copy of the generator, the generator
    itself is advanced past them (one 64-bit step per double) and then supplies the letters; `chunk` is a multiple of
    8 rows so that every chunk of uint8 letters consumes whole 64-bit outputs."""
    assert chunk % 8 == 0
    rng = np.random.Generator(np.random.PCG64(seed))
    genome = _ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
    frag = rng.integers(frag_min, frag_max + 1, size=n_pairs, dtype=np.int64)
    pos = (rng.random(n_pairs) * (genome_len - frag)).astype(np.int64)
    strand = rng.integers(0, 2, size=n_pairs, dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    out = []
    for mate in (0, 1):
        res = np.empty((n_pairs, read_len), dtype=np.uint8)
        bg_u = np.random.PCG64()
        bg_u.state = rng.bit_generator.state          # the uniforms: n * L doubles from here
        rng_u = np.random.Generator(bg_u)
        st0 = rng.bit_generator.state
        rng.bit_generator.advance(n_pairs * read_len)  # ... and the letters follow them
        st1 = rng.bit_generator.state                  # (advance() drops a cached 32-bit half, which the doubles would have left alone)
        st1["has_uint32"], st1["uinteger"] = st0["has_uint32"], st0["uinteger"]
        rng.bit_generator.state = st1
        for s in range(0, n_pairs, chunk):
            e = min(n_pairs, s + chunk)
            left = genome[pos[s:e, None] + ar[None, :]]
            right = _COMP[genome[(pos[s:e] + frag[s:e] - read_len)[:, None] + ar[None, :]][:, ::-1]]
            fwd = (strand[s:e, None] == 0) if mate == 0 else (strand[s:e, None] != 0)
            r = np.where(fwd, left, right)
            u = rng_u.random(size=(e - s, read_len))
            alt = _ACGT[rng.integers(0, 4, size=(e - s, read_len), dtype=np.uint8)]
            r = np.where(u < sub_rate, alt, r)
            res[s:e] = np.where((u >= sub_rate) & (u < sub_rate + n_rate), np.uint8(ord("N")), r)
        out.append(res)
    return out[0], out[1]


def synth_mixed_lengths(n_reads: int = 1500, genome_len: int = 40000, seed: int = 7):
    """Very short (20-31 bp, shorter than the b-mer), very long (4200-6000 bp, beyond the LDS staging
    buffer) and ordinary reads with N runs.  Returns (ids, seqs, quals) lists of bytes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = _ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
    ids, seqs, quals = [], [], []
    for i in range(n_reads):
        r = rng.random()
        L = int(rng.integers(20, 32)) if r < 0.4 else int(rng.integers(4200, 6000)) if r < 0.45 else int(rng.integers(60, 200))
        pos = int(rng.integers(0, genome_len - L))
        s = g[pos:pos + L].copy()
        if rng.random() < 0.3:
            s[int(rng.integers(0, L)):][:int(rng.integers(1, 12))] = ord("N")
        ids.append(b"@x%d" % i)
        seqs.append(s.tobytes())
        quals.append(b"I" * L)
    return ids, seqs, quals
