"""Row N4 on the GPU: the read-id stream from fqsx_idg_* (one wavefront per worker; csrc/fqsx_idk.h) against the host coder
fqsx_id_* -- which the full-file fixtures pin byte for byte to the reference's id streams (tests/test_fqs_file.py,
tests/test_gpu_fullsize.py; with the GPU coder in the loop there, the files stay identical).  Emulation build here, the HIP
build under -m gpu."""
import numpy as np
import pytest

from conftest import EMU_LIB
from fqsqueezer_amd import hostpipe as hp
from fqsqueezer_amd.codec import IdCodec, FqsxError


def _ids(kind, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    if kind == "srr":        # the bench generator's ids: one numeric field counting up, twice
        out = [b"@SRR000001.%d %d/1" % (i + 1, i + 1) for i in range(n)]
    elif kind == "illumina":  # instrument:run:flowcell:lane:tile:x:y with slowly changing fields, jumps and sign changes
        tile, x, y = 1101, 1000, 2000
        for i in range(n):
            if rng.random() < 0.02:
                tile += int(rng.integers(1, 4)); x = int(rng.integers(1000, 3000))
            x += int(rng.integers(-300, 70000)) if rng.random() < 0.3 else 1
            y = int(rng.integers(0, 250000))
            inst = b"M0%d" % (1 + (i // 700) % 3)
            out.append(b"@%s:17:000000000-A1B2C:%d:%d:%d:%d 1:N:0:%s" % (inst, 1 + (i // 1500) % 2, tile, max(x, 0), y, b"ACGT" if i % 50 else b"ACGTTGCA"))
    else:                     # changing token shapes: long digit runs (not numeric), empty tokens, one-off literals, big / negative deltas
        for i in range(n):
            r = rng.random()
            if r < 0.1:
                out.append(b"@read_%d//%012d.x" % (i, int(rng.integers(0, 10 ** 11))))
            elif r < 0.2:
                out.append(b"@q%d-%d-%d" % (int(rng.integers(0, 10)), int(rng.integers(0, 2 ** 40)), i))
            elif r < 0.3:
                out.append(b"@q%d-%d-%d" % (int(rng.integers(0, 10)), int(rng.integers(0, 70000)), n - i))
            else:
                out.append(b"@lib%s.%d.%d" % (b"A" if i % 7 else b"Bc", i * 3, int(rng.integers(0, 300))))
    return out


def _arrays(ids):
    lines = [x + b"\n" for x in ids]
    off = np.zeros(len(lines) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in lines])
    return np.frombuffer(b"".join(lines), dtype=np.uint8), off


def _compare(lib, device, T, id_mode, kind, paired, n=3000, blocks=4):
    header = hp.make_header(T, "pe_sorted" if paired else "se_sorted", 1, id_mode=id_mode)
    host, gpu = IdCodec(header, lib_path=lib), IdCodec(header, lib_path=lib, device=device)
    ids = _ids(kind, n, 5)
    if paired:   # mates: mostly the typical .../1 .../2 pair, sometimes not
        both = []
        for i, x in enumerate(ids):
            both.append(x + b"/1")
            both.append((x if i % 11 else x[:-1] + b"Z") + b"/2")
        ids = both
    per = len(ids) // blocks // 2 * 2
    for b in range(blocks):
        a, off = _arrays(ids[b * per:(b + 1) * per])
        want, got = host.encode_block(a, off, paired), gpu.encode_block(a, off, paired)
        assert got == want, f"block {b}: the GPU id stream differs from the host coder's"
        assert sum(len(s) for s in got) >= 8 * T
    host.close(); gpu.close()


CASES = [(3, "lossless", "srr", False), (4, "lossless", "illumina", False), (2, "lossless", "odd", False), (3, "lossless", "illumina", True),
         (2, "lossless", "odd", True), (3, "instrument", "illumina", False), (2, "instrument", "illumina", True), (5, "instrument", "srr", False)]


@pytest.mark.parametrize("T,id_mode,kind,paired", CASES)
def test_emu_id_kernel_equals_host_coder(built, T, id_mode, kind, paired):
    _compare(EMU_LIB, 0, T, id_mode, kind, paired, n=1200, blocks=3)


@pytest.mark.gpu
@pytest.mark.parametrize("T,id_mode,kind,paired", CASES + [(64, "lossless", "illumina", False), (64, "instrument", "illumina", True)])
def test_gpu_id_kernel_equals_host_coder(T, id_mode, kind, paired):
    _compare(None, 0, T, id_mode, kind, paired, n=20000, blocks=5)


def test_emu_id_kernel_reports_what_it_cannot_stage(built):
    header = hp.make_header(2, "se_sorted", 1, id_mode="lossless")
    gpu = IdCodec(header, lib_path=EMU_LIB, device=0)
    a, off = _arrays([b"@" + b"x" * 1500] * 8)
    with pytest.raises(FqsxError):
        gpu.encode_block(a, off, False)
