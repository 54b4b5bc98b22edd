"""FASTQ ingest.  The oracle restates the reference's std::getline loop (ReadData::loadFromFastqFile,
src/ReadData.cpp:86-221; that translation unit needs Boost and cannot be built here: parity unpinned against reference
bytes, pinned against hand-derived expectations of the getline rules below); the GPU parser must reproduce it byte for
byte, including the reads' 2-bit folding."""
import numpy as np
import pytest

from tests import oracle_lib

FOLD = {0: "A", 1: "T", 2: "C", 3: "G"}


def fold(b):
    return "".join(FOLD[(c & 2) | ((c & 4) >> 2)] for c in b)


# (text, expected list of base lines) -- derived by hand from std::getline semantics
CASES = [
    (b"@r0\nACGT\n+\n!!!!\n", [b"ACGT"]),
    (b"@r0\nACGT\n+\n!!!!", [b"ACGT"]),                                   # no final newline
    (b"@r0\nACGT\n+\n!!!!\n@r1\nTTGCA\n+\n#####\n", [b"ACGT", b"TTGCA"]),
    (b"@r0\r\nACGT\r\n+\r\n!!!!\r\n", [b"ACGT\r"]),                       # CRLF: the '\r' is a base (folds to T)
    (b"@r0\nACGT\n+\n!!!!\n\n", [b"ACGT", b""]),                          # trailing blank line = a name line of a record without bases
    (b"@r0\nACGT\n+\n!!!!\n@r1", [b"ACGT", b""]),                         # truncated record: name only
    (b"@r0\nACGT\n+\n!!!!\n@r1\nGG", [b"ACGT", b"GG"]),                   # truncated record: unterminated base line
    (b"@r0\nACGT\n+\n!!!!\n@r1\nGG\n+", [b"ACGT", b"GG"]),
    (b"@r0\n\n+\n\n", [b""]),                                              # empty read
    (b"\n", [b""]),                                                        # a single empty name line
    (b"@r0\nacgtnNxX\n+\n........\n", [b"acgtnNxX"]),                      # lower case, N, other bytes fold through baseToInt
]


def test_oracle_follows_getline():
    orc = oracle_lib.Oracle()
    for text, want in CASES:
        st, ln = orc.fastq_index(text)
        got = [text[int(s):int(s) + int(l)] for s, l in zip(st, ln)]
        assert got == want, (text, got, want)
    st, ln = orc.fastq_index(b"")
    assert len(st) == 0


def synth_fastq(seed, n, crlf=False, final_newline=True):
    rng = np.random.RandomState(seed)
    parts = []
    for i in range(n):
        ln = int(rng.choice([0, 1, 3, 31, 32, 33, 150, 4000, 20000])) + int(rng.randint(0, 40))
        seq = "".join("ACGTNacgtn"[c] for c in rng.randint(0, 10, size=ln))
        parts += ["@read%d some description" % i, seq, "+", "I" * ln]
    nl = "\r\n" if crlf else "\n"
    t = nl.join(parts) + (nl if final_newline else "")
    return t.encode()


@pytest.mark.gpu
def test_gpu_ingest_equals_oracle():
    import nanospring_amd as ns
    orc = oracle_lib.Oracle()
    g = ns.NsGpu()
    texts = [t for t, _ in CASES] + [synth_fastq(1, 300), synth_fastq(2, 257, crlf=True), synth_fastq(3, 64, final_newline=False),
                                     synth_fastq(4, 1000)[:-7], synth_fastq(5, 3) + b"\n\n\n"]
    for text in texts:
        st, ln = orc.fastq_index(text)
        n = g.load_fastq(text)
        assert n == len(st), (text[:60], n, len(st))
        assert g.num_bases == int(ln.sum())
        for r in range(n):
            want = fold(text[int(st[r]):int(st[r]) + int(ln[r])])
            assert g.get_read(r) == want, (text[:60], r)
    # newlines on and around the 16-byte (lane) and 1 KiB (wave) boundaries of the parser
    for ln in (1008, 1019, 1020, 1021, 1022, 1023, 1024, 1025, 2043, 2044, 2045, 4091, 4092, 4093):
        name = b"@r"
        seq = (b"ACGT" * (ln // 4 + 1))[:ln - len(name) - 1]          # the base line ends exactly at byte ln - 1 ... + 1
        for pad in (0, 1, 2, 15, 16, 17):
            text = name + b"\n" + seq + b"N" * pad + b"\n+\n" + b"I" * (len(seq) + pad) + b"\n@s\nGATTACA\n+\nIIIIIII\n"
            st, ln2 = orc.fastq_index(text)
            assert g.load_fastq(text) == len(st) == 2
            for r in range(2):
                assert g.get_read(r) == fold(text[int(st[r]):int(st[r]) + int(ln2[r])]), (ln, pad, r)
    with pytest.raises(ns.NsGpuError):
        g.load_fastq(b"")
    # the loaded reads feed the path like any others: sketches equal those of the same reads loaded as strings
    text = synth_fastq(7, 120)
    st, ln = orc.fastq_index(text)
    salts = ns.mt19937_64_salts(60)
    g.load_fastq(text)
    a = g.sketch(salts)
    g.load_reads([text[int(s):int(s) + int(l)] for s, l in zip(st, ln)])
    b = g.sketch(salts)
    assert np.array_equal(a, b)
    g.close()


@pytest.mark.gpu
def test_gpu_chunked_ingest_equals_one_call():
    """nsgpu_load_fastq_begin / _chunk / _end: pieces cut anywhere (inside names, base lines, between the CR and the LF, right behind
    a record) must give the reads of one call over the whole text, for well-formed and for truncated files."""
    import nanospring_amd as ns
    g, h = ns.NsGpu(), ns.NsGpu()
    rng = np.random.RandomState(11)
    small = [t for t, _ in CASES] + [synth_fastq(5, 3) + b"\n\n\n", synth_fastq(6, 4, crlf=True), synth_fastq(8, 5, final_newline=False)]
    for text in small:
        if not text:
            continue
        n = h.load_fastq(text)
        want = [h.get_read(r) for r in range(n)]
        cuts = range(0, len(text) + 1) if len(text) < 400 else sorted(set(rng.randint(0, len(text) + 1, size=60).tolist()))
        for c in cuts:
            assert g.load_fastq_chunks([text[:c], text[c:]]) == n, (text[:40], c)
            assert [g.get_read(r) for r in range(n)] == want, (text[:40], c)
    for seed, nrec in ((21, 400), (22, 1500)):
        text = synth_fastq(seed, nrec, crlf=seed == 22)[:-3 if seed == 22 else None]
        n = h.load_fastq(text)
        for trial in range(3):
            k = int(rng.randint(2, 40))
            cs = [0] + sorted(rng.randint(0, len(text) + 1, size=k).tolist()) + [len(text)]
            pieces = [text[a:b] for a, b in zip(cs[:-1], cs[1:])]
            assert g.load_fastq_chunks(pieces) == n
            assert g.num_bases == h.num_bases
            for r in list(range(0, n, 17)) + [n - 1]:
                assert g.get_read(r) == h.get_read(r), (seed, trial, r)
        # and they feed the path: same sketches
        salts = ns.mt19937_64_salts(60)
        assert np.array_equal(g.sketch(salts), h.sketch(salts))
    with pytest.raises(ns.NsGpuError):
        g.load_fastq_chunks([b"", b""])
    g.close(); h.close()


@pytest.mark.gpu
def test_gpu_gzip_file_ingest_equals_text(tmp_path):
    """nsgpu_load_fastq_file: ReadData::loadFromFile with gzip_flag (src/ReadData.cpp:95-101) -- a .fastq.gz (BASELINE cfg1 is one), a
    file of several concatenated gzip members, a plain file and detection by content all give the reads of the decompressed text;
    pieces smaller than the file (NSGPU_FASTQ_PIECE_MB is read once per process, so the piece size is the default here and the
    multi-piece path is covered by the chunk test above); a truncated gzip stream and a missing file are errors."""
    import gzip
    import nanospring_amd as ns
    g, h = ns.NsGpu(), ns.NsGpu()
    text = synth_fastq(31, 800, crlf=False)
    n = h.load_fastq(text)
    want = [h.get_read(r) for r in range(0, n, 13)]
    gz = tmp_path / "reads.fastq.gz"
    gz.write_bytes(gzip.compress(text, 6))
    multi = tmp_path / "multi.fastq.gz"
    cut = len(text) // 3
    multi.write_bytes(gzip.compress(text[:cut]) + gzip.compress(text[cut:2 * cut + 5]) + gzip.compress(text[2 * cut + 5:]))
    plain = tmp_path / "reads.fastq"
    plain.write_bytes(text)
    for path, flag in ((gz, 1), (gz, -1), (multi, 1), (plain, 0), (plain, -1)):
        assert g.load_fastq_file(str(path), flag) == n, (path, flag)
        assert g.num_bases == h.num_bases
        assert [g.get_read(r) for r in range(0, n, 13)] == want, (path, flag)
    bad = tmp_path / "cut.fastq.gz"
    bad.write_bytes(gz.read_bytes()[:-200])
    with pytest.raises(ns.NsGpuError):
        g.load_fastq_file(str(bad), 1)
    with pytest.raises(ns.NsGpuError):
        g.load_fastq_file(str(tmp_path / "missing.fastq.gz"), 1)
    g.close(); h.close()
