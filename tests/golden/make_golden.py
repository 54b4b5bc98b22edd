#!/usr/bin/env python3
"""Generates the committed golden vectors from the REFERENCE's own objects
(oracle/_ref, built by oracle/Makefile from /root/reference where it lies).

    python tests/golden/make_golden.py            # needs /root/reference (this container)

Outputs (data only: inputs + the reference's outputs):
    minhash_small.npz   reads, salts, queries -> sketches, DnaBitset bytes, per-table
                        id lists, getFilteredReads results     (P1, P2, P3 of SURVEY 8c)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import oracle_lib  # noqa: E402
from nanospring_amd.filter import mt19937_64_salts  # noqa: E402

COMP = {"A": "T", "T": "A", "C": "G", "G": "C"}


def revcomp(s):
    return "".join(COMP.get(c, c) for c in reversed(s))


def mutate(rng, s, p):
    out = []
    for c in s:
        u = rng.random_sample()
        if u < p:
            out.append("ACGT"[rng.randint(4)])
        elif u < 2 * p:
            out.append("ACGT"[rng.randint(4)])
            out.append(c)
        elif u < 3 * p:
            continue
        else:
            out.append(c)
    return "".join(out)


def minhash_reads(seed=1234, k=23):
    """~70 reads: overlapping noisy reads of a 40 kb genome with a planted tandem
    repeat, plus the edge cases of SURVEY A2 (len < k-1, == k-1, == k, 31, 32),
    N / lowercase bytes (folded by baseToInt), exact duplicates and a homopolymer."""
    rng = np.random.RandomState(seed)
    g = "".join("ACGT"[i] for i in rng.randint(0, 4, size=40000))
    unit = "".join("ACGT"[i] for i in rng.randint(0, 4, size=37))
    g = g[:15000] + unit * 60 + g[15000:]
    reads = []
    for _ in range(56):
        ln = int(max(300, rng.gamma(2.0, 1500.0)))
        st = rng.randint(0, len(g) - ln)
        s = g[st:st + ln]
        if rng.randint(2):
            s = revcomp(s)
        reads.append(mutate(rng, s, 0.01))
    reads += ["", "A", g[:k - 3], g[:k - 2], g[100:100 + k - 1], g[200:200 + k], g[300:300 + k + 1], g[400:431], g[500:532]]
    reads += [g[1000:1400].replace("A", "N", 3), g[2000:2300].lower(), reads[0], reads[0], "A" * 500, "AC" * 300]
    return g, reads


def minhash_case():
    k, n, thr = 23, 60, 6
    g, reads = minhash_reads(k=k)
    salts = mt19937_64_salts(n, 12345)
    queries = []
    for r in (0, 1, 2, 3, 10, 20, 30):
        queries += [reads[r], revcomp(reads[r])]
    queries += [g[15000:17000], g[14000:16000], revcomp(g[5000:9000]), g[0:32], g[40:40 + k - 1], g[40:40 + k - 2], "", "ACGT" * 10,
                reads[0][:500], "A" * 200]
    ref = oracle_lib.run_nsref(reads, queries, k, n, thr, salts)
    assert ref["unpack_ok"].all()
    rb, roff = oracle_lib.concat(reads)
    qb, qoff = oracle_lib.concat(queries)
    fl_off = np.zeros(len(queries) + 1, dtype=np.uint64)
    fl_off[1:] = np.cumsum([len(x) for x in ref["filter"]])
    fl_ids = np.concatenate(ref["filter"]) if fl_off[-1] else np.zeros(0, np.uint32)
    # tables: per (j, r) the id list of sketch[r][j]; store as CSR
    tl = [x for row in ref["tables"] for x in row]
    t_off = np.zeros(len(tl) + 1, dtype=np.uint64)
    t_off[1:] = np.cumsum([len(x) for x in tl])
    t_ids = np.concatenate(tl)
    np.savez_compressed(os.path.join(HERE, "minhash_small.npz"), k=k, n=n, thr=thr, salts=salts,
                        read_bases=rb[:int(roff[-1])], read_off=roff, query_bases=qb[:int(qoff[-1])], query_off=qoff,
                        sketches=ref["sketches"], packed=ref["packed"], qsketch=ref["qsketch"],
                        filter_off=fl_off, filter_ids=fl_ids, table_off=t_off, table_ids=t_ids.astype(np.uint32))
    print("minhash_small.npz:", len(reads), "reads,", len(queries), "queries,",
          int(fl_off[-1]), "filter ids, max list", max(len(x) for x in tl))




def ksw2_case():
    """Raw ksw_extd2_sse calls (P4k of SURVEY 8c): inputs -> ez fields + CIGAR from the reference kernel."""
    from tests.test_ksw2_oracle import cases
    cs = cases(77, 120)
    seqs, so, prm, ezs, cig, co = [], [0], [], [], [], [0]
    for q, t, w, zdrop, eb, flag in cs:
        ez, c = oracle_lib.ref_ksw(q, t, w, zdrop, eb, flag)
        seqs += [q, t]
        so += [so[-1] + len(q), so[-1] + len(q) + len(t)]
        prm.append([w, zdrop, eb, flag])
        ezs.append(list(ez))
        cig.append(c)
        co.append(co[-1] + len(c))
    np.savez_compressed(os.path.join(HERE, "ksw2_cases.npz"), n=len(cs), seqs=np.concatenate(seqs).astype(np.uint8),
                        seq_off=np.array(so, dtype=np.int64), params=np.array(prm, dtype=np.int32), ez=np.array(ezs, dtype=np.int64),
                        cigar=np.concatenate(cig).astype(np.uint32), cigar_off=np.array(co, dtype=np.int64))
    print("ksw2_cases.npz:", len(cs), "calls,", sum(e[1] for e in ezs), "z-dropped")


def align_case():
    """(reference string, query) pairs -> what the reference's minimap2 returns for reg[0] under NanoSpring's
    call sequence (src/ConsensusGraph.cpp:195-217): P4 of SURVEY 8c, minus alignRead's own Edit list (its
    translation unit needs Boost and cannot be built here; that conversion is pinned by the reference's
    CHECKS invariant applyEdits(ref, script) == read instead)."""
    from tests.align_cases import pairs, make_genome, mutate as mut2
    ps = [(r[:15000], q) for r, q in pairs(4242, 72, big=False)]
    rng = np.random.RandomState(99)
    gg = make_genome(rng, 140000)
    for _ in range(3):
        a = rng.randint(0, len(gg) - 9000)
        ps.append((gg, mut2(rng, gg[a:a + rng.randint(3000, 9000)], 0.03)))
    refs, ref_id = [], []
    for r, _ in ps:
        if r not in refs:
            refs.append(r)
        ref_id.append(refs.index(r))
    rb, ro = oracle_lib.concat(refs)
    qb, qo = oracle_lib.concat([q for _, q in ps])
    fields = ["hits", "rs", "re", "qs", "qe", "blen", "mlen", "n_ambi", "dp_max", "n_cigar", "mid_occ"]
    vals, cig, co = [], [], [0]
    for r, q in ps:
        d = oracle_lib.ref_mm2_align(r, q)
        vals.append([d[f] for f in fields])
        cig.append(d["cigar"])
        co.append(co[-1] + len(d["cigar"]))
    np.savez_compressed(os.path.join(HERE, "align_pairs.npz"), ref_bases=rb[:int(ro[-1])], ref_off=ro, qry_bases=qb[:int(qo[-1])], qry_off=qo,
                        pair_ref=np.array(ref_id, dtype=np.uint32), fields=np.array(fields), values=np.array(vals, dtype=np.int64),
                        cigar=np.concatenate(cig).astype(np.uint32), cigar_off=np.array(co, dtype=np.int64))
    v = np.array(vals)
    print("align_pairs.npz:", len(ps), "pairs,", int((v[:, 0] > 0).sum()), "with hits,", int((v[:, 0] > 1).sum()), "multi-hit")

def sketch_cases(seed=77):
    """Sequences for mm_sketch: random DNA, reads with N / lowercase / other bytes, tandem repeats and homopolymers
    (hash ties inside a window), palindrome-rich stretches (the strand-unknown `continue`), lengths around k and w+k."""
    rng = np.random.RandomState(seed)
    out = []
    for it in range(48):
        ln = int(rng.choice([0, 1, 7, 19, 20, 21, 68, 69, 70, 71, 200, 1500, 6000, 15000]))
        s = "".join("ACGT"[i] for i in rng.randint(0, 4, size=ln))
        if it % 5 == 1 and ln > 30:
            p = rng.randint(0, ln - 20)
            s = s[:p] + "N" * int(rng.randint(1, 12)) + s[p:]
        if it % 7 == 2:
            s = s.lower()
        if it % 7 == 3 and ln > 10:
            s = s[:ln // 2] + "xU-u" + s[ln // 2:]
        if it % 6 == 4:
            unit = "".join("ACGT"[i] for i in rng.randint(0, 4, size=int(rng.randint(1, 40))))
            s = s[:ln // 3] + unit * int(rng.randint(5, 120)) + s[ln // 3:]
        if it % 9 == 0:
            s = "AT" * int(rng.randint(5, 200)) + s
        if it % 11 == 5:
            s = s + "A" * int(rng.randint(30, 400)) + s[:50]
        out.append(s)
    return out


SKETCH_WK = [(50, 20), (10, 15), (5, 11), (1, 8), (255, 28), (19, 28), (64, 20), (65, 17)]


def sketch_case():
    """sequences -> the reference's mm_sketch output (minimap2/sketch.c via oracle/_ref/libmm2ref.so), for the
    (w, k) pairs of SKETCH_WK: the golden vectors of the batched GPU sketch."""
    seqs = sketch_cases()
    sb, so = oracle_lib.concat(seqs)
    xs, offs = [], [0]
    for w, k in SKETCH_WK:
        for s in seqs:
            # the reference asserts len > 0 (sketch.c:84); its callers skip empty sequences
            xy = oracle_lib.ref_mm_sketch(s, w, k) if s else np.zeros((0, 2), dtype=np.uint64)
            xs.append(xy.reshape(-1))
            offs.append(offs[-1] + len(xy))
    np.savez_compressed(os.path.join(HERE, "mm_sketch_cases.npz"), bases=sb[:int(so[-1])], off=so, wk=np.array(SKETCH_WK, dtype=np.int32),
                        xy=np.concatenate(xs).astype(np.uint64), xy_off=np.array(offs, dtype=np.int64))
    print("mm_sketch_cases.npz:", len(seqs), "sequences x", len(SKETCH_WK), "(w,k) pairs,", offs[-1], "minimizers")


def edit_scripts(seed=5, n_cases=160):
    """Raw edit scripts as read2EditScript produces them (SAME runs, INSERT with the base, DELETE with '-'): mixes of isolated
    edits, adjacent insert/delete runs of unequal length (-> substitutions + remainder), leading / trailing edits, empty."""
    rng = np.random.RandomState(seed)
    out = []
    for c in range(n_cases):
        ops, orig_len = [], 0
        n_blocks = int(rng.randint(0, 12)) if c else 0
        for _ in range(n_blocks):
            if rng.rand() < 0.8:
                num = int(rng.randint(1, 40))
                ops.append((0, 0, num)); orig_len += num
            ni, nd = int(rng.choice([0, 0, 1, 1, 2, 3, 7])), int(rng.choice([0, 0, 1, 1, 2, 3, 7]))
            block = [(1, ord("ACGT"[rng.randint(4)]), 0)] * 0
            block = [(1, ord("ACGT"[rng.randint(4)]), 0) for _ in range(ni)] + [(2, ord("-"), 0) for _ in range(nd)]
            rng.shuffle(block)
            ops += [tuple(int(v) for v in b) for b in block]
            orig_len += nd
        orig = "".join("ACGT"[i] for i in rng.randint(0, 4, size=orig_len))
        out.append((ops, orig))
    return out


def edits_case():
    """raw scripts -> what the reference's own Edit::optimizeEditScript and Edits::applyEdits (src/Edits.cpp, include/Edits.h,
    through oracle/_ref/nsref_edits) make of them: golden vectors of row a15."""
    import struct
    import subprocess
    import tempfile
    cases = edit_scripts()
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(struct.pack("<I", len(cases)))
            for ops, orig in cases:
                f.write(struct.pack("<II", len(ops), len(orig)) + orig.encode())
                for t, b, num in ops:
                    f.write(struct.pack("<BBI", t, b, num))
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "nsref_edits"), fi, fo], check=True)
        raw = open(fo, "rb").read()
    pos = 0
    rec = {"in_types": [], "in_bases": [], "in_nums": [], "in_off": [0], "orig": [], "orig_off": [0], "dis": [], "out_types": [], "out_bases": [], "out_nums": [],
           "out_off": [0], "applied_raw": [], "applied_opt": [], "app_off": [0]}
    for ops, orig in cases:
        rec["in_types"] += [o[0] for o in ops]; rec["in_bases"] += [o[1] for o in ops]; rec["in_nums"] += [o[2] for o in ops]
        rec["in_off"].append(len(rec["in_types"]))
        rec["orig"].append(orig); rec["orig_off"].append(rec["orig_off"][-1] + len(orig))
        dis, n_new = struct.unpack_from("<QI", raw, pos); pos += 12
        rec["dis"].append(dis)
        for _ in range(n_new):
            t, b, num = struct.unpack_from("<BBI", raw, pos); pos += 6
            rec["out_types"].append(t); rec["out_bases"].append(b); rec["out_nums"].append(num)
        rec["out_off"].append(len(rec["out_types"]))
        for key in ("applied_raw", "applied_opt"):
            (ln,) = struct.unpack_from("<I", raw, pos); pos += 4
            rec[key].append(raw[pos:pos + ln].decode()); pos += ln
        assert rec["applied_raw"][-1] == rec["applied_opt"][-1]
        rec["app_off"].append(rec["app_off"][-1] + len(rec["applied_raw"][-1]))
    assert pos == len(raw)
    np.savez_compressed(os.path.join(HERE, "edit_cases.npz"), in_types=np.array(rec["in_types"], np.uint8), in_bases=np.array(rec["in_bases"], np.uint8),
                        in_nums=np.array(rec["in_nums"], np.uint32), in_off=np.array(rec["in_off"], np.int64),
                        orig=np.frombuffer("".join(rec["orig"]).encode(), np.uint8), orig_off=np.array(rec["orig_off"], np.int64),
                        dis=np.array(rec["dis"], np.uint64), out_types=np.array(rec["out_types"], np.uint8), out_bases=np.array(rec["out_bases"], np.uint8),
                        out_nums=np.array(rec["out_nums"], np.uint32), out_off=np.array(rec["out_off"], np.int64),
                        applied=np.frombuffer("".join(rec["applied_opt"]).encode(), np.uint8), app_off=np.array(rec["app_off"], np.int64))
    print("edit_cases.npz:", len(cases), "scripts,", len(rec["in_types"]), "raw ops ->", len(rec["out_types"]), "optimised ops")


if __name__ == "__main__":
    if not oracle_lib.have_nsref():
        sys.exit("oracle/_ref/nsref missing: run `make -C oracle` where /root/reference exists")
    which = sys.argv[1:] or ["minhash", "ksw2", "align", "sketch", "edits"]
    if "minhash" in which:
        minhash_case()
    if "ksw2" in which:
        ksw2_case()
    if "align" in which:
        align_case()
    if "sketch" in which:
        sketch_case()
    if "edits" in which:
        edits_case()
