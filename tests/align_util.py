import os

import numpy as np

from tests.golden_util import split, GOLD

FIELDS = ["hits", "rs", "re", "qs", "qe", "blen", "mlen", "n_ambi", "dp_max", "n_cigar"]


def load_align_golden():
    z = np.load(os.path.join(GOLD, "align_pairs.npz"))
    d = {k: z[k] for k in z.files}
    d["refs"] = split(d["ref_bases"], d["ref_off"])
    d["qrys"] = split(d["qry_bases"], d["qry_off"])
    d["fields"] = [str(f) for f in d["fields"]]
    return d


def apply_edits(ref_window, edits):
    """Edits::applyEdits (include/Edits.h:73-94) for scripts of SAME / INSERT / DELETE."""
    out, p = [], 0
    for t, b, n in edits:
        if t == 0:
            out.append(ref_window[p:p + n]); p += n
        elif t == 1:
            out.append(chr(b))
        elif t == 2:
            assert ref_window[p] == chr(b)
            p += 1
    return "".join(out), p


def check_alignread_invariant(ref, qry, d, edits):
    """The reference's own CHECKS block (src/Consensus.cpp:280-317): the edit script applied to the
    consensus window [max(beginOffset,0), len + min(endOffset,0)) reproduces the aligned part of the read;
    plus the offset rules of src/ConsensusGraph.cpp:284-357."""
    assert d["ok"]
    bo, eo = d["begin_offset"], d["end_offset"]
    orig = ref[(bo if bo > 0 else 0):len(ref) + (0 if eo > 0 else eo)]
    target = qry[(0 if bo > 0 else -bo):len(qry) - (eo if eo > 0 else 0)]
    got, used = apply_edits(orig, edits)
    assert used == len(orig)
    assert got == target
    assert d["rel_pos"] == d["rs"] - d["qs"]
    assert bo == (d["rs"] if d["rs"] > 0 else -d["qs"])
    assert eo == (d["re"] - len(ref) if d["re"] < len(ref) else len(qry) - d["qe"])
