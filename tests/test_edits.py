"""Row a15: Edit::optimizeEditScript / Edits::applyEdits (src/Edits.cpp:23-60, include/Edits.h:73-94).  Golden vectors come
from the reference's own object (oracle/_ref/nsref_edits, tests/golden/make_golden.py edits); the product's
optimize_edit_script (consensus.cpp) and the tests' apply_edits (tests/align_util.py) must reproduce them."""
import os

import numpy as np

from tests import host_lib

HERE = os.path.dirname(os.path.abspath(__file__))


def apply_edits(orig, edits):
    """Edits::applyEdits (include/Edits.h:73-94), all four edit types, exactly as written there (a DELETE's character is not
    looked at: emission scripts carry '-')."""
    out, p = [], 0
    for t, b, n in edits:
        if t == 0:
            out.append(orig[p:p + n]); p += n
        elif t == 1:
            out.append(chr(b))
        elif t == 2:
            p += 1
        else:
            out.append(chr(b)); p += 1
    return "".join(out), p


def test_optimize_edit_script_equals_reference_object():
    z = np.load(os.path.join(HERE, "golden", "edit_cases.npz"))
    io, oo, ao, go = z["in_off"], z["out_off"], z["app_off"], z["orig_off"]
    n_sub = 0
    for c in range(len(io) - 1):
        t, b, m = z["in_types"][io[c]:io[c + 1]], z["in_bases"][io[c]:io[c + 1]], z["in_nums"][io[c]:io[c + 1]]
        dis, ot, ob, om = host_lib.optimize_edits(t, b, m)
        assert dis == int(z["dis"][c]), c
        wt, wb, wm = z["out_types"][oo[c]:oo[c + 1]], z["out_bases"][oo[c]:oo[c + 1]], z["out_nums"][oo[c]:oo[c + 1]]
        assert np.array_equal(ot, wt), c
        assert np.array_equal(om[ot == 0], wm[wt == 0]), c                    # run lengths of SAME
        assert np.array_equal(ob[(ot == 1) | (ot == 3)], wb[(wt == 1) | (wt == 3)]), c   # inserted / substituted bases
        n_sub += int((wt == 3).sum())
        # the tests' applyEdits restatement against the reference's result, on the raw and on the optimised script
        orig = bytes(z["orig"][go[c]:go[c + 1]]).decode()
        want = bytes(z["applied"][ao[c]:ao[c + 1]]).decode()
        for ty, ba, nu in ((t, b, m), (ot, ob, om)):
            got, used = apply_edits(orig, [(int(x), int(y), int(w)) for x, y, w in zip(ty, ba, nu)])
            assert got == want and used == len(orig), c
    assert n_sub > 100
