import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def split(bases, off):
    b = bytes(bases)
    return [b[int(off[i]):int(off[i + 1])].decode() for i in range(len(off) - 1)]


def load_minhash():
    z = np.load(os.path.join(GOLD, "minhash_small.npz"))
    d = {k: z[k] for k in z.files}
    d["reads"] = split(d["read_bases"], d["read_off"])
    d["queries"] = split(d["query_bases"], d["query_off"])
    d["k"], d["n"], d["thr"] = int(d["k"]), int(d["n"]), int(d["thr"])
    return d
