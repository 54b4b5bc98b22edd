"""SURVEY 8 f4 (block sorter): the restatement in tests/oracle_lib.py (bwt_naive) against the reference's own libbsc
(oracle/_ref/backendref bwt = bsc_bwt_encode, libbsc/bwt/bwt.cpp:46-79), and the published rate rule."""
import numpy as np

import nanospring_amd as ns
from tests import oracle_lib


def cases():
    rng = np.random.RandomState(5)
    yield b"banana$banana$ban"
    yield bytes(rng.choice(list(b"ACGT"), 1000).tolist())
    yield b"ACGT" * 300
    yield bytes(3000)
    yield bytes(rng.randint(0, 256, 2500).astype(np.uint8).tolist())
    yield (b"ACGTTGCA" * 40 + b"\n") * 9
    yield bytes([255]) * 17 + bytes([0]) * 16


def test_naive_bwt_equals_the_reference_block_sorter():
    for t in cases():
        b, idx, aux = oracle_lib.bwt_naive(t)
        wb, widx, waux = oracle_lib.ref_bwt(t)
        assert b == wb and idx == widx and aux == waux, len(t)
        assert len(aux) == (len(t) - 1) // ns.bsc_aux_rate(len(t))


def test_rate_rule():
    assert [ns.bsc_aux_rate(n) for n in (16, 31, 32, 1000, 48 << 20)] == [2, 2, 4, 64, 1 << 22]
