"""Independent Python restatement of the reference's decoder for one thread's stream set
(Decompressor::decompress inner loop + generateRead, src/Decompressor.cpp:105-172, 252-314).
Test infrastructure: it is the executable spec our output streams must satisfy."""
import struct

COMP = bytes.maketrans(b"ATCG", b"TAGC")


def _varint(buf, p):
    v = shift = 0
    while True:
        b = buf[p]; p += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, p


def decode(streams):
    """streams: dict with genome, lone, id, pos, type, base, complement (bytes). Returns {read id: bytes}."""
    out = {}
    idb, pos, typ, base, comp = (streams[k] for k in ("id", "pos", "type", "base", "complement"))
    ip = pp = tp = bp = cp = 0
    genomes = streams["genome"].split(b"\n")[:-1] if streams["genome"] else []
    for g in genomes:
        rid = 0
        while True:
            c = comp[cp:cp + 1]; cp += 1
            if c == b"\n":
                break
            rid = (rid + struct.unpack_from("<I", idb, ip)[0]) & 0xFFFFFFFF; ip += 4
            cur, pp = _varint(pos, pp)
            n_start, pp = _varint(pos, pp)
            read = bytearray(base[bp:bp + n_start]); bp += n_start
            while True:
                same, pp = _varint(pos, pp)
                read += g[cur:cur + same]; cur += same
                t = typ[tp:tp + 1]; tp += 1
                if t == b"\n":
                    break
                if t == b"d":
                    cur += 1
                elif t == b"i":
                    read += base[bp:bp + 1]; bp += 1
                elif t == b"s":
                    cur += 1
                    read += base[bp:bp + 1]; bp += 1
                else:
                    raise ValueError("bad edit type %r" % t)
            n_end, pp = _varint(pos, pp)
            read += base[bp:bp + n_end]; bp += n_end
            r = bytes(read)
            if c == b"c":
                r = r[::-1].translate(COMP)
            assert rid not in out
            out[rid] = r
    rid = 0
    lones = streams["lone"].split(b"\n")[:-1] if streams["lone"] else []
    for l in lones:
        rid = (rid + struct.unpack_from("<I", idb, ip)[0]) & 0xFFFFFFFF; ip += 4
        assert rid not in out
        out[rid] = l
    assert ip == len(idb) and pp == len(pos) and tp == len(typ) and bp == len(base) and cp == len(comp)
    return out


def fold(b):
    return bytes(b"ATCG"[(c & 2) | ((c & 4) >> 2)] for c in b)
