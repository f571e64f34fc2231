"""serde forms of BaseGraph: serde_json and bincode restatements, checked against bytes derived by
hand from the derive field order (src/graph.rs:43-50, src/dna_string.rs:72-76,762-767, src/lib.rs:577-580) and the formats'
published rules, plus round trips.  Host-side: runs without a GPU."""
import json
import struct

import numpy as np
import pytest

from pkg import dbg


def small_graph(stranded=False):
    seqs = [[0, 1, 2, 3, 0, 1], [3, 3, 2], [1] * 40]
    return dbg.BaseGraph(3, dbg.PackedDnaStringSet.from_seqs(seqs), np.array([0x80, 0x21, 0xFF], np.uint8), np.array([7, 65535, 300], np.uint32),
                         stranded)


def test_bincode_bytes_by_hand():
    g = small_graph(True)
    b = dbg.graph_serialize(g, dbg.SERDE_BINCODE, data_width=2)
    words = [int(w) for w in g.sequences.words]
    assert len(words) == 2                                            # 49 bases
    want = struct.pack("<Q2Q", 2, *words) + struct.pack("<Q", 49)                       # DnaString { storage: Vec<u64>, len: usize }
    want += struct.pack("<Q3Q", 3, 0, 6, 9) + struct.pack("<Q3I", 3, 6, 3, 40)          # start: Vec<usize>, length: Vec<u32>
    want += struct.pack("<Q3B", 3, 0x80, 0x21, 0xFF)                                    # exts: Vec<Exts{val: u8}>
    want += struct.pack("<Q3H", 3, 7, 65535, 300)                                       # data: Vec<u16>
    want += b"\x01"                                                                     # stranded: bool; PhantomData: nothing
    assert b == want
    assert words[0] >> 52 == 0b000110110001                           # ACGTAC...: base 0 in the top bit pair (dna_string.rs:383-399)


def test_json_text_by_hand():
    g = small_graph(False)
    t = dbg.graph_serialize(g, dbg.SERDE_JSON).decode()
    w = [int(x) for x in g.sequences.words]
    assert t == ('{"sequences":{"sequence":{"storage":[%d,%d],"len":49},"start":[0,6,9],"length":[6,3,40]},'
                 '"exts":[{"val":128},{"val":33},{"val":255}],"data":[7,65535,300],"stranded":false,"phantom":null}' % (w[0], w[1]))
    assert json.loads(t)["sequences"]["sequence"]["len"] == 49


@pytest.mark.parametrize("fmt,dw", [(0, 4), (1, 4), (1, 2), (1, 1), (1, 0), (0, 0)])
def test_round_trip(fmt, dw):
    rng = np.random.default_rng(fmt * 10 + dw)
    seqs = [rng.integers(0, 4, size=int(rng.integers(5, 200))) for _ in range(50)]
    data = rng.integers(0, 1 << (8 * dw), size=50).astype(np.uint32) if dw else np.zeros(50, np.uint32)
    g = dbg.BaseGraph(5, dbg.PackedDnaStringSet.from_seqs(seqs), rng.integers(0, 256, size=50).astype(np.uint8), data, bool(fmt))
    b = dbg.graph_serialize(g, fmt, data_width=dw)
    h = dbg.graph_deserialize(b, 5, fmt, data_width=dw)
    a, c = g.arrays(), h.arrays()
    for name in ("words", "start", "length", "exts", "data"):
        assert np.array_equal(a[name], c[name]), name
    assert a["n_bases"] == c["n_bases"] and g.stranded == h.stranded
    assert dbg.graph_serialize(h, fmt, data_width=dw) == b
    empty = dbg.BaseGraph(5, dbg.PackedDnaStringSet(), np.zeros(0, np.uint8), np.zeros(0, np.uint32), False)
    assert len(dbg.graph_deserialize(dbg.graph_serialize(empty, fmt, data_width=dw), 5, fmt, data_width=dw)) == 0


def test_malformed_input_is_an_error():
    g = small_graph()
    b = dbg.graph_serialize(g, dbg.SERDE_BINCODE, data_width=4)
    for bad in (b[:-3], b + b"\x00", b"\xff" * 8 + b[8:]):
        with pytest.raises(dbg.DbgError):
            dbg.graph_deserialize(bad, 3, dbg.SERDE_BINCODE, data_width=4)
    t = dbg.graph_serialize(g, dbg.SERDE_JSON)
    for bad in (t[:-1], t.replace(b'"exts"', b'"ext"'), t.replace(b"null}", b"nul}")):
        with pytest.raises(dbg.DbgError):
            dbg.graph_deserialize(bad, 3, dbg.SERDE_JSON)
    with pytest.raises(dbg.DbgError):                                  # 65535 does not fit a u8 data column
        dbg.graph_serialize(g, dbg.SERDE_BINCODE, data_width=1)


def test_hostile_input_is_rejected_not_trusted():
    """Cases a reviewer fed to the reader: u64 wrap in the node bounds check, integers beyond the field's
    type in JSON (serde_json rejects them), and a Vec<()> length with no bytes behind it."""
    # bincode: start = 2^64 - 2, length = 3 wraps to 1 <= n_bases
    blob = struct.pack("<Q1Q", 1, 0) + struct.pack("<Q", 8)
    blob += struct.pack("<Q1Q", 1, (1 << 64) - 2) + struct.pack("<Q1I", 1, 3)
    blob += struct.pack("<Q1B", 1, 0) + struct.pack("<Q1I", 1, 0) + b"\x00"
    with pytest.raises(dbg.DbgError):
        dbg.graph_deserialize(blob, 3, dbg.SERDE_BINCODE, data_width=4)
    g = small_graph()
    t = dbg.graph_serialize(g, dbg.SERDE_JSON)
    for bad in (t.replace(b'"length":[6,', b'"length":[4294967302,'),          # > u32
                t.replace(b'{"val":128}', b'{"val":256}'),                     # > u8
                t.replace(b'"data":[7,', b'"data":[4294967303,'),              # > u32
                t.replace(b'"len":49', b'"len":99999999999999999999999'),      # > u64
                t.replace(b'"len":49', b'"len":49.0')):
        assert bad != t
        with pytest.raises(dbg.DbgError):
            dbg.graph_deserialize(bad, 3, dbg.SERDE_JSON)
    with pytest.raises(dbg.DbgError):                                          # u16 data column: 70000 does not fit
        dbg.graph_deserialize(t.replace(b'"data":[7,', b'"data":[70000,'), 3, dbg.SERDE_JSON, data_width=2)
    # bincode Vec<()>: a length of 2^39 units with nothing behind it must not be allocated
    unit = dbg.graph_serialize(g, dbg.SERDE_BINCODE, data_width=0)
    assert unit[-9:-1] == struct.pack("<Q", 3)
    with pytest.raises(dbg.DbgError):
        dbg.graph_deserialize(unit[:-9] + struct.pack("<Q", 1 << 39) + unit[-1:], 3, dbg.SERDE_BINCODE, data_width=0)
