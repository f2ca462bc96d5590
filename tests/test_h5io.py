"""emei_amd/h5io.py — the reference's dataset container (HDF5 via h5py: zoo/util.py:108-111 writes, emei/core.py:61-81 reads)
without h5py.  Three kinds of evidence:
  1. a file REAL h5py wrote (tests/golden/h5py_written.h5, made by oracle/gen_h5_golden.py) is read back value for value;
  2. a file written here has the HDF5 File Format Specification's fixed fields at the specified offsets;
  3. where the image has libhdf5 tools / h5py (this one does, under /opt/conda), they read a file written here bit for bit,
     and a file they write now is read here.
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from emei_amd import h5io
from oracle.gen_h5_golden import expected  # the arrays the fixture was written from (functions of their shapes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONDA_PY = "/opt/conda/bin/python3.9"
H5DUMP = "/opt/conda/bin/h5dump"


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype.newbyteorder("=") == b.dtype.newbyteorder("=") and np.array_equal(a, b)


def _six(n=64, seed=0):
    rng = np.random.default_rng(seed)
    return {"observations": rng.normal(size=(n, 4)).astype(np.float32), "next_observations": rng.normal(size=(n, 4)).astype(np.float32),
            "actions": rng.integers(0, 2, (n, 1)).astype(np.float32), "rewards": rng.normal(size=n).astype(np.float32),
            "dones": (rng.random(n) < 0.1).astype(np.float32), "timeouts": (rng.random(n) < 0.05).astype(np.float32)}


def test_reads_a_file_written_by_real_h5py():
    got = h5io.read_h5(os.path.join(ROOT, "tests", "golden", "h5py_written.h5"))
    want = expected()
    assert set(got) == set(want)  # nested group "infos/episode" included, 12 links in the root = two symbol nodes
    for k in want:
        assert _same(got[k], want[k]), k
    assert got["extra_scalar"].shape == () and got["extra_be"].dtype == np.float64


def test_round_trip_and_dtypes(tmp_path):
    d = _six()
    d.update(f64=np.linspace(0, 1, 12).reshape(3, 2, 2), i32=np.arange(-3, 4, dtype=np.int32), u8=np.arange(5, dtype=np.uint8),
             f16=np.arange(4, dtype=np.float16), scalar=np.float32(2.5), empty=np.zeros((0, 4), np.float32), flag=np.array([True, False]))
    p = tmp_path / "a.h5"
    size = h5io.write_h5(p, d)
    assert size == os.path.getsize(p)
    back = h5io.read_h5(p)
    assert set(back) == set(d)
    for k, v in d.items():
        want = np.asarray(v).astype(np.uint8) if np.asarray(v).dtype == np.bool_ else v
        assert _same(back[k], want), k
    for bad in ({"a/b": np.zeros(2)}, {"": np.zeros(2)}, {"c": np.array(["x"])}, {"c": np.zeros(2, dtype=">f4")}):
        with pytest.raises((ValueError, TypeError)):
            h5io.write_h5(tmp_path / "bad.h5", bad)
    with pytest.raises(ValueError):
        (tmp_path / "junk.h5").write_bytes(b"not hdf5" * 20)
        h5io.read_h5(tmp_path / "junk.h5")


def test_fixed_fields_of_the_format_specification(tmp_path):
    """HDF5 File Format Specification 3.0: II.A superblock version 0, III.A v1 B-tree node, III.C symbol node, III.D local
    heap, IV.A object header version 1 and the four dataset messages — every field whose value the format fixes."""
    d = _six(8)
    p = tmp_path / "s.h5"
    eof = h5io.write_h5(p, d)
    b = p.read_bytes()
    u8 = lambda o: struct.unpack_from("<Q", b, o)[0]
    # superblock
    assert b[:8] == b"\x89HDF\r\n\x1a\n"
    assert b[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])  # versions 0, sizes of offsets / lengths 8
    assert struct.unpack_from("<HHI", b, 16) == (4, 16, 0)  # group leaf K, internal K, consistency flags
    assert (u8(24), u8(32), u8(40), u8(48)) == (0, h5io.UNDEF, eof, h5io.UNDEF)  # base, free space, END OF FILE, driver block
    name_off, root_hdr, cache = u8(56), u8(64), struct.unpack_from("<I", b, 72)[0]
    btree, heap = u8(80), u8(88)
    assert (name_off, root_hdr, cache) == (0, 96, 1)
    # root object header: version 1, one message, reference count 1; Symbol Table message 0x0011 of 16 bytes
    assert struct.unpack_from("<BBHII", b, 96) == (1, 0, 1, 1, 24)
    assert struct.unpack_from("<HHB", b, 112) == (0x11, 16, 0) and (u8(120), u8(128)) == (btree, heap)
    # B-tree node
    assert b[btree:btree + 4] == b"TREE" and struct.unpack_from("<BBH", b, btree + 4) == (0, 0, 1)
    assert (u8(btree + 8), u8(btree + 16)) == (h5io.UNDEF, h5io.UNDEF) and u8(btree + 24) == 0
    snod, last_key = u8(btree + 32), u8(btree + 40)
    assert heap == btree + 24 + 33 * 8 + 32 * 8  # the node is allocated whole: 2K + 1 keys, 2K children
    # local heap
    assert b[heap:heap + 4] == b"HEAP" and b[heap + 4:heap + 8] == bytes(4)
    seg_size, free_head, seg = u8(heap + 8), u8(heap + 16), u8(heap + 24)
    assert seg == heap + 32 and seg_size % 8 == 0 and free_head < seg_size
    assert u8(seg + free_head) == 1 and u8(seg + free_head + 8) == seg_size - free_head  # one free block: next = H5HL_FREE_NULL
    name = lambda off: b[seg + off:b.index(b"\0", seg + off)].decode()
    assert name(0) == "" and name(last_key) == "timeouts"  # the largest name closes the only child
    # symbol node: sorted names, cache type 0
    assert b[snod:snod + 4] == b"SNOD" and struct.unpack_from("<BBH", b, snod + 4) == (1, 0, 6)
    names, headers = [], []
    for e in range(6):
        off, hdr, ctype = struct.unpack_from("<QQI", b, snod + 8 + 40 * e)
        names.append(name(off)), headers.append(hdr)
        assert ctype == 0 and off % 8 == 0
    assert names == sorted(d)
    # a dataset's object header
    h = headers[names.index("observations")]
    ver, _, nmsg, refc, hsize = struct.unpack_from("<BBHII", b, h)
    assert (ver, nmsg, refc) == (1, 4, 1) and hsize % 8 == 0
    msgs, q = {}, h + 16
    for _ in range(nmsg):
        t, s, fl = struct.unpack_from("<HHB", b, q)
        assert s % 8 == 0
        msgs[t] = (fl, b[q + 8:q + 8 + s])
        q += 8 + s
    assert q == h + 16 + hsize and sorted(msgs) == [0x1, 0x3, 0x5, 0x8]
    sp = msgs[0x1][1]
    assert sp[:4] == bytes([1, 2, 1, 0]) and struct.unpack_from("<4Q", sp, 8) == (8, 4, 8, 4)  # v1, rank 2, max dims = dims
    ty = msgs[0x3][1]
    # class 1 (float) version 1; bit field: little endian, mantissa normalisation "implied" (0x20), sign bit 31; size 4;
    # properties: bit offset 0, precision 32, exponent at 23 of 8 bits, mantissa at 0 of 23 bits, bias 127
    assert ty[:8] == bytes([0x11, 0x20, 31, 0, 4, 0, 0, 0]) and struct.unpack_from("<HHBBBBI", ty, 8) == (0, 32, 23, 8, 0, 23, 127)
    assert msgs[0x5][1][:8] == bytes([2, 2, 2, 1, 0, 0, 0, 0])
    lay = msgs[0x8][1]
    addr, nbytes = struct.unpack_from("<QQ", lay, 2)
    assert lay[:2] == bytes([3, 1]) and nbytes == 8 * 4 * 4 and addr % 8 == 0 and addr + nbytes <= eof
    assert np.array_equal(np.frombuffer(b, np.float32, 32, addr).reshape(8, 4), d["observations"])
    # byte for byte what real h5py puts in the same places (fixture): superblock but the EOF address, root header, B-tree head
    g = open(os.path.join(ROOT, "tests", "golden", "h5py_written.h5"), "rb").read()
    assert g[:40] == b[:40] and g[48:136] == b[48:136] and g[136:144] == b[136:142] + g[142:144] and g[680:688] == b[680:688]


def test_many_datasets_use_several_symbol_nodes(tmp_path):
    d = {f"k{i:03d}": np.full((3,), i, np.float32) for i in range(70)}
    h5io.write_h5(tmp_path / "m.h5", d)
    back = h5io.read_h5(tmp_path / "m.h5")
    assert list(back) == sorted(d) and all(back[k][0] == int(k[1:]) for k in d)
    with pytest.raises(ValueError):
        h5io.write_h5(tmp_path / "x.h5", {f"k{i}": np.zeros(1) for i in range(257)})


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="no libhdf5 tools in this image")
def test_libhdf5_tools_read_a_file_written_here(tmp_path):
    d = _six(16)
    p = tmp_path / "t.h5"
    h5io.write_h5(p, d)
    out = subprocess.check_output([H5DUMP, "-H", str(p)], text=True)
    for k in d:
        assert f'DATASET "{k}"' in out
    assert "H5T_IEEE_F32LE" in out and "SIMPLE { ( 16, 4 ) / ( 16, 4 ) }" in out
    raw = tmp_path / "obs.bin"
    subprocess.check_call([H5DUMP, "-d", "/observations", "-b", "LE", "-o", str(raw), str(p)], stdout=subprocess.DEVNULL)
    assert np.array_equal(np.fromfile(raw, np.float32).reshape(16, 4), d["observations"])


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with h5py in this image")
def test_real_h5py_reads_what_is_written_here_and_the_reverse(tmp_path):
    """Both directions against h5py itself, with the reference's own calls: load_h5_data's visititems + f[k][:] (core.py:61-81)
    on a file written here; save_as_h5's f[key] = array (zoo/util.py:108-111) read here."""
    d = _six(40, seed=3)
    d["scalar"] = np.float64(1.25)
    mine, theirs, ref = tmp_path / "mine.h5", tmp_path / "theirs.h5", tmp_path / "ref.npz"
    h5io.write_h5(mine, d)
    np.savez(ref, **d)
    code = f"""
import h5py, numpy as np
e = np.load({str(ref)!r})
keys, data = [], {{}}
with h5py.File({str(mine)!r}, "r") as f:
    f.visititems(lambda name, item: keys.append(name) if isinstance(item, h5py.Dataset) else None)
    for k in keys:
        try:
            data[k] = f[k][:]
        except ValueError:
            data[k] = f[k][()]
assert sorted(keys) == sorted(e.files), keys
for k in keys:
    assert np.array_equal(data[k], e[k]) and data[k].dtype == e[k].dtype and np.shape(data[k]) == e[k].shape, k
with h5py.File({str(theirs)!r}, "w") as f:
    for k in e.files:
        f[k] = e[k]
print("ok")
"""
    assert subprocess.check_output([CONDA_PY, "-c", code], text=True).strip() == "ok"
    back = h5io.read_h5(theirs)
    assert set(back) == set(d) and all(_same(back[k], d[k]) for k in d)
