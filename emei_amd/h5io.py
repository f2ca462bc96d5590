"""A minimal HDF5 writer / reader for the reference's offline-dataset container — no h5py needed.

The reference writes its datasets with ``h5py.File(path, "w")`` and ``file[key] = array`` per key
(zoo/util.py:108-111) and reads them back with ``visititems`` + ``f[k][:]`` (emei/core.py:61-81).  h5py's defaults
for that call are the oldest on-disk structures of the HDF5 File Format Specification (version 3.0), and those are
all this module writes:

    superblock version 0 (8-byte offsets and lengths, group leaf K = 4, internal K = 16)                  III.A / II.A
    root group: version-1 object header with ONE Symbol Table message -> v1 B-tree ("TREE", type 0) whose
        leaf children are symbol-table nodes ("SNOD", <= 2K = 8 entries, sorted by name) + a local heap
        ("HEAP") that holds the link names                                                                III.B-D
    one dataset per key: version-1 object header with Dataspace (v1, with max dims), Datatype (v1: IEEE
        float or two's-complement integer, little endian), Fill Value (v2, default) and Data Layout
        (v3, CONTIGUOUS) messages; the raw array bytes, C order                                           IV.A.2

``write_h5`` lays the blocks out in the order h5py itself does (root header at 96, B-tree at 136, heap at 680), so a
file written here is byte-compatible where the format fixes the bytes (tests/test_h5io.py checks those fields against
the specification's offsets, and — where the image offers libhdf5 — that ``h5dump`` / real ``h5py`` read the arrays
back bit for bit).  ``read_h5`` reads what h5py writes by default: the structures above plus object-header
continuation blocks, multi-level group B-trees, nested groups, big-endian numbers and COMPACT layouts; chunked or
filtered datasets raise NotImplementedError (the reference never writes them).
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16
_FREE_NULL = 1  # local heap: "no further free block" (H5HL_FREE_NULL)

MSG_NIL, MSG_DATASPACE, MSG_DATATYPE, MSG_FILL, MSG_LAYOUT, MSG_CONT, MSG_SYMTAB = 0x0, 0x1, 0x3, 0x5, 0x8, 0x10, 0x11


def _pad8(n):
    return (n + 7) & ~7


# ---------------------------------------------------------------------------------------------------------------
# writer
def _datatype_message(dt):
    """Datatype message, version 1 (spec IV.A.2.d): class + version byte, 3 class bit-field bytes, size, properties."""
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise ValueError("little-endian arrays only")
    if dt.kind == "f" and dt.itemsize in (2, 4, 8):
        exp_bits, mant_bits = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = dt.itemsize * 8
        # bit field: byte order LE (bit 0 = 0), no padding, mantissa normalisation 2 = "msb implied" (bits 4-5), sign location
        head = struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize)
        props = struct.pack("<HHBBBBI", 0, bits, mant_bits, exp_bits, 0, mant_bits, (1 << (exp_bits - 1)) - 1)
        return head + props
    if dt.kind in "iub" and dt.itemsize in (1, 2, 4, 8):
        signed = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBI", 0x10, signed, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    raise TypeError(f"dtype {dt} is not supported by this writer (floats and integers only)")


def _message(mtype, data, flags=0):
    data = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages):
    body = b"".join(messages)
    # version 1, reserved, number of messages, reference count 1, header size; 4 bytes pad the prefix to 16
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def _dataset_header(arr, data_addr):
    nd = arr.ndim
    dims = struct.pack(f"<{nd}Q", *arr.shape)
    space = struct.pack("<BBB5x", 1, nd, 1) + dims + dims  # version 1, rank, flags = max dims present (= dims)
    layout = struct.pack("<BBQQ", 3, 1, data_addr if arr.nbytes else UNDEF, arr.nbytes)  # version 3, class 1 = contiguous
    return _object_header([
        _message(MSG_DATASPACE, space),
        _message(MSG_DATATYPE, _datatype_message(arr.dtype), flags=1),       # constant message, as h5py marks it
        _message(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),   # v2: alloc late, write if-set, defined, size 0
        _message(MSG_LAYOUT, layout),
    ])


def write_h5(path, arrays):
    """Write {name: array} as the datasets of the root group of a new HDF5 file (what ``h5py.File(path, "w")`` +
    ``file[name] = array`` per key produce, zoo/util.py:108-111).  Names are plain link names (no "/")."""
    items = []
    for name, a in arrays.items():
        if not isinstance(name, str) or not name or "/" in name or "\0" in name:
            raise ValueError(f"dataset name {name!r}: a non-empty link name without '/' is required")
        a = np.asarray(a)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)  # h5py stores bool as an enum; the reference's keys are all floats
        items.append((name.encode("utf-8"), a.copy(order="C")))  # (np.ascontiguousarray would turn a 0-d array into 1-d)
    items.sort(key=lambda kv: kv[0])  # symbol-table entries are ordered by strcmp of the link names
    if len(items) > 2 * LEAF_K * 2 * INTERNAL_K:
        raise ValueError("too many datasets for a single-level group B-tree")

    # ---- fixed front: superblock 0..96, root object header 96..136, B-tree node 136..680, heap header 680..712
    root_hdr_addr, btree_addr = 96, 136
    btree_size = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8  # header + 2K+1 keys + 2K children
    heap_addr = btree_addr + btree_size
    heap_data_addr = heap_addr + 32
    # heap data segment: "" at offset 0, then the names, each NUL-terminated and padded to 8; one free block closes it
    heap = bytearray(8)
    name_off = []
    for name, _ in items:
        name_off.append(len(heap))
        heap += name + b"\0" * (_pad8(len(name) + 1) - len(name))
    free_off = len(heap)
    heap_size = max(_pad8(free_off + 16), 88)  # room for a free-list block (16 bytes); h5py's initial size is 88
    heap += struct.pack("<QQ", _FREE_NULL, heap_size - free_off) + b"\0" * (heap_size - free_off - 16)
    # symbol nodes
    groups = [list(range(i, min(i + 2 * LEAF_K, len(items)))) for i in range(0, len(items), 2 * LEAF_K)] or [[]]
    snod_size = 8 + 2 * LEAF_K * 40
    snod_addr = [heap_data_addr + heap_size + i * snod_size for i in range(len(groups))]
    pos = snod_addr[-1] + snod_size
    hdr_addr, hdr_bytes = [], []
    for name, a in items:  # object headers first (their size does not depend on the data address), data after them
        hdr_addr.append(pos)
        pos += len(_dataset_header(a, 0))
    data_addr = []
    for name, a in items:
        pos = _pad8(pos)
        data_addr.append(pos)
        pos += a.nbytes
    eof = pos
    for (name, a), da in zip(items, data_addr):
        hdr_bytes.append(_dataset_header(a, da))

    out = bytearray()
    # superblock, version 0 (spec II.A): versions, sizes, K values, flags, base / free-space / EOF / driver addresses, root entry
    out += SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQII", 0, root_hdr_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)  # cache type 1: B-tree + heap
    assert len(out) == 96
    out += _object_header([_message(MSG_SYMTAB, struct.pack("<QQ", btree_addr, heap_addr))])
    assert len(out) == btree_addr
    # B-tree node: group node (type 0), level 0; key[0] = "" (offset 0), key[i + 1] = the largest name in child i
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(groups) if items else 0, UNDEF, UNDEF) + struct.pack("<Q", 0)
    if items:
        for g, addr in zip(groups, snod_addr):
            node += struct.pack("<QQ", addr, name_off[g[-1]])
    out += node + b"\0" * (btree_size - len(node))
    assert len(out) == heap_addr
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, heap_size, free_off, heap_data_addr) + heap
    for g in groups:
        sn = b"SNOD" + struct.pack("<BBH", 1, 0, len(g))
        for i in g:
            sn += struct.pack("<QQII16x", name_off[i], hdr_addr[i], 0, 0)  # cache type 0: nothing cached
        out += sn + b"\0" * (snod_size - len(sn))
    for h in hdr_bytes:
        out += h
    for (name, a), da in zip(items, data_addr):
        out += b"\0" * (da - len(out))
        out += a.tobytes()
    assert len(out) == eof
    with open(path, "wb") as f:
        f.write(out)
    return eof


# ---------------------------------------------------------------------------------------------------------------
# reader
class _File:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file (no signature at offset 0; user blocks are not supported)")
        ver = buf[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"superblock version {ver}: only the version 0 / 1 layout of libver='earliest' is read")
        self.so, self.sl = buf[13], buf[14]
        if (self.so, self.sl) != (8, 8):
            raise NotImplementedError("only 8-byte offsets / lengths")
        p = 24 + (4 if ver == 1 else 0)  # v1 adds indexed-storage K + 2 reserved bytes
        self.base, _, self.eof, _ = struct.unpack_from("<QQQQ", buf, p)
        self.root_header = struct.unpack_from("<Q", buf, p + 32 + 8)[0]  # root symbol-table entry: name offset, header address, ...
        sym = [d for t, _, d in self.messages(self.base + self.root_header) if t == MSG_SYMTAB]
        if not sym:
            raise NotImplementedError("root group without a Symbol Table message (new-style groups are not read)")
        self.root_btree, self.root_heap = struct.unpack_from("<QQ", sym[0])

    def messages(self, addr):
        """(type, flags, data bytes) of every message of the version-1 object header at addr, continuation blocks followed"""
        b = self.b
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise NotImplementedError(f"object header version {ver} (only version 1, libver='earliest')")
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                t, s, fl = struct.unpack_from("<HHB", b, p)
                data = bytes(b[p + 8:p + 8 + s])
                out.append((t, fl, data))
                if t == MSG_CONT:
                    caddr, clen = struct.unpack_from("<QQ", data)
                    blocks.append((self.base + caddr, clen))
                p += 8 + s
        return out

    def heap_name(self, heap_addr, off):
        b = self.b
        heap_addr += self.base
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        data_addr = self.base + struct.unpack_from("<Q", b, heap_addr + 24)[0]
        start = data_addr + off
        end = start
        while b[end] != 0:
            end += 1
        return bytes(b[start:end]).decode("utf-8")

    def links(self, btree_addr, heap_addr):
        """[(name, object header address)] of a symbol-table group, in B-tree order"""
        b = self.b
        btree_addr += self.base
        if b[btree_addr:btree_addr + 4] != b"TREE":
            raise ValueError("bad B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", b, btree_addr + 4)
        if ntype != 0:
            raise ValueError("not a group B-tree node")
        out = []
        for i in range(used):
            child = struct.unpack_from("<Q", b, btree_addr + 24 + 8 + 16 * i)[0]
            if level > 0:
                out += self.links(child, heap_addr)
                continue
            child += self.base
            if b[child:child + 4] != b"SNOD":
                raise ValueError("bad symbol node signature")
            n = struct.unpack_from("<H", b, child + 6)[0]
            for e in range(n):
                noff, oaddr = struct.unpack_from("<QQ", b, child + 8 + 40 * e)
                out.append((self.heap_name(heap_addr, noff), oaddr))
        return out


def _numpy_dtype(data):
    cls, ver = data[0] & 0x0F, data[0] >> 4
    bf0, size = data[1], struct.unpack_from("<I", data, 4)[0]
    order = ">" if bf0 & 1 else "<"
    if cls == 1:
        if size not in (2, 4, 8):
            raise NotImplementedError(f"float of {size} bytes")
        return np.dtype(f"{order}f{size}")
    if cls == 0:
        return np.dtype(f"{order}{'i' if bf0 & 0x08 else 'u'}{size}")
    raise NotImplementedError(f"datatype class {cls} (only fixed-point and floating-point are read)")


def _read_dataset(f, msgs):
    shape = dtype = layout = None
    for t, _, d in msgs:
        if t == MSG_DATASPACE:
            ver, rank = d[0], d[1]
            off = 8 if ver == 1 else 4
            shape = struct.unpack_from(f"<{rank}Q", d, off) if rank else ()
        elif t == MSG_DATATYPE:
            dtype = _numpy_dtype(d)
        elif t == MSG_LAYOUT:
            layout = d
    if shape is None or dtype is None or layout is None:
        raise ValueError("dataset header without dataspace / datatype / layout message")
    if layout[0] != 3:
        raise NotImplementedError(f"data layout message version {layout[0]}")
    count = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    if layout[1] == 1:  # contiguous
        addr, size = struct.unpack_from("<QQ", layout, 2)
        if addr == UNDEF or count == 0:
            return np.zeros(shape, dtype.newbyteorder("="))
        raw = f.b[f.base + addr:f.base + addr + count * dtype.itemsize]
    elif layout[1] == 0:  # compact: the bytes sit in the message
        size = struct.unpack_from("<H", layout, 2)[0]
        raw = layout[4:4 + size]
    else:
        raise NotImplementedError("chunked / filtered datasets are not read by this module (the reference writes contiguous ones)")
    a = np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)
    return a.astype(dtype.newbyteorder("="), copy=True)


def read_h5(path):
    """{"group/dataset": array} of every dataset in the file — the dict emei/core.py:61-81 ``load_h5_data`` builds."""
    with open(path, "rb") as fh:
        f = _File(memoryview(fh.read()))
    out = {}

    def walk(btree, heap, prefix):
        for name, oaddr in f.links(btree, heap):
            msgs = f.messages(f.base + oaddr)
            sym = [d for t, _, d in msgs if t == MSG_SYMTAB]
            if sym:
                walk(*struct.unpack_from("<QQ", sym[0]), prefix + name + "/")
            else:
                out[prefix + name] = _read_dataset(f, msgs)

    walk(f.root_btree, f.root_heap, "")
    return out
