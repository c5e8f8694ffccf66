"""Minimal HDF5 subset reader and writer (no h5py in this environment).

Enough of the format for the mesh files the reference ships next to its XDMF descriptors
(reference: examples/emix-simulations/run_EMIx_simulation.py:162-168 reads them through dolfin's XDMFFile):
superblock version 0, version-1 object headers, symbol-table groups, contiguous or chunked datasets with version-1
chunk B-trees, fixed-point / IEEE float types, deflate (+ shuffle) filters.  Everything else raises.

`H5Writer` writes the same subset (nested symbol-table groups, contiguous datasets) for the result files of
`Solver.init_h5_savefile / save_h5` (reference: src/knpemidg/solver.py:1214-1242: /mesh, /subdomains, /surfaces and the time series
/concentrations/vector_n, /elim_concentration/vector_n, /potential/vector_n).
"""
import struct
import zlib

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


class H5File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise H5Error("not an HDF5 file")
        if b[8] != 0:
            raise H5Error("superblock version %d not supported" % b[8])
        if b[13] != 8 or b[14] != 8:
            raise H5Error("only 8-byte offsets / lengths are supported")
        # root group symbol table entry follows base / free-space / eof / driver addresses
        ste = 24 + 32
        _, _, cache_type = struct.unpack_from("<QQI", b, ste)
        if cache_type != 1:
            raise H5Error("root group without cached symbol table")
        btree, heap = struct.unpack_from("<QQ", b, ste + 24)
        self.datasets = {}
        self._walk("", btree, heap)

    def _walk(self, prefix, btree, heap):
        """Collect every object below a group: name -> object header address (nested groups as 'a/b/c')."""
        b = self.buf
        heap_data = self._local_heap(heap)
        for name_off, ohdr in self._group_entries(btree):
            end = b.index(b"\0", heap_data + name_off)
            name = prefix + b[heap_data + name_off:end].decode()
            sub = [d for t, d in self._messages(ohdr) if t == 0x11]
            if sub:
                self._walk(name + "/", *struct.unpack_from("<QQ", sub[0], 0))
            else:
                self.datasets[name] = ohdr

    # -- groups ------------------------------------------------------------------------------------
    def _local_heap(self, addr):
        b = self.buf
        if b[addr:addr + 4] != b"HEAP":
            raise H5Error("bad local heap")
        return struct.unpack_from("<Q", b, addr + 24)[0]

    def _group_entries(self, addr):
        b = self.buf
        if b[addr:addr + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                off = addr + 8 + 40 * i
                yield struct.unpack_from("<QQ", b, off)
            return
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise H5Error("bad group B-tree node")
        n = struct.unpack_from("<H", b, addr + 6)[0]
        for i in range(n):
            child = struct.unpack_from("<Q", b, addr + 24 + 8 + 16 * i)[0]
            yield from self._group_entries(child)

    # -- object headers ----------------------------------------------------------------------------
    def _messages(self, addr):
        b = self.buf
        if b[addr] != 1:
            raise H5Error("object header version %d not supported" % b[addr])
        nmsg, _, size = struct.unpack_from("<HII", b, addr + 2)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            while left >= 8 and len(out) < nmsg:
                mtype, msize, _ = struct.unpack_from("<HHB", b, pos)
                data = b[pos + 8:pos + 8 + msize]
                if mtype == 0x10:
                    blocks.append(struct.unpack_from("<QQ", data, 0))
                out.append((mtype, data))
                pos += 8 + msize
                left -= 8 + msize
        return out

    def read(self, name):
        name = name.lstrip("/")
        if name not in self.datasets:
            raise KeyError(name)
        shape = dtype = layout = None
        filters = []
        for mtype, d in self._messages(self.datasets[name]):
            if mtype == 0x01:
                ver, rank, flags = d[0], d[1], d[2]
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, d, off)
            elif mtype == 0x03:
                cls = d[0] & 0x0F
                bits0 = d[1]
                size = struct.unpack_from("<I", d, 4)[0]
                if bits0 & 1:
                    raise H5Error("big-endian data not supported")
                if cls == 0:
                    dtype = np.dtype("<%s%d" % ("i" if bits0 & 8 else "u", size))
                elif cls == 1:
                    dtype = np.dtype("<f%d" % size)
                else:
                    raise H5Error("datatype class %d not supported" % cls)
            elif mtype == 0x08:
                if d[0] != 3:
                    raise H5Error("data layout version %d not supported" % d[0])
                if d[1] == 1:
                    layout = ("contiguous",) + struct.unpack_from("<QQ", d, 2)
                elif d[1] == 2:
                    nd = d[2]
                    bt = struct.unpack_from("<Q", d, 3)[0]
                    layout = ("chunked", bt, struct.unpack_from("<%dI" % nd, d, 11))
                else:
                    raise H5Error("compact layout not supported")
            elif mtype == 0x0B:
                if d[0] != 1:
                    raise H5Error("filter pipeline version %d not supported" % d[0])
                pos = 8
                for _ in range(d[1]):
                    fid, nlen, _, ncd = struct.unpack_from("<HHHH", d, pos)
                    pos += 8 + ((nlen + 7) // 8) * 8 + 4 * (ncd + (ncd & 1))
                    filters.append(fid)
        if shape is None or dtype is None or layout is None:
            raise H5Error("dataset %s: incomplete header" % name)
        n = int(np.prod(shape))
        if layout[0] == "contiguous":
            return np.frombuffer(self.buf, dtype=dtype, count=n, offset=layout[1]).reshape(shape).copy()
        for fid in filters:
            if fid not in (1, 2):
                raise H5Error("filter %d not supported" % fid)
        chunk = layout[2][:-1]
        out = np.zeros(shape, dtype=dtype)
        for offs, raw in self._chunks(layout[1], len(shape)):
            for fid in reversed(filters):
                if fid == 1:
                    raw = zlib.decompress(raw)
                else:                                   # shuffle: bytes of equal significance stored together
                    a = np.frombuffer(raw, dtype=np.uint8).reshape(dtype.itemsize, -1)
                    raw = np.ascontiguousarray(a.T).tobytes()
            blk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(chunk))).reshape(chunk)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
            out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks(self, addr, rank):
        b = self.buf
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise H5Error("bad chunk B-tree node")
        level = b[addr + 5]
        n = struct.unpack_from("<H", b, addr + 6)[0]
        ksize = 8 + 8 * (rank + 1)
        pos = addr + 24
        for _ in range(n):
            csize, _ = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from("<%dQ" % rank, b, pos + 8)
            child = struct.unpack_from("<Q", b, pos + ksize)[0]
            if level == 0:
                yield offs, b[child:child + csize]
            else:
                yield from self._chunks(child, rank)
            pos += ksize + 8


def read_xdmf_mesh(xdmf_path):
    """(coords f64[Nv, d], cells i64[Nc, d+1], {attribute name: array}) of a single-grid XDMF file whose heavy data sit in
    HDF5 (the layout meshio / dolfin write; reference meshes: examples/emix-simulations/meshes/.../mesh.xdmf)."""
    import os
    import re
    txt = open(xdmf_path).read()
    items = re.findall(r"<(Geometry|Topology|Attribute)([^>]*)>\s*<DataItem[^>]*>([^<:]+):([^<]+)</DataItem>", txt)
    files = {}
    coords = cells = None
    attrs = {}
    for kind, tagattrs, fname, dset in items:
        fname = fname.strip()
        if fname not in files:
            files[fname] = H5File(os.path.join(os.path.dirname(os.path.abspath(xdmf_path)), fname))
        arr = files[fname].read(dset.strip().lstrip("/"))
        if kind == "Geometry":
            coords = np.ascontiguousarray(arr, dtype=np.float64)
        elif kind == "Topology":
            cells = np.ascontiguousarray(arr, dtype=np.int64)
        else:
            m = re.search(r'Name="([^"]+)"', tagattrs)
            attrs[m.group(1) if m else "attribute%d" % len(attrs)] = arr
    if coords is None or cells is None:
        raise H5Error("XDMF file without geometry / topology")
    return coords, cells, attrs


# ------------------------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------------------------
class H5Writer:
    """Streams datasets to disk as they arrive (raw data appended to the file) and writes all metadata -- nested groups as
    version-1 B-tree / local heap / symbol-table nodes, version-1 object headers with dataspace, datatype, fill-value and
    contiguous-layout messages -- when the file is closed.  Superblock version 0, little endian, 8-byte offsets."""

    _K_LEAF = 4096                     # symbols per symbol-table node <= 2 K: one node per group (time series of any length)

    def __init__(self, path):
        self.path = path
        self.fh = open(path, "wb")
        self.fh.write(b"\0" * 96)      # superblock, patched on close
        self.items = {}                # path -> (offset, shape, dtype)

    def write(self, name, array):
        a = np.ascontiguousarray(array)
        if a.dtype.kind not in "iuf" or a.dtype.itemsize not in (1, 2, 4, 8):
            raise H5Error("dtype %s not supported" % a.dtype)
        a = a.astype(a.dtype.newbyteorder("<"), copy=False)
        name = name.strip("/")
        if name in self.items or not name:
            raise H5Error("dataset %r already written" % name)
        self._align()
        off = self.fh.tell()
        self.fh.write(a.tobytes())
        self.items[name] = (off, a.shape, a.dtype)

    def _align(self):
        pad = (-self.fh.tell()) % 8
        if pad:
            self.fh.write(b"\0" * pad)

    def _put(self, blob):
        self._align()
        off = self.fh.tell()
        self.fh.write(blob)
        return off

    @staticmethod
    def _msg(mtype, data):
        data = data + b"\0" * ((-len(data)) % 8)
        return struct.pack("<HHB3x", mtype, len(data), 0) + data

    def _object_header(self, msgs):
        body = b"".join(msgs)
        return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body

    def _dataset_header(self, off, shape, dtype):
        n = int(np.prod(shape)) * dtype.itemsize if len(shape) else dtype.itemsize
        space = struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)
        if dtype.kind == "f":
            bits = dtype.itemsize * 8
            mant, expo = (52, 11) if bits == 64 else (23, 8)
            dt = struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dtype.itemsize) + \
                struct.pack("<HHBBBBI", 0, bits, mant, expo, 0, mant, (1 << (expo - 1)) - 1)
        else:
            dt = struct.pack("<BBBBI", 0x10, 0x08 if dtype.kind == "i" else 0x00, 0, 0, dtype.itemsize) + \
                struct.pack("<HH", 0, dtype.itemsize * 8)
        fill = struct.pack("<BBBB", 2, 2, 0, 0)
        layout = struct.pack("<BBQQ", 3, 1, off if n else _UNDEF, n)
        return self._object_header([self._msg(0x01, space), self._msg(0x03, dt), self._msg(0x05, fill), self._msg(0x08, layout)])

    def _write_group(self, tree):
        """tree: {name: subtree dict | (off, shape, dtype)}.  Returns (object header address, btree address, heap address)."""
        names = sorted(tree, key=lambda s: s.encode())
        if len(names) > 2 * self._K_LEAF:
            raise H5Error("too many entries in one group")
        entries = []
        for nm in names:
            node = tree[nm]
            if isinstance(node, dict):
                oh, bt, hp = self._write_group(node)
                entries.append((nm, oh, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((nm, self._put(self._dataset_header(*node)), 0, b"\0" * 16))
        heap = bytearray(8)                                         # offset 0: the empty name
        offs = []
        for nm, *_ in entries:
            offs.append(len(heap))
            raw = nm.encode() + b"\0"
            heap += raw + b"\0" * ((-len(raw)) % 8)
        heap += b"\0" * 16                                          # a free block keeps library writers happy
        heap_data = self._put(bytes(heap))
        heap_addr = self._put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), len(heap) - 16, heap_data))
        # the free block header inside the data segment: next free = 1 (none), size
        self.fh.seek(heap_data + len(heap) - 16)
        self.fh.write(struct.pack("<QQ", 1, 16))
        self.fh.seek(0, 2)
        snod = b"SNOD" + struct.pack("<BxH", 1, len(entries))
        for (nm, oh, cache, scratch), o in zip(entries, offs):
            snod += struct.pack("<QQI4x", o, oh, cache) + scratch
        snod += b"\0" * (40 * (2 * self._K_LEAF - len(entries)))
        snod_addr = self._put(snod)
        K_INT = 16
        bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, _UNDEF, _UNDEF) + struct.pack("<QQQ", 0, snod_addr, offs[-1] if offs else 0)
        bt += b"\0" * (16 * (2 * K_INT - 1))
        bt_addr = self._put(bt)
        oh_addr = self._put(self._object_header([self._msg(0x11, struct.pack("<QQ", bt_addr, heap_addr))]))
        return oh_addr, bt_addr, heap_addr

    def close(self):
        if self.fh is None:
            return
        tree = {}
        for name, item in self.items.items():
            node = tree
            parts = name.split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
                if not isinstance(node, dict):
                    raise H5Error("%r is both a dataset and a group" % part)
            node[parts[-1]] = item
        oh, bt, hp = self._write_group(tree)
        self._align()
        eof = self.fh.tell()
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self._K_LEAF, 16, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
        sb += struct.pack("<QQI4xQQ", 0, oh, 1, bt, hp)
        assert len(sb) == 96
        self.fh.seek(0)
        self.fh.write(sb)
        self.fh.close()
        self.fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
