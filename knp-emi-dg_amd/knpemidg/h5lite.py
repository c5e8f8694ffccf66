"""Minimal read-only HDF5 subset reader (no h5py in this environment).

Enough of the format for the mesh files the reference ships next to its XDMF descriptors
(reference: examples/emix-simulations/run_EMIx_simulation.py:162-168 reads them through dolfin's XDMFFile):
superblock version 0, version-1 object headers, symbol-table groups, contiguous or chunked datasets with version-1
chunk B-trees, fixed-point / IEEE float types, deflate (+ shuffle) filters.  Everything else raises.
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
        heap_data = self._local_heap(heap)
        for name_off, ohdr in self._group_entries(btree):
            end = b.index(b"\0", heap_data + name_off)
            self.datasets[b[heap_data + name_off:end].decode()] = ohdr

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
