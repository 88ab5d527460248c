"""A small, self-contained HDF5 reader / writer (pure Python + numpy) with the part of h5py's surface that
PGDrome's result files need.

Why it exists: the reference keeps its heavy result data in HDF5 - ``dolfin.HDF5File`` / ``dolfin.XDMFFile``
write it, ``h5py`` reads it back (/root/reference/pgdrome/model.py:162-196, 264-306, 470-560) - and neither
library is part of this image.  Result files written here are real HDF5 (checked against libhdf5's own
``h5dump`` / ``h5ls`` where those tools exist, tests/test_h5lite.py), so they open in ParaView, h5py and real
PGDrome, and files written by those (contiguous, compact or chunked - optionally deflate / shuffle - datasets
of integers, floats and fixed strings, old- and new-style groups, attributes) are read here.

Writer: superblock version 0, object headers version 1, groups as symbol tables (local heap + one-level
B-tree), contiguous datasets, attributes version 1 - the "earliest" format every HDF5 release reads.

    with File(path, "w") as f:
        f.create_dataset("/Mesh/0/mesh/geometry", data=coords)
        f["/Mesh/0/mesh/topology"].attrs["celltype"] = "interval"
    with File(path, "r") as f:
        geom = np.array(f.get("Mesh/0/mesh/geometry"))

Format reference: "HDF5 File Format Specification Version 3.0" (The HDF Group; public).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


# ===================================================================== datatypes
def _dtype_message(dt: np.dtype) -> bytes:
    """Datatype message (version 1) of a numpy dtype: fixed-point, IEEE float or fixed-length string."""
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits = (0x08 if dt.kind == "i" else 0x00)            # byte order little-endian, signed flag
        return struct.pack("<BBBBI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 8:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            return struct.pack("<BBBBI", 0x10 | 1, 0x20, 63, 0, 8) + props
        props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        return struct.pack("<BBBBI", 0x10 | 1, 0x20, 31, 0, 4) + props
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x10 | 3, 0x00, 0, 0, max(dt.itemsize, 1))     # null-terminated ASCII
    raise H5Error("h5lite cannot store dtype %r" % (dt,))


def _parse_datatype(buf: bytes):
    """(numpy dtype or None, description) of a datatype message."""
    cls = buf[0] & 0x0F
    b0, b1 = buf[1], buf[2]
    size = struct.unpack_from("<I", buf, 4)[0]
    if cls == 0:
        order = ">" if (b0 & 1) else "<"
        return np.dtype("%s%s%d" % (order, "i" if (b0 & 0x08) else "u", size)), "int"
    if cls == 1:
        order = ">" if (b0 & 1) else "<"
        if size not in (2, 4, 8):
            raise H5Error("floating-point type of %d bytes" % size)
        return np.dtype("%sf%d" % (order, size)), "float"
    if cls == 3:
        return np.dtype("S%d" % size), "string"
    if cls == 9:
        base, _ = _parse_datatype(buf[8:])
        kind = b0 & 0x0F
        return None, ("vlen-string" if kind == 1 else "vlen")
    if cls == 6:
        raise H5Error("compound datatypes are not supported by h5lite")
    if cls == 8:        # enum: read as its base integer
        return _parse_datatype(buf[8:])
    raise H5Error("datatype class %d is not supported by h5lite" % cls)


# ===================================================================== writer
class _Node:
    """In-memory tree of the file being written."""

    def __init__(self, name, parent=None):
        self.name, self.parent = name, parent
        self.children = {}          # groups only
        self.data = None            # datasets only: numpy array
        self.attrs = {}
        self.addr = None

    @property
    def is_dataset(self):
        return self.data is not None


class AttributeManager:
    def __init__(self, store, writable):
        self._store, self._writable = store, writable

    def __getitem__(self, k):
        return self._store[k]

    def __setitem__(self, k, v):
        if not self._writable:
            raise H5Error("file is open read-only")
        self._store[str(k)] = v

    def __contains__(self, k):
        return k in self._store

    def get(self, k, default=None):
        return self._store.get(k, default)

    def keys(self):
        return self._store.keys()

    def items(self):
        return self._store.items()

    def __iter__(self):
        return iter(self._store)

    def __len__(self):
        return len(self._store)


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * ((-len(b)) % 8)


def _attr_value_array(v):
    if isinstance(v, str):
        v = v.encode("utf-8")
    if isinstance(v, bytes):
        return np.array(v, dtype="S%d" % max(len(v) + 1, 1)), ()      # room for the terminator
    a = np.asarray(v)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "b":
        a = a.astype(np.int8)
    if a.dtype.kind == "O":
        raise H5Error("object arrays cannot be stored as attributes")
    return np.ascontiguousarray(a).reshape(a.shape), a.shape


def _dataspace_message(shape) -> bytes:
    if shape == ():
        return struct.pack("<BBBBI", 1, 0, 0, 0, 0)
    return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attribute_message(name: str, value) -> bytes:
    arr, shape = _attr_value_array(value)
    nm = name.encode("utf-8") + b"\x00"
    dt = _dtype_message(arr.dtype)
    ds = _dataspace_message(shape)
    head = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds))
    return head + _pad8(nm) + _pad8(dt) + _pad8(ds) + arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHBBBB", mtype, len(body), flags, 0, 0, 0) + body


def _object_header(messages) -> bytes:
    blob = b"".join(messages)
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(blob)) + b"\x00" * 4 + blob


class _Writer:
    def __init__(self, path):
        self.path = path
        self.root = _Node("")

    # ---- tree building
    def node(self, name, create_groups=False):
        parts = [p for p in name.split("/") if p]
        cur = self.root
        for i, p in enumerate(parts):
            nxt = cur.children.get(p)
            if nxt is None:
                if not create_groups:
                    return None
                if cur.is_dataset:
                    raise H5Error("%r is a dataset, not a group" % cur.name)
                nxt = _Node(p, cur)
                cur.children[p] = nxt
            cur = nxt
        return cur

    # ---- serialisation
    def write(self):
        leaf_k = 4
        stack = [self.root]
        while stack:
            n = stack.pop()
            if not n.is_dataset:
                leaf_k = max(leaf_k, (len(n.children) + 1) // 2)
                stack.extend(n.children.values())
        if leaf_k > 32000:
            raise H5Error("too many links in one group for h5lite's writer")
        self.leaf_k, self.internal_k = leaf_k, 16
        self.buf = bytearray(96)                       # superblock placeholder
        root_addr, btree, heap = self._emit_group(self.root)
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0)
        sb += struct.pack("<HHI", self.leaf_k, self.internal_k, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[0:96] = sb
        # through a temporary file in the same directory + os.replace: a failure while writing leaves the old file as it was
        import os
        tmp = "%s.h5lite-tmp-%d" % (self.path, os.getpid())
        try:
            with open(tmp, "wb") as f:
                f.write(bytes(self.buf))
            os.replace(tmp, self.path)
        except BaseException:
            try:
                os.unlink(tmp)
            except OSError:
                pass
            raise

    def _alloc(self, blob: bytes) -> int:
        self.buf.extend(b"\x00" * ((-len(self.buf)) % 8))
        addr = len(self.buf)
        self.buf.extend(blob)
        return addr

    def _attr_messages(self, n):
        return [_message(0x000C, _attribute_message(k, v)) for k, v in n.attrs.items()]

    def _emit_dataset(self, n) -> int:
        arr = np.asarray(n.data)
        arr = np.ascontiguousarray(arr).reshape(arr.shape)        # (ascontiguousarray alone turns 0-d into 1-d)
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf-8")
        if arr.dtype.kind == "b":
            arr = arr.astype(np.int8)
        arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
        raw = arr.tobytes()
        data_addr = self._alloc(raw) if raw else UNDEF
        msgs = [_message(0x0001, _dataspace_message(arr.shape)),
                _message(0x0003, _dtype_message(arr.dtype), flags=1),
                _message(0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),                    # fill value: default, late alloc
                _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, len(raw)))]          # layout v3, contiguous
        msgs += self._attr_messages(n)
        n.addr = self._alloc(_object_header(msgs))
        return n.addr

    def _emit_group(self, n):
        names = sorted(n.children)                       # strcmp order = byte order of the UTF-8 names
        entries = []
        for nm in names:
            ch = n.children[nm]
            if ch.is_dataset:
                entries.append((nm, self._emit_dataset(ch), 0, 0, 0))
            else:
                a, bt, hp = self._emit_group(ch)
                entries.append((nm, a, 1, bt, hp))
        # local heap: offset 0 holds the empty string, then the link names, each 8-byte aligned
        seg = bytearray(b"\x00" * 8)
        offs = []
        for nm in names:
            offs.append(len(seg))
            seg.extend(_pad8(nm.encode("utf-8") + b"\x00"))
        free_off = len(seg)
        seg.extend(struct.pack("<QQ", 1, 16))             # one free block closes the segment (next = none, 16 bytes)
        seg_addr = self._alloc(bytes(seg))
        heap = self._alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(seg), free_off, seg_addr))
        # one symbol-table node (2 * leaf_k entries allocated) under a one-entry B-tree
        snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(entries)))
        for (nm, addr, cache, bt, hp), off in zip(entries, offs):
            snod += struct.pack("<QQII", off, addr, cache, 0) + (struct.pack("<QQ", bt, hp) if cache else b"\x00" * 16)
        snod += b"\x00" * (8 + 2 * self.leaf_k * 40 - len(snod))
        snod_addr = self._alloc(bytes(snod))
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, UNDEF, UNDEF))
        if entries:
            node += struct.pack("<QQQ", 0, snod_addr, offs[-1])
        node += b"\x00" * (24 + (2 * self.internal_k + 1) * 8 + 2 * self.internal_k * 8 - len(node))
        btree = self._alloc(bytes(node))
        msgs = [_message(0x0011, struct.pack("<QQ", btree, heap))] + self._attr_messages(n)
        n.addr = self._alloc(_object_header(msgs))
        return n.addr, btree, heap


# ===================================================================== reader
class _Reader:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        b = self.b
        base = 0
        while b[base:base + 8] != SIGNATURE:
            base = 512 if base == 0 else base * 2       # the superblock may sit at 0, 512, 1024, ...
            if base > len(b):
                raise H5Error("%s is not an HDF5 file" % path)
        ver = b[base + 8]
        if ver in (0, 1):
            self.so, self.sl = b[base + 13], b[base + 14]
            p = base + 16 + 8 + (4 if ver == 1 else 0)
            self.base_addr = self._off(p)
            p += 4 * self.so
            self.root = self._off(p + self.so)           # root symbol-table entry: link name offset, header address
        elif ver in (2, 3):
            self.so, self.sl = b[base + 9], b[base + 10]
            p = base + 12
            self.base_addr = self._off(p)
            self.root = self._off(p + 3 * self.so)
        else:
            raise H5Error("superblock version %d" % ver)
        if self.so != 8 or self.sl != 8:
            raise H5Error("h5lite reads files with 8-byte offsets and lengths only")

    def _off(self, p):
        return struct.unpack_from("<Q", self.b, p)[0]

    # ---- object headers
    def messages(self, addr):
        """[(type, flags, body bytes)] of the object header at addr (versions 1 and 2, continuation blocks followed)."""
        b = self.b
        addr += self.base_addr
        out = []
        if b[addr:addr + 4] == b"OHDR":
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szf = 1 << (flags & 3)
            size = int.from_bytes(b[p:p + szf], "little")
            p += szf
            blocks = [(p, size)]
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end:
                    mtype, msize, mflags = b[p], struct.unpack_from("<H", b, p + 1)[0], b[p + 3]
                    p += 4 + (2 if flags & 0x04 else 0)
                    body = b[p:p + msize]
                    p += msize
                    if mtype == 0x10:
                        o, ln = struct.unpack_from("<QQ", body)
                        blocks.append((o + self.base_addr + 4, ln - 8))          # skip "OCHK", drop the checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, body))
            return out
        if b[addr] != 1:
            raise H5Error("object header version %d at %d" % (b[addr], addr))
        nmsg, = struct.unpack_from("<H", b, addr + 2)
        size, = struct.unpack_from("<I", b, addr + 8)
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg + 64:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, p)
                body = b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x10:
                    o, ln = struct.unpack_from("<QQ", body)
                    blocks.append((o + self.base_addr, ln))
                elif mtype != 0:
                    out.append((mtype, mflags, body))
        return out

    # ---- groups
    def links(self, msgs):
        """{name: object header address} of a group."""
        out = {}
        for mtype, _, body in msgs:
            if mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", body)
                hb = heap + self.base_addr
                if self.b[hb:hb + 4] != b"HEAP":
                    raise H5Error("bad local heap")
                seg = self._off(hb + 24) + self.base_addr
                self._walk_group_btree(btree, seg, out)
            elif mtype == 0x06:
                name, addr = self._link_message(body)
                if addr is not None:
                    out[name] = addr
            elif mtype == 0x02:
                fheap = struct.unpack_from("<Q", body, 2 + (8 if body[1] & 1 else 0))[0]
                if fheap != UNDEF:
                    raise H5Error("groups with dense link storage (fractal heap) are not supported by h5lite")
        return out

    def _walk_group_btree(self, addr, seg, out):
        b = self.b
        p = addr + self.base_addr
        if b[p:p + 4] != b"TREE" or b[p + 4] != 0:
            raise H5Error("bad group B-tree node")
        level, used = b[p + 5], struct.unpack_from("<H", b, p + 6)[0]
        p += 8 + 16
        for i in range(used):
            child = self._off(p + 8 + 16 * i)
            if level > 0:
                self._walk_group_btree(child, seg, out)
                continue
            q = child + self.base_addr
            if b[q:q + 4] != b"SNOD":
                raise H5Error("bad symbol-table node")
            n, = struct.unpack_from("<H", b, q + 6)
            for e in range(n):
                off, hdr = struct.unpack_from("<QQ", b, q + 8 + 40 * e)
                end = b.index(b"\x00", seg + off)
                out[b[seg + off:end].decode("utf-8")] = hdr

    def _link_message(self, body):
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        szf = 1 << (flags & 3)
        ln = int.from_bytes(body[p:p + szf], "little")
        p += szf
        name = body[p:p + ln].decode("utf-8")
        p += ln
        if ltype != 0:
            return name, None                              # soft / external links are not followed
        return name, struct.unpack_from("<Q", body, p)[0]

    # ---- attributes
    def attributes(self, msgs):
        out = {}
        for mtype, _, body in msgs:
            if mtype != 0x0C:
                continue
            ver = body[0]
            nlen, tlen, slen = struct.unpack_from("<HHH", body, 2)
            p = 8 + (1 if ver == 3 else 0)
            pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
            name = body[p:p + nlen].split(b"\x00")[0].decode("utf-8")
            p += pad(nlen)
            tmsg = body[p:p + tlen]
            p += pad(tlen)
            smsg = body[p:p + slen]
            p += pad(slen)
            shape = self._parse_dataspace(smsg)
            try:
                dt, kind = _parse_datatype(tmsg)
            except H5Error:
                continue
            if dt is None:                                  # variable-length string: global heap reference(s)
                if kind != "vlen-string":
                    continue
                vals = [self._global_heap_string(body, p + 16 * i) for i in range(int(np.prod(shape)) if shape else 1)]
                out[name] = vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
                continue
            n = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(body, dtype=dt, count=n, offset=p)
            out[name] = self._present(arr, shape, kind)
        return out

    @staticmethod
    def _present(arr, shape, kind):
        if kind == "string":
            vals = [v.split(b"\x00")[0].decode("utf-8", "replace") for v in arr.tolist()]
            return vals[0] if shape == () else np.array(vals).reshape(shape)
        arr = arr.astype(arr.dtype.newbyteorder("="))
        return arr.reshape(shape)[()] if shape == () else arr.reshape(shape).copy()

    def _global_heap_string(self, body, p):
        ln, addr, idx = struct.unpack_from("<IQI", body, p)
        b = self.b
        q = addr + self.base_addr
        if b[q:q + 4] != b"GCOL":
            raise H5Error("bad global heap collection")
        size = self._off(q + 8)
        r = q + 16
        while r < q + size:
            oid, _, _, osz = struct.unpack_from("<HHIQ", b, r)
            if oid == idx:
                return b[r + 16:r + 16 + ln].decode("utf-8", "replace")
            if oid == 0:
                break
            r += 16 + ((osz + 7) & ~7)
        raise H5Error("global heap object %d not found" % idx)

    @staticmethod
    def _parse_dataspace(body):
        ver, rank = body[0], body[1]
        if ver == 1:
            p = 8
        elif ver == 2:
            if body[3] == 2:
                return None                                 # null dataspace
            p = 4
        else:
            raise H5Error("dataspace message version %d" % ver)
        return tuple(struct.unpack_from("<Q", body, p + 8 * i)[0] for i in range(rank))

    # ---- datasets
    def dataset_info(self, msgs):
        shape = dt = kind = layout = None
        filters = []
        for mtype, _, body in msgs:
            if mtype == 0x01:
                shape = self._parse_dataspace(body)
            elif mtype == 0x03:
                dt, kind = _parse_datatype(body)
            elif mtype == 0x08:
                layout = body
            elif mtype == 0x0B:
                filters = self._parse_filters(body)
        if layout is None or dt is None and kind is None:
            return None
        return shape, dt, kind, layout, filters

    @staticmethod
    def _parse_filters(body):
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid, = struct.unpack_from("<H", body, p)
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen, = struct.unpack_from("<H", body, p)
                p += 2
            _, ncv = struct.unpack_from("<HH", body, p)
            p += 4
            if nlen:
                p += (nlen + 7) & ~7 if ver == 1 else nlen
            cv = struct.unpack_from("<%dI" % ncv, body, p)
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
            out.append((fid, cv))
        return out

    def read_dataset(self, msgs):
        shape, dt, kind, layout, filters = self.dataset_info(msgs)
        if dt is None:
            raise H5Error("datasets of variable-length type are not supported by h5lite")
        if shape is None:
            return np.zeros((0,), dtype=dt)
        n = int(np.prod(shape)) if shape else 1
        ver, cls = layout[0], layout[1]
        if ver not in (3, 4):
            raise H5Error("data layout message version %d" % ver)
        if cls == 0:
            size, = struct.unpack_from("<H", layout, 2)
            raw = layout[4:4 + size]
        elif cls == 1:
            addr, size = struct.unpack_from("<QQ", layout, 2)
            raw = b"\x00" * (n * dt.itemsize) if addr == UNDEF else self.b[addr + self.base_addr:addr + self.base_addr + size]
        elif cls == 2:
            if ver != 3:
                raise H5Error("chunked datasets of the version-4 layout (libver latest) are not supported by h5lite")
            nd = layout[2]
            btree, = struct.unpack_from("<Q", layout, 3)
            cdims = struct.unpack_from("<%dI" % nd, layout, 11)
            return self._read_chunked(shape, dt, kind, btree, cdims[:-1], filters)
        else:
            raise H5Error("data layout class %d" % cls)
        arr = np.frombuffer(raw, dtype=dt, count=n)
        return self._present_data(arr, shape, kind)

    @staticmethod
    def _present_data(arr, shape, kind):
        arr = arr.reshape(shape) if shape else arr.reshape(())
        if kind == "string":
            return arr.copy()
        return arr.astype(arr.dtype.newbyteorder("="))

    def _read_chunked(self, shape, dt, kind, btree, cdims, filters):
        out = np.zeros(shape, dtype=dt)
        if btree == UNDEF:
            return self._present_data(out, shape, kind)
        rank = len(shape)
        csize = int(np.prod(cdims)) * dt.itemsize

        def walk(addr):
            b = self.b
            p = addr + self.base_addr
            if b[p:p + 4] != b"TREE" or b[p + 4] != 1:
                raise H5Error("bad chunk B-tree node")
            level, used = b[p + 5], struct.unpack_from("<H", b, p + 6)[0]
            p += 24
            ksz = 8 + 8 * (rank + 1)
            for i in range(used):
                q = p + i * (ksz + 8)
                nbytes, mask = struct.unpack_from("<II", b, q)
                offs = struct.unpack_from("<%dQ" % rank, b, q + 8)
                child = self._off(q + ksz)
                if level > 0:
                    walk(child)
                    continue
                raw = b[child + self.base_addr:child + self.base_addr + nbytes]
                for k, (fid, cv) in reversed(list(enumerate(filters))):
                    if mask & (1 << k):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cv[0] if cv else dt.itemsize
                        a = np.frombuffer(raw, dtype=np.uint8)
                        m = a.size // es
                        raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                    elif fid == 3:
                        raw = raw[:-4]
                    else:
                        raise H5Error("filter %d is not supported by h5lite" % fid)
                chunk = np.frombuffer(raw[:csize], dtype=dt).reshape(cdims)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        walk(btree)
        return self._present_data(out, shape, kind)


# ===================================================================== the h5py-like surface
class Dataset:
    def __init__(self, file, name, reader=None, msgs=None, node=None):
        self.file, self.name = file, name
        self._reader, self._msgs, self._node = reader, msgs, node
        self._cache = None

    def _data(self):
        if self._node is not None:
            return self._node.data
        if self._cache is None:
            self._cache = self._reader.read_dataset(self._msgs)
        return self._cache

    @property
    def shape(self):
        return self._data().shape

    @property
    def dtype(self):
        return self._data().dtype

    @property
    def attrs(self):
        if self._node is not None:
            return AttributeManager(self._node.attrs, True)
        return AttributeManager(self._reader.attributes(self._msgs), False)

    def __array__(self, dtype=None, copy=None):
        a = self._data()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        return self._data()[key]

    def __len__(self):
        return self._data().shape[0]


class Group:
    def __init__(self, file, name, reader=None, msgs=None, node=None):
        self.file, self.name = file, name or "/"
        self._reader, self._msgs, self._node = reader, msgs, node

    # ---- reading
    def _links(self):
        if self._node is not None:
            return {k: v for k, v in self._node.children.items()}
        return self._reader.links(self._msgs)

    def keys(self):
        return list(self._links().keys())

    def __iter__(self):
        return iter(self.keys())

    def __contains__(self, name):
        return self.get(name) is not None

    def get(self, name, default=None):
        parts = [p for p in str(name).split("/") if p]
        cur = self if not str(name).startswith("/") else self.file._root()
        for p in parts:
            if not isinstance(cur, Group):
                return default
            cur = cur._child(p)
            if cur is None:
                return default
        return cur

    def __getitem__(self, name):
        out = self.get(name)
        if out is None:
            raise KeyError("%r not found in %r" % (name, self.name))
        return out

    def _child(self, p):
        path = (self.name.rstrip("/") + "/" + p)
        if self._node is not None:
            n = self._node.children.get(p)
            if n is None:
                return None
            return Dataset(self.file, path, node=n) if n.is_dataset else Group(self.file, path, node=n)
        addr = self._links().get(p)
        if addr is None:
            return None
        msgs = self._reader.messages(addr)
        if self._reader.dataset_info(msgs) is not None:
            return Dataset(self.file, path, self._reader, msgs)
        return Group(self.file, path, self._reader, msgs)

    @property
    def attrs(self):
        if self._node is not None:
            return AttributeManager(self._node.attrs, True)
        return AttributeManager(self._reader.attributes(self._msgs), False)

    # ---- writing
    def _wnode(self):
        if self._node is None:
            raise H5Error("file is open read-only")
        return self._node

    def create_group(self, name):
        w = self.file._writer
        base = self.name if not str(name).startswith("/") else ""
        self._wnode()
        n = w.node(base + "/" + name, create_groups=True)
        return Group(self.file, base.rstrip("/") + "/" + name.strip("/"), node=n)

    def require_group(self, name):
        return self.create_group(name)

    def create_dataset(self, name, shape=None, dtype=None, data=None, **_ignored):
        self._wnode()
        if data is None:
            data = np.zeros(shape if shape is not None else (), dtype=dtype or np.float64)
        data = np.asarray(data)
        if dtype is not None:
            data = data.astype(dtype)
        if data.dtype.kind == "O":
            raise H5Error("object arrays cannot be stored")
        base = self.name if not str(name).startswith("/") else ""
        full = base.rstrip("/") + "/" + str(name).strip("/")
        parent, leaf = full.rsplit("/", 1)
        w = self.file._writer
        pn = w.node(parent, create_groups=True)
        if leaf in pn.children:
            raise H5Error("%r exists already" % full)
        n = _Node(leaf, pn)
        n.data = np.array(data, copy=True)
        pn.children[leaf] = n
        return Dataset(self.file, full, node=n)


class File(Group):
    """``File(path, "r")`` reads, ``File(path, "w")`` builds the file in memory and writes it on close, ``File(path, "a")`` does the
    same starting from what the file already holds."""

    def __init__(self, path, mode="r", **_ignored):
        self.filename, self.mode = str(path), mode
        self._writer = None
        self._open = False          # armed only when construction has succeeded: close() / __del__ of a half-built object write nothing
        if mode == "r":
            try:
                rd = _Reader(self.filename)
            except OSError as e:
                raise OSError("unable to open %s: %s" % (self.filename, e)) from e
            Group.__init__(self, self, "/", rd, rd.messages(rd.root))
        elif mode in ("w", "w-", "x", "a", "r+"):
            import os
            self._writer = _Writer(self.filename)
            Group.__init__(self, self, "/", node=self._writer.root)
            if mode in ("a", "r+") and os.path.exists(self.filename):
                # APPEND: what the file holds is read into the writer's tree (groups, datasets, attributes) and goes out again,
                # together with what is added, when the file is closed - the writer builds files whole.  The writer is armed
                # only after ALL of it has been adopted: if the reader cannot represent something (a dtype or layout
                # of a file written by libhdf5 / dolfin, a truncated file) the open fails and the file stays byte-identical.
                try:
                    with File(self.filename, "r") as old:
                        self._adopt(old, self)
                except BaseException:
                    self._writer = None
                    self._node = None
                    raise
            elif mode == "r+":
                raise OSError("unable to open %s: no such file" % self.filename)
        else:
            raise H5Error("h5lite opens files with mode 'r', 'w' or 'a' (got %r)" % (mode,))
        self._open = True

    @staticmethod
    def _adopt(src, dst):
        for k, v in src.attrs.items():
            dst.attrs[k] = v
        for k in src.keys():
            c = src._child(k)
            if c is None:
                raise H5Error("cannot append to %s: link %r of %s cannot be read, and rewriting the file would drop it"
                              % (src.file.filename, k, src.name))
            if isinstance(c, Dataset):
                d = dst.create_dataset(k, data=np.array(c))
                for ak, av in c.attrs.items():
                    d.attrs[ak] = av
            else:
                File._adopt(c, dst.create_group(k))

    def _root(self):
        return self

    def close(self):
        if self._open and self._writer is not None:
            self._writer.write()
        self._open = False

    def flush(self):
        if self._writer is not None:
            self._writer.write()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    def visit_datasets(self):
        """[(path, Dataset)] of every dataset, depth first (h5ls -r)."""
        out = []

        def rec(g):
            for k in sorted(g.keys()):
                c = g._child(k)
                if isinstance(c, Dataset):
                    out.append((c.name, c))
                elif c is not None:
                    rec(c)
        rec(self)
        return out
