"""Dataset readers - host-side I/O in front of the accelerated path, no third-party packages.

* ``BagReader``: rosbag v2.0 files, the input of the reference's offline harness (offline_odometry.cpp:64-68: topics
  /Navtech/Polar (sensor_msgs/Image, mono8) and /gt (nav_msgs/Odometry)). Chunks may be uncompressed, bz2 or lz4 (LZ4 frames decoded
  by a block decoder written here: the image carries no lz4 module).
* ``read_oxford_png``: one sweep of the Oxford Radar RobotCar dataset in its native PNG layout (one row per azimuth:
  8 bytes timestamp, 2 bytes encoder count, 1 byte valid flag, then the power readings) - the format the reference's
  README lists as future work.
* ``BagWriter`` / ``write_png_gray8``: minimal writers of the same formats, used to build test fixtures.

``polar_image`` puts a message into the rows-=-azimuth layout the filter expects, as radarDriver does
(radar_driver.cpp:74-111: Oxford bags are already azimuth-major, other datasets are rotated).
"""
import bz2
import struct
import zlib

import numpy as np

# ---------------------------------------------------------------------------------------------------------------
# PNG (8-bit grayscale, non-interlaced)
# ---------------------------------------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _png_chunks(data):
    if data[:8] != _PNG_SIG:
        raise ValueError("not a PNG file")
    off = 8
    while off < len(data):
        n, typ = struct.unpack(">I4s", data[off:off + 8])
        yield typ, data[off + 8:off + 8 + n]
        off += 12 + n


def read_png_gray8(path_or_bytes):
    """-> uint8 [H, W]"""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    idat, hdr = [], None
    for typ, body in _png_chunks(data):
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
    if hdr is None:
        raise ValueError("PNG without IHDR")
    W, H, depth, ctype, comp, flt, interlace = hdr
    if depth != 8 or ctype != 0 or interlace != 0:
        raise ValueError("only 8-bit grayscale non-interlaced PNGs are supported (got depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(H, W + 1)
    out = np.zeros((H, W), dtype=np.uint8)
    prev = np.zeros(W, dtype=np.int32)
    for y in range(H):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:  # Up
            cur = (line + prev) & 255
        elif ft == 1:  # Sub: running sum modulo 256
            cur = np.cumsum(line) & 255
        else:  # Average / Paeth depend on the pixel to the left: sequential
            cur = np.zeros(W, dtype=np.int32)
            left = upleft = 0
            for x in range(W):
                up = int(prev[x])
                if ft == 3:
                    pred = (left + up) >> 1
                elif ft == 4:
                    p = left + up - upleft
                    pa, pb, pc = abs(p - left), abs(p - up), abs(p - upleft)
                    pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else upleft)
                else:
                    raise ValueError("bad PNG filter type %d" % ft)
                left = (int(line[x]) + pred) & 255
                cur[x] = left
                upleft = up
        out[y] = cur
        prev = cur
    return out


def write_png_gray8(path, img, filter_type=0):
    """minimal writer (fixtures): filter_type 0 (None), 1 (Sub) or 2 (Up) for every row"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    a = img.astype(np.int32)
    if filter_type == 1:
        a = np.concatenate([a[:, :1], np.diff(a, axis=1)], axis=1) & 255
    elif filter_type == 2:
        a = np.concatenate([a[:1], np.diff(a, axis=0)], axis=0) & 255
    elif filter_type != 0:
        raise ValueError("writer supports filter types 0..2")
    raw = np.concatenate([np.full((H, 1), filter_type, dtype=np.uint8), a.astype(np.uint8)], axis=1).tobytes()

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


OXFORD_META_BYTES = 11  # 8 timestamp + 2 azimuth encoder + 1 valid


def read_oxford_png(path_or_bytes):
    """-> dict(polar uint8 [A, R] rows = azimuth, timestamps int64 [A] (us), azimuths float64 [A] (rad), valid bool [A])"""
    img = read_png_gray8(path_or_bytes)
    if img.shape[1] <= OXFORD_META_BYTES:
        raise ValueError("image too narrow for the Oxford radar layout")
    meta = np.ascontiguousarray(img[:, :OXFORD_META_BYTES])
    ts = meta[:, :8].copy().view("<i8")[:, 0]
    enc = meta[:, 8:10].copy().view("<u2")[:, 0]
    return {"polar": np.ascontiguousarray(img[:, OXFORD_META_BYTES:]), "timestamps": ts,
            "azimuths": enc.astype(np.float64) / 5600.0 * 2.0 * np.pi,  # Navtech CTS350-X: 5600 encoder counts per revolution
            "valid": meta[:, 10] == 255}


def oxford_png_rows(polar, timestamps=None, encoder=None, valid=None):
    """inverse of read_oxford_png's split (fixtures): -> uint8 [A, 11 + R]"""
    polar = np.ascontiguousarray(polar, dtype=np.uint8)
    A = polar.shape[0]
    ts = np.arange(A, dtype="<i8") if timestamps is None else np.asarray(timestamps, dtype="<i8")
    enc = (np.arange(A) * 14).astype("<u2") if encoder is None else np.asarray(encoder, dtype="<u2")
    va = np.full(A, 255, dtype=np.uint8) if valid is None else np.where(valid, 255, 0).astype(np.uint8)
    meta = np.concatenate([ts.view(np.uint8).reshape(A, 8), enc.view(np.uint8).reshape(A, 2), va.reshape(A, 1)], axis=1)
    return np.concatenate([meta, polar], axis=1)


# ---------------------------------------------------------------------------------------------------------------
# rosbag v2.0
# ---------------------------------------------------------------------------------------------------------------
# LZ4 frame format (rosbag "lz4" chunks are roslz4 streams = LZ4 frames, magic 0x184D2204), standard library only.
# ---------------------------------------------------------------------------------------------------------------
_LZ4_MAGIC = 0x184D2204


def lz4_block_decompress(src, max_out=None, history=b""):
    """one LZ4 block (sequences of literals + matches) -> bytes. history: the (up to 64 KB of) output in front of this block that
    its matches may reach into (frames with linked blocks)"""
    out = bytearray(history)
    base = len(out)
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = src[i]; i += 1
                ll += b
                if b != 255:
                    break
        if i + ll > n:
            raise ValueError("lz4: literal run past the end of the block")
        out += src[i:i + ll]; i += ll
        if i >= n:
            break  # the last sequence has literals only
        off = src[i] | (src[i + 1] << 8); i += 2
        if off == 0 or off > len(out):
            raise ValueError("lz4: bad match offset")
        ml = (tok & 15) + 4
        if (tok & 15) == 15:
            while True:
                b = src[i]; i += 1
                ml += b
                if b != 255:
                    break
        start = len(out) - off
        if off >= ml:
            out += out[start:start + ml]
        else:  # overlapping match: the pattern of `off` bytes repeats
            pat = bytes(out[start:])
            out += (pat * (ml // off + 1))[:ml]
        if max_out is not None and len(out) - base > max_out:
            raise ValueError("lz4: block larger than announced")
    return bytes(out[base:])


def lz4_frame_decompress(data):
    """LZ4 frame (possibly several, concatenated) -> bytes. Checksums are skipped, not verified."""
    out, i, n = [], 0, len(data)
    while i + 4 <= n:
        magic, = struct.unpack_from("<I", data, i); i += 4
        if 0x184D2A50 <= magic <= 0x184D2A5F:  # skippable frame
            sz, = struct.unpack_from("<I", data, i); i += 4 + sz
            continue
        if magic != _LZ4_MAGIC:
            raise ValueError("not an LZ4 frame (magic %08x)" % magic)
        flg, bd = data[i], data[i + 1]; i += 2
        if (flg >> 6) != 1:
            raise ValueError("lz4 frame version %d" % (flg >> 6))
        block_checksum, content_size, content_checksum, dict_id = (flg >> 4) & 1, (flg >> 3) & 1, (flg >> 2) & 1, flg & 1
        block_max = {4: 1 << 16, 5: 1 << 18, 6: 1 << 20, 7: 1 << 22}.get((bd >> 4) & 7)
        if block_max is None:
            raise ValueError("lz4 frame: bad block size code")
        i += 8 * content_size + 4 * dict_id + 1  # + header checksum byte
        linked = not (flg >> 5) & 1  # (roslz4 writes independent blocks; the lz4 command line tool links them by default)
        hist = b""
        while True:
            sz, = struct.unpack_from("<I", data, i); i += 4
            if sz == 0:
                break
            raw = sz >> 31
            sz &= 0x7FFFFFFF
            blk = data[i:i + sz]; i += sz + 4 * block_checksum
            out.append(bytes(blk) if raw else lz4_block_decompress(blk, block_max, hist))
            if linked:
                hist = (hist + out[-1])[-65536:]
        i += 4 * content_checksum
    return b"".join(out)


def lz4_block_compress(src):
    """greedy hash-chain-free LZ4 block compressor (fixtures: produces literal runs, matches and overlapping matches)"""
    n, out, anchor, i, table = len(src), bytearray(), 0, 0, {}

    def emit(lit_end, mlen, off):
        ll = lit_end - anchor
        tok_l = 15 if ll >= 15 else ll
        tok_m = 0 if mlen == 0 else (15 if mlen - 4 >= 15 else mlen - 4)
        out.append((tok_l << 4) | tok_m)
        if ll >= 15:
            r = ll - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(src[anchor:lit_end])
        if mlen:
            out.append(off & 255); out.append(off >> 8)
            if mlen - 4 >= 15:
                r = mlen - 4 - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)
    while i + 12 < n:  # the last 12 bytes are literals (end-of-block rules)
        key = bytes(src[i:i + 4])
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and src[cand + m] == src[i + m]:
                m += 1
            emit(i, m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(n, 0, 0)
    return bytes(out)


def lz4_frame_compress(data, block_size=1 << 22):
    """LZ4 frame with independent 4 MB blocks, no checksums (the header checksum byte is not verified by readers here;
    0 is written) -- fixtures only"""
    out = [struct.pack("<IBBB", _LZ4_MAGIC, 0x60, 0x70, 0)]
    for a in range(0, len(data), block_size):
        raw = data[a:a + block_size]
        c = lz4_block_compress(raw)
        out.append(struct.pack("<I", len(c)) + c if len(c) < len(raw) else struct.pack("<I", len(raw) | 0x80000000) + bytes(raw))
    out.append(struct.pack("<I", 0))
    return b"".join(out)


# ---------------------------------------------------------------------------------------------------------------
_BAG_MAGIC = b"#ROSBAG V2.0\n"
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 2, 3, 4, 5, 6, 7


def _parse_header(buf):
    out, off = {}, 0
    while off < len(buf):
        n, = struct.unpack_from("<I", buf, off)
        field = buf[off + 4:off + 4 + n]
        k, _, v = field.partition(b"=")
        out[k.decode()] = v
        off += 4 + n
    return out


def _records(buf, off=0, end=None):
    end = len(buf) if end is None else end
    while off + 8 <= end:
        hl, = struct.unpack_from("<I", buf, off)
        hdr = _parse_header(buf[off + 4:off + 4 + hl])
        dl, = struct.unpack_from("<I", buf, off + 4 + hl)
        d0 = off + 8 + hl
        yield hdr, buf[d0:d0 + dl]
        off = d0 + dl


class _Cur:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u8(self):
        v = self.b[self.o]; self.o += 1; return v

    def u32(self):
        v, = struct.unpack_from("<I", self.b, self.o); self.o += 4; return v

    def f64(self, n=1):
        v = struct.unpack_from("<%dd" % n, self.b, self.o); self.o += 8 * n; return v if n > 1 else v[0]

    def string(self):
        n = self.u32(); v = self.b[self.o:self.o + n]; self.o += n; return v.decode(errors="replace")

    def raw(self, n):
        v = self.b[self.o:self.o + n]; self.o += n; return v


def _std_header(c):
    seq, sec, nsec = c.u32(), c.u32(), c.u32()
    return {"seq": seq, "stamp": sec * 1000000000 + nsec, "frame_id": c.string()}


def decode_image(data):
    """sensor_msgs/Image -> dict(header, height, width, encoding, step, data uint8 [height, step])"""
    c = _Cur(data)
    h = _std_header(c)
    height, width = c.u32(), c.u32()
    enc = c.string()
    big = c.u8()
    step = c.u32()
    n = c.u32()
    arr = np.frombuffer(c.raw(n), dtype=np.uint8)
    return {"header": h, "height": height, "width": width, "encoding": enc, "is_bigendian": big, "step": step,
            "data": arr.reshape(height, step) if height * step == n else arr}


def decode_odometry(data):
    """nav_msgs/Odometry -> dict(header, child_frame_id, position (3), orientation xyzw (4), pose_cov 6x6, twist (6), twist_cov)"""
    c = _Cur(data)
    h = _std_header(c)
    child = c.string()
    pos = np.array(c.f64(3)); quat = np.array(c.f64(4)); pcov = np.array(c.f64(36)).reshape(6, 6)
    tw = np.array(c.f64(6)); tcov = np.array(c.f64(36)).reshape(6, 6)
    return {"header": h, "child_frame_id": child, "position": pos, "orientation": quat, "pose_cov": pcov, "twist": tw, "twist_cov": tcov}


def odometry_to_xyt(msg):
    """planar pose as the reference reduces it (offline_odometry.cpp:83-90): yaw of the quaternion, x, y"""
    x, y, z, w = msg["orientation"]
    yaw = np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return np.array([msg["position"][0], msg["position"][1], yaw])


class BagReader:
    """Iterates (topic, datatype, time_ns, raw message bytes) in file order; ``topics`` filters like rosbag::TopicQuery."""

    def __init__(self, path):
        import mmap
        self._fh = open(path, "rb")
        self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)  # recordings are tens of GB: map, do not read
        if self.buf[:len(_BAG_MAGIC)] != _BAG_MAGIC:
            raise ValueError("not a rosbag v2.0 file")
        self.connections = {}

    def _conn(self, hdr, data):
        cid, = struct.unpack("<I", hdr["conn"])
        info = _parse_header(data)
        self.connections[cid] = {"topic": hdr["topic"].decode(), "type": info.get("type", b"").decode(), "md5sum": info.get("md5sum", b"").decode()}

    def messages(self, topics=None):
        topics = None if topics is None else set(topics)
        for hdr, data in _records(self.buf, len(_BAG_MAGIC)):
            op = hdr["op"][0]
            if op == OP_CONNECTION:
                self._conn(hdr, data)
            elif op == OP_CHUNK:
                comp = hdr["compression"].decode()
                if comp == "none":
                    body = data
                elif comp == "bz2":
                    body = bz2.decompress(data)
                elif comp == "lz4":
                    body = lz4_frame_decompress(data)
                    want, = struct.unpack("<I", hdr["size"])
                    if len(body) != want:
                        raise ValueError("lz4 chunk: %d bytes after decompression, header says %d" % (len(body), want))
                else:
                    raise NotImplementedError("rosbag chunk compression '%s' (none, bz2 and lz4 are supported)" % comp)
                for h2, d2 in _records(body):
                    op2 = h2["op"][0]
                    if op2 == OP_CONNECTION:
                        self._conn(h2, d2)
                    elif op2 == OP_MSG:
                        cid, = struct.unpack("<I", h2["conn"])
                        sec, nsec = struct.unpack("<II", h2["time"])
                        con = self.connections[cid]
                        if topics is None or con["topic"] in topics:
                            yield con["topic"], con["type"], sec * 1000000000 + nsec, d2

    def sweeps_and_gt(self, image_topic="/Navtech/Polar", gt_topic="/gt"):
        """the reference's replay loop (offline_odometry.cpp:67-127): yields ("gt", t, xyt) and ("image", t, decoded image)"""
        for topic, typ, t, data in self.messages([image_topic, gt_topic]):
            if topic == gt_topic:
                yield "gt", t, odometry_to_xyt(decode_odometry(data))
            else:
                yield "image", t, decode_image(data)


def polar_image(msg, dataset="oxford"):
    """radarDriver::Callback* (radar_driver.cpp:74-111): rows = azimuth, cols = range. Oxford bags already have that layout;
    the other datasets are stored range-major and rotated by 90 degrees counter-clockwise (:84)."""
    if msg["encoding"] not in ("mono8", "8UC1"):
        raise ValueError("expected a mono8 radar image, got '%s'" % msg["encoding"])
    img = msg["data"][:, :msg["width"]]
    if dataset == "oxford":
        return np.ascontiguousarray(img)
    return np.ascontiguousarray(np.rot90(img, 1))  # cv::ROTATE_90_COUNTERCLOCKWISE


class BagWriter:
    """minimal rosbag v2.0 writer (fixtures): a chunk is closed when it passes chunk_threshold bytes (rosbag's default is
    768 KiB) or at flush(); compression none, bz2 or lz4; no index records."""

    def __init__(self, path, compression="none", chunk_threshold=768 * 1024):
        self.fh = open(path, "wb")
        self.compression = compression
        self.chunk_threshold, self.chunk_bytes = chunk_threshold, 0
        self.conns, self.chunk, self.nchunks = {}, [], 0
        self.fh.write(_BAG_MAGIC)
        hdr = self._header({"op": bytes([OP_BAG_HEADER]), "index_pos": struct.pack("<Q", 0), "conn_count": struct.pack("<I", 0), "chunk_count": struct.pack("<I", 0)})
        pad = 4096 - 4 - len(hdr) - 4
        self.fh.write(struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", pad) + b" " * pad)

    @staticmethod
    def _header(fields):
        out = b""
        for k, v in fields.items():
            f = k.encode() + b"=" + v
            out += struct.pack("<I", len(f)) + f
        return out

    def _record(self, fields, data):
        h = self._header(fields)
        return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data

    def write(self, topic, datatype, time_ns, data, md5sum="*"):
        if topic not in self.conns:
            cid = len(self.conns)
            self.conns[topic] = cid
            info = self._header({"topic": topic.encode(), "type": datatype.encode(), "md5sum": md5sum.encode(), "message_definition": b""})
            self.chunk.append(self._record({"op": bytes([OP_CONNECTION]), "conn": struct.pack("<I", cid), "topic": topic.encode()}, info))
        sec, nsec = divmod(int(time_ns), 1000000000)
        self.chunk.append(self._record({"op": bytes([OP_MSG]), "conn": struct.pack("<I", self.conns[topic]), "time": struct.pack("<II", sec, nsec)}, data))
        self.chunk_bytes += len(self.chunk[-1])
        if self.chunk_bytes >= self.chunk_threshold:
            self.flush()

    def flush(self):
        if not self.chunk:
            return
        body = b"".join(self.chunk)
        comp = bz2.compress(body) if self.compression == "bz2" else (lz4_frame_compress(body) if self.compression == "lz4" else body)
        self.chunk_bytes = 0
        self.fh.write(self._record({"op": bytes([OP_CHUNK]), "compression": self.compression.encode(), "size": struct.pack("<I", len(body))}, comp))
        self.chunk, self.nchunks = [], self.nchunks + 1

    def close(self):
        self.flush()
        self.fh.close()


def encode_image(img, stamp_ns, seq=0, frame_id="navtech", encoding="mono8"):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    sec, nsec = divmod(int(stamp_ns), 1000000000)
    fid, enc = frame_id.encode(), encoding.encode()
    return (struct.pack("<III", seq, sec, nsec) + struct.pack("<I", len(fid)) + fid + struct.pack("<II", img.shape[0], img.shape[1]) +
            struct.pack("<I", len(enc)) + enc + struct.pack("<BI", 0, img.shape[1]) + struct.pack("<I", img.size) + img.tobytes())


def encode_odometry(xyt, stamp_ns, seq=0, frame_id="world", child="navtech"):
    sec, nsec = divmod(int(stamp_ns), 1000000000)
    fid, ch = frame_id.encode(), child.encode()
    half = 0.5 * float(xyt[2])
    return (struct.pack("<III", seq, sec, nsec) + struct.pack("<I", len(fid)) + fid + struct.pack("<I", len(ch)) + ch +
            struct.pack("<7d", float(xyt[0]), float(xyt[1]), 0.0, 0.0, 0.0, np.sin(half), np.cos(half)) + struct.pack("<36d", *([0.0] * 36)) +
            struct.pack("<6d", *([0.0] * 6)) + struct.pack("<36d", *([0.0] * 36)))
