"""Dataset readers: rosbag v2.0 and Oxford radar PNG, against fixtures written on the fly."""
import struct
import zlib

import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import readers, synth


@pytest.mark.parametrize("ft", [0, 1, 2])
def test_png_roundtrip(tmp_path, ft):
    rng = np.random.default_rng(ft)
    img = rng.integers(0, 256, size=(37, 211), dtype=np.uint8)
    readers.write_png_gray8(tmp_path / "a.png", img, filter_type=ft)
    assert np.array_equal(readers.read_png_gray8(tmp_path / "a.png"), img)


def test_png_average_and_paeth_filters(tmp_path):
    """rows filtered with types 3 and 4 (written here by the definition of the filters, PNG spec section 9)"""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, size=(6, 50), dtype=np.uint8)
    rows = []
    for y in range(6):
        ft = 3 if y % 2 == 0 else 4
        line = bytearray([ft])
        for x in range(50):
            a = int(img[y, x - 1]) if x > 0 else 0
            b = int(img[y - 1, x]) if y > 0 else 0
            c = int(img[y - 1, x - 1]) if (x > 0 and y > 0) else 0
            if ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line.append((int(img[y, x]) - pred) & 255)
        rows.append(bytes(line))

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    data = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 50, 6, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b""))
    assert np.array_equal(readers.read_png_gray8(data), img)


def test_oxford_png_layout(tmp_path):
    polar = synth.world_scan(synth.World(3), 0, 400, 3768, np.float32(0.0438), seed=1)
    ts = 1547131046353776 + np.arange(400) * 625
    enc = (np.arange(400) * 14) % 5600
    rows = readers.oxford_png_rows(polar, ts, enc, valid=np.arange(400) % 50 != 7)
    assert rows.shape == (400, 3779)
    readers.write_png_gray8(tmp_path / "1547131046353776.png", rows, filter_type=2)
    got = readers.read_oxford_png(tmp_path / "1547131046353776.png")
    assert np.array_equal(got["polar"], polar) and got["polar"].shape == (400, 3768)
    assert np.array_equal(got["timestamps"], ts)
    assert np.allclose(got["azimuths"], enc / 5600.0 * 2 * np.pi)
    assert got["valid"].sum() == 392


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_bag_roundtrip_and_replay_order(tmp_path, compression):
    imgs, gt = synth.world_sequence(3, A=40, R=64, seed=2)
    w = readers.BagWriter(tmp_path / "radar.bag", compression=compression)
    t0 = 1547131046000000000
    for i in range(3):
        w.write("/gt", "nav_msgs/Odometry", t0 + i * 250000000, readers.encode_odometry(gt[i], t0 + i * 250000000, seq=i))
        w.write("/Navtech/Polar", "sensor_msgs/Image", t0 + i * 250000000 + 1000, readers.encode_image(imgs[i], t0 + i * 250000000 + 1000, seq=i))
        w.write("/other", "std_msgs/String", t0 + i, struct.pack("<I", 2) + b"hi")
        if i == 1:
            w.flush()  # two chunks
    w.close()
    bag = readers.BagReader(tmp_path / "radar.bag")
    events = list(bag.sweeps_and_gt())
    assert [e[0] for e in events] == ["gt", "image"] * 3  # /other filtered out like rosbag::TopicQuery (offline_odometry.cpp:67-68)
    for i in range(3):
        kind, t, xyt = events[2 * i]
        assert t == t0 + i * 250000000 and np.allclose(xyt, gt[i], atol=1e-12)
        kind, t, msg = events[2 * i + 1]
        assert msg["encoding"] == "mono8" and msg["height"] == 40 and msg["width"] == 64 and msg["header"]["seq"] == i
        assert np.array_equal(readers.polar_image(msg, "oxford"), imgs[i])
        assert np.array_equal(readers.polar_image(msg, "mulran"), np.rot90(imgs[i], 1))
    assert bag.connections[0]["type"] in ("nav_msgs/Odometry", "sensor_msgs/Image")
    assert len(list(bag.messages())) == 9


def test_bag_rejects_other_files_and_unknown_compression(tmp_path):
    (tmp_path / "x.bag").write_bytes(b"not a bag")
    with pytest.raises(ValueError):
        readers.BagReader(tmp_path / "x.bag")
    w = readers.BagWriter(tmp_path / "l.bag", compression="none")
    w.write("/gt", "nav_msgs/Odometry", 1, readers.encode_odometry([0, 0, 0], 1))
    w.close()
    data = (tmp_path / "l.bag").read_bytes().replace(b"compression=none", b"compression=zstd")
    (tmp_path / "l.bag").write_bytes(data)
    with pytest.raises(NotImplementedError):
        list(readers.BagReader(tmp_path / "l.bag").messages())


def test_lz4_frame_known_answers_and_roundtrip():
    """LZ4 frames as liblz4 / roslz4 write them. Known answers assembled by hand from the format specification (frame:
    magic 0x184D2204, FLG, BD, header checksum, blocks, end mark; block: token, literals, 16-bit offset, match length)."""
    # uncompressed block inside a frame (high bit of the block size)
    f = struct.pack("<IBBB", 0x184D2204, 0x60, 0x40, 0x82) + struct.pack("<I", 5 | 0x80000000) + b"hello" + struct.pack("<I", 0)
    assert readers.lz4_frame_decompress(f) == b"hello"
    # one compressed block: 1 literal 'a', match offset 1 length 15 + 4 + 2 = 21 (overlapping: run of 'a'), then 5 literals
    blk = bytes([0x1F, ord("a"), 0x01, 0x00, 0x02, 0x50]) + b"vwxyz"
    assert readers.lz4_block_decompress(blk) == b"a" * 22 + b"vwxyz"
    # content size + content checksum + block checksum flags are skipped over correctly
    f2 = (struct.pack("<IBB", 0x184D2204, 0x60 | 0x10 | 0x08 | 0x04, 0x40) + struct.pack("<Q", 27) + b"\x00" +
          struct.pack("<I", len(blk)) + blk + b"\xAA\xBB\xCC\xDD" + struct.pack("<I", 0) + b"\x11\x22\x33\x44")
    assert readers.lz4_frame_decompress(f2) == b"a" * 22 + b"vwxyz"
    # two frames back to back and a skippable frame in between
    skip = struct.pack("<II", 0x184D2A50, 3) + b"xyz"
    assert readers.lz4_frame_decompress(f + skip + f2) == b"hello" + b"a" * 22 + b"vwxyz"
    # literal length 15 + 255 + 3 = 273 with length-extension bytes
    lit = bytes(range(256)) + bytes(range(17))
    blk2 = bytes([0xF0, 255, 3]) + lit
    assert readers.lz4_block_decompress(blk2) == lit
    with pytest.raises(ValueError):
        readers.lz4_block_decompress(bytes([0x10, ord("a"), 0x05, 0x00]))  # offset beyond what has been written
    # the fixture compressor against the decoder on radar-like and repetitive data
    rng = np.random.default_rng(0)
    for data in (b"", b"abc", bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), b"ab" * 40000 + bytes(rng.integers(0, 4, 5000, dtype=np.uint8)),
                 synth.world_scan(synth.World(3), 2, 40, 512).tobytes()):
        assert readers.lz4_frame_decompress(readers.lz4_frame_compress(data, block_size=1 << 16)) == data
    assert len(readers.lz4_frame_compress(b"ab" * 40000)) < 2000


@pytest.mark.parametrize("compress_level,optimize", [(0, False), (1, False), (6, False), (9, True)])
def test_png_written_by_pillow_oxford_layout(tmp_path, compress_level, optimize):
    """An encoder that is not ours: Pillow (libpng-style adaptive per-row filters - all five types occur -, several IDAT chunks,
    different zlib levels incl. stored blocks) writes the 400 x 3779 Oxford radar layout (README:133; 11 metadata columns in front
    of the 3768 range bins); read_png_gray8 / read_oxford_png must give back exactly what PIL.Image.open decodes."""
    PIL = pytest.importorskip("PIL.Image")
    polar = synth.world_scan(synth.World(5), 1, 400, 3768, np.float32(0.0438), seed=2)
    ts = 1547131046353776 + np.arange(400) * 625
    rows = readers.oxford_png_rows(polar, ts, (np.arange(400) * 14) % 5600)
    f = tmp_path / "pil.png"
    PIL.fromarray(rows, mode="L").save(f, format="PNG", compress_level=compress_level, optimize=optimize)
    ref = np.array(PIL.open(f))
    assert ref.shape == (400, 3779) and np.array_equal(ref, rows)
    got = readers.read_png_gray8(f)
    assert np.array_equal(got, ref)
    ox = readers.read_oxford_png(f)
    assert np.array_equal(ox["polar"], polar) and np.array_equal(ox["timestamps"], ts)
    # which row filters and how many IDAT chunks the file really has (the test means something only if they vary)
    data = f.read_bytes()
    idat = [body for t, body in readers._png_chunks(data) if t == b"IDAT"]
    raw = zlib.decompress(b"".join(idat))
    filters = {raw[y * 3780] for y in range(400)}
    assert filters <= {0, 1, 2, 3, 4}
    if compress_level in (6, 9):
        assert len(filters) >= 2  # adaptive filtering chose more than one type on a radar image


def test_png_smooth_image_by_pillow_uses_every_filter_type(tmp_path):
    """a smooth gradient + noise image makes libpng-style heuristics pick Sub / Up / Average / Paeth rows; ours decodes them all"""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    y, x = np.mgrid[0:300, 0:500]
    img = ((np.sin(x / 17.0) * 60 + np.cos(y / 9.0) * 50 + 128) + rng.normal(0, 1.5, (300, 500))).clip(0, 255).astype(np.uint8)
    img[::7] = rng.integers(0, 256, (len(img[::7]), 500), dtype=np.uint8)  # noisy rows: filter None
    f = tmp_path / "g.png"
    PIL.fromarray(img, mode="L").save(f, format="PNG", compress_level=6)
    raw = zlib.decompress(b"".join(body for t, body in readers._png_chunks(f.read_bytes()) if t == b"IDAT"))
    assert len({raw[r * 501] for r in range(300)}) >= 4
    assert np.array_equal(readers.read_png_gray8(f), np.array(PIL.open(f)))


def test_png_rejects_what_the_oxford_sdk_never_writes(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    f = tmp_path / "rgb.png"
    PIL.fromarray(np.zeros((4, 4, 3), dtype=np.uint8), mode="RGB").save(f)
    with pytest.raises(ValueError):
        readers.read_png_gray8(f)
    f16 = tmp_path / "g16.png"
    PIL.fromarray(np.zeros((4, 4), dtype=np.uint16)).save(f16)
    with pytest.raises(ValueError):
        readers.read_png_gray8(f16)


def test_lz4_frame_published_format_vectors():
    """Frames reconstructed by hand from the LZ4 frame format description (v1.6.x): an empty frame (header + end mark only), with
    and without a content checksum (xxh32 of nothing = 0x02CC5D05), an uncompressed-block frame, a frame with block-independence
    off and 64 KB maximum block size; `xxhash` (in the image) checks the header checksum byte our decoder skips over."""
    xxhash = pytest.importorskip("xxhash")

    def header(flg, bd, extra=b""):
        desc = bytes([flg, bd]) + extra
        return struct.pack("<I", 0x184D2204) + desc + bytes([(xxhash.xxh32(desc, seed=0).intdigest() >> 8) & 0xFF])
    end = struct.pack("<I", 0)
    assert readers.lz4_frame_decompress(header(0x60, 0x40) + end) == b""                                        # version 01, independent blocks
    assert readers.lz4_frame_decompress(header(0x64, 0x40) + end + struct.pack("<I", 0x02CC5D05)) == b""        # + content checksum
    assert header(0x64, 0x40)[-1] == 0xA7  # the byte the specification's own example of this descriptor carries
    payload = bytes(range(200)) * 3
    f = header(0x40, 0x40) + struct.pack("<I", len(payload) | 0x80000000) + payload + end                        # linked blocks flag, stored block
    assert readers.lz4_frame_decompress(f) == payload
    f = header(0x68, 0x40, struct.pack("<Q", len(payload))) + struct.pack("<I", len(payload) | 0x80000000) + payload + end  # content size field
    assert readers.lz4_frame_decompress(f) == payload
    assert readers.lz4_frame_decompress(b"") == b""
    # linked blocks: the second block's only match reaches 8 bytes back into the first block's output
    b1 = bytes([0x80]) + b"ABCDEFGH"                               # 8 literals, end of block
    b2 = bytes([0x04, 0x08, 0x00, 0x50]) + b"vwxyz"                # no literals, match offset 8 length 8, then 5 literals
    f = header(0x40, 0x40) + struct.pack("<I", len(b1)) + b1 + struct.pack("<I", len(b2)) + b2 + end
    assert readers.lz4_frame_decompress(f) == b"ABCDEFGH" + b"ABCDEFGH" + b"vwxyz"
    with pytest.raises(ValueError):  # the same blocks declared independent: the match has nothing to refer to
        readers.lz4_frame_decompress(header(0x60, 0x40) + struct.pack("<I", len(b1)) + b1 + struct.pack("<I", len(b2)) + b2 + end)
