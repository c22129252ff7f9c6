"""Split JPEG decode: the host entropy decoder (libaqjpeg.so) + the oracle's pixel half against Pillow's
libjpeg-turbo on the CPU -- which PINS the oracle to the library [UPSTREAM cv2.imread] uses -- and the HIP pixel half against both on the
GPU.  Byte for byte everywhere."""
import io
import os

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jpeg(img, **kw):
    bio = io.BytesIO()
    Image.fromarray(img).save(bio, format="JPEG", **kw)
    return bio.getvalue()


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def _cases():
    from aquaculture_amd import tiles
    rng = np.random.default_rng(5)
    noise = lambda h, w: rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
    smooth = lambda h, w: np.clip(np.add.outer(np.arange(h) * 0.7, np.arange(w) * 0.4)[..., None] + rng.normal(0, 6, (h, w, 3)) + [20, 70, 110], 0, 255).astype(np.uint8)
    return [
        ("ocean tile 640, q75 (the GDAL default of reference tile_tifs.py:74)", _jpeg(tiles.synthetic_tile(3, 640), quality=75)),
        ("cage tile 640", _jpeg(tiles.synthetic_tile(19, 640), quality=75)),
        ("noise 128x192 q90", _jpeg(noise(128, 192), quality=90)),
        ("noise 100x150 q50 (ragged: partial MCUs on both edges)", _jpeg(noise(100, 150), quality=50)),
        ("noise 33x17 q80", _jpeg(noise(33, 17), quality=80)),
        ("smooth 641x1023", _jpeg(smooth(641, 1023), quality=75)),
        ("restart markers every 3 MCUs", _jpeg(noise(64, 64), quality=75, restart_marker_blocks=3)),
        ("restart markers every row", _jpeg(smooth(96, 160), quality=85, restart_marker_rows=1)),
        ("optimised Huffman tables, q95", _jpeg(noise(48, 80), quality=95, optimize=True)),
        ("one MCU, q10", _jpeg(noise(16, 16), quality=10)),
        ("saturated colours (range limit in every stage)", _jpeg(np.tile(np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]], np.uint8).repeat(8, 0).repeat(8, 1), (4, 4, 1)), quality=100)),
    ]


@pytest.fixture(scope="module")
def jpeg_lib():
    from aquaculture_amd import build, jpeg
    build.build()
    return jpeg


def test_libaqjpeg_exports_what_its_header_declares(jpeg_lib):
    import re
    hdr = open(os.path.join(ROOT, "include", "aq_jpeg.h")).read()
    names = set(re.findall(r"^int\s+(aq_jpeg_\w+)\s*\(", hdr, re.M))                 # (the device-side entry aq_jpeg_huffman_decode lives in aq_engine.h / libaqengine.so)
    assert names == {"aq_jpeg_decode_coeffs", "aq_jpeg_scan", "aq_jpeg_prepare", "aq_jpeg_prepare_files"}
    lib = jpeg_lib.load_lib()
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.parametrize("idx", range(11))
def test_entropy_decoder_plus_oracle_equals_pillow(jpeg_lib, idx):
    from oracle import jpeg_oracle as J
    name, data = _cases()[idx]
    ref = _pil(data)
    H, W = ref.shape[:2]
    coef = np.zeros(jpeg_lib.coef_count(H, W), np.int16)
    qt = np.zeros((3, 64), np.uint16)
    rc, info = jpeg_lib.decode_coeffs(data, coef, qt)
    assert rc == 0 and (info.height, info.width) == (H, W), (name, rc)
    got = J.decode_from_coeffs(coef, qt, W, H, info.mcu_cols, info.mcu_rows)
    assert np.array_equal(got, ref), f"{name}: {int((got != ref).sum())} bytes differ"


def test_files_outside_the_split_decoder_are_refused_not_misread(jpeg_lib):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 255, (32, 48, 3), dtype=np.uint8)
    for kw in (dict(subsampling=0), dict(subsampling=1), dict(progressive=True)):
        assert jpeg_lib.scan(_jpeg(img, **kw)) is None, kw
    grey = _jpeg(img[..., 0])
    assert jpeg_lib.scan(grey) is None                                   # one component: the caller's software path
    good = _jpeg(img)
    assert jpeg_lib.scan(good) is not None
    coef = np.zeros(jpeg_lib.coef_count(32, 48), np.int16)
    qt = np.zeros((3, 64), np.uint16)
    for cut in (len(good) // 2, len(good) - 3, 30):                      # truncated files are CORRUPT (next test: exactly where Pillow raises)
        rc, _ = jpeg_lib.decode_coeffs(good[:cut], coef, qt)
        assert rc == -2, rc
    rc, _ = jpeg_lib.decode_coeffs(good, coef[:100], qt)
    assert rc == -3                                                      # too small a buffer is reported, not overrun
    assert jpeg_lib.decode_coeffs(b"not a jpeg at all", coef, qt)[0] == -2


def test_oversubscribed_huffman_table_is_rejected_before_it_is_built(jpeg_lib):
    """a DHT segment whose counts over-subscribe a code length used to index past look[512] on the stack while the table
    was being built (SOI + DHT with counts[0] = 200 and 200 values; reached through aq_jpeg_scan on file-controlled bytes).  The table is
    now refused before anything is written; the sanitizer build of test_entropy_decoder_survives_mutated_files_under_asan runs the same
    bytes plus DHT-count mutations that keep the segment length consistent."""
    import ctypes as C
    lib = jpeg_lib.load_lib()
    for counts in ([200] + [0] * 15, [2, 5] + [0] * 14, [0] * 8 + [255] + [0] * 7, [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3]):
        n = sum(counts)
        poc = bytes([0xFF, 0xD8, 0xFF, 0xC4]) + (2 + 17 + n).to_bytes(2, "big") + bytes([0x00] + counts) + bytes(i & 255 for i in range(n))
        assert lib.aq_jpeg_scan(poc, len(poc), C.byref(jpeg_lib.JpegInfo())) == -2, counts
    # a FULL but legal table (the all-ones code included at length 16) is still accepted up to the missing frame header
    counts = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    ok = bytes([0xFF, 0xD8, 0xFF, 0xC4]) + (2 + 17 + 12).to_bytes(2, "big") + bytes([0x00] + counts) + bytes(range(12))
    assert lib.aq_jpeg_scan(ok, len(ok), C.byref(jpeg_lib.JpegInfo())) == -2          # (no SOF / SOS: corrupt for another reason, no crash)


@pytest.mark.parametrize("kw", [dict(quality=75), dict(quality=60, restart_marker_blocks=2), dict(quality=90, optimize=True)])
def test_truncated_or_interrupted_scans_fail_exactly_where_pillow_fails(jpeg_lib, kw):
    """after EOF or an unexpected marker the bit reader supplies zero bits; the decoder used to turn those into
    'valid' grey blocks and return OK, so a half-written tile got (missing) labels and a done-manifest entry.  Every prefix of a file now
    gets the status Pillow gives it: OK only for the complete file, EOI included; a marker in the middle of the entropy-coded data too."""
    rng = np.random.default_rng(7)
    img = np.clip(rng.normal(128, 40, (48, 80, 3)), 0, 255).astype(np.uint8)
    data = _jpeg(img, **kw)
    coef = np.zeros(jpeg_lib.coef_count(48, 80), np.int16)
    qt = np.zeros((3, 64), np.uint16)

    def pil_ok(d):
        try:
            Image.open(io.BytesIO(d)).load()
            return True
        except Exception:
            return False

    cuts = list(range(4, 40)) + list(range(len(data) - 300, len(data) + 1))
    for cut in cuts:
        rc, _ = jpeg_lib.decode_coeffs(data[:cut], coef, qt)
        assert (rc == 0) == pil_ok(data[:cut]), (cut, len(data), rc)
    assert jpeg_lib.decode_coeffs(data, coef, qt)[0] == 0
    sos = data.index(b"\xff\xda")
    mid = sos + (len(data) - sos) // 2
    for marker in (b"\xff\xd9", b"\xff\xc4", b"\xff\xd3" if "restart_marker_blocks" not in kw else b"\xff\xd9"):
        broken = data[:mid] + marker + data[mid + 2:]
        assert jpeg_lib.decode_coeffs(broken, coef, qt)[0] == -2, marker


def test_rgb_signalled_files_go_to_the_software_decoder(jpeg_lib):
    """libjpeg / Pillow do NOT colour-convert three-component files whose Adobe APP14 marker says transform 0, or which
    have no JFIF / Adobe marker and component ids 'R','G','B'.  The device half always converts YCbCr -> RGB, so the header scan must
    hand such files to the software path.  (Pillow writes JFIF + ids 1,2,3; the variants are made by editing those bytes.)"""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 255, (32, 48, 3), dtype=np.uint8)
    good = _jpeg(img, quality=80)
    assert jpeg_lib.scan(good) is not None
    app0 = good.index(b"\xff\xe0")
    app0_len = int.from_bytes(good[app0 + 2:app0 + 4], "big")
    no_jfif = good[:app0] + good[app0 + 2 + app0_len:]
    assert jpeg_lib.scan(no_jfif) is not None                            # no marker, ids 1,2,3: YCbCr by libjpeg's rule
    adobe = lambda t: b"\xff\xee" + (14).to_bytes(2, "big") + b"Adobe" + bytes([0, 100, 0, 0, 0, 0, t])
    assert jpeg_lib.scan(no_jfif[:2] + adobe(0) + no_jfif[2:]) is None   # Adobe transform 0: stored RGB
    assert jpeg_lib.scan(no_jfif[:2] + adobe(1) + no_jfif[2:]) is not None
    assert jpeg_lib.scan(good[:2] + adobe(0) + good[2:]) is not None     # JFIF wins over Adobe (jdapimin.c)
    sof = no_jfif.index(b"\xff\xc0")
    sos = no_jfif.index(b"\xff\xda")
    rgb_ids = bytearray(no_jfif)
    for k, ch in enumerate(b"RGB"):
        rgb_ids[sof + 4 + 6 + 3 * k] = ch
        rgb_ids[sos + 4 + 1 + 2 * k] = ch
    assert jpeg_lib.scan(bytes(rgb_ids)) is None
    ref = np.asarray(Image.open(io.BytesIO(bytes(rgb_ids))).convert("RGB"))
    assert not np.array_equal(ref, _pil(good))                            # and Pillow indeed decodes it differently


@pytest.mark.gpu
def test_device_pixel_half_equals_pillow(jpeg_lib, lib):
    """aq_jpeg_idct_rgb on batches of same-size images: byte-identical to Pillow (and so to the oracle)."""
    import torch
    from aquaculture_amd import engine, tiles
    rng = np.random.default_rng(11)
    groups = {
        (640, 640): [_jpeg(tiles.synthetic_tile(i, 640), quality=75) for i in (0, 3, 19, 40)],
        (1024, 1024): [_jpeg(tiles.synthetic_tile(i, 1024), quality=75) for i in (1, 19)],
        (100, 150): [_jpeg(rng.integers(0, 255, (100, 150, 3), dtype=np.uint8), quality=q) for q in (30, 75, 98)],
        (33, 17): [_jpeg(rng.integers(0, 255, (33, 17, 3), dtype=np.uint8), quality=80)],
        (64, 64): [_jpeg(np.tile(np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]], np.uint8).repeat(8, 0).repeat(8, 1), (4, 4, 1)), quality=100)],
    }
    for (H, W), files in groups.items():
        n = jpeg_lib.coef_count(H, W)
        coef = np.zeros((len(files), n), np.int16)
        qt = np.zeros((len(files), 3, 64), np.uint16)
        for i, data in enumerate(files):
            rc, info = jpeg_lib.decode_coeffs(data, coef[i], qt[i])
            assert rc == 0
        off = torch.arange(len(files), dtype=torch.int64) * n
        got = engine.jpeg_idct_rgb(torch.from_numpy(coef).cuda(), off.cuda(), torch.from_numpy(qt.view(np.int16)).cuda(), H, W).cpu().numpy()
        for i, data in enumerate(files):
            ref = _pil(data)
            assert np.array_equal(got[i], ref), f"{H}x{W} image {i}: {int((got[i] != ref).sum())} bytes differ"


def test_entropy_decoder_survives_mutated_files_under_asan(tmp_path):
    """The host decoder parses files it did not write: tests/fuzz_jpeg.c, built with AddressSanitizer + UBSan (CPU build only, as the pool
    asks), decodes thousands of mutated baseline JPEGs into an output buffer of EXACTLY the size the header promises -- flips, truncations,
    damaged tables and lengths, stray markers, runs of 0xFF: every one is either decoded or rejected with a code, none reads or writes out
    of bounds."""
    import shutil
    import subprocess
    from aquaculture_amd import tiles
    cc = shutil.which("gcc")
    if not cc:
        pytest.skip("no gcc")
    rng = np.random.default_rng(1)
    files = []
    for name, img, kw in (("tile.jpg", tiles.synthetic_tile(3, 256), dict(quality=75)),
                          ("noise.jpg", rng.integers(0, 255, (100, 150, 3), dtype=np.uint8), dict(quality=50)),
                          ("restart.jpg", rng.integers(0, 255, (64, 64, 3), dtype=np.uint8), dict(quality=90, restart_marker_blocks=3))):
        (tmp_path / name).write_bytes(_jpeg(img, **kw))
        files.append(str(tmp_path / name))
    exe = str(tmp_path / "fuzz_jpeg")
    r = subprocess.run([cc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, os.path.join(ROOT, "tests", "fuzz_jpeg.c"),
                        os.path.join(ROOT, "aquaculture_amd", "csrc", "jpeg_coef.c")], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "1500"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_jpeg:" in r.stdout, r.stdout[-500:] + r.stderr[-3000:]
    decoded, rejected = (int(x) for x in __import__("re").findall(r"(\d+) decoded, (\d+) rejected", r.stdout)[0])
    assert decoded > 500 and rejected > 500          # the mutations reach both outcomes


def test_header_scan_in_processes_equals_the_in_process_one(jpeg_lib, tmp_path):
    """Long sweeps scan their files' headers in child processes (jpeg.scan_files_in_processes): same answers as scan_file, file by file --
    sizes for what the split decoder reads, None for a progressive file and for a file that is no JPEG."""
    from aquaculture_amd import tiles
    for i in range(25):
        (tmp_path / f"{i:03d}.jpeg").write_bytes(_jpeg(tiles.synthetic_tile(i, 64 + 16 * (i % 3)), quality=75))
    (tmp_path / "zz_progressive.jpeg").write_bytes(_jpeg(tiles.synthetic_tile(1, 64), quality=75, progressive=True))
    (tmp_path / "zzz_text.jpeg").write_bytes(b"not a jpeg")
    paths = sorted(str(tmp_path / f) for f in os.listdir(tmp_path))
    want = [jpeg_lib.scan_file(p) for p in paths]
    assert want[:3] == [(64, 64), (80, 80), (96, 96)] and want[-2:] == [None, None]
    assert jpeg_lib.scan_files_in_processes(paths, 3) == want
    assert jpeg_lib.scan_files_in_processes(paths[:1], 8) == want[:1]


def _scan_bytes(data):
    """The entropy-coded bytes of a baseline file, from behind the SOS header up to (not including) EOI."""
    sos = data.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    end = data.rindex(b"\xff\xd9")
    return data[start:end]


@pytest.mark.parametrize("idx", range(11))
def test_gpu_decode_preparation_is_the_scan_minus_stuffing_and_restart_markers(jpeg_lib, idx):
    """aq_jpeg_prepare (host side of the GPU entropy decode): putting the stuffed zeros and the RSTn markers back into what it wrote gives the
    file's scan bytes exactly; segments are 4-byte aligned, zero-padded, one per restart interval; tables and sizes equal the header scan's."""
    name, data = _cases()[idx]
    ref = _pil(data)
    H, W = ref.shape[:2]
    b = jpeg_lib.GpuDecodeBatch(2, H, W, bytes_per_image=jpeg_lib.stream_capacity(H, W, 12))
    assert b.add(1, data) == 0, name
    sg = b._segs[1]
    rebuilt = bytearray()
    for i, s in enumerate(sg):
        off, ln = int(s["stream_off"]), int(s["stream_len"])
        assert off % 16 == 0 and off >= b.per and not b.streams[off + ln:off + ln + 8].any(), name
        rebuilt += bytes(b.streams[off:off + ln]).replace(b"\xff", b"\xff\x00")
        if i + 1 < len(sg):
            rebuilt += bytes([0xFF, 0xD0 + (i & 7)])
    assert bytes(rebuilt) == _scan_bytes(data), name
    info = jpeg_lib.scan(data)
    assert np.array_equal(b.qt[1], np.ctypeslib.as_array(info.qt).reshape(3, 64))
    total = info.mcu_cols * info.mcu_rows
    assert int(sg["n_mcu"].sum()) == total and int(sg["mcu0"][0]) == 0 and (np.diff(sg["mcu0"].astype(np.int64)) == sg["n_mcu"][:-1]).all()
    assert (sg["coef_off"] == b.nco).all()


def test_gpu_decode_preparation_refuses_what_the_host_decoder_refuses(jpeg_lib):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 255, (32, 48, 3), dtype=np.uint8)
    b = jpeg_lib.GpuDecodeBatch(1, 32, 48, bytes_per_image=jpeg_lib.stream_capacity(32, 48, 12))
    for kw in (dict(subsampling=0), dict(progressive=True)):
        assert b.add(0, _jpeg(img, **kw)) == -1, kw
    assert b.add(0, _jpeg(img[..., 0])) == -1
    good = _jpeg(img, quality=90, restart_marker_blocks=2)
    assert b.add(0, good) == 0
    for cut in (len(good) // 2, len(good) - 2, 40):
        assert b.add(0, good[:cut]) == -2, cut                     # truncated: no EOI
    sos = good.index(b"\xff\xda")
    mid = sos + (len(good) - sos) // 2
    assert b.add(0, good[:mid] + b"\xff\xc4" + good[mid + 2:]) == -2      # a stray marker inside the scan
    rst = good.index(b"\xff\xd0", sos)
    assert b.add(0, good[:rst] + b"\xff\xd3" + good[rst + 2:]) == -2      # restart markers out of sequence
    with pytest.raises(ValueError):
        b.add(0, _jpeg(rng.integers(0, 255, (48, 48, 3), dtype=np.uint8)))  # another size than the batch's


@pytest.mark.gpu
def test_gpu_entropy_decode_equals_the_host_decoder(jpeg_lib, lib):
    """aq_jpeg_huffman_decode (one lane per restart segment) on batches of same-size files: the coefficient buffers equal
    aq_jpeg_decode_coeffs's, value for value -- and therefore, through aq_jpeg_idct_rgb, Pillow's pixels byte for byte."""
    import torch
    from aquaculture_amd import engine, tiles
    rng = np.random.default_rng(11)
    noise = lambda h, w: rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
    groups = {
        (640, 640): [_jpeg(tiles.synthetic_tile(i, 640), quality=75) for i in (0, 3, 19, 40)] + [_jpeg(noise(640, 640), quality=q) for q in (50, 95)],
        (1024, 1024): [_jpeg(tiles.synthetic_tile(i, 1024), quality=75) for i in (1, 19)] + [_jpeg(noise(1024, 1024), quality=75, optimize=True)],
        (100, 150): [_jpeg(noise(100, 150), quality=q) for q in (30, 75, 98)] + [_jpeg(noise(100, 150), quality=60, restart_marker_blocks=3)],
        (33, 17): [_jpeg(noise(33, 17), quality=80)],
        (64, 64): [_jpeg(noise(64, 64), quality=75, restart_marker_blocks=3), _jpeg(noise(64, 64), quality=85, restart_marker_rows=1),
                   _jpeg(np.tile(np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]], np.uint8).repeat(8, 0).repeat(8, 1), (4, 4, 1)), quality=100)],
    }
    for (H, W), files in groups.items():
        # 70 images: more than one wave of segments, files repeated
        files = (files * 70)[:70]
        n = jpeg_lib.coef_count(H, W)
        want = np.zeros((len(files), n), np.int16)
        qt = np.zeros((len(files), 3, 64), np.uint16)
        for i, data in enumerate(files):
            rc, _ = jpeg_lib.decode_coeffs(data, want[i], qt[i])
            assert rc == 0
        b = jpeg_lib.GpuDecodeBatch(len(files), H, W, bytes_per_image=jpeg_lib.stream_capacity(H, W, 12))
        for i, data in enumerate(files):
            assert b.add(i, data) == 0
        segs, sets, first = b.finish()
        coef = torch.zeros((len(files), n), dtype=torch.int16, device="cuda")
        st = engine.jpeg_huffman_decode(torch.from_numpy(b.streams).cuda(), torch.from_numpy(segs.view(np.uint8).reshape(-1, 32)).cuda(),
                                        torch.from_numpy(sets).cuda(), coef.view(-1))
        assert int(st.abs().max()) == 0, (H, W, st.cpu().tolist())
        got = coef.cpu().numpy()
        assert np.array_equal(b.qt, qt)
        bad = [i for i in range(len(files)) if not np.array_equal(got[i], want[i])]
        assert not bad, f"{H}x{W}: images {bad[:5]} differ ({int((got != want).sum())} values)"
        off = torch.arange(len(files), dtype=torch.int64) * n
        rgb = engine.jpeg_idct_rgb(coef.view(-1), off.cuda(), torch.from_numpy(b.qt.view(np.int16)).cuda(), H, W).cpu().numpy()
        for i in (0, len(files) - 1):
            assert np.array_equal(rgb[i], _pil(files[i]))


@pytest.mark.gpu
def test_gpu_entropy_decode_flags_damaged_scans(jpeg_lib, lib):
    """Bits flipped inside the entropy-coded data (markers intact, so the host preparation accepts the file): every segment either decodes to
    SOME coefficients or reports status 2 -- the kernel terminates, stays inside its buffers (a canary behind them is intact), and a scan cut
    short inside its data (the EOI kept) is reported as consumed-past-the-end."""
    import torch
    from aquaculture_amd import engine
    rng = np.random.default_rng(5)
    H, W = 96, 128
    good = _jpeg(rng.integers(0, 255, (H, W, 3), dtype=np.uint8), quality=80)
    sos = good.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(good[sos + 2:sos + 4], "big")
    files = [good]
    for k in range(40):
        d = bytearray(good)
        for _ in range(1 + k % 5):
            pos = int(rng.integers(start, len(good) - 2))
            d[pos] = (d[pos] ^ (1 << int(rng.integers(0, 8)))) & 0xFE or 0x01          # never create 0xFF (a marker) by accident
        files.append(bytes(d))
    files.append(good[:start + (len(good) - start) // 3] + b"\xff\xd9")                # a third of the scan, then EOI
    b = jpeg_lib.GpuDecodeBatch(len(files), H, W, bytes_per_image=jpeg_lib.stream_capacity(H, W, 12))
    keep = [i for i, d in enumerate(files) if b.add(i, d) == 0]
    assert 0 in keep and len(files) - 1 in keep and len(keep) > 30
    # compact the accepted files into slots 0.. (finish() wants a prefix)
    b2 = jpeg_lib.GpuDecodeBatch(len(keep), H, W, bytes_per_image=jpeg_lib.stream_capacity(H, W, 12))
    for j, i in enumerate(keep):
        assert b2.add(j, files[i]) == 0
    segs, sets, first = b2.finish()
    n = jpeg_lib.coef_count(H, W)
    coef = torch.zeros(len(keep) * n + 4096, dtype=torch.int16, device="cuda")
    coef[len(keep) * n:] = 12345
    st = engine.jpeg_huffman_decode(torch.from_numpy(b2.streams).cuda(), torch.from_numpy(segs.view(np.uint8).reshape(-1, 32)).cuda(),
                                    torch.from_numpy(sets).cuda(), coef).cpu().numpy()
    assert set(st.tolist()) <= {0, 2} and st[0] == 0 and st[-1] == 2
    assert bool((coef[len(keep) * n:] == 12345).all())
    host_ok = []
    tmp, qt = np.zeros(n, np.int16), np.zeros((3, 64), np.uint16)
    for j, i in enumerate(keep):
        host_ok.append(jpeg_lib.decode_coeffs(files[i], tmp, qt)[0] == 0)
    assert [s == 0 for s in st.tolist()] == host_ok                                     # the same files are refused as by the host decoder
