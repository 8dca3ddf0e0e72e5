"""The JPEG 2000 MSB payload (ref encode.py:137, decode.py:69-73) through liblbdrn_jp2.so =
OpenJPEG behind a C ABI.  Host code: the container-level tests run without a GPU, the CLI round trip needs one.
Skipped where OpenJPEG was not found at build time."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def jp2():
    from lbdrn_hip import jp2 as mod
    if not mod.available():
        pytest.skip("liblbdrn_jp2.so not built (OpenJPEG absent)")
    return mod


def test_jp2_library_exports_every_declared_symbol(jp2):
    hdr = open(os.path.join(ROOT, "include", "lbdrn_jp2.h")).read()
    declared = set(re.findall(r"\b(lbdrn_jp2_[a-z_0-9]+)\s*\(", hdr))
    assert declared == {"lbdrn_jp2_last_error", "lbdrn_jp2_encode", "lbdrn_jp2_info", "lbdrn_jp2_decode", "lbdrn_jp2_free",
                        "lbdrn_jp2_set_threads"}
    L = ctypes.CDLL(jp2._PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_lossless_round_trip_of_multiband_planes(jp2):
    """uint8 and uint16 planes, one to sixteen bands, sizes below / above the 1024-pixel tile edge, full 16-bit range:
    decode(encode(x)) == x with the dtype kept (the reference's MSB raster is Byte when max <= 255, LBDRNdataset.py:100)."""
    from lbdrn_hip import container
    rng = np.random.default_rng(0)
    cases = [(np.uint8, 255, (3, 37, 41)), (np.uint16, 312, (8, 70, 90)), (np.uint16, 65535, (1, 5, 7)),
             (np.uint8, 255, (2, 1100, 1030)), (np.uint16, 2047, (16, 33, 65)), (np.uint8, 1, (4, 1, 9))]
    for dt, hi, shape in cases:
        x = rng.integers(0, hi + 1, shape).astype(dt)
        if shape[1] > 64:   # spatially correlated content as well as noise
            x[0] = (np.add.outer(np.arange(shape[1]), np.arange(shape[2])) % (hi + 1)).astype(dt)
        b = jp2.encode(x)
        assert b[:12] == jp2.SIGNATURE and jp2.is_jp2(b)
        y = jp2.decode(b)
        assert y.dtype == dt and np.array_equal(x, y), (dt, shape)
        # through the container: the payload is recognised by its signature, whatever wrote it
        z = container.decode_base(container.encode_base(x, codec="jp2"))
        assert z.dtype == dt and np.array_equal(x, z)
    flat = np.zeros((2, 64, 64), np.uint16)
    assert len(jp2.encode(flat)) < 1000      # an all-zero plane costs its headers


def test_worker_threads_change_no_byte(jp2):
    """lbdrn_jp2_set_threads: OpenJPEG codes a call's code blocks on worker threads; the stream and the decoded values are
    those of the single-threaded call (tiled and untiled rasters)."""
    rng = np.random.default_rng(8)
    old = jp2.set_threads(1)
    try:
        for shape in ((8, 300, 260), (3, 1100, 1030)):
            x = (rng.integers(0, 300, shape) + np.arange(shape[2])[None, None, :] // 7).astype(np.uint16)
            one = jp2.encode(x)
            assert jp2.set_threads(4) == 1
            four = jp2.encode(x)
            assert four == one and np.array_equal(jp2.decode(one), x)
            assert jp2.set_threads(1) == 4
            assert np.array_equal(jp2.decode(four), x)
        assert jp2.set_threads(100000) == 1 and jp2.set_threads(1) == 1      # (out of range: refused, setting unchanged)
    finally:
        jp2.set_threads(old)
    assert 0 <= jp2.default_threads() <= 8


def test_damaged_and_foreign_streams_are_refused(jp2):
    from lbdrn_hip import container
    x = np.arange(3 * 20 * 30, dtype=np.uint16).reshape(3, 20, 30)
    b = jp2.encode(x)
    with pytest.raises(jp2.Jp2Error):
        jp2.decode(b[:len(b) // 2])
    # what OpenJPEG said reaches the caller whichever of the codec's threads met it: a per-call context, not the reporting
    # thread's thread-local buffer (csrc/jp2_shim.c: err_ctx)
    big = jp2.encode(np.random.default_rng(1).integers(0, 300, (4, 300, 300)).astype(np.uint16))
    old = jp2.set_threads(8)
    try:
        for cut in (len(big) // 2, 200):
            with pytest.raises(jp2.Jp2Error) as e:
                jp2.decode(big[:cut])
            assert "openjpeg: " in str(e.value) and not str(e.value).endswith(("decoding failed", "cannot read the header", ": "))
    finally:
        jp2.set_threads(old)
    with pytest.raises(jp2.Jp2Error):
        jp2.decode(b"not a jpeg 2000 stream at all")
    with pytest.raises(ValueError):
        container.decode_base(b"XXXX" + b"\0" * 40)
    with pytest.raises(ValueError):
        jp2.encode(np.zeros((1, 4, 4), np.int32))


def _pillow_jp2():
    try:
        from PIL import Image, features
    except ImportError:
        pytest.skip("Pillow is not installed")
    if not features.check("jpg_2000"):
        pytest.skip("this Pillow has no JPEG 2000 codec")
    return Image, features.version("jpg_2000")


def test_interop_with_a_second_openjpeg_build(jp2, tmp_path):
    """VERDICT round 4 item 5, the reviewer's hand check as a test: Pillow bundles an OpenJPEG of its own (2.5.x; the shim
    links the image's 2.4.0), so streams that cross between the two are checked by an independent build of the codec --
    what `jp2.encode` writes Pillow decodes bit-exactly (1 band x 16 bit, 3 bands x 8 bit), and what Pillow writes
    reversibly (one tile and several, raw codestream and .jp2 file) `jp2.decode` reads bit-exactly.  That is as far as
    the reference's own payload (GDAL -> OpenJPEG, encode.py:137 / decode.py:69-73) can be pinned without GDAL."""
    import io
    Image, version = _pillow_jp2()
    rng = np.random.default_rng(7)
    smooth = (np.add.outer(np.arange(150) * 37, np.arange(211) * 11) % 9000).astype(np.uint16)
    g16 = (smooth + rng.integers(0, 40, smooth.shape)).astype(np.uint16)                       # [H, W], up to 9039
    rgb8 = rng.integers(0, 256, (97, 130, 3)).astype(np.uint8)
    rgb8[..., 1] = (np.add.outer(np.arange(97), np.arange(130)) % 256).astype(np.uint8)
    # ours -> Pillow
    im = Image.open(io.BytesIO(jp2.encode(g16[None])))
    im.load()
    assert im.mode in ("I;16", "I;16L", "I;16B", "I") and np.array_equal(np.asarray(im).astype(np.uint16), g16), (version, im.mode)
    im = Image.open(io.BytesIO(jp2.encode(np.ascontiguousarray(rgb8.transpose(2, 0, 1)))))
    im.load()
    assert im.mode == "RGB" and np.array_equal(np.asarray(im), rgb8), (version, im.mode)
    # Pillow -> ours: reversible wavelet, no colour transform on the multi-band image, single- and multi-tile, both containers
    survived = 0
    for arr, planes in ((g16, g16[None]), (rgb8, rgb8.transpose(2, 0, 1))):
        for ext in ("jp2", "j2k"):
            for tile in (None, (64, 64)):
                path = tmp_path / f"p_{arr.ndim}_{ext}_{'t' if tile else 'o'}.{ext}"
                kw = dict(irreversible=False, mct=0)
                if tile:
                    kw["tile_size"] = tile
                Image.fromarray(arr).save(str(path), **kw)
                raw = path.read_bytes()
                assert jp2.is_jp2(raw)
                got = jp2.decode(raw)
                # the judge of a stream is the OTHER build's decoder (Pillow 12's tiled writer mangles 16-bit input --
                # its own decoder does not give the source back either --, so the source is compared where it survives)
                back = np.asarray(Image.open(str(path)))
                back = back[None] if back.ndim == 2 else back.transpose(2, 0, 1)
                assert got.dtype == planes.dtype and np.array_equal(got, back.astype(planes.dtype)), (version, ext, tile, arr.shape)
                if np.array_equal(back, planes):
                    survived += 1
    assert survived >= 6, survived    # (every case but the two tiled 16-bit ones)


@pytest.mark.gpu
def test_cli_round_trip_with_the_jpeg2000_payload(jp2, dev, tmp_path):
    """LBDRN_BASE_CODEC=jp2: encode.py writes the MSB planes as a JP2 file inside the .bin (where the reference's
    gdal_translate output sits), decode.py recognises it by its signature; the decoded raster equals the one of the default
    (LBB2) payload bit for bit, the metrics are logged, and the JP2 payload alone decodes to img >> K."""
    from lbdrn_hip import container, raster_io
    from lbdrn_hip.synth import synthetic_tile
    img = synthetic_tile(21, 4, 40, 56)
    src = tmp_path / "tile.tif"
    raster_io.write_raster(str(src), img)
    recs = {}
    for codec_name in ("jp2", "LBB2"):
        env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "lbdrn-msic_amd"), LBDRN_BASE_CODEC=codec_name, LBDRN_REPORT_BOTH_BPSP="1")
        out = tmp_path / codec_name
        subprocess.run([sys.executable, os.path.join(ROOT, "lbdrn-msic_amd", "encode.py"), "-i", str(src), "-o", str(out),
                        "-K", "5", "-D", "2", "-bs", "256", "-e", "2"], check=True, env=env, capture_output=True)
        outdir = out / "tile_r1_K5_bc64_nl2_D2_prec16_lr0.001_bs256_e2"
        raw = (outdir / "tile.bin").read_bytes()
        n, sr, w, h, K, bc, nl, D, nn, base = container.unpack_header(raw)
        payload = raw[n + nn[0]:]
        assert len(payload) == base[0]
        assert jp2.is_jp2(payload) == (codec_name == "jp2")
        msb = container.decode_base(payload)
        assert np.array_equal(msb, img >> 5) and msb.dtype == np.uint16      # max 312 > 255: UInt16 like the reference's raster
        import decode as dec_mod
        sys.argv = ["decode.py"]
        assert dec_mod.main(["-i", str(outdir / "tile.bin")]) == 0
        recs[codec_name] = raster_io.read_raster(str(outdir / "tile_recon.tif"))
        log = (outdir / "encode.txt").read_text()
        assert re.search(r"MSB: (\d+) bytes", log)
        # LBDRN_REPORT_BOTH_BPSP: the other format's size rides along, in a record the summary's patterns do not match
        other = re.search(r"MSB as (jp2|LBB2): (\d+) bytes: bpsp=", log)
        assert other and other.group(1) == ("LBB2" if codec_name == "jp2" else "jp2") and len(re.findall(r"MSB: \d+ bytes", log)) == 1
    assert np.array_equal(recs["jp2"], recs["LBB2"]) and np.array_equal(recs["jp2"] >> 5, img >> 5)
    # the JP2 payloads are coded on a host thread WHILE the GPU fits (encode.BasePayloadsAhead, round 6): the same .bin byte
    # for byte as coding them after the fits (LBDRN_JP2_AHEAD=0), one tile or four
    assert "coded beside the fit" in (tmp_path / "jp2" / "tile_r1_K5_bc64_nl2_D2_prec16_lr0.001_bs256_e2" / "encode.txt").read_text()
    bins = {}
    for ahead in ("1", "0"):
        env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "lbdrn-msic_amd"), LBDRN_BASE_CODEC="jp2", LBDRN_JP2_AHEAD=ahead)
        out = tmp_path / f"sr2_ahead{ahead}"
        subprocess.run([sys.executable, os.path.join(ROOT, "lbdrn-msic_amd", "encode.py"), "-i", str(src), "-o", str(out),
                        "-K", "5", "-D", "2", "-bs", "256", "-e", "2", "-sr", "2"], check=True, env=env, capture_output=True)
        bins[ahead] = (out / "tile_r2_K5_bc64_nl2_D2_prec16_lr0.001_bs256_e2" / "tile.bin").read_bytes()
    assert bins["1"] == bins["0"] and len(bins["1"]) > 1000
