"""LBB2 MSB-plane payload: the HIP encoder/decoder against the sequential restatement in oracle/plane_codec.c
(byte identity both ways), lossless round trips over edge shapes and value ranges, rejection of damaged
streams, and the full-size tile."""
import struct
import time

import numpy as np
import pytest
import torch

import oracle as O
from lbdrn_hip import container, ops
from lbdrn_hip.synth import synthetic_tile

pytestmark = pytest.mark.gpu


def _oracle_body(x):
    counts, words = O.plane_encode(x)
    return counts.astype("<u4").tobytes() + words.astype("<u4").tobytes()


def _cases():
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:150, 0:200]
    yield "one pixel", np.array([[[513]]], np.uint16)
    yield "one column", rng.integers(0, 300, (2, 130, 1)).astype(np.uint16)
    yield "one row", rng.integers(0, 300, (1, 1, 257)).astype(np.uint16)
    yield "zeros", np.zeros((2, 70, 90), np.uint16)
    yield "constant", np.full((1, 65, 64), 40000, np.uint16)
    yield "ramp", ((xx * 3 + yy * 5) % 65536).astype(np.uint16)[None]
    yield "smooth", (2000 + 1500 * np.sin(xx / 30.0) * np.cos(yy / 20.0)).astype(np.uint16)[None]
    yield "noise 8 bit", rng.integers(0, 256, (3, 64, 64)).astype(np.uint16)
    yield "noise 16 bit", rng.integers(0, 65536, (2, 129, 67)).astype(np.uint16)
    yield "sparse spikes", ((rng.random((1, 200, 130)) < 0.02) * rng.integers(0, 65536, (1, 200, 130))).astype(np.uint16)
    yield "wraparound", np.where(rng.random((1, 90, 70)) < 0.5, 0, 65535).astype(np.uint16)
    yield "synthetic msb", (synthetic_tile(2, 4, 300, 333) >> 5).astype(np.uint16)
    yield "synthetic msb K9", (synthetic_tile(3, 2, 260, 200) >> 9).astype(np.uint16)


@pytest.mark.parametrize("name,x", list(_cases()), ids=[n for n, _ in _cases()])
def test_body_is_byte_identical_to_oracle_and_round_trips(dev, name, x):
    x_d = ops.to_device_u16(x, dev)
    body = ops.plane_encode(x_d)
    assert body == _oracle_body(x)
    C, H, W = x.shape
    assert np.array_equal(ops.from_device_u16(ops.plane_decode(body, C, H, W, dev)), x)
    ns = C * ((W + 63) // 64)
    counts = np.frombuffer(body[:4 * ns], "<u4")
    words = np.frombuffer(body[4 * ns:], "<u4")
    assert np.array_equal(O.plane_decode(counts, words, C, H, W), x)      # the oracle reads the GPU's stream


def test_damaged_streams_are_rejected_not_trusted(dev):
    x = (synthetic_tile(4, 2, 140, 100) >> 4).astype(np.uint16)
    C, H, W = x.shape
    body = ops.plane_encode(ops.to_device_u16(x, dev))
    with pytest.raises(ops._lib.LbdrnError):
        ops.plane_decode(body[:8], C, H, W, dev)                  # cannot even hold the counts
    with pytest.raises(ops._lib.LbdrnError):
        ops.plane_decode(body[:-40], C, H, W, dev)                # counts promise more words than there are
    ns = C * ((W + 63) // 64)
    bad = bytearray(body)
    struct.pack_into("<I", bad, 0, struct.unpack_from("<I", bad, 0)[0] - 3)   # a strip claims fewer words
    with pytest.raises(ops._lib.LbdrnError):
        ops.plane_decode(bytes(bad), C, H, W, dev)
    bad = bytearray(body)
    struct.pack_into("<I", bad, 4 * ns, 0x77)                     # k0 out of range in a strip's first word
    with pytest.raises(ops._lib.LbdrnError):
        ops.plane_decode(bytes(bad), C, H, W, dev)
    rng = np.random.default_rng(1)
    for _ in range(4):                                            # noise instead of words: must terminate
        bad = bytearray(body)
        bad[4 * ns + 8:] = rng.integers(0, 256, len(bad) - 4 * ns - 8, dtype=np.uint8).tobytes()
        try:
            ops.plane_decode(bytes(bad), C, H, W, dev)
        except ops._lib.LbdrnError:
            pass


def test_container_round_trip_uint8_and_uint16(dev):
    for K, dtype in ((3, np.uint16), (9, np.uint8)):
        msb = (synthetic_tile(7, 3, 100, 140) >> K).astype(dtype)
        payload = container.encode_base(msb)
        assert payload[:4] == b"LBB2"
        back = container.decode_base(payload)
        assert back.dtype == dtype and np.array_equal(back, msb)
    legacy = container.encode_base(msb, codec="LBB1")            # the portable host codec still decodes
    assert legacy[:4] == b"LBB1" and np.array_equal(container.decode_base(legacy), msb)


def test_full_size_tile(dev):
    """8 x 2048 x 2048 at K=5: byte identity with the oracle, lossless, and far below the fit time."""
    msb = (synthetic_tile(0, 8, 2048, 2048) >> 5).astype(np.uint16)
    x_d = ops.to_device_u16(msb, dev)
    ops.plane_encode(x_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    body = ops.plane_encode(x_d)
    t_enc = time.perf_counter() - t0
    assert body == _oracle_body(msb)
    t0 = time.perf_counter()
    back = ops.plane_decode(body, 8, 2048, 2048, dev)
    torch.cuda.synchronize()
    t_dec = time.perf_counter() - t0
    assert torch.equal(back, x_d)
    bpsp = 8 * len(body) / msb.size
    print(f"LBB2 full tile: {bpsp:.3f} bpsp, encode {t_enc * 1e3:.1f} ms, decode {t_dec * 1e3:.1f} ms (host-inclusive)")
    assert bpsp < 3.2 and t_enc < 0.5 and t_dec < 0.5


def test_random_shapes_and_statistics_fuzz(dev):
    """Random geometry and pixel statistics (smooth, noisy, sparse, saturated, mixtures per band): byte identity
    with the oracle and lossless both ways.  LBDRN_FUZZ_SOAK multiplies the case count."""
    import os
    soak = int(os.environ.get("LBDRN_FUZZ_SOAK", "1"))
    rng = np.random.default_rng(2024)
    for it in range(12 * soak):
        C, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 300)), int(rng.integers(1, 200))
        yy, xx = np.mgrid[0:H, 0:W]
        bands = []
        for c in range(C):
            kind = int(rng.integers(0, 6))
            if kind == 0:
                b = rng.integers(0, int(rng.choice([2, 16, 300, 65536])), (H, W))
            elif kind == 1:
                b = 3000 + 2500 * np.sin(xx / rng.uniform(3, 60)) * np.cos(yy / rng.uniform(3, 60)) + rng.normal(0, rng.uniform(0, 6), (H, W))
            elif kind == 2:
                b = (rng.random((H, W)) < rng.uniform(0, 0.1)) * rng.integers(0, 65536, (H, W))
            elif kind == 3:
                b = np.full((H, W), int(rng.integers(0, 65536)))
            elif kind == 4:
                b = (xx * int(rng.integers(0, 700)) + yy * int(rng.integers(0, 700))) % 65536
            else:
                b = np.where((xx // 7 + yy // 5) % 2 == 0, 65535, 0) + rng.integers(-1, 2, (H, W))
            bands.append(np.clip(np.rint(b), 0, 65535).astype(np.uint16))
        x = np.stack(bands)
        body = ops.plane_encode(ops.to_device_u16(x, dev))
        assert body == _oracle_body(x), (it, x.shape)
        assert np.array_equal(ops.from_device_u16(ops.plane_decode(body, C, H, W, dev)), x), (it, x.shape)
