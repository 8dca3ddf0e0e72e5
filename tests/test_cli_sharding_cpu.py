"""Host logic of the multi-GPU CLIs on CPU (gloo, world_size 2): tile-sharded encode.py / decode.py give the
files of the serial run, the RNG replay that makes this possible, the sweep's job dealing and the results
CSV.  The GPU fits themselves are replaced by a stand-in whose output depends on the random draws made for each
tile (codec.draw_fit) -- what is under test here is everything around them."""
import hashlib
import os
import re
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lbdrn-msic_amd")
for p in (PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def test_skip_fit_rng_consumes_what_a_fit_consumes():
    """Generator state after skip_fit_rng == after the torch port of the reference's train() (real
    DataLoader, real model construction) on the same shapes."""
    import torch_port
    import oracle as O
    from lbdrn_hip import codec
    rng = np.random.default_rng(0)
    img = rng.integers(0, 4000, (3, 9, 11)).astype(np.uint16)
    for epochs, vd in ((3, 1), (1, 1), (4, 2)):
        torch.manual_seed(77)
        torch_port.fit(img, 3, 1, 32, 2, 1e-3, 40, epochs, O.FeatCfg(), 0, True, vd)
        after_fit = torch.get_rng_state()
        torch.manual_seed(77)
        codec.skip_fit_rng(3 * 9, 32, 3, 2, epochs, vd)
        assert torch.equal(torch.get_rng_state(), after_fit), (epochs, vd)


def _stub_train_tiles(args, tiles, draws):
    """Stands where encode.train_tiles stands (the GPU fits): one result per tile that depends on the tile's
    pixels and on the draws made for it -- so a wrong draw order on any rank changes the payload."""
    out = []
    for (path, img), dr in zip(tiles, draws):
        h = hashlib.sha256(dr.params.numpy().tobytes() + repr(dr.train_seeds).encode()).digest()
        out.append((h + hashlib.sha256(img.tobytes()).digest(), img.tobytes()[: 50 + img.shape[2]]))
    return out


def _stub_report(args, res, base_ahead=None):
    import logger
    nn, base = res
    logger.log.info(f"nn: {len(nn)} bytes, bpsp=0.5")
    logger.log.info(f"MSB: {len(base)} bytes: bpsp=0.25")
    return nn, base


def _stub_apply(base, params, K, D, bc, nl, cfg=None, device=None, **kw):
    return (np.asarray(base).astype(np.uint16) << K) + 1


def _encode_worker(rank, world, port, src, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import encode
    encode.train_tiles, encode.report_and_pack = _stub_train_tiles, _stub_report
    rc = encode.main(["-i", src, "-o", out_dir, "-sr", "3", "-K", "4", "-e", "2", "-bs", "64"])
    assert rc == 0


def _decode_worker(rank, world, port, bin_path, org):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import decode
    from lbdrn_hip import codec, container
    codec.apply_image = _stub_apply
    container.decode_weights = lambda payload, expected=None: np.zeros(4, np.float32)
    container.decode_base = lambda payload, device=None, keep_on_device=False: np.frombuffer(payload[:48], np.uint8).reshape(2, 4, 6).copy()
    assert decode.main(["-i", bin_path, "-org", org]) == 0


def _records(path):
    with open(path) as f:
        return [re.sub(r"^\[[^\]]*\] ", "", line.rstrip("\n")) for line in f]


def test_tile_sharded_encode_equals_serial(tmp_path, monkeypatch):
    import encode
    rng = np.random.default_rng(4)
    src = str(tmp_path / "scene.npy")
    np.save(src, rng.integers(0, 9000, (2, 31, 40)).astype(np.uint16))
    monkeypatch.setattr(encode, "train_tiles", _stub_train_tiles)
    monkeypatch.setattr(encode, "report_and_pack", _stub_report)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert encode.main(["-i", src, "-o", str(tmp_path / "serial"), "-sr", "3", "-K", "4", "-e", "2", "-bs", "64"]) == 0
    port = 29600 + os.getpid() % 300
    mp.spawn(_encode_worker, args=(2, port, src, str(tmp_path / "sharded")), nprocs=2, join=True)
    sub = "scene_r3_K4_bc64_nl2_D2_prec16_lr0.001_bs64_e2"
    a = (tmp_path / "serial" / sub / "scene.bin").read_bytes()
    b = (tmp_path / "sharded" / sub / "scene.bin").read_bytes()
    assert a == b and len(a) > 9 * 64
    ra, rb = _records(tmp_path / "serial" / sub / "encode.txt"), _records(tmp_path / "sharded" / sub / "encode.txt")
    strip = lambda recs: [r.replace(str(tmp_path / "serial"), "X").replace(str(tmp_path / "sharded"), "X")
                          for r in recs if not r.startswith("Time elapsed")]
    assert strip(ra) == strip(rb)                      # same records, in tile order
    assert sum(r.startswith("nn: ") for r in rb) == 9 and rb[-1].startswith("Time elapsed: ")


def test_tile_sharded_decode_equals_serial(tmp_path, monkeypatch):
    import decode
    from lbdrn_hip import codec, container
    tiles = 4
    nn, base = [b"w" * 10] * tiles, [bytes(range(t, t + 48)) + b"pad" * t for t in range(tiles)]
    blob = container.pack_header(2, 12, 8, 3, 64, 2, 2, [len(x) for x in nn], [len(x) for x in base])
    for x, y in zip(nn, base):
        blob += x + y
    org = str(tmp_path / "org.npy")
    np.save(org, np.random.default_rng(1).integers(0, 2000, (2, 8, 12)).astype(np.uint16))
    outs = {}
    for mode in ("serial", "sharded"):
        d = tmp_path / mode
        d.mkdir()
        (d / "img.bin").write_bytes(blob)
        if mode == "serial":
            monkeypatch.setattr(codec, "apply_image", _stub_apply)
            monkeypatch.setattr(container, "decode_weights", lambda payload, expected=None: np.zeros(4, np.float32))
            monkeypatch.setattr(container, "decode_base",
                                lambda payload, device=None, keep_on_device=False: np.frombuffer(payload[:48], np.uint8).reshape(2, 4, 6).copy())
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                monkeypatch.delenv(k, raising=False)
            assert decode.main(["-i", str(d / "img.bin"), "-org", org]) == 0
        else:
            mp.spawn(_decode_worker, args=(2, 29900 + os.getpid() % 90, str(d / "img.bin"), org), nprocs=2, join=True)
        outs[mode] = [r for r in _records(d / "decode.txt") if r.startswith(("MSE", "PSNR", "Total size"))]
        assert len(outs[mode]) == 3
    assert outs["serial"] == outs["sharded"]


def test_sweep_points_and_legacy_arguments():
    import sweep
    a = sweep.parse(["3", "2", "256", "2", "0.001", "8192", "10", "1", "outputs-x"])
    assert (a.D, a.base_channel, a.num_layers, a.lr, a.batch_size, a.epochs, a.split_ratio, a.output_dir) == \
        (2, 256, 2, "0.001", 8192, 10, 1, "outputs-x")
    pts = sweep.points(a)
    assert len(pts) == 13 * 6 and pts[0][1] == 1 and pts[5][1] == 6 and pts[6][0] != pts[0][0]   # run.sh order
    assert pts[0][0].endswith("TRIPLESAT_2_MS_L1_20191107021947_001FFCVI_002_0120200811001001_001.tif")
    assert pts[-1][0] == "data/GF-dataset/GF-6/GF6-PMS/GF6_PMS_Sample_D.tif"
    b = sweep.parse(["--images", "a.tif", "b.tif", "--k", "2", "3", "-bc", "32"])
    assert sweep.points(b) == [("a.tif", 2), ("a.tif", 3), ("b.tif", 2), ("b.tif", 3)]
    from lbdrn_hip import shard
    dealt = [shard.assign(len(pts), r, 8) for r in range(8)]
    assert sorted(i for part in dealt for i in part) == list(range(78))
    assert max(map(len, dealt)) - min(map(len, dealt)) <= 1


def test_results_summary_csv(tmp_path):
    import csv
    import results_summary as R
    out = tmp_path / "outs"
    names = ["imgA", "imgB"]
    for name in names:
        for K in (1, 2):
            d = out / f"{name}_r1_K{K}_bc64_nl2_D2_prec16_lr0.001_bs8192_e10"
            d.mkdir(parents=True)
            (d / "decode.txt").write_text(
                "[t] Binstream: x\n[t] Time elapsed: 1.5\n"
                f"[t] MSE: {10.5 * K}\n[t] PSNR: {60.25 - K}\n[t] Total size: {1000 * K} bytes, bpsp={0.125 * K}\n")
            (d / "encode.txt").write_text("[t] nn: 300 bytes, bpsp=0.01\n[t] MSB: 700 bytes: bpsp=0.02\n[t] Time elapsed: 9.0\n")
    (out / "imgB_r1_K2_bc64_nl2_D2_prec16_lr0.001_bs8192_e10" / "decode.txt").unlink()   # a missing point
    path = R.save_to_csv(["-o", str(out), "--files"] + names + ["--k", "1", "3"])
    assert path.endswith("results_r1_bc64_nl2_D2_prec16_lr0.001_bs8192_e10.csv")
    rows = list(csv.reader(open(path)))
    assert rows[0] == ["K"] + [f"{n}_{m}" for n in names for m in ("MSE", "PSNR", "bpsp", "bits")]
    assert rows[1] == ["K1", "10.5", "59.25", "0.125", "8000.0", "10.5", "59.25", "0.125", "8000.0"]
    assert rows[2] == ["K2", "21.0", "58.25", "0.25", "16000.0", "", "", "", ""]
    assert rows[3] == ["K3"] + [""] * 8
    assert R.extract_metrics1(str(out / "imgA_r1_K1_bc64_nl2_D2_prec16_lr0.001_bs8192_e10" / "encode.txt")) == (2400.0, 5600.0)
    # default image lists: the GF-6 samples are tabulated only for the headline configuration
    a = R.parse_args(["-o", "o"])
    assert len(R.default_files(a, "o/results_r1_bc64_nl2_D2_prec16_lr0.001_bs8192_e10.csv")) == 13
    assert len(R.default_files(a, "o/results_r1_bc256_nl2_D2_prec16_lr0.001_bs8192_e10.csv")) == 5
