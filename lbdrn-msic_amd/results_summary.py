"""Collect the rate-distortion numbers of a sweep into one CSV (ref results_summary.py:79-137).

Same command line, same CSV name (`results_r{sr}_bc{bc}_nl{nl}_D{D}_prec{prec}_lr{lr}_bs{bs}_e{e}.csv` in the
output directory), same layout -- one row per K ("K1".."K11"), four columns per image (`<image>_MSE`,
`_PSNR`, `_bpsp`, `_bits`) -- and the same log records scraped (`MSE:`, `PSNR:`, `bpsp=`,
`Total size: N bytes` from decode.txt; `nn: N bytes`, `MSB: N bytes`, `Time elapsed:` for the printed
encode-side breakdown).  `--files` / `--k` replace the reference's hard-wired image list and K range.
"""
import argparse
import csv
import os
import re
import sys

TRIPLESAT = [
    "TRIPLESAT_2_MS_L1_20191107021947_001FFCVI_002_0120200811001001_001",
    "TRIPLESAT_2_MS_L1_20191107021950_001FFCVI_003_0120200811001001_002",
    "TRIPLESAT_2_MS_L1_20191107021954_001FFCVI_004_0120200811001001_001",
    "TRIPLESAT_2_MS_L1_20200109023258_002115VI_003_0120200811001001_001",
    "TRIPLESAT_2_MS_L1_20200109023301_002115VI_004_0120200811001001_001",
]
GF6 = [f"GF6_{sensor}_Sample_{s}" for sensor in ("WFI", "PMS") for s in "ABCD"]
METRICS = ("MSE", "PSNR", "bpsp", "bits")

# record -> (metric, scale); the last match in a file wins, like the reference's line loop
DECODE_RECORDS = {
    "MSE": (re.compile(r"MSE: (\d+\.\d+)"), 1.0),
    "PSNR": (re.compile(r"PSNR: (\d+\.\d+)"), 1.0),
    "bpsp": (re.compile(r"bpsp=(\d+\.\d+)"), 1.0),
    "bits": (re.compile(r"Total size: (\d+) bytes"), 8.0),
}
ELAPSED = re.compile(r"Time elapsed: (\d+\.\d+)")
ENCODE_RECORDS = {"nn": re.compile(r"nn: (\d+) bytes"), "MSB": re.compile(r"MSB: (\d+) bytes")}


def run_dir(args, name, K):
    return (f"{args.output_dir}/{name}_r{args.split_ratio}_K{K}_bc{args.base_channel}_nl{args.num_layers}"
            f"_D{args.D}_prec{args.precision}_lr{args.lr}_bs{args.batch_size}_e{args.epochs}")


def extract_metrics(file_path):
    """decode.txt -> {"MSE", "PSNR", "bpsp", "bits"} (those present)."""
    found = {}
    with open(file_path) as f:
        for line in f:
            for metric, (pattern, scale) in DECODE_RECORDS.items():
                m = pattern.search(line)
                if m:
                    found[metric] = scale * float(m.group(1))
            m = ELAPSED.search(line)
            if m:
                print(f"decTime: {m.group(1)}")
    return found


def extract_metrics1(file_path):
    """encode.txt -> prints the payload split; returns (nn_bits, MSB_bits)."""
    bits = {"nn": 0.0, "MSB": 0.0}
    with open(file_path) as f:
        for line in f:
            for key, pattern in ENCODE_RECORDS.items():
                m = pattern.search(line)
                if m:
                    bits[key] = 8 * float(m.group(1))
                    print(f"{key} bits: {bits[key]}")
            m = ELAPSED.search(line)
            if m:
                print(f"encTime: {m.group(1)}")
    total = bits["nn"] + bits["MSB"]
    if total:
        print(100 * bits["MSB"] / total)
    return bits["nn"], bits["MSB"]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Results Summary")
    p.add_argument("-o", "--output_dir", default="outputs-rel-colors-D2", type=str, help="output dir")
    p.add_argument("-sr", "--split_ratio", type=int, default=1, help="tile size (default: 1)")
    p.add_argument("-bc", "--base_channel", type=int, default=64, help="base channel (default: 64)")
    p.add_argument("-nl", "--num_layers", type=int, default=2, help="Number of layers (default: 2)")
    p.add_argument("-D", "--D", type=int, default=2, help="#neighbors (2D+1)^2")
    p.add_argument("-prec", "--precision", type=int, default=16, help=" (default: 16)")
    p.add_argument("-lr", "--lr", type=float, default=1e-3, help="learning rate (default: 1e-3)")
    p.add_argument("-bs", "--batch_size", type=int, default=8192, help="batch size (default: 8192)")
    p.add_argument("-e", "--epochs", type=int, default=10, help="number of epochs to train (default: 10)")
    p.add_argument("--files", nargs="*", default=None,
                   help="image names (without extension); default: the reference's GF-2 / GF-6 lists")
    p.add_argument("--k", nargs=2, type=int, default=(1, 11), metavar=("FIRST", "LAST"),
                   help="K range, inclusive (default: 1 11)")
    return p.parse_args(argv)


def default_files(args, csv_file):
    # the reference tabulates the GF-6 images only for its headline configuration (results_summary.py:95-110)
    headline = f"{args.output_dir}/results_r1_bc64_nl2_D2_prec16_lr0.001_bs8192_e10.csv"
    return TRIPLESAT + GF6 if csv_file == headline else list(TRIPLESAT)


def save_to_csv(argv=None):
    args = parse_args(argv)
    csv_file = (f"{args.output_dir}/results_r{args.split_ratio}_bc{args.base_channel}_nl{args.num_layers}"
                f"_D{args.D}_prec{args.precision}_lr{args.lr}_bs{args.batch_size}_e{args.epochs}.csv")
    files = args.files if args.files else default_files(args, csv_file)
    os.makedirs(args.output_dir, exist_ok=True)
    with open(csv_file, "w", newline="") as out:
        writer = csv.writer(out)
        writer.writerow(["K"] + [f"{name}_{metric}" for name in files for metric in METRICS])
        for K in range(args.k[0], args.k[1] + 1):
            row = [f"K{K}"]
            for i, name in enumerate(files):
                print(f"Processing {name} for K={K}")
                folder = run_dir(args, name, K)
                try:
                    found = extract_metrics(os.path.join(folder, "decode.txt"))
                    row += [found[m] for m in METRICS]
                except (OSError, KeyError):
                    print("decode.txt does not exists.")
                    row += [None] * len(METRICS)
                if i == 0 and K == 3 and os.path.exists(os.path.join(folder, "encode.txt")):
                    extract_metrics1(os.path.join(folder, "encode.txt"))
            writer.writerow(row)
    print(f"All results have been written to {csv_file}")
    return csv_file


if __name__ == "__main__":
    save_to_csv()
    sys.exit(0)
