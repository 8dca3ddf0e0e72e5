"""Generate tests/golden/summary.json by RUNNING the reference's results_summary.extract_metrics on log files
written here (authoring container only; needs /root/reference).  Data only: the log texts (this repo's own
sample records in the formats the CLIs write) and the dictionaries the reference's parser returns for them."""
import json
import os
import sys
import tempfile

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

LOGS = {
    "complete": "[2024-10-22 10:00:00,000] Binstream: outputs/x/x.bin\n"
                "[2024-10-22 10:00:01,000] Recon: outputs/x/x_recon.tif\n"
                "[2024-10-22 10:00:01,500] Time elapsed: 1.5003\n"
                "[2024-10-22 10:00:02,000] MSE: 81.19549560546875\n"
                "[2024-10-22 10:00:02,000] PSNR: 60.90478515625\n"
                "[2024-10-22 10:00:02,000] Total size: 12912168 bytes, bpsp=3.078500747680664\n",
    "no_metrics": "[t] Binstream: a.bin\n[t] Time elapsed: 0.25\n",
    "twice": "[t] MSE: 1.5\n[t] PSNR: 70.25\n[t] Total size: 10 bytes, bpsp=0.5\n"
             "[t] MSE: 2.5\n[t] PSNR: 68.0\n[t] Total size: 20 bytes, bpsp=0.75\n",
    "integers_do_not_match": "[t] MSE: 3\n[t] PSNR: 60\n[t] Total size: 7 bytes, bpsp=1\n",
}


def main():
    sys.path.insert(0, REF)
    import results_summary as RS
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, text in LOGS.items():
            path = os.path.join(d, name + ".txt")
            with open(path, "w") as f:
                f.write(text)
            out[name] = {"log": text, "metrics": RS.extract_metrics(path)}
    with open(os.path.join(OUT, "summary.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: v["metrics"] for k, v in out.items()})


if __name__ == "__main__":
    main()
