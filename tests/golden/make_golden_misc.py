"""Generate tests/golden/misc.json by RUNNING the reference's constants.py and logger.py (authoring container
only; needs /root/reference): the module-level switches, and the layout of a record written through
create_logger()/log.info() with every digit replaced by 'd' (the timestamp itself is not data)."""
import json
import os
import re
import sys
import tempfile

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    import constants as RC
    import logger as RL
    consts = {k: getattr(RC, k) for k in dir(RC) if k.isupper()}
    with tempfile.TemporaryDirectory() as d:
        RL.create_logger(os.path.join(d, "sub", "dir"), "run.txt", log_file_only=True)
        RL.log.info("MSE: 1.25")
        RL.log.info(f"Total size: {1234} bytes, bpsp={0.5}")
        RL.destroy_logger()
        with open(os.path.join(d, "sub", "dir", "run.txt")) as f:
            lines = f.read().splitlines()
    layout = [re.sub(r"\d", "d", re.match(r"^\[[^\]]*\]", ln).group(0)) + ln[ln.index("]") + 1:] for ln in lines]
    with open(os.path.join(OUT, "misc.json"), "w") as f:
        json.dump({"constants": consts, "log_layout": layout}, f, indent=1, sort_keys=True)
    print(consts, layout)


if __name__ == "__main__":
    main()
