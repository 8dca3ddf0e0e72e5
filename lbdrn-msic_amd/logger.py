"""Run log of the codec CLIs: every record goes to stdout and to a text file in the output directory.

Interface kept from the reference (ref logger.py:9-18): module attribute `log` with `.info()`,
`create_logger(folder, file_name, log_file_only=False)`, `destroy_logger()`, record layout
"[timestamp] message".  The message texts are an interface of their own: results_summary.py in the
reference scrapes `MSE:`, `PSNR:`, `bpsp=`, `Total size: N bytes`, `Time elapsed:`, `nn: N bytes` and
`MSB: N bytes` out of these files.
"""
import logging
import pathlib
import sys

RECORD_LAYOUT = "[%(asctime)s] %(message)s"
log = None


class _RunLog(logging.Logger):
    """A private logger (not the root logger): repeated CLI runs in one process do not stack handlers."""

    def __init__(self, targets):
        super().__init__("lbdrn", level=logging.INFO)
        fmt = logging.Formatter(RECORD_LAYOUT)
        for h in targets:
            h.setFormatter(fmt)
            self.addHandler(h)

    def shutdown(self):
        while self.handlers:
            h = self.handlers.pop()
            h.close()


def create_logger(exp_folder, file_name, log_file_only=False):
    global log
    if log is not None:
        log.shutdown()
    targets = [] if log_file_only else [logging.StreamHandler(sys.stdout)]
    if file_name != "":
        path = pathlib.Path(exp_folder) / file_name
        path.parent.mkdir(parents=True, exist_ok=True)
        targets.append(logging.FileHandler(path, mode="w"))
    log = _RunLog(targets)


def destroy_logger():
    if log is not None:
        log.shutdown()
