"""Run log of the codec CLIs: every record goes to stdout and to a text file in the output directory.

Interface kept from the reference (ref logger.py:9-18): module attribute `log` with `.info()`,
`create_logger(folder, file_name, log_file_only=False)`, `destroy_logger()`, record layout
"[timestamp] message".  The message texts are an interface of their own: results_summary.py in the
reference scrapes `MSE:`, `PSNR:`, `bpsp=`, `Total size: N bytes`, `Time elapsed:`, `nn: N bytes` and
`MSB: N bytes` out of these files.
"""
import contextlib
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


class _Collector(logging.Handler):
    def __init__(self, sink):
        super().__init__()
        self.sink = sink
        self.setFormatter(logging.Formatter(RECORD_LAYOUT))

    def emit(self, record):
        self.sink.append(self.format(record))


@contextlib.contextmanager
def capture(exclusive=True):
    """Collect the formatted records written inside the block (a rank of a multi-GPU run keeps the records
    of its tiles until rank 0 writes them out in tile order).  exclusive: the other handlers stay silent."""
    lines = []
    held = log.handlers[:] if exclusive else []
    for h in held:
        log.removeHandler(h)
    collector = _Collector(lines)
    log.addHandler(collector)
    try:
        yield lines
    finally:
        log.removeHandler(collector)
        for h in held:
            log.addHandler(h)


def replay(lines):
    """Write records captured elsewhere, already formatted, to this run's targets."""
    for h in log.handlers:
        stream = getattr(h, "stream", None)
        if stream is not None:
            for line in lines:
                stream.write(line + "\n")
            h.flush()


def destroy_logger():
    if log is not None:
        log.shutdown()
