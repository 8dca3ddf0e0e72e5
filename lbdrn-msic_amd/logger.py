"""stdout + file logger with the reference's record format (ref logger.py:9-18); the text of the
records is an interface: results_summary.py scrapes it with regular expressions."""
import logging
import os
import sys

log = None


def create_logger(exp_folder, file_name, log_file_only=False):
    global log
    handlers = []
    if not log_file_only:
        handlers.append(logging.StreamHandler(sys.stdout))
    if file_name:
        target = os.path.join(exp_folder, file_name)
        os.makedirs(os.path.dirname(target) or ".", exist_ok=True)
        handlers.append(logging.FileHandler(target, mode="w"))
    for h in list(logging.root.handlers):
        logging.root.removeHandler(h)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s", handlers=handlers)
    log = logging.getLogger()


def destroy_logger():
    for h in list(log.handlers):
        h.close()
        log.removeHandler(h)
