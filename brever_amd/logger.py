"""Root logger to stdout + file with an optional rank tag (brever/logger.py:5-37)."""
import logging
import sys


def set_logger(log_file=None, ddp=False, rank=0, level=logging.INFO):
    tag = f'[rank {rank}] ' if ddp else ''
    fmt = logging.Formatter(f'%(asctime)s [%(levelname)s] {tag}%(message)s')
    root = logging.getLogger()
    root.setLevel(level)
    for h in list(root.handlers):
        root.removeHandler(h)
    sh = logging.StreamHandler(sys.stdout)
    sh.setFormatter(fmt)
    root.addHandler(sh)
    if log_file is not None:
        fh = logging.FileHandler(log_file)
        fh.setFormatter(fmt)
        root.addHandler(fh)
