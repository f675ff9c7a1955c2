"""Logging helpers - counterpart of BOBE/utils/log.py (same function names; the package logs under ``bobe_amd.<name>``)."""
from __future__ import annotations

import logging
import sys


def get_logger(name: str) -> logging.Logger:
    """BOBE/utils/log.py:102-115."""
    return logging.getLogger(f"bobe_amd.{name}")


def setup_logging(verbosity="INFO", log_file=None):
    """BOBE/utils/log.py:30-100: the package logger's level, a stdout handler (INFO and below) and a stderr handler
    (WARNING and above), optionally a file.  Idempotent: handlers installed by an earlier call are replaced."""
    root = logging.getLogger("bobe_amd")
    level = getattr(logging, str(verbosity).upper(), logging.INFO)
    root.setLevel(level)
    for h in [h for h in root.handlers if getattr(h, "_bobe_amd", False)]:
        root.removeHandler(h)
    fmt = logging.Formatter("%(asctime)s %(name)s %(levelname)s: %(message)s")
    out = logging.StreamHandler(sys.stdout)
    out.addFilter(lambda r: r.levelno < logging.WARNING)
    err = logging.StreamHandler(sys.stderr)
    err.setLevel(logging.WARNING)
    handlers = [out, err]
    if log_file is not None:
        handlers.append(logging.FileHandler(log_file))
    for h in handlers:
        h.setFormatter(fmt)
        h._bobe_amd = True
        root.addHandler(h)
    return root


def update_verbosity(verbosity):
    """BOBE/utils/log.py:117-: change the level of the package logger."""
    logging.getLogger("bobe_amd").setLevel(getattr(logging, str(verbosity).upper(), logging.INFO))
