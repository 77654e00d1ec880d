"""`logger.log` with the reference's text format (guided_diffusion/logger.py:247): one line per call to
stdout and, after configure(dir), to <dir>/log.txt (users grep it for "top", GD/README.md:24)."""
import os
import sys

_file = None
_dir = None


def configure(dir=None, **_):
    global _file, _dir
    if dir:
        os.makedirs(dir, exist_ok=True)
        _dir = dir
        _file = open(os.path.join(dir, "log.txt"), "a")


def get_dir():
    return _dir


def log(*args):
    line = " ".join(str(a) for a in args)
    print(line, file=sys.stdout, flush=True)
    if _file is not None:
        _file.write(line + "\n")
        _file.flush()
