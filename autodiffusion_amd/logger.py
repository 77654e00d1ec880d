"""`logger.log` with the reference's text format (guided_diffusion/logger.py:247): one line per call to
stdout and, after configure(dir), to <dir>/log.txt (users grep it for "top", GD/README.md:24).

Ranks > 0 write `<dir>/log-rank%03i.txt` and nothing to stdout, as the reference's configure() does for them
(logger.py:456-464): the 'epoch = i : top k result' / 'No.j ... fid' lines appear once, in rank 0's log.txt."""
import os
import sys

_file = None
_dir = None
_stdout = True


def _rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:  # noqa: BLE001 -- logging must not depend on the process group
        pass
    return int(os.environ.get("RANK", "0"))


def configure(dir=None, **_):
    global _file, _dir, _stdout
    rank = _rank()
    _stdout = rank == 0
    if dir:
        os.makedirs(dir, exist_ok=True)
        _dir = dir
        if _file is not None:
            _file.close()
        _file = open(os.path.join(dir, "log.txt" if rank == 0 else "log-rank%03i.txt" % rank), "a")


def get_dir():
    return _dir


def log(*args):
    line = " ".join(str(a) for a in args)
    if _stdout:
        print(line, file=sys.stdout, flush=True)
    if _file is not None:
        _file.write(line + "\n")
        _file.flush()


def warn(*args):
    """Like log(), but to stderr (stdout may be a machine-read channel: bench.py prints ONE JSON line there)."""
    line = " ".join(str(a) for a in args)
    print(line, file=sys.stderr, flush=True)
    if _file is not None:
        _file.write(line + "\n")
        _file.flush()
