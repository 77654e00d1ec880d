"""Process-group helpers with the reference's names (guided_diffusion/dist_util.py:21-63).

One process per GPU; backend "nccl" on ROCm IS RCCL (xGMI intra-node), gloo without a GPU.
Checkpoints are read with torch.load from local disk on every rank (the reference's blobfile
indirection and its sync_params broadcast are training-time features, out of scope).
"""
import os

import torch
import torch.distributed as dist

GPUS_PER_NODE = 8


def setup_dist():
    if dist.is_initialized():
        return
    # the pool's host driver offers only dmabuf IPC: with the legacy IPC mode RCCL's intra-node transport setup (and any
    # CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid argument`
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12345")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    # ADM_DIST_BACKEND=gloo: rehearse several ranks on ONE GPU (RCCL refuses two ranks per device); real runs use nccl = RCCL
    backend = os.environ.get("ADM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(dev())
    if backend == "nccl":   # bind the communicator to this rank's GPU at once (eager RCCL init: a bad device shows up here, not in the first collective)
        dist.init_process_group(backend=backend, init_method="env://", device_id=dev())
    else:
        dist.init_process_group(backend=backend, init_method="env://")


def dev():
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        return torch.device(f"cuda:{local % max(1, min(GPUS_PER_NODE, torch.cuda.device_count()))}")
    return torch.device("cpu")


def load_state_dict(path, **kwargs):
    kwargs.setdefault("map_location", "cpu")
    kwargs.setdefault("weights_only", True)
    return torch.load(path, **kwargs)


def collectives_on(group=None):
    """True when the data path's collectives must run: a process group with more than one rank -- or ANY initialised group when
    ADM_FORCE_COLLECTIVES=1 (bench.py --force-dist, tests/test_hip_rccl.py): a world-size-1 `nccl` group then takes every
    all_gather / all_reduce / barrier of this code base through RCCL on a one-GPU box instead of the single-rank short-cuts."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("ADM_FORCE_COLLECTIVES", "0") == "1"


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0
