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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12345")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    # ADM_DIST_BACKEND=gloo: rehearse several ranks on ONE GPU (RCCL refuses two ranks per device); real runs use nccl = RCCL
    backend = os.environ.get("ADM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(dev())
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


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0
