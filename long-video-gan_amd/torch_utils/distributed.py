"""Process-group bootstrap for one-process-per-GPU data parallelism over RCCL/xGMI
(reference torch_utils/distributed.py:19-73). backend "nccl" IS RCCL on ROCm; `gloo` is used
when no GPU is visible so the same code path runs in CPU tests."""

import os

import torch
import torch.distributed as dist

from . import custom_ops, training_stats

_sync_device = None


def get_local_rank() -> int:
    return int(os.environ.get('LOCAL_RANK', '0'))


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init(temp_dir: str = None, backend: str = None):
    """Fill in single-process defaults for the torchrun environment, bind this process to
    its GPU and create the default process group."""
    global _sync_device
    # 127.0.0.1, not "localhost": container hostnames do not always resolve.
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29500 + (os.getpid() % 16384)))
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('LOCAL_RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC only on this driver

    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(get_local_rank())
    if backend is None:
        backend = 'nccl' if on_gpu else 'gloo'
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method='env://')

    _sync_device = (torch.device('cuda') if on_gpu else torch.device('cpu')) if get_world_size() > 1 else None
    training_stats.init_multiprocessing(rank=get_rank(), sync_device=_sync_device)
    if get_rank() != 0:
        custom_ops.verbosity = 'none'
