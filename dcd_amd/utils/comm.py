"""Process-group helpers with the reference's names (DGDE/utils/comm.py:18-80)."""
import torch.distributed as dist


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def get_rank():
    return dist.get_rank() if _ready() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    """Barrier across all ranks (no-op for a single process)."""
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()
