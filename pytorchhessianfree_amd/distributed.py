"""Data-parallel sum of the curvature partials across ranks.

Default: ``torch.distributed.all_reduce`` (backend ``nccl`` is RCCL on ROCm; ``gloo``
in the CPU tests).  Optional (``HF_RCCL_DIRECT=1``): the library's own RCCL
communicator (``hf_comm_*`` / ``hf_allreduce_sum`` of include/hf_pcg.h), which
enqueues the collective on the SAME stream as the PCG kernels -- no hand-off to
the process group's stream and back.  The communicator is bootstrapped through
the existing process group (rank 0's unique id is broadcast), so both paths need
``torch.distributed`` to be initialised.
"""

import ctypes
import os

import torch

from . import _lib

_comms = {}


class _DirectComm:
    def __init__(self, group):
        dist = torch.distributed
        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.hf_comm_unique_id(uid), "hf_comm_unique_id")
        payload = [uid.raw if rank == 0 else None]
        dist.broadcast_object_list(payload, src=dist.get_global_rank(group, 0), group=group)
        self.handle = _lib.c_void_p()
        _lib.check(lib.hf_comm_create(ctypes.byref(self.handle), payload[0], world, rank),
                   "hf_comm_create")
        self.lib = lib

    def all_reduce_sum(self, t):
        _lib.check(
            self.lib.hf_allreduce_sum(self.handle, _lib.c_void_p(t.data_ptr()), t.numel(),
                                      _lib.dtype_code(t.dtype), _lib.current_stream_ptr(t.device)),
            "hf_allreduce_sum")
        return t


def use_direct_rccl(t):
    return bool(os.environ.get("HF_RCCL_DIRECT")) and t.is_cuda and t.is_contiguous()


def all_reduce_sum(t, group):
    """In-place sum of ``t`` over ``group`` (``None``: single process, no-op)."""
    if group is None:
        return t
    if use_direct_rccl(t):
        comm = _comms.get(id(group))
        if comm is None:
            comm = _comms[id(group)] = _DirectComm(group)
        return comm.all_reduce_sum(t)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t
