"""TEST INFRASTRUCTURE: host-side stand-ins for the RCCL exchange steps of pandora_amd.comm.Comm.

The product transport is RCCL inside libpandora_amd.so and nothing else (pandora_amd/comm.py has no switch).  RCCL refuses two
ranks on one device and needs a GPU at all, so the exchange STEPS of pandora_amd.dist (which buffer is reduced with which
operator, who owns which rows) are exercised on test boxes through these subclasses, which a test constructs explicitly:

  TcpComm   the exchange buffer visits the host (pmx_xbuf_download / _upload) and is reduced through the rendezvous socket of
            rank 0 - several ranks on the one GPU of a test box, or host-only arithmetic with engine=None
  GlooComm  the same through an initialised torch.distributed gloo group (CPU tests, world_size 2)

bench.py reaches them only through its explicit `--test-comm tests.transports:TcpComm` hook.
"""
import os

import numpy as np

from pandora_amd.comm import Comm, Rendezvous, env_world
from pandora_amd.dist import shard_range


class _HostComm(Comm):
    """exchange buffers go through the host; subclasses say how host arrays are reduced / gathered"""

    nranks_note = "host test transport"

    def _bootstrap(self, addr, port):
        raise NotImplementedError

    @property
    def nranks(self):
        return self.world

    def _gather_bytes(self, raw):
        raise NotImplementedError

    def allreduce_xbuf(self, which, op):
        if self.world == 1 and not self.always:
            return
        host = self.engine.xbuf_download(which)
        self.engine.xbuf_upload(which, np.asarray(self.host_allreduce(host, op)).astype(host.dtype, copy=False))

    def allgather_rows(self, H, with_itp):
        if self.world == 1 and not self.always:
            return
        lo, hi = shard_range(H, self.world, self.rank)
        for which in ("full_disp", "full_validity") + (("full_itp",) if with_itp else ()):
            full = self.engine.xbuf_download(which).reshape(H, -1)
            parts = self._gather_bytes(np.ascontiguousarray(full[lo:hi]).tobytes())
            for r, blob in enumerate(parts):
                rlo, rhi = shard_range(H, self.world, r)
                full[rlo:rhi] = np.frombuffer(blob, full.dtype).reshape(rhi - rlo, -1)
            self.engine.xbuf_upload(which, full)

    def gather_rows(self, H, with_itp, root=0):
        self.allgather_rows(H, with_itp)  # everybody gets everything

    def close(self):
        if self.rdv is not None:
            self.rdv.close()


class TcpComm(_HostComm):
    def _bootstrap(self, addr, port):
        port = int(os.environ.get("PANDORA_COMM_PORT", port + 1))
        self.rdv = Rendezvous(self.rank, self.world, addr, port)

    def host_allreduce(self, arr, op):
        if self.world == 1:
            return np.asarray(arr)
        return self.rdv.allreduce(np.asarray(arr), op)

    def barrier(self):
        if self.world == 1:
            return
        if self.engine is not None:
            self.engine.sync()
        self.rdv.barrier()

    def _gather_bytes(self, raw):
        return self.rdv.allgather(raw)


class GlooComm(_HostComm):
    def __init__(self, engine=None, always=False):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("the gloo test transport needs an initialised torch.distributed group")
        self._dist = dist
        erank, eworld, _, _, _ = env_world()
        self.engine, self.always, self.rdv = engine, always, None
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def host_allreduce(self, arr, op):
        if self.world == 1:
            return np.asarray(arr)
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        red = {"min": self._dist.ReduceOp.MIN, "sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX}[op]
        self._dist.all_reduce(t, op=red)
        return t.numpy()

    def barrier(self):
        if self.world == 1:
            return
        if self.engine is not None:
            self.engine.sync()
        self._dist.barrier()

    def _gather_bytes(self, raw):
        parts = [None] * self.world
        self._dist.all_gather_object(parts, raw)
        return parts
