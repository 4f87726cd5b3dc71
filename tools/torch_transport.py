"""A scarplet_amd.dist transport over torch.distributed (launcher-side code:
bench.py, tools/dist_check.py and the gloo tests use it; the package itself
imports no process-group library)."""
import numpy as np


class TorchTransport(object):
    """broadcast_bytes / exchange / gather of scarplet_amd/dist.py's transport
    protocol on an initialised torch.distributed process group (gloo for host
    arrays)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def broadcast_bytes(self, payload):
        box = [payload]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def exchange(self, sends, recvs):
        import torch
        reqs, bufs = [], []
        for (peer, tag, arr) in sends:
            reqs.append(self.dist.isend(torch.from_numpy(np.ascontiguousarray(arr)), dst=peer,
                                        group=self.group, tag=tag))
        for (peer, tag, shape) in recvs:
            t = torch.empty(tuple(shape), dtype=torch.float64)
            reqs.append(self.dist.irecv(t, src=peer, group=self.group, tag=tag))
            bufs.append(t)
        for r in reqs:
            r.wait()
        return [t.numpy() for t in bufs]

    def gather(self, obj, dst):
        rank = self.dist.get_rank(self.group)
        out = [None] * self.dist.get_world_size(self.group) if rank == dst else None
        self.dist.gather_object(obj, out, dst=dst, group=self.group)
        return out
