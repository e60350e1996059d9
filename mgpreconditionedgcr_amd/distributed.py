"""Multi-GPU host side: one process per GPU, row-block partition.

`Comm` wraps the library's communicator.  Two transports:
  * Comm.rccl(...)  — RCCL over xGMI, created inside libmgcr_hip.so; the 128-byte unique id is
    broadcast with torch.distributed (any backend) by the launcher;
  * Comm.host(...)  — host-staged callbacks over a torch.distributed (gloo) group: slow, for
    bring-up and tests (several ranks may share one GPU, or no GPU at all for the plan logic).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ALLREDUCE_CB, EXCHANGE_CB, check
from .api import Operator, c128


class Comm:
    def __init__(self, handle, rank, size, keep=()):
        self.h = handle
        self.rank, self.size = rank, size
        self._keep = list(keep)

    @classmethod
    def rccl(cls, dist):
        """Create the RCCL communicator; `dist` is an initialised torch.distributed module/group."""
        import torch
        _lib.init()
        rank, size = dist.get_rank(), dist.get_world_size()
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            check(_lib.lib().mgcr_rccl_unique_id(buf))
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        ident = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        check(_lib.lib().mgcr_comm_create_rccl(rank, size, ident, C.byref(h)))
        return cls(h, rank, size)

    @classmethod
    def host(cls, dist):
        """Host-staged transport over torch.distributed (gloo)."""
        import torch
        rank, size = dist.get_rank(), dist.get_world_size()

        def allreduce(user, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                t = torch.from_numpy(a.copy())
                dist.all_reduce(t)
                a[:] = t.numpy()
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("allreduce callback failed:", e)
                return 1

        def exchange(user, npeers, peers, send, scount, recv, rcount):
            try:
                reqs, outs = [], []
                for p in range(npeers):
                    if rcount[p]:
                        t = torch.empty(rcount[p], dtype=torch.float64)
                        reqs.append(dist.irecv(t, src=peers[p]))
                        outs.append((p, t))
                for p in range(npeers):
                    if scount[p]:
                        a = np.ctypeslib.as_array(send[p], shape=(scount[p],))
                        reqs.append(dist.isend(torch.from_numpy(a.copy()), dst=peers[p]))
                for r in reqs:
                    r.wait()
                for p, t in outs:
                    np.ctypeslib.as_array(recv[p], shape=(rcount[p],))[:] = t.numpy()
                return 0
            except Exception as e:
                print("exchange callback failed:", e)
                return 1

        ar, ex = ALLREDUCE_CB(allreduce), EXCHANGE_CB(exchange)
        h = C.c_void_p()
        check(_lib.lib().mgcr_comm_create_host(rank, size, ar, ex, None, C.byref(h)))
        return cls(h, rank, size, keep=(ar, ex))

    @property
    def allreduce_kind(self):
        """How a solve sums its per-iteration scalars over the ranks: "host", "rccl" or "peer-write"
        (decided by a self-test when the first distributed operator is created on this communicator)."""
        k = C.c_int32()
        check(_lib.lib().mgcr_comm_allreduce_kind(self.h, C.byref(k)))
        return ("host", "rccl", "peer-write")[k.value]

    def allreduce_sum(self, values):
        """In-place-style sum over the ranks of a few host doubles (collective); returns the summed array."""
        buf = np.ascontiguousarray(values, np.float64).copy()
        check(_lib.lib().mgcr_comm_allreduce_sum(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size))
        return buf

    def dot(self, a, b):
        """Global <a, b> (conj on a) of two Fields distributed over this communicator."""
        d = a.dot(b)
        s = self.allreduce_sum([d.real, d.imag])
        return complex(s[0], s[1])

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().mgcr_comm_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Plan:
    """Partition plan of one row block (pure host logic, no GPU needed)."""

    def __init__(self, comm, n_global, row0, rowptr, col_global):
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        col_global = np.ascontiguousarray(col_global, np.int64)
        self.nloc = rowptr.size - 1
        self.nnz = int(rowptr[-1])
        h = C.c_void_p()
        check(_lib.lib().mgcr_plan_create(comm.h, n_global, row0, self.nloc, rowptr.ctypes.data, col_global.ctypes.data, C.byref(h)))
        self.h = h
        nh, npeers, ib, ie = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
        check(_lib.lib().mgcr_plan_info(h, C.byref(nh), C.byref(npeers), C.byref(ib), C.byref(ie)))
        self.n_halo, self.npeers = nh.value, npeers.value
        self.interior = (ib.value, ie.value)
        self.peers = np.empty(self.npeers, np.int32)
        self.send_counts = np.empty(self.npeers, np.int64)
        self.recv_counts = np.empty(self.npeers, np.int64)
        check(_lib.lib().mgcr_plan_peers(h, self.peers.ctypes.data, self.send_counts.ctypes.data, self.recv_counts.ctypes.data))
        self.col_local = np.empty(self.nnz, np.int64)
        check(_lib.lib().mgcr_plan_local_columns(h, self.col_local.ctypes.data))
        self.halo_globals = np.empty(self.n_halo, np.int64)
        check(_lib.lib().mgcr_plan_halo_globals(h, self.halo_globals.ctypes.data))
        self.send_rows = []
        for p in range(self.npeers):
            idx = np.empty(self.send_counts[p], np.int64)
            check(_lib.lib().mgcr_plan_send_indices(h, p, idx.ctypes.data))
            self.send_rows.append(idx)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().mgcr_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


class DistSparse(Operator):
    """Row block [row0, row0 + nrow_local) of a distributed Sparse; column indices are GLOBAL.
    Fields it applies to hold this rank's nrow_local entries."""

    def __init__(self, comm, n_global, row0, rowptr, col_global, val):
        super().__init__()
        _lib.init()
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        col_global = np.ascontiguousarray(col_global, np.int64)
        val = np.ascontiguousarray(val, c128)
        h = C.c_void_p()
        check(_lib.lib().mgcr_dcsr_create(comm.h, n_global, row0, rowptr.size - 1, rowptr.ctypes.data,
                                          col_global.ctypes.data, val.ctypes.data, C.byref(h)))
        self.h = h
        self._keep.append(comm)
        self.row0 = int(row0)
        self.comm = comm   # Fields this operator applies to are distributed over it (global dot products: Comm.dot)
        self._nnz = int(rowptr[-1])

    @property
    def halo_kind(self):
        """How the halo travels before an apply: "host", "rccl" or "peer-write" (self-tested at creation)."""
        k = C.c_int32()
        check(_lib.lib().mgcr_op_halo_kind(self.h, C.byref(k)))
        return ("host", "rccl", "peer-write")[k.value]


class DistHierarchicalSparse(Operator):
    """Block rows [brow0, brow0 + nbrow_local) of a distributed HierarchicalSparse (block-CSR of dense bs x bs blocks,
    src/HierarchicalSparse.h:22-48,101-161), from this rank's (block_row, block_col, block) triplets: block rows LOCAL
    (0 .. nbrow_local), block columns GLOBAL; duplicates kept and summed at apply time, triplets sorted stably by
    (row, col) like the single-GPU constructor.  Fields hold this rank's nbrow_local * bs entries."""

    def __init__(self, comm, nb_global, brow0, nbrow_local, rows_local, cols_global, blocks):
        super().__init__()
        _lib.init()
        rows_local = np.ascontiguousarray(rows_local, np.int64)
        cols_global = np.ascontiguousarray(cols_global, np.int64)
        blocks = np.ascontiguousarray(blocks, c128)
        nt = rows_local.size
        bs = int(round(np.sqrt(blocks.size // max(nt, 1)))) if nt else 1
        if nt and bs * bs * nt != blocks.size:
            raise _lib.MgcrError(1, "blocks must hold ntriplets square blocks")
        order = np.argsort(rows_local * int(nb_global) + cols_global, kind="stable")
        browptr = np.zeros(nbrow_local + 1, np.int32)
        np.add.at(browptr, rows_local + 1, 1)
        np.cumsum(browptr, out=browptr)
        bcol = np.ascontiguousarray(cols_global[order])
        blk = np.ascontiguousarray(blocks.reshape(nt, bs, bs)[order]) if nt else blocks
        h = C.c_void_p()
        check(_lib.lib().mgcr_dbcsr_create(comm.h, nb_global, brow0, nbrow_local, bs, browptr.ctypes.data, bcol.ctypes.data,
                                           blk.ctypes.data, C.byref(h)))
        self.h = h
        self._keep.append(comm)
        self.comm, self.bs = comm, bs
        self.row0 = int(brow0) * bs

    @property
    def halo_kind(self):
        k = C.c_int32()
        check(_lib.lib().mgcr_op_halo_kind(self.h, C.byref(k)))
        return ("host", "rccl", "peer-write")[k.value]
