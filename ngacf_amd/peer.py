"""Peer-store all-gather for the node-partitioned path (ngacf_amd/dist.py): `PeerComm` stands where `Comm` issues an RCCL
all-gather per half-table and stage.  A stage at Gowalla scale moves ~1 MB per rank, and a collective LAUNCH costs more than the
traversal it should hide behind; here a rank's block goes into its peers' tables by a 64-workgroup copy kernel
through hipIpc-mapped pointers, followed by one flag word per peer, and a consumer's traversal is preceded by a one-wave kernel
that waits on its own flag words (csrc/peer.hip: ngacf_peer_push / ngacf_peer_wait).  No host, no collective on the path; the
epochs live in device memory, so the captured hipGraph of a pass replays unchanged.

What has to hold for that, and how it is kept:
  symmetric tables   every gathered table (H of a stage; Gn and t of a backward stage) comes from a POOL of library-owned
                     allocations (ngacf_peer_alloc) whose hipIpc handles were exchanged once (all_gather_object on the process
                     group: host work, done on first use = the eager warm-up pass, never inside a capture).  All ranks run the
                     same program, so the k-th acquisition of a key picks the k-th buffer everywhere; a buffer goes back to the
                     pool when the autograd node that leased it is done (`_Lease`).
  no early overwrite a peer writes stage s of pass k+1 into my table only after it has consumed my stage s-1 of pass k+1 (s > 0)
                     or, for s = 0, after it has finished pass k, whose last stage needed a block that I sent AFTER my own
                     stage-0 traversals of pass k: with at least two stages a table is never written while its previous
                     content is still being read.  One-stage stacks are refused.
  control block      the flag / epoch words live in uncached (fine-grained) device memory (ngacf_peer_alloc_ctrl): they are polled
                     by running kernels while other agents write them.
  visibility         push: system-scope write-through stores -> vmcnt(0) -> last-workgroup ticket -> system-scope store of the
                     epoch per peer (no cache-wide write-back: it would flush the L2 under the traversal running beside it);
                     wait: system-scope loads of the flag words; the traversal is a NEW kernel (its start drops what its caches
                     held of the table).  A peer that never arrives costs `timeout_s`, sets an error word (check()) and the
                     kernel leaves: wrong numbers reported by the host, never a hung GPU.
  a step boundary    in training, a peer's next forward may only start after my backward has finished reading the tables it
                     will overwrite: the per-step all-reduce of the weight gradients (reduce_grads) is that barrier.
  sequential forwards a wait kernel SPINS: everything behind it in the same hardware queue waits with it.  As long as a rank has
                     its compute stream, this push stream and a warm-up stream, a push never lands behind a wait that needs a
                     peer's LATER push.  The step's two forwards as parallel branches (tools/peer_branches_probe.py) do not
                     finish: the working explanation is that branch, push and warm-up streams alias onto HIP's few hardware
                     queues and pushes sit behind the other branch's spinning wait on both ranks, so only time-outs move the
                     exchange (unproven: GPU_MAX_HW_QUEUES=8 changes nothing).  The sharded step keeps its forwards one after
                     the other with this transport too.
The push runs on a side stream (under the other half's traversal, like the asynchronous all-gather it replaces; overlap=False:
in line on the current stream, no fork); join() brings it back before a pass ends.  Everything that is not a pool table
(state-dict / evaluation gathers) takes the base class's path.  Validated as 2-4 processes on ONE GPU (tests/test_gpu_dist.py,
tools/peer_probe.py); between GPUs it has never run: opt-in (NGACF_PEER_GATHER=1 / peer_store=True).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from .dist import Comm, _Handle

SLOTS = 1024                      # flag slots: two per pool buffer (user half, item half)


class _Raw:
    """library-owned device memory as a __cuda_array_interface__ object (torch.as_tensor makes a non-owning view of it)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class _Buf:
    def __init__(self, ptr, nbytes, peers, index, device):
        self.ptr, self.nbytes, self.peers, self.index = ptr, nbytes, peers, index     # peers: rank -> this process's mapping
        self.bytes_view = torch.as_tensor(_Raw(ptr, nbytes), device=device)
        self.busy = False


class _Lease:
    """The pool buffers one autograd node holds; release() (idempotent; also on garbage collection) returns them."""

    def __init__(self):
        self.bufs = []

    def release(self):
        for b in self.bufs:
            b.busy = False
        self.bufs = []

    # No __del__: "the k-th acquisition of a key is the k-th buffer on every rank" holds only while buffers return to the pool at
    # the SAME program points on every rank.  release() is called at such points (the end of a no-grad forward, the backward of
    # the node that leased them); a lease that is dropped without it (a forward whose backward never runs) keeps its buffers --
    # a garbage collector returning them at a rank-dependent moment would let ranks pick different buffers, and a peer would
    # then write into a table that is still being read.  An abandoned lease costs memory and, in a loop, ends in the pool's
    # "out of flag slots" error: loud, never a silent wrong answer.


class PeerComm(Comm):
    peer_store = True

    def __init__(self, part, group=None, device=None, timeout_s=None):
        super().__init__(part, group)
        if part.P - 1 > _lib.MAX_PEERS:
            raise ValueError("peer-store all-gather: at most %d ranks" % (_lib.MAX_PEERS + 1))
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.rank, self.W = part.rank, part.P
        if timeout_s is None:       # a straggler's plan build, a checkpoint write or a GC pause must not read as "peer lost"
            timeout_s = float(os.environ.get("NGACF_PEER_TIMEOUT_S", "60"))
        self._refuse_unvalidated_topology()
        self.others = [r for r in range(self.W) if r != self.rank]
        self.side = torch.cuda.Stream(self.device)
        self.timeout_ticks = int(timeout_s * 100e6)           # s_memrealtime: 100 MHz
        self._bufs, self._pool, self._by_ptr, self._ar = [], {}, {}, {}
        # control block: flags[SLOTS][W] (written by the peers) | sent[SLOTS] | expect[SLOTS] | ticket[SLOTS] | err
        self._ctrl = self._symmetric(4 * (SLOTS * self.W + 3 * SLOTS + 1), ctrl=True)
        self.pushes = 0
        if self.W > 1:
            dist.barrier(group=self.group)                    # nobody pushes before every rank has mapped every control block

    def _refuse_unvalidated_topology(self):
        """The transport has only ever run as several processes on ONE GPU (one L2 serves all writers).  Between GPUs the flag
        polls and the traversals' reads of peer-written tables rest on memory-model behaviour that is not validated here (xGMI
        write acknowledgement order; a GPU's L2 against a peer's stores): refused until a two-GPU run of tools/peer_soak.py has
        passed on the machine -- NGACF_PEER_ALLOW_MULTI_GPU=1 is that operator's statement."""
        if self.W <= 1:
            return
        props = torch.cuda.get_device_properties(self.device)
        me = (os.uname().nodename, str(getattr(props, "uuid", "")) or "%s/%d" % (props.name, self.device.index))
        ids = [None] * self.W
        dist.all_gather_object(ids, me, group=self.group)
        if len(set(ids)) > 1 and os.environ.get("NGACF_PEER_ALLOW_MULTI_GPU") != "1":
            raise RuntimeError("peer-store all-gather between different GPUs (%d distinct devices among %d ranks) is not validated: "
                               "run `python tools/peer_soak.py --world 2` with one rank per GPU on this machine first, then set "
                               "NGACF_PEER_ALLOW_MULTI_GPU=1; the default transport (RCCL all-gather) needs nothing"
                               % (len(set(ids)), self.W))

    # ---- symmetric memory -------------------------------------------------------------------------------------------------
    def _symmetric(self, nbytes, ctrl=False):
        """nbytes of zeroed device memory on every rank, each rank holding a mapping of every peer's (collective, host side).
        ctrl: the control block -- uncached (fine-grained) memory, polled by running kernels while peers write it."""
        if 2 * len(self._bufs) + 1 >= SLOTS:                  # (every buffer owns two flag slots: 2 * index, 2 * index + 1)
            raise RuntimeError("peer-store pool: out of flag slots (%d buffers; leases dropped without release()?)" % len(self._bufs))
        ptr = ctypes.c_void_p()
        if ctrl:
            _lib.check(_lib.lib.ngacf_peer_alloc_ctrl(nbytes, ctypes.byref(ptr)), "ngacf_peer_alloc_ctrl")
        else:
            _lib.check(_lib.lib.ngacf_peer_alloc(nbytes, ctypes.byref(ptr)), "ngacf_peer_alloc")
        peers = {}
        if self.W > 1:
            h = ctypes.create_string_buffer(_lib.PEER_HANDLE_BYTES)
            _lib.check(_lib.lib.ngacf_peer_export(ptr, h), "ngacf_peer_export")
            handles = [None] * self.W
            dist.all_gather_object(handles, bytes(h.raw), group=self.group)
            for r in self.others:
                q = ctypes.c_void_p()
                _lib.check(_lib.lib.ngacf_peer_import(ctypes.create_string_buffer(handles[r], _lib.PEER_HANDLE_BYTES), ctypes.byref(q)),
                           "ngacf_peer_import (rank %d's memory)" % r)
                peers[r] = q.value
        b = _Buf(ptr.value, nbytes, peers, len(self._bufs), self.device)
        self._bufs.append(b)
        return b

    def _ctrl_addr(self, base, what, slot):
        off = {"flags": slot * self.W, "sent": SLOTS * self.W + slot, "expect": SLOTS * self.W + SLOTS + slot,
               "ticket": SLOTS * self.W + 2 * SLOTS + slot, "err": SLOTS * self.W + 3 * SLOTS}[what]
        return base + 4 * off

    # ---- the pool ---------------------------------------------------------------------------------------------------------
    def new_lease(self):
        return _Lease()

    def table(self, key, shape, dtype, device, lease=None):
        """A gathered table: the first free pool buffer of `key` (a new symmetric allocation when there is none)."""
        if lease is None:
            raise ValueError("peer-store tables are leased")
        key = (key, tuple(shape), dtype)
        nbytes = int(torch.empty(0, dtype=dtype).element_size())
        for n in shape:
            nbytes *= int(n)
        free = [b for b in self._pool.setdefault(key, []) if not b.busy]
        if free:
            b = free[0]
        else:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("peer-store table %r first needed inside a stream capture: run one eager pass first" % (key,))
            b = self._symmetric(max(nbytes, 16))
            self._pool[key].append(b)
            self._by_ptr[b.ptr] = b
        b.busy = True
        lease.bufs.append(b)
        return b.bytes_view[:nbytes].view(dtype).view(*shape)

    # ---- the exchange -----------------------------------------------------------------------------------------------------
    def start_gather(self, t, *fulls):
        if not self.active:
            return _Handle()
        bufs = [self._by_ptr.get(f.data_ptr()) for f in fulls]
        if any(b is None for b in bufs) or len(fulls) > _lib.PEER_MAX_REGIONS:
            return super().start_gather(t, *fulls)            # not a pool table: the collective path
        p = self.part
        lo, n = (p.rank * p.cu, p.cu) if t == "user" else (p.Up + p.rank * p.ci, p.ci)
        regions = []
        for f, b in zip(fulls, bufs):
            rowb = f.stride(0) * f.element_size()
            regions.append((b, lo * rowb, n * rowb))
        slot = 2 * bufs[0].index + (0 if t == "user" else 1)
        self._push(regions, slot, side=self.overlap)
        h = _Handle(done=lambda slot=slot: self._wait(slot))
        if not self.overlap:
            h.wait()
        return h

    def _push(self, regions, slot, side):
        """regions: [(pool buffer, byte offset, bytes)]: the same span of every peer's copy of the buffer receives mine; then the
        flag word of (slot, this rank) goes up on every peer.  side: on the side stream, behind everything the current stream
        holds so far (under the other half's traversal); else in line."""
        regs = (_lib.PeerRegion * len(regions))()
        for k, (b, off, nbytes) in enumerate(regions):
            regs[k].src = b.ptr + off
            regs[k].bytes = nbytes
            for j, r in enumerate(self.others):
                regs[k].dst[j] = b.peers[r] + off
        flags = (ctypes.c_void_p * max(len(self.others), 1))()
        for j, r in enumerate(self.others):                   # my word in peer r's flag array
            flags[j] = self._ctrl_addr(self._ctrl.peers[r], "flags", slot) + 4 * self.rank
        cur = torch.cuda.current_stream(self.device)
        st = cur
        if side:
            ev = torch.cuda.Event()
            ev.record(cur)
            self.side.wait_event(ev)
            st = self.side
        _lib.check(_lib.lib.ngacf_peer_push(regs, len(regions), len(self.others), flags, self._ctrl_addr(self._ctrl.ptr, "sent", slot),
                                            self._ctrl_addr(self._ctrl.ptr, "ticket", slot), ctypes.c_void_p(st.cuda_stream)),
                   "ngacf_peer_push")
        self.launches += 1
        self.pushes += 1

    def _wait(self, slot):
        _lib.check(_lib.lib.ngacf_peer_wait(self._ctrl_addr(self._ctrl.ptr, "flags", slot), self.W, self.rank,
                                            self._ctrl_addr(self._ctrl.ptr, "expect", slot), self._ctrl_addr(self._ctrl.ptr, "err", 0),
                                            self.timeout_ticks, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                   "ngacf_peer_wait")

    def all_reduce_sum(self, *ts):
        """Sum over the ranks without a collective: every rank stores its tensor into row `rank` of a [W, n] buffer on every
        peer, waits for the others' rows and adds the W rows in rank order -- the same order on every rank, so the results are
        bit-identical everywhere (the row exchange of the scores and the stage weights' gradient sum: with them the whole step
        of the node-partitioned loop is free of RCCL).  Push and wait go IN LINE on the caller's stream: a rank's push always
        precedes its own wait in one hardware queue, so the peer's wait is always served.
        Two buffers per (size, dtype) taken in turn: a rank may start the next sum while a slow peer still adds the rows of
        this one -- it cannot start the one after before that peer has sent its share of the next."""
        if not self.active or any((not t.is_cuda) or (t.numel() * t.element_size()) % 4 or not t.is_contiguous() for t in ts):
            return super().all_reduce_sum(*ts)
        for t in ts:
            n = t.numel()
            key = ("all_reduce", n, t.dtype)
            st = self._ar.get(key)
            if st is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("peer-store all-reduce of %d x %s first needed inside a stream capture: run one eager "
                                       "step first" % (n, t.dtype))
                st = self._ar[key] = {"bufs": [self._symmetric(max(self.W * n * t.element_size(), 16)) for _ in range(2)], "turn": 0}
            b = st["bufs"][st["turn"]]
            st["turn"] ^= 1
            rows = b.bytes_view[:self.W * n * t.element_size()].view(t.dtype).view(self.W, n)
            rows[self.rank].copy_(t.reshape(-1))
            slot = 2 * b.index
            self._push([(b, self.rank * n * t.element_size(), n * t.element_size())], slot, side=False)
            self._wait(slot)
            acc = rows[0].clone() if self.W == 1 else rows[0] + rows[1]
            for r in range(2, self.W):
                acc += rows[r]
            t.copy_(acc.view_as(t))
        return ts[0] if len(ts) == 1 else ts

    def join(self):
        """The side stream's pushes rejoin the current stream (end of a pass; before a capture ends)."""
        torch.cuda.current_stream(self.device).wait_stream(self.side)

    def check(self):
        """Synchronises; raises if a wait ever timed out (a peer never published its block)."""
        torch.cuda.synchronize(self.device)
        n = SLOTS * self.W + 3 * SLOTS
        err = int(self._ctrl.bytes_view[4 * n:4 * n + 4].view(torch.int32).item())
        if err:
            raise RuntimeError("peer-store all-gather: a wait timed out (a peer did not publish its block)")

    def close(self):
        torch.cuda.synchronize(self.device)
        if self.W > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
        for b in self._bufs:
            for q in b.peers.values():
                _lib.lib.ngacf_peer_close(ctypes.c_void_p(q))
        if self.W > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
        for b in self._bufs:
            _lib.lib.ngacf_peer_free(ctypes.c_void_p(b.ptr))
        self._bufs, self._pool, self._by_ptr, self._ar = [], {}, {}, {}
