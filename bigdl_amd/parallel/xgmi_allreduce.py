"""All-reduce over IPC-mapped peer buffers (kernels: ``csrc/hip/allreduce_xgmi.hip``): a one-shot form for the
small, latency-bound all-reduces of a tensor-parallel decode step, and a two-shot (reduce-scatter + all-gather) form
for the multi-megabyte all-reduces of the prompt pass.

Every rank allocates one fine-grained device buffer, the ranks exchange ``hipIpcMemHandle``s through the (RCCL or
gloo) process group, and each call is a single kernel launch on the current stream -- capturable into the decode
hipGraph, because the call counter lives in device memory.  Ranks must sit in one node (xGMI / same host); anything
else, other dtypes or larger messages fall back to ``torch.distributed.all_reduce``.

On by default for tensor-parallel engines (``shard_engine``; ``BIGDL_XGMI_ALLREDUCE=0`` turns it off) behind a
self-check at start-up: ``XgmiAllReduce.create_verified`` runs both kernels on known data, compares with the process
group's own all-reduce on every rank, and hands back None (RCCL stays in charge) unless every rank agrees that the IPC
path set up and summed correctly.
"""
from __future__ import annotations

import contextlib
import ctypes
import socket

import torch
import torch.distributed as dist

from ..ops import hip

DEFAULT_MAX_BYTES = 256 * 1024


class _Failed:
    """What a rank puts into a set-up exchange in place of its contribution when its local step raised."""

    def __init__(self, why: str):
        self.why = why


class OneShotAllReduce:
    """All ranks construct this together; the constructor either returns on every rank or raises on every rank."""

    def __init__(self, group=None, max_bytes: int = DEFAULT_MAX_BYTES, device=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.max_bytes = (max_bytes + 63) // 64 * 64
        self._own, self._opened = None, []
        # Every rank runs the same sequence of collectives whatever fails locally: a local error is recorded, the
        # ranks exchange what they have, and all of them raise together after the exchange.
        hosts = self._exchange(socket.gethostname())
        if len(set(hosts)) != 1:
            raise RuntimeError("one-shot all-reduce needs all ranks in one node")
        if self.world > 16:
            raise RuntimeError("one-shot all-reduce supports at most 16 ranks")
        L = self.lib = hip.lib()
        L.bigdl_ar_buffer_bytes.restype = ctypes.c_int64
        L.bigdl_ar_buffer_bytes.argtypes = [ctypes.c_int, ctypes.c_int64]
        L.bigdl_ar_alloc.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]
        L.bigdl_ar_free.argtypes = [ctypes.c_void_p]
        L.bigdl_ar_ipc_handle.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.bigdl_ar_ipc_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        L.bigdl_ar_ipc_close.argtypes = [ctypes.c_void_p]
        L.bigdl_allreduce_oneshot.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.bigdl_ar2_buffer_bytes.restype = ctypes.c_int64
        L.bigdl_ar2_buffer_bytes.argtypes = [ctypes.c_int, ctypes.c_int64]
        L.bigdl_allreduce_twoshot.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.max_bytes = self._round_max_bytes(self.max_bytes)
        with (torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()):
            # phase 1: own buffer + its handle
            mine = None
            try:
                mine = self._alloc_and_export()
            except Exception as e:
                mine = _Failed(f"rank {self.rank}: {e}")
            handles = self._exchange(mine)
            self._raise_together(handles)
            # phase 2: map every peer's buffer
            opened = None
            try:
                ptrs = self._open_peers(handles)
            except Exception as e:
                opened = _Failed(f"rank {self.rank}: {e}")
            self._raise_together(self._exchange(opened))
            self._peers = (ctypes.c_int64 * self.world)(*ptrs)
            # phase 3: per-rank device state
            state = None
            try:
                self.epoch = torch.zeros(self._epoch_words(), dtype=torch.int32, device=self.device)
                self.error = torch.zeros(1, dtype=torch.int32, device=self.device)
            except Exception as e:
                state = _Failed(f"rank {self.rank}: {e}")
            self._raise_together(self._exchange(state))   # doubles as "every buffer is mapped everywhere"

    def _exchange(self, obj):
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _raise_together(self, gathered):
        bad = [g.why for g in gathered if isinstance(g, _Failed)]
        if bad:
            self.close()
            raise RuntimeError("IPC all-reduce set-up failed: " + "; ".join(bad))

    def _alloc_and_export(self) -> bytes:
        own = ctypes.c_void_p()
        if self.lib.bigdl_ar_alloc(self._buffer_bytes(), ctypes.byref(own)) != 0:
            raise RuntimeError("fine-grained buffer allocation failed")
        self._own = own
        handle = ctypes.create_string_buffer(64)
        if self.lib.bigdl_ar_ipc_handle(own, handle) != 0:
            raise RuntimeError("hipIpcGetMemHandle failed")
        return handle.raw

    def _open_peers(self, handles):
        ptrs = []
        for r, h in enumerate(handles):
            if r == self.rank:
                ptrs.append(self._own.value)
                continue
            p = ctypes.c_void_p()
            if self.lib.bigdl_ar_ipc_open(h, ctypes.byref(p)) != 0:
                raise RuntimeError(f"hipIpcOpenMemHandle failed for rank {r}")
            self._opened.append(p)
            ptrs.append(p.value)
        return ptrs

    def _round_max_bytes(self, n):
        return n

    def _buffer_bytes(self):
        return self.lib.bigdl_ar_buffer_bytes(self.world, self.max_bytes)

    def _epoch_words(self):
        return 1

    def usable(self, t: torch.Tensor) -> bool:
        return (t.device.type == "cuda" and t.dtype == torch.float16 and t.is_contiguous() and t.numel() % 8 == 0
                and t.numel() * 2 <= self.max_bytes and t.data_ptr() % 16 == 0)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the group."""
        if not self.usable(t):
            dist.all_reduce(t, group=self.group)
            return t
        rc = self.lib.bigdl_allreduce_oneshot(self._peers, self.rank, self.world, t.data_ptr(), t.data_ptr(), t.numel(),
                                              self.max_bytes, self.epoch.data_ptr(), self.error.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"allreduce_oneshot: native launcher rejected the call (code {rc})")
        return t

    __call__ = all_reduce

    def check(self):
        """Raise if a call gave up waiting for a peer (synchronises the device)."""
        if int(self.error.item()) != 0:
            raise RuntimeError("one-shot all-reduce: a peer did not publish its data in time")

    def close(self):
        for p in getattr(self, "_opened", []):
            self.lib.bigdl_ar_ipc_close(p)
        self._opened = []
        if getattr(self, "_own", None) is not None:
            self.lib.bigdl_ar_free(self._own)
            self._own = None


class TwoShotAllReduce(OneShotAllReduce):
    """Reduce-scatter + all-gather by direct peer stores for the large all-reduces of a tensor-parallel prompt pass
    (kernel ``allreduce_twoshot_kernel``): each GPU moves 2 (W-1)/W of the message once over all its xGMI links.
    Same set-up, fallback and error handling as the one-shot form; 64 workgroups, one launch, capturable."""

    def __init__(self, group=None, max_bytes: int = 64 << 20, device=None):
        super().__init__(group, max_bytes, device)

    def _round_max_bytes(self, n):
        q = 64 * self.world
        return (n + q - 1) // q * q

    def _buffer_bytes(self):
        return self.lib.bigdl_ar2_buffer_bytes(self.world, self.max_bytes)

    def _epoch_words(self):
        return self.lib.bigdl_ar2_epoch_words()

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if not self.usable(t):
            dist.all_reduce(t, group=self.group)
            return t
        rc = self.lib.bigdl_allreduce_twoshot(self._peers, self.rank, self.world, t.data_ptr(), t.data_ptr(), t.numel(),
                                              self.max_bytes, self.epoch.data_ptr(), self.error.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"allreduce_twoshot: native launcher rejected the call (code {rc})")
        return t

    __call__ = all_reduce


class XgmiAllReduce:
    """Size-switched front: one-shot below ``small_bytes`` (decode), two-shot up to ``large_bytes`` (prompt), RCCL
    above.  This is what ``shard_engine(..., fast_allreduce=True)`` installs."""

    def __init__(self, group=None, small_bytes: int = DEFAULT_MAX_BYTES, large_bytes: int = 64 << 20, device=None):
        self.group = group
        self.small = OneShotAllReduce(group, small_bytes, device)
        try:
            self.large = TwoShotAllReduce(group, large_bytes, device)
        except Exception:
            self.small.close()
            raise

    @classmethod
    def create_verified(cls, group=None, device=None, **kw):
        """Set up, run a self-check against ``dist.all_reduce`` and return the object, or None when any rank failed to
        set up or saw a wrong sum.  Every rank takes the same decision after every step: the constructors raise on
        all ranks or on none, and each self-check size ends in a vote, so no rank is left inside a collective that a
        failed peer never enters."""
        import warnings
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

        def all_ok(ok: int) -> bool:
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        def give_up(obj, why):
            if obj is not None:
                obj.close()
            warnings.warn("IPC all-reduce disabled, using the process group's all-reduce"
                          + (f" ({why})" if why else " (a peer rank failed its self-check)"))
            return None

        try:
            obj = cls(group, device=device, **kw)
        except Exception as e:            # IPC not permitted, ranks on different hosts, allocation failure, ...
            return give_up(None, f"set-up failed: {e}")
        rank = dist.get_rank(group)
        for n in (4096, 1 << 19):                             # 8 KB: one-shot; 1 MB: two-shot
            ok, why = 1, ""
            base = torch.arange(n, device=dev, dtype=torch.float32).remainder_(13).add_(rank + 1).to(torch.float16)
            want = base.clone()
            dist.all_reduce(want, group=group)
            try:
                got = base.clone()
                for _ in range(3):                            # repeated calls walk the epoch counters
                    got.copy_(base)
                    obj(got)
                obj.check()
                if not torch.equal(got, want):
                    ok, why = 0, f"wrong sum at {n} elements"
            except Exception as e:
                ok, why = 0, f"self-check raised: {e}"
            if not all_ok(ok):
                return give_up(obj, why)
        return obj

    def usable(self, t):
        return self.small.usable(t) or self.large.usable(t)

    def __call__(self, t):
        return self.small(t) if self.small.usable(t) else self.large(t)

    all_reduce = __call__

    def check(self):
        self.small.check()
        self.large.check()

    def close(self):
        self.small.close()
        self.large.close()
