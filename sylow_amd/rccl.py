"""A native `ncclComm_t` per rank for the C ABI's aggregate entry points (`sylow_hip_all_valid`,
`sylow_hip_pairing_product_all`, `sylow_hip_bls_aggregate_verify_batch`: include/sylow_hip.h), built the way a non-Python host would:
rank 0 draws an `ncclUniqueId`, the 128 bytes travel through whatever transport the host already has (here: the existing
torch.distributed process group), every rank calls `ncclCommInitRank`.  RCCL is the copy torch already loaded (the same one
collective.hip binds with dlopen), so there is one RCCL in the process.

Every step that could leave ranks waiting for each other is agreed on first (MIN all-reduce of an "I am fine" word over the
process group), so a rank that cannot load RCCL makes ALL ranks fall back instead of hanging the others inside the collective
`ncclCommInitRank`."""
from __future__ import annotations

import ctypes
import os


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def warm_file(path: str) -> None:
    """Read a shared object front to back once.  RCCL's 300+ MB fat binary is demand-paged: on a box whose image is not in the page
    cache yet, the first communicator faults it in 4 KB at a time (minutes -- measured 416 s on one fresh MI355X box against 1 s
    warm); one sequential read with read-ahead brings it in at disk speed.  Best effort, never an error."""
    try:
        with open(path, "rb", buffering=0) as f:
            while f.read(1 << 24):
                pass
    except OSError:
        pass


def quiet_init_env() -> None:
    """ONE-NODE launchers only (bench.py and the tests call it explicitly; the library never does): defaults, never overriding the caller's
    environment, that keep communicator construction short when every rank is local -- the path's collectives are a 4-byte all-reduce and a
    384-byte all-gather, so the MSCCL / MSCCL++ algorithm stores RCCL would otherwise parse at init are of no use, and the bootstrap socket
    needs no interface scan (loopback).  On a multi-node process group NCCL_SOCKET_IFNAME=lo would make rank 0's bootstrap listen where remote
    ranks cannot reach it, which is why `_load_rccl` / `NativeComm` leave the environment alone."""
    os.environ.setdefault("RCCL_MSCCL_ENABLE", "0")
    os.environ.setdefault("RCCL_MSCCLPP_ENABLE", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def _load_rccl():
    """The RCCL shared object of this process: torch's bundled copy when torch is importable (its soname is what collective.hip
    dlopens too), else the system one.  Does not touch the environment (see quiet_init_env)."""
    names = []
    try:
        import torch
        names.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    except Exception:  # noqa: BLE001
        pass
    names += ["librccl.so.1", "librccl.so"]
    last = None
    for name in names:
        try:
            if os.path.isfile(name):
                warm_file(name)
            lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            lib.ncclCommInitRank, lib.ncclGetUniqueId, lib.ncclCommCount, lib.ncclCommDestroy, lib.ncclGetErrorString   # AttributeError if absent
        except (OSError, AttributeError) as e:
            last = e
            continue
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        return lib
    raise OSError(f"no librccl found: {last}")


class NativeComm:
    """`comm.value` is the raw `ncclComm_t` (an int) the C ABI takes as `void* comm`; `ranks` is `ncclCommCount`."""

    def __init__(self, lib, comm, ranks):
        self._lib, self._comm, self.ranks = lib, comm, ranks

    @property
    def value(self) -> int:
        return self._comm.value

    def destroy(self):
        if self._comm is not None and self._comm.value:
            self._lib.ncclCommDestroy(self._comm)
        self._comm = None

    @classmethod
    def single(cls) -> "NativeComm":
        """A one-rank communicator without any process group (a single-GPU host that still wants the RCCL code path)."""
        lib = _load_rccl()
        uid = _UniqueId()
        rc = lib.ncclGetUniqueId(ctypes.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId: {lib.ncclGetErrorString(rc).decode()}")
        return cls._init(lib, uid, 1, 0)

    @classmethod
    def _init(cls, lib, uid, world, rank):
        comm = ctypes.c_void_p()
        rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank: {lib.ncclGetErrorString(rc).decode()}")
        cnt = ctypes.c_int(0)
        rc = lib.ncclCommCount(comm, ctypes.byref(cnt))
        if rc != 0:
            raise RuntimeError(f"ncclCommCount: {lib.ncclGetErrorString(rc).decode()}")
        return cls(lib, comm, int(cnt.value))

    @classmethod
    def from_process_group(cls, dist) -> tuple["NativeComm | None", str | None]:
        """(communicator, None) on every rank, or (None, reason) on every rank -- never a mix.  `dist` is torch.distributed with an
        initialised process group whose backend is nccl (one GPU per rank; RCCL refuses two ranks on one device)."""
        import torch
        from . import sharding
        world, rank = dist.get_world_size(), dist.get_rank()
        dev = sharding.collective_device(dist)

        def all_fine(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        lib, err = None, None
        try:
            lib = _load_rccl()
        except Exception as e:  # noqa: BLE001 -- whatever went wrong here must not leave the other ranks waiting
            err = f"{type(e).__name__}: {e}"
        if not all_fine(lib is not None):
            return None, err or "another rank could not load RCCL"
        # rank 0's unique id (+ a status byte) to everybody
        buf = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = _UniqueId()
            rc = lib.ncclGetUniqueId(ctypes.byref(uid))
            # c_char arrays stop at the first NUL when read through .internal: copy the raw 128 bytes instead
            raw = ctypes.string_at(ctypes.addressof(uid), 128) if rc == 0 else bytes(128)
            buf = torch.tensor([1 if rc == 0 else 0] + list(raw), dtype=torch.uint8, device=dev)
        dist.broadcast(buf, src=0)
        host = bytes(buf.cpu().tolist())
        if host[0] != 1:
            return None, "ncclGetUniqueId failed on rank 0"
        uid = _UniqueId()
        ctypes.memmove(ctypes.addressof(uid), host[1:129], 128)
        comm, err = None, None
        try:
            comm = cls._init(lib, uid, world, rank)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if not all_fine(comm is not None and comm.ranks == world):
            if comm is not None:
                comm.destroy()
            return None, err or "ncclCommInitRank failed on another rank"
        return comm, None
