import ctypes, os, sys
os.environ.setdefault("NCCL_DEBUG", "WARN")
import torch
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
for name in ("librccl.so.1",):
    lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]
uid = UniqueId()
lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
lib.ncclGetErrorString.restype = ctypes.c_char_p
lib.ncclGetLastError.restype = ctypes.c_char_p
print("uid", lib.ncclGetUniqueId(ctypes.byref(uid)))
comm = ctypes.c_void_p()
rc = lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0)
print("init rc", rc, lib.ncclGetErrorString(rc), lib.ncclGetLastError(None))
# alternative: ncclCommInitAll
comms = (ctypes.c_void_p * 1)()
devs = (ctypes.c_int * 1)(0)
rc = lib.ncclCommInitAll(comms, 1, devs)
print("initAll rc", rc, lib.ncclGetErrorString(rc), lib.ncclGetLastError(None))
