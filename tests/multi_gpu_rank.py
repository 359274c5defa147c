"""One rank of the multi-GPU pre-flight (tests/test_gpu_multi_gpu.py launches N of these under torch.distributed.run, one per GPU,
backend nccl = RCCL over xGMI).  Exercises, with N > 1 REAL ranks, the three native collective entry points of the C ABI on this
rank's own ncclComm_t (include/sylow_hip.h: sylow_hip_all_valid, sylow_hip_pairing_product_all, sylow_hip_bls_aggregate_verify_batch):
 * rank r draws its own pairs / keys / messages (seeded by r); every rank can re-derive every other rank's inputs, so rank 0 checks
   the union's glued pairing against the ORACLE (test infrastructure: this file lives under tests/) and against a one-GPU product;
 * a signature planted bad on the LAST rank must flip the AND on every rank; the aggregate verifier must agree;
 * prints ONE JSON line from rank 0 with per-rank kernel milliseconds and the communicator's construction time."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

G1 = [1, 2]
G2 = [0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
      0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2,
      0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
      0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B]


def main():
    import torch
    import torch.distributed as dist

    from helpers import SEED, pack
    import sylow_amd
    from sylow_amd.rccl import NativeComm, quiet_init_env

    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    quiet_init_env()                                   # one node (this launcher's choice, not the library's)
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    eng = sylow_amd.Engine(local_rank)
    t0 = time.perf_counter()
    comm, err = NativeComm.from_process_group(dist)
    comm_init_s = time.perf_counter() - t0
    assert comm is not None, err
    assert comm.ranks == world

    npairs, nsig = 5, 64

    def rank_pairs(r):
        ka, kb = eng.xoshiro_fp_soa(SEED + 900 + r, npairs).T.copy(), eng.xoshiro_fp_soa(SEED + 950 + r, npairs).T.copy()
        p, _ = eng.g1_scalar_mul(np.tile(pack(G1, 8), (npairs, 1)), ka)
        q, _ = eng.g2_scalar_mul(np.tile(pack(G2, 16), (npairs, 1)), kb)
        return p, q

    p, q = rank_pairs(rank)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    gt, is_one = eng.pairing_product_all(p, q, comm=comm.value)
    ev[1].record()

    # signatures: rank r signs nsig messages; the last rank may carry a planted bad one
    sk = eng.xoshiro_fp_soa(SEED + 1000 + rank, nsig).T.copy()
    msgs = [bytes([rank, i & 255]) * (1 + i % 7) for i in range(nsig)]
    sig, _ = eng.bls_sign(sk, msgs)
    pk, _ = eng.g2_scalar_mul(np.tile(pack(G2, 16), (nsig, 1)), sk)
    out = {}
    for plant in (False, True):
        s = sig.copy()
        if plant and rank == world - 1:
            s[7] = s[8]
        ok = eng.bls_verify(pk, msgs, s)
        flags = eng.to_device(ok)
        ev[2].record()
        out["all_valid_planted" if plant else "all_valid"] = eng.all_valid(flags, comm=comm.value)
        ev[3].record()
        _, agg = eng.bls_aggregate_verify(pk, msgs, s, comm=comm.value)
        out["aggregate_planted" if plant else "aggregate"] = int(agg)
    torch.cuda.synchronize()
    ms = torch.tensor([ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])], dtype=torch.float64, device="cuda")
    allms = [torch.zeros_like(ms) for _ in range(world)]
    dist.all_gather(allms, ms)

    # every rank holds the same Gt value: compare through the process group (MIN of an equality flag)
    mine = torch.from_numpy(gt.astype(np.int64)).cuda()
    ref0 = mine.clone()
    dist.broadcast(ref0, src=0)
    same = torch.tensor([int(torch.equal(mine, ref0))], dtype=torch.int32, device="cuda")
    dist.all_reduce(same, op=dist.ReduceOp.MIN)

    if rank == 0:
        from oracle import coracle as C                    # the checker
        allp, allq = zip(*[rank_pairs(r) for r in range(world)])
        allp, allq = np.concatenate(allp), np.concatenate(allq)
        m = allp.shape[0]
        one = np.zeros((m, 4), dtype=np.uint64); one[:, 0] = 1
        exp = C.glued_pairing(np.concatenate([allp, one], axis=1), np.concatenate([allq, one, np.zeros((m, 4), dtype=np.uint64)], axis=1),
                              np.array([0, m], dtype=np.uint64))
        local, _ = eng.pairing_product(allp, allq)
        out.update({"world": world, "rccl_ranks": comm.ranks, "comm_init_s": comm_init_s,
                    "product_all_equals_oracle": int(np.array_equal(gt, exp)), "product_all_equals_one_gpu_product": int(np.array_equal(gt, local)),
                    "product_all_same_on_every_rank": int(same.item()), "product_is_one": int(is_one),
                    "per_rank_ms": {"pairing_product_all": [float(x[0]) for x in allms], "all_valid": [float(x[1]) for x in allms]}})
        print(json.dumps(out), flush=True)
    torch.cuda.synchronize()
    comm.destroy()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
