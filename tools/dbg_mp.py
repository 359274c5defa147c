import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from helpers import *
from oracle import pyref as R, coracle as C
import sylow_amd
eng=sylow_amd.Engine(0)
G1=[1,2]; G2=list(R.G2_GEN_AFF[0])+list(R.G2_GEN_AFF[1])
ONE4=np.array([[1,0,0,0]],dtype=np.uint64)
proj1=lambda xy: np.concatenate([xy,np.repeat(ONE4,xy.shape[0],0)],axis=1)
proj2=lambda xy: np.concatenate([xy,np.repeat(ONE4,xy.shape[0],0),np.zeros((xy.shape[0],4),dtype=np.uint64)],axis=1)
rng=Xoshiro(SEED+40)
for ks in ([1],[2],[3],[1,1],[2,2],[1,2],[4],[5],[2,1]):
    n=sum(ks); off=np.concatenate([[0],np.cumsum(ks)]).astype(np.uint64)
    p,_=eng.g1_scalar_mul(np.repeat(pack(G1,8),n,0),limbs([rng.fp() for _ in range(n)]))
    q,_=eng.g2_scalar_mul(np.repeat(pack(G2,16),n,0),limbs([rng.fp() for _ in range(n)]))
    gt,_=eng.multi_pairing(p,q,off)
    exp=C.glued_pairing(proj1(p),proj2(q),off)
    print(ks, [bool(np.array_equal(gt[i],exp[i])) for i in range(len(ks))])
