import sys, os
sys.path[:0]=['csparse.py_amd','tests']
import numpy as np, _csx, synth
_csx.init(0); lib=_csx.lib()
for n,pc,seed in ((30000,64,5),(1000,32,9),(200,64,1),(5000,7,3)):
    h=_csx.new_handle(); _csx.check(lib.csx_gen_grand_uniform(n,pc,seed,h),"gen")
    p=np.empty(n+1,np.int32); i=np.empty(n*pc,np.int32); x=np.empty(n*pc)
    _csx.check(lib.csx_csc_download(h,_csx.pi(p),_csx.pi(i),_csx.pd(x)),"dl")
    Ap,Ai,Ax=synth.grand_uniform(n,pc,seed)
    print(n,pc,(p==Ap).all(),(i==Ai).all(),x.tobytes()==Ax.tobytes())
