import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch, ctypes as C, time
from ilqr_amd import _lib
from test_eigvals_restatement import stage_matrices
g = np.load(os.path.join(R,'tests/golden/eig_decisions.npz'))
L=_lib.lib()
for tag in ("car","quat"):
    Q0, Pm, rounds, end = g[tag+"_Q0"], g[tag+"_P"], g[tag+"_rounds"], g[tag+"_end"]
    mats=[]
    for s in range(len(Q0)):
        m,w = stage_matrices(Q0[s],Pm[s],int(rounds[s]),int(end[s]))
        if len(m): mats.append(m[np.isfinite(m).all(axis=(1,2))])
    mats=np.concatenate(mats); w=mats.shape[1]
    reps = max(1, (1<<20)//len(mats))
    mats=np.concatenate([mats]*reps); B=len(mats)
    q=torch.as_tensor(np.ascontiguousarray(mats.reshape(B,w*w).T),device="cuda:0"); out=torch.zeros(B,dtype=torch.int32,device="cuda:0")
    for it in range(3):
        torch.cuda.synchronize(); t=time.time()
        L.ilqr_eigvals_negative(w,C.c_void_p(q.data_ptr()),B,B,C.c_void_p(out.data_ptr()),None); torch.cuda.synchronize()
        dt=time.time()-t
    print(tag,"w",w,"B",B,"%.2f ms"%(dt*1e3),"-> %.1f ns per matrix, %.1f us per wave-pass at %d waves"%(dt/B*1e9, dt/(B/64)*1e6*min(B/64,1024*2)/1.0 if False else dt*1e6/(B/64/1024), B//64))
