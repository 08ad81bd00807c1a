import sys, numpy as np, ctypes as C
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import golden_util as gu
from oracle_lib import Oracle, lib, dgbsv, _pd
ne, L = 6, 30
g, states = gu.make_grid(ne, L, 6)
o = Oracle(g); o.set_state(0, states)
for _ in range(3): o.step_ars343(300.0)
n = 3*(L+1); ldab = 13; kv = 8
nz = np.zeros((3, 9)); cnt = np.zeros(3); piv = np.zeros(5); lz = np.zeros((3,5)); pvt=np.zeros((3,5))
P = o.cpatches[0]
node = o.node[0][0]; redge = o.redge[0][0]
x0 = np.zeros(n); F = np.zeros(n); AB = np.zeros(n*ldab)
for iA in range(1, 1+4*ne, 3):
    for iB in range(1, 1+4*ne, 3):
        lib().orc_column_f_jacobian(C.byref(o.cgrid), C.byref(P), _pd(node), _pd(redge), iA, iB, C.c_double(65.0), _pd(x0), _pd(F), _pd(AB))
        info, x, ipiv = dgbsv(AB.reshape(n, ldab), F, 4, 4)
        ab = AB.reshape(n, ldab)
        # dgbsv wrapper copies; redo factorization on a copy to get U
        ab2 = ab.copy(); b2 = F.copy(); ip = np.zeros(n, dtype=np.int32)
        lib().orc_dgbsv(n, 4, 4, _pd(ab2), ldab, ip.ctypes.data_as(C.POINTER(C.c_int)), _pd(b2))
        for j in range(n):
            piv[min(ip[j]-j,4)] += 1
            pvt[j%3, min(ip[j]-j,4)] += 1
            for i in range(1,5):
                if j+i < n and ab2[j, kv+i] != 0.0: lz[j%3, i] += 1
            cnt[j % 3] += 1
            for c in range(9):
                if j + c < n and ab2[j+c, kv-c] != 0.0: nz[j % 3, c] += 1
print("pivot displacement histogram:", piv/piv.sum())
print("U nonzero fraction by row type (P,W,R) and offset 0..8:")
print(np.round(nz/cnt[:,None], 3))
print("L multiplier nonzero fraction by row type and i=1..4:")
print(np.round(lz/cnt[:,None],3))
print("pivot displacement by row type:")
print(np.round(pvt/cnt[:,None],3))
