import numpy as np, sympy as sp
from newton import AM, ROWS
zI=np.array([1,0,0,0,1,0,0,0,1,1.0])
A2=(AM*2).round().astype(int)   # integer matrices (2 A_i)
G=sp.Matrix([[int(v) for v in (A2[i]@zI.astype(int))] for i in range(22)]).T   # 10 x 22
ns=G.nullspace()
print("nullspace dim", len(ns))
mats=[]
for v in ns:
    den=sp.ilcm(*[sp.fraction(x)[1] for x in v]); vi=[int(x*den) for x in v]
    M=sum(c*A2[i] for i,c in enumerate(vi))
    mats.append(M)
# remove dependency: rank
F=np.array([m.reshape(-1) for m in mats],dtype=float)
print("rank of matrices", np.linalg.matrix_rank(F))
for k,m in enumerate(mats):
    nz=[(i,j,m[i,j]) for i in range(10) for j in range(i,10) if m[i,j]!=0]
    print(k, "nnz(upper)", len(nz), nz)
