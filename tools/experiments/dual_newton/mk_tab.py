import numpy as np, sympy as sp
from newton import AM
zI=np.array([1,0,0,0,1,0,0,0,1,1]); A2=(AM*2).round().astype(int)
G=sp.Matrix([[int(v) for v in (A2[i]@zI)] for i in range(22)]).T
mats=[]
for v in G.nullspace():
    den=sp.ilcm(*[sp.fraction(x)[1] for x in v]); vi=[int(x*den) for x in v]
    mats.append(sum(c*A2[i] for i,c in enumerate(vi)))
# greedy independent subset, sparsest first
order=sorted(range(len(mats)),key=lambda k:(np.count_nonzero(mats[k]),k)); sel=[]
for k in order:
    F=np.array([mats[j].reshape(-1) for j in sel+[k]],dtype=float)
    if np.linalg.matrix_rank(F)==len(sel)+1: sel.append(k)
sel=sorted(sel); assert len(sel)==14
tabs=[]
for k in sel:
    m=mats[k]; g=np.gcd.reduce(np.abs(m[m!=0])); m=m//g
    tabs.append([(i,j,float(m[i,j])) for i in range(10) for j in range(i,10) if m[i,j]!=0])
T=np.eye(10)-np.outer(zI,zI)/4.0
tabs.append([(i,j,float(T[i,j])) for i in range(10) for j in range(i,10) if T[i,j]!=0])
open("basis_tab.py","w").write("TAB="+repr(tabs)+"\n")
print("selected",sel,[len(t) for t in tabs])
