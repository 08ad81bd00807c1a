"""CPU-only evidence for DESIGN.md section 2: the C oracle (own dgbtf2-order band LU, own geometry) against the real
reference (MKL dgbsv) over 100 ARS343 steps at ne15 L30 -- i.e. how far two correct CPU implementations drift apart.
Needs oracle/_ref (build container only).  Prints per-variable max-norm relative differences (U V rhotheta W rho)."""
import sys, os, subprocess, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import golden_util as gu, tmxd
from oracle_lib import Oracle
ne,L,dt=15,30,300.0
g,_=gu.make_grid(ne,L,6)
out='/tmp/ref15.tmxd'
env=dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
subprocess.run(['/root/repo/oracle/_ref/tmx_ref_dump',"--mode","steps","--ne",str(ne),"--levels",str(L),"--dt",str(dt),"--warm","2","--steps","100","--every","100","--geom","0","--out",out],env=env,check=True,stdout=subprocess.DEVNULL)
rd=tmxd.read(out)
start=gu.full_states(rd,"warm",6); end=gu.full_states(rd,"step100",6)
o=Oracle(g); o.set_state(0,start)
for _ in range(100): assert o.step_ars343(dt)==0
print("oracle vs reference ne15 L30 100 steps:", gu.prognostic_errors(o.get_state(0), end))
wmax=max(np.abs(e[3][:,1:-1,1:-1]).max() for n,e in end); print("max |W|", wmax)
os.remove(out)
