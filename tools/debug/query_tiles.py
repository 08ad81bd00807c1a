import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_util as gu
from tempestmodel_amd.engine import Engine
for ne, npatch in ((4, 24), (8, 24), (6, 6), (8, 6), (12, 24)):
    g, _ = gu.make_grid(ne, 4, npatch)
    for n in (2, 3, 4, 8):
        if n > npatch: continue
        out = []
        for r in range(n):
            e = Engine(g, rank=r, n_ranks=n)
            out.append((e.info(7), e.info(8)))
            e.close()
        print(ne, npatch, n, out, flush=True)
