#!/usr/bin/env python3
"""A/B timing of single kernels on ONE box in ONE process (devices differ by ~10 %, so variants are only
comparable inside one run).  Usage: python tests/kernel_ab.py VAR=a,b [VAR2=..] -- op [ne] [L]
   op in: vi, hv, dss, hypervis, lincomb, step"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    args = sys.argv[1:]
    sep = args.index("--")
    variants = [a.split("=") for a in args[:sep]]
    op = args[sep + 1]
    ne = int(args[sep + 2]) if len(args) > sep + 2 else 30
    L = int(args[sep + 3]) if len(args) > sep + 3 else 30
    from tempestmodel_amd.cubed_sphere import CubedSphereGrid, BaroclinicWaveJW
    from tempestmodel_amd.engine import Engine
    grid = CubedSphereGrid(ne, L, 30000.0, ppd=2)
    states = grid.evaluate_test_case(BaroclinicWaveJW(0.0, 30000.0, "exp"))
    names = [v[0] for v in variants]
    for combo in itertools.product(*[v[1].split(",") for v in variants]):
        for n, val in zip(names, combo):
            os.environ[n] = val
        e = Engine(grid)
        e.upload_state(0, states)
        for _ in range(3):
            e.step_ars343(150.0)
        e.sync()
        best = 1e9
        for rep in range(5):
            e.profile_enable(True); e.profile_reset()
            t0 = time.perf_counter()
            for _ in range(10):
                if op == "vi":
                    e.v_step_implicit(0, 2, 65.0)
                elif op == "hv":
                    e.copy_data(0, 1); e.h_step_explicit(0, 1, 65.0)
                elif op == "dss":
                    e.apply_dss(0)
                elif op == "step":
                    e.step_ars343(150.0)
            e.sync()
            wall = (time.perf_counter() - t0) / 10
            prof = e.profile()
            e.profile_enable(False)
            best = min(best, wall)
        print(dict(zip(names, combo)), "wall/iter %.1f us" % (best * 1e6), {k: "%.1f us" % (1e3 * v[0] / max(v[1], 1)) for k, v in prof.items() if v[1]})
        e.close()


if __name__ == "__main__":
    main()
