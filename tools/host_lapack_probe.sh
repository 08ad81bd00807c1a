#!/bin/bash
# Does the reference built against MKL equal the one built against OpenBLAS's netlib LAPACK on THIS host?
# (100 ARS343 steps at ne4 L10; prints the final checksums of both binaries.)
export MKL_THREADING_LAYER=SEQUENTIAL OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1
grep -m1 "model name" /proc/cpuinfo
for b in tmx_ref_dump tmx_ref_dump_lapack2; do
  echo "== $b"; oracle/_ref/$b --mode steps --ne 4 --levels 10 --dt 200 --warm 2 --steps 100 --every 100 --geom 0 | grep "checksum step100"
done
