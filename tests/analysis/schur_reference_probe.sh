#!/bin/bash
# does the reference's reduced (Schur-complement) vertical solve run?  Build container only (needs
# oracle/_ref/tmx_ref_dump, i.e. /root/reference).  The driver constructs VerticalDynamicsSchur exactly as
# TempestInitialize.h:388-396 does for --vmethod SCHUR, on a finite-volume vertical discretisation of order 2 (the only
# one its constructor accepts: VerticalDynamicsSchur.cpp:203-214; LinearColumnOperatorFEM.cpp:227 wants an even order).
# Writes tests/analysis/schur_reference_probe.log
cd "$(dirname "$0")/../.."
export OMP_NUM_THREADS=1 MKL_THREADING_LAYER=SEQUENTIAL
LOG=tests/analysis/schur_reference_probe.log
: > $LOG
run() {
	echo "== tmx_ref_dump --mode steps --steps 1 --vmethod schur --geom 0 $* $EXTRA" >> $LOG
	timeout 300 oracle/_ref/tmx_ref_dump --mode steps --steps 1 --vmethod schur --geom 0 --out /tmp/schur_probe.tmxd "$@" 2>&1 \
		| awk '/nan/ { n++ } !/nan/ && !/^[-0-9. e+]*$/ { print } END { print "   lines holding NaN in the reference own diagnostic dump: " n+0 }' | tail -4 >> $LOG
	echo "   exit status ${PIPESTATUS[0]}" >> $LOG
}
run --ne 4 --levels 10 --dt 200
run --ne 4 --levels 30 --dt 100
run --ne 4 --levels 10 --dt 200 --pert none
run --ne 2 --levels 8 --dt 50 --case heldsuarez
echo "== vertical order 4 (TMX_REF_SCHUR_VORDER=4)" >> $LOG
TMX_REF_SCHUR_VORDER=4 run --ne 4 --levels 12 --dt 200
echo "== the same with Charney-Phillips staggering (TMX_REF_VSTAGGER_CPH=1)" >> $LOG
TMX_REF_VSTAGGER_CPH=1 run --ne 4 --levels 10 --dt 200
# round 3: the remaining switches of the reference's command line that reach this class
echo "== --vstagger LEV (TMX_REF_VSTAGGER=lev)" >> $LOG
TMX_REF_VSTAGGER=lev run --ne 4 --levels 10 --dt 200
echo "== --vstagger INT (TMX_REF_VSTAGGER=int)" >> $LOG
TMX_REF_VSTAGGER=int run --ne 4 --levels 10 --dt 200
echo "== --vmassfluxlevels (TMX_REF_SCHUR_MASSFLUX=1)" >> $LOG
TMX_REF_SCHUR_MASSFLUX=1 run --ne 4 --levels 10 --dt 200
echo "== --vmassfluxlevels with --vstagger LEV" >> $LOG
TMX_REF_SCHUR_MASSFLUX=1 TMX_REF_VSTAGGER=lev run --ne 4 --levels 10 --dt 200
echo "== --norefstate (TMX_REF_SCHUR_NOREF=1)" >> $LOG
TMX_REF_SCHUR_NOREF=1 run --ne 4 --levels 10 --dt 200
echo "== --norefstate with --vstagger LEV, small time step" >> $LOG
TMX_REF_SCHUR_NOREF=1 TMX_REF_VSTAGGER=lev run --ne 4 --levels 10 --dt 10
echo "== fully explicit vertical mode (--explicitvertical): the solve is never called" >> $LOG
run --ne 4 --levels 10 --dt 1 --explicitvertical
cat $LOG
