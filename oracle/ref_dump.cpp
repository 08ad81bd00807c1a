// oracle/ref_dump.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Dump / timing driver that LINKS the real reference (paullric/tempestmodel, compiled from
// /root/reference by oracle/Makefile into oracle/_ref/libtempestref.a) and only *calls* its
// public classes.  No reference source is copied: the Jablonowski-Williamson test-case class is
// pulled in by #include of the reference's own translation unit where it lies
// (test/nonhydro_sphere/BaroclinicWaveJWTest.cpp, its main() renamed away).
//
// What it produces (TMXD container, see tests/tmxd.py for the reader):
//   --mode percall : geometry + operators + the inputs/outputs of each hot-path call
//                    (H.StepExplicit, V.StepExplicit, ApplyDSS, V.StepImplicit,
//                     H.StepAfterSubCycle) made one at a time on the reference objects.
//   --mode steps   : instance-0 state after each of N real TimestepScheme::Step calls
//                    (+ area-weighted checksums), optionally with geometry.
//   --mode time    : wall time per step of the real reference (CPU baseline "reference").
//   --mode kessler : state + tracers before / after one KesslerPhysics::Perform (supercell case)
//   --case heldsuarez | supercell, --physics hs | kessler : the stock HeldSuarezTest / SupercellTest set-ups with
//                    their WorkflowProcess performed after every step, as Model::Go does (Model.cpp:470-481).
//
// Call sites mirrored (not copied): Model::Go init sequence (src/atm/Model.cpp:316-356),
// _TempestSetupCubedSphereModel (src/atm/TempestInitialize.h:476-586).

#define main tmx_unused_jw_main
#include "BaroclinicWaveJWTest.cpp"
#undef main
#define main tmx_unused_sw2_main
#include "SWTest2.cpp"
#undef main
#define main tmx_unused_schar_main
#include "ScharMountainSphereTest.cpp"
#undef main
#define main tmx_unused_heldsuarez_main
#include "HeldSuarezTest.cpp"
#undef main
// BASELINE config 4: the DCMIP2016 supercell with its Kessler microphysics.  SupercellTest.cpp and KesslerPhysics.cpp
// are the reference's own translation units (test/dcmip2016), their Fortran halves (interface/supercell_test.f90,
// interface/kessler.f90) are compiled by oracle/Makefile with /opt/rocm/bin/amdflang -O3 (= mk/defs.make:35).
#define main tmx_unused_supercell_main
#include "SupercellTest.cpp"
#undef main
#include "KesslerPhysics.h"

#include "test_inputs.h"

#include "GridPatchGLL.h"
#include "GridPatchCSGLL.h"
#include "HeldSuarezPhysics.h"
#include "CubedSphereTrans.h"
#include "PolynomialInterp.h"
#include "LinearColumnOperatorFEM.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>

///////////////////////////////////////////////////////////////////////////////
// TMXD writer

struct TmxdWriter {
	FILE * fp;
	TmxdWriter() : fp(NULL) {}
	void open(const std::string & path) {
		fp = fopen(path.c_str(), "wb");
		if (!fp) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
		fwrite("TMXD0001", 1, 8, fp);
	}
	void close() { if (fp) fclose(fp); fp = NULL; }
	void rec(const std::string & name, int dtype, const std::vector<size_t> & dims, const void * data) {
		if (!fp) return;
		unsigned int nl = name.size();
		fwrite(&nl, 4, 1, fp);
		fwrite(name.data(), 1, nl, fp);
		unsigned int dt = dtype, nd = dims.size();
		fwrite(&dt, 4, 1, fp);
		fwrite(&nd, 4, 1, fp);
		size_t tot = 1;
		for (size_t d = 0; d < dims.size(); d++) {
			unsigned long long v = dims[d];
			fwrite(&v, 8, 1, fp);
			tot *= dims[d];
		}
		fwrite(data, (dtype == 0) ? 8 : 4, tot, fp);
	}
	void f64(const std::string & name, const std::vector<size_t> & dims, const double * p) { rec(name, 0, dims, p); }
	void i32(const std::string & name, const std::vector<size_t> & dims, const int * p) { rec(name, 1, dims, p); }
	void scalar(const std::string & name, double v) { std::vector<size_t> d; d.push_back(1); f64(name, d, &v); }
	void iscalar(const std::string & name, int v) { std::vector<size_t> d; d.push_back(1); i32(name, d, &v); }
};

static std::vector<size_t> D1(size_t a) { std::vector<size_t> d; d.push_back(a); return d; }
static std::vector<size_t> D2(size_t a, size_t b) { std::vector<size_t> d = D1(a); d.push_back(b); return d; }
static std::vector<size_t> D3(size_t a, size_t b, size_t c) { std::vector<size_t> d = D2(a,b); d.push_back(c); return d; }
static std::vector<size_t> D4(size_t a, size_t b, size_t c, size_t e) { std::vector<size_t> d = D3(a,b,c); d.push_back(e); return d; }

static void dump2(TmxdWriter & w, const std::string & n, const DataArray2D<double> & a) {
	w.f64(n, D2(a.GetRows(), a.GetColumns()), &(a[0][0]));
}
static void dump3(TmxdWriter & w, const std::string & n, const DataArray3D<double> & a) {
	w.f64(n, D3(a.GetSize(0), a.GetSize(1), a.GetSize(2)), &(a[0][0][0]));
}
static void dump4(TmxdWriter & w, const std::string & n, const DataArray4D<double> & a) {
	w.f64(n, D4(a.GetSize(0), a.GetSize(1), a.GetSize(2), a.GetSize(3)), &(a[0][0][0][0]));
}

static void dumpOp(TmxdWriter & w, const std::string & n, const LinearColumnOperator & op) {
	const DataArray2D<double> & c = op.GetCoeffs();
	dump2(w, "op/" + n + "/coeff", c);
	std::vector<int> b(c.GetRows()), e(c.GetRows());
	for (size_t k = 0; k < c.GetRows(); k++) { b[k] = op.GetIxBegin()[k]; e[k] = op.GetIxEnd()[k]; }
	w.i32("op/" + n + "/begin", D1(b.size()), &b[0]);
	w.i32("op/" + n + "/end", D1(e.size()), &e[0]);
}

static std::string pname(int p) { char b[32]; snprintf(b, 32, "p%d/", p); return std::string(b); }

///////////////////////////////////////////////////////////////////////////////

static void dumpState(TmxdWriter & w, GridGLL * pGrid, const std::string & tag, int ix) {
	for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
		GridPatch * pPatch = pGrid->GetActivePatch(n);
		int p = pPatch->GetPatchIndex();
		dump4(w, "state/" + tag + "/" + pname(p) + "node", pPatch->GetDataState(ix, DataLocation_Node));
		dump4(w, "state/" + tag + "/" + pname(p) + "redge", pPatch->GetDataState(ix, DataLocation_REdge));
		if (pPatch->GetDataTracers(ix).GetSize(0) > 0)
			dump4(w, "state/" + tag + "/" + pname(p) + "tracers", pPatch->GetDataTracers(ix));
	}
}

static void dumpGeometry(TmxdWriter & w, Model & model, GridGLL * pGrid) {
	const PhysicalConstants & phys = model.GetPhysicalConstants();
	w.scalar("phys/earth_radius", phys.GetEarthRadius());
	w.scalar("phys/g", phys.GetG());
	w.scalar("phys/omega", phys.GetOmega());
	w.scalar("phys/alpha", phys.GetAlpha());
	w.scalar("phys/Rd", phys.GetR());
	w.scalar("phys/cp", phys.GetCp());
	w.scalar("phys/cv", phys.GetCv());
	w.scalar("phys/p0", phys.GetP0());
	w.scalar("grid/ztop", pGrid->GetZtop());
	w.scalar("grid/reference_length", pGrid->GetReferenceLength());
	w.iscalar("grid/has_rayleigh", pGrid->HasRayleighFriction() ? 1 : 0);
	w.iscalar("grid/has_uniform_diffusion", pGrid->HasUniformDiffusion() ? 1 : 0);

	int L = pGrid->GetRElements();
	w.f64("grid/reta_levels", D1(L), &(pGrid->GetREtaLevels()[0]));
	w.f64("grid/reta_interfaces", D1(L+1), &(pGrid->GetREtaInterfaces()[0]));
	w.f64("grid/reta_levels_normarea", D1(L), &(pGrid->GetREtaLevelsNormArea()[0]));
	w.f64("grid/reta_interfaces_normarea", D1(L+1), &(pGrid->GetREtaInterfacesNormArea()[0]));

	dump2(w, "op/dx_basis_1d", pGrid->GetDxBasis1D());
	dump2(w, "op/stiffness_1d", pGrid->GetStiffness1D());
	w.f64("op/gll_weights_1d", D1(pGrid->GetGLLWeights1D().GetRows()), &(pGrid->GetGLLWeights1D()[0]));

	dumpOp(w, "interp_node_to_redge", pGrid->GetOpInterpNodeToREdge());
	dumpOp(w, "interp_redge_to_node", pGrid->GetOpInterpREdgeToNode());
	dumpOp(w, "diff_node_to_node", pGrid->GetOpDiffNodeToNode());
	dumpOp(w, "diff_node_to_redge", pGrid->GetOpDiffNodeToREdge());
	dumpOp(w, "diff_redge_to_node", pGrid->GetOpDiffREdgeToNode());
	dumpOp(w, "diff_redge_to_redge", pGrid->GetOpDiffREdgeToREdge());
	dumpOp(w, "diffdiff_node_to_node", pGrid->GetOpDiffDiffNodeToNode());
	dumpOp(w, "diffdiff_redge_to_redge", pGrid->GetOpDiffDiffREdgeToREdge());
	dumpOp(w, "penalty_left", pGrid->GetOpPenaltyNodeToNode().GetLeftOp());
	dumpOp(w, "penalty_right", pGrid->GetOpPenaltyNodeToNode().GetRightOp());

	for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
		GridPatchGLL * pPatch = dynamic_cast<GridPatchGLL*>(pGrid->GetActivePatch(n));
		const PatchBox & box = pPatch->GetPatchBox();
		int p = pPatch->GetPatchIndex();
		std::string pn = pname(p);
		int ibox[8] = {
			box.GetPanel(), box.GetHaloElements(),
			box.GetAGlobalInteriorBegin(), box.GetAGlobalInteriorEnd(),
			box.GetBGlobalInteriorBegin(), box.GetBGlobalInteriorEnd(),
			box.GetATotalWidth(), box.GetBTotalWidth()};
		w.i32(pn + "box", D1(8), ibox);
		int inb[8];
		for (int d = 0; d < 8; d++) inb[d] = pPatch->GetNeighborPanel((Direction)d);
		w.i32(pn + "neighbor_panels", D1(8), inb);
		w.scalar(pn + "element_delta_a", pPatch->GetElementDeltaA());
		w.scalar(pn + "element_delta_b", pPatch->GetElementDeltaB());
		w.f64(pn + "a_nodes", D1(pPatch->GetANodes().GetRows()), &(pPatch->GetANodes()[0]));
		w.f64(pn + "b_nodes", D1(pPatch->GetBNodes().GetRows()), &(pPatch->GetBNodes()[0]));
		dump2(w, pn + "lon", pPatch->GetLongitude());
		dump2(w, pn + "lat", pPatch->GetLatitude());
		dump2(w, pn + "jacobian2d", pPatch->GetJacobian2D());
		dump3(w, pn + "contra_metric_2d_a", pPatch->GetContraMetric2DA());
		dump3(w, pn + "contra_metric_2d_b", pPatch->GetContraMetric2DB());
		dump3(w, pn + "jacobian", pPatch->GetJacobian());
		dump3(w, pn + "jacobian_redge", pPatch->GetJacobianREdge());
		dump4(w, pn + "contra_metric_a", pPatch->GetContraMetricA());
		dump4(w, pn + "contra_metric_b", pPatch->GetContraMetricB());
		dump4(w, pn + "contra_metric_xi", pPatch->GetContraMetricXi());
		dump4(w, pn + "contra_metric_a_redge", pPatch->GetContraMetricAREdge());
		dump4(w, pn + "contra_metric_b_redge", pPatch->GetContraMetricBREdge());
		dump4(w, pn + "contra_metric_xi_redge", pPatch->GetContraMetricXiREdge());
		dump4(w, pn + "deriv_r_node", pPatch->GetDerivRNode());
		dump4(w, pn + "deriv_r_redge", pPatch->GetDerivRREdge());
		dump3(w, pn + "element_area_node", pPatch->GetElementAreaNode());
		dump3(w, pn + "element_area_redge", pPatch->GetElementAreaREdge());
		dump2(w, pn + "topography", pPatch->GetTopography());
		dump3(w, pn + "topography_deriv", pPatch->GetTopographyDeriv());
		dump2(w, pn + "coriolis_f", pPatch->GetCoriolisF());
		dump3(w, pn + "z_levels", pPatch->GetZLevels());
		dump3(w, pn + "z_interfaces", pPatch->GetZInterfaces());
		dump4(w, pn + "ref_node", pPatch->GetReferenceState(DataLocation_Node));
		dump4(w, pn + "ref_redge", pPatch->GetReferenceState(DataLocation_REdge));
		if (pGrid->GetModel().GetEquationSet().GetTracers() > 0) dump4(w, pn + "ref_tracers", pPatch->GetReferenceTracers());
		if (pGrid->HasRayleighFriction()) {
			dump3(w, pn + "rayleigh_node", pPatch->GetRayleighStrength(DataLocation_Node));
			dump3(w, pn + "rayleigh_redge", pPatch->GetRayleighStrength(DataLocation_REdge));
		}
	}
}

// Probe the reference's covector panel transform with unit vectors at every halo node of every
// patch edge that borders a different panel (same call TransformHaloVelocities makes,
// GridPatchCSGLL.cpp:1783-1924) and record the 2x2 matrices.
static void dumpHaloTransforms(TmxdWriter & w, GridGLL * pGrid) {
	for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
		GridPatchGLL * pPatch = dynamic_cast<GridPatchGLL*>(pGrid->GetActivePatch(n));
		const PatchBox & box = pPatch->GetPatchBox();
		int p = pPatch->GetPatchIndex();
		const DataArray1D<double> & dA = pPatch->GetANodes();
		const DataArray1D<double> & dB = pPatch->GetBNodes();
		// edges: 0=right,1=top,2=left,3=bottom (Direction enum order)
		for (int e = 0; e < 4; e++) {
			int ixPanel = pPatch->GetNeighborPanel((Direction)e);
			if (ixPanel == box.GetPanel()) continue;
			bool alongB = (e == 0) || (e == 2);
			int nAlong = alongB ? box.GetBTotalWidth() : box.GetATotalWidth();
			int fixed;
			if (e == 0) fixed = box.GetAInteriorEnd();
			else if (e == 1) fixed = box.GetBInteriorEnd();
			else if (e == 2) fixed = box.GetAInteriorBegin() - 1;
			else fixed = box.GetBInteriorBegin() - 1;
			std::vector<double> m(nAlong * 4);
			for (int s = 0; s < nAlong; s++) {
				int i = alongB ? fixed : s;
				int j = alongB ? s : fixed;
				double X = tan(dA[i]), Y = tan(dB[j]);
				double a0 = 1.0, b0 = 0.0, a1 = 0.0, b1 = 1.0;
				CubedSphereTrans::CoVecPanelTrans(ixPanel, box.GetPanel(), a0, b0, X, Y);
				CubedSphereTrans::CoVecPanelTrans(ixPanel, box.GetPanel(), a1, b1, X, Y);
				// out = M * in, column 0 = image of (1,0), column 1 = image of (0,1)
				m[4*s+0] = a0; m[4*s+1] = a1; m[4*s+2] = b0; m[4*s+3] = b1;
			}
			char nm[64]; snprintf(nm, 64, "halo_trans/p%d/e%d", p, e);
			w.f64(nm, D3(nAlong, 2, 2), &m[0]);
			snprintf(nm, 64, "halo_trans/p%d/e%d_srcpanel", p, e);
			w.iscalar(nm, ixPanel);
		}
	}
}

static void printChecksums(Grid * pGrid, const char * tag, TmxdWriter * w) {
	DataArray1D<double> dSums;
	pGrid->Checksum(DataType_State, dSums, 0, ChecksumType_Sum);
	printf("checksum %s", tag);
	for (size_t c = 0; c < dSums.GetRows(); c++) printf(" %.15e", dSums[c]);
	printf("\n");
	if (w && w->fp) w->f64(std::string("checksum/") + tag, D1(dSums.GetRows()), &dSums[0]);
}

///////////////////////////////////////////////////////////////////////////////

int main(int argc, char ** argv) {
	TempestInitialize(&argc, &argv);

	std::string mode = "steps", out = "", scheme = "ars343", pert = "exp", tcase = "jw";
	int ne = 4, levels = 10, nsteps = 3, npatch = 6, nwarm = 0, geom = 1, every = 1, ntracers = 0, hvorder = 4;
	double dt = 200.0, ztop = 30000.0, planet = 1.0, udiffS = 0.0, udiffV = 0.0;
	bool fExplicitVertical = false, fZtopGiven = false;
	std::string physics = "none", vmethod = "v1";
	double nuArg = -1.0, moisten = 0.0, offCentering = 0.0;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		const char * v = (i + 1 < argc) ? argv[i+1] : "";
		if (a == "--mode") { mode = v; i++; }
		else if (a == "--out") { out = v; i++; }
		else if (a == "--scheme") { scheme = v; i++; }
		else if (a == "--pert") { pert = v; i++; }
		else if (a == "--case") { tcase = v; i++; }
		else if (a == "--ne") { ne = atoi(v); i++; }
		else if (a == "--levels") { levels = atoi(v); i++; }
		else if (a == "--steps") { nsteps = atoi(v); i++; }
		else if (a == "--warm") { nwarm = atoi(v); i++; }
		else if (a == "--patches") { npatch = atoi(v); i++; }
		else if (a == "--geom") { geom = atoi(v); i++; }
		else if (a == "--every") { every = atoi(v); i++; }
		else if (a == "--tracers") { ntracers = atoi(v); i++; }
		else if (a == "--hypervis") { hvorder = atoi(v); i++; }      // hyperviscosity order 4 (default) or 2
		else if (a == "--dt") { dt = atof(v); i++; }
		else if (a == "--ztop") { ztop = atof(v); fZtopGiven = true; i++; }
		else if (a == "--nu") { nuArg = atof(v); i++; }               // hyperviscosity coefficient (all three), default per case
		else if (a == "--moisten") { moisten = atof(v); i++; }        // --mode kessler: second call on a moistened copy
		else if (a == "--offcentering") { offCentering = atof(v); i++; }   // TimestepSchemeStrang ctor argument
		else if (a == "--vmethod") { vmethod = v; i++; }              // v1 (VerticalDynamicsFEM) | schur (VerticalDynamicsSchur, with --vdisc FV)
		else if (a == "--physics") { physics = v; i++; }              // hs | kessler: WorkflowProcess after every step
		else if (a == "--planet") { planet = atof(v); i++; }          // reduced-planet factor X (with --udiff)
		else if (a == "--udiff") { udiffS = atof(v); udiffV = atof((i + 2 < argc) ? argv[i+2] : "0"); i += 2; }   // uniform diffusion K_scalar K_vector
		else if (a == "--explicitvertical") { fExplicitVertical = true; }
		else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
	}

try {
	AnnounceSetVerbosityLevel(0);

	const bool fSW = (tcase == "sw2");
	const bool fHS = (tcase == "heldsuarez"), fSC = (tcase == "supercell");
	if (fSW) { levels = 1; }
	if (fHS && !fZtopGiven) ztop = 35000.0;             // HeldSuarezTest.cpp:336
	if (fSC) {
		if (!fZtopGiven) ztop = 20000.0;                // DCMIP2016 supercell: 40 levels of 500 m
		ntracers = 3;
		if (planet == 1.0) planet = 120.0;              // SupercellTest.cpp:249
		if (!fExplicitVertical) { fprintf(stderr, "supercell needs --explicitvertical (VerticalDynamicsFEM.cpp:3914-3917)\n"); return 2; }
	}
	// tracers are declared on the EquationSet before the Model exists (as test/dcmip2016/SupercellTest.cpp:271-281 does)
	EquationSet eqn(fSW ? EquationSet::ShallowWaterEquations : EquationSet::PrimitiveNonhydrostaticEquations);
	if (fSC) {
		eqn.InsertTracer("RhoQv", "RhoQv"); eqn.InsertTracer("RhoQc", "RhoQc"); eqn.InsertTracer("RhoQr", "RhoQr");   // SupercellTest.cpp:271-275
	} else
	for (int c = 0; c < ntracers; c++) {
		char nm[16]; snprintf(nm, 16, "RhoQ%d", c);
		eqn.InsertTracer(nm, nm);
	}
	UserDataMeta metaUserData;
	if (fSC) metaUserData.InsertDataItem2D("PRECT");    // SupercellTest.cpp:279
	Model model(eqn, metaUserData);

	int isec = (int)dt;
	int iusec = (int)((dt - isec) * 1.0e6 + 0.5);
	Time timeDeltaT(0, 0, 0, isec, iusec, Time::CalendarNoLeap, Time::TypeDelta);
	// End time == start time: Model::Go() then performs exactly its initialisation sequence
	// (Model.cpp:343-366) and returns before the time loop; steps are driven from here.
	model.SetDeltaT(timeDeltaT);
	model.SetEndTime(model.GetStartTime());

	// Method of lines (defaults of TempestInitialize.h:112-144)
	if (scheme == "ars343") model.SetTimestepScheme(new TimestepSchemeARS343(model));
	else if (scheme == "ars232") model.SetTimestepScheme(new TimestepSchemeARS232(model));
	else if (scheme == "ark232") model.SetTimestepScheme(new TimestepSchemeARK232(model));
	else if (scheme == "ars222") model.SetTimestepScheme(new TimestepSchemeARS222(model));
	else if (scheme == "ars443") model.SetTimestepScheme(new TimestepSchemeARS443(model));
	else if (scheme == "strang") model.SetTimestepScheme(new TimestepSchemeStrang(model, offCentering));
	else if (scheme == "strang_fe") model.SetTimestepScheme(new TimestepSchemeStrang(model, offCentering, TimestepSchemeStrang::ForwardEuler));
	else if (scheme == "strang_rk4") model.SetTimestepScheme(new TimestepSchemeStrang(model, offCentering, TimestepSchemeStrang::RungeKutta4));
	else if (scheme == "strang_ssp3") model.SetTimestepScheme(new TimestepSchemeStrang(model, offCentering, TimestepSchemeStrang::RungeKuttaSSP3));
	else if (scheme == "strang_ssprk53") model.SetTimestepScheme(new TimestepSchemeStrang(model, offCentering, TimestepSchemeStrang::RungeKuttaSSPRK53));
	else { fprintf(stderr, "bad scheme\n"); return 2; }

	// order 2 is plain viscosity: coefficients of a sensible magnitude for it (the 1e15 defaults belong to order 4)
	// supercell: the DCMIP2016 set-up has the uniform diffusion as its only dissipation (--nu* 0 on the reference's
	// command line; with SupercellTest.cpp:285's reference length / X the 1e15 defaults would be scaled UP by X^3.2)
	const double dNu = (nuArg >= 0.0) ? nuArg : (fSC ? 0.0 : ((hvorder == 2) ? 2.0e5 : 1.0e15));
	HorizontalDynamicsFEM * pH = new HorizontalDynamicsFEM(model, 4, hvorder, dNu, dNu, dNu, 0.0);
	model.SetHorizontalDynamics(pH);
	// levels == 1 -> VerticalDynamicsStub, as TempestInitialize.h:362-365
	VerticalDynamics * pV = fSW ? (VerticalDynamics *)(new VerticalDynamicsStub(model))
	                : (vmethod == "schur") ? (VerticalDynamics *)(new VerticalDynamicsSchur(model, 4, getenv("TMX_REF_SCHUR_VORDER") ? atoi(getenv("TMX_REF_SCHUR_VORDER")) : 2, 0, fExplicitVertical,
	                                                              getenv("TMX_REF_SCHUR_NOREF") == NULL /* --norefstate */, getenv("TMX_REF_SCHUR_MASSFLUX") != NULL /* --vmassfluxlevels */))
	                            : (VerticalDynamics *)(new VerticalDynamicsFEM(model, 4, 1, 0, fExplicitVertical, true, false));
	model.SetVerticalDynamics(pV);

	GridCSGLL * pGrid = new GridCSGLL(model);
	pGrid->DefineParameters();
	pGrid->SetParameters(levels, (npatch < 6) ? 6 : npatch, ne, 4, 4, (vmethod == "schur") ? (getenv("TMX_REF_SCHUR_VORDER") ? atoi(getenv("TMX_REF_SCHUR_VORDER")) : 2) : 1,
		(vmethod == "schur") ? Grid::VerticalDiscretization_FiniteVolume : Grid::VerticalDiscretization_FiniteElement,
		// probes of the Schur path only (tests/analysis/schur_reference_probe.sh): --vstagger of TempestInitialize.h:137
		getenv("TMX_REF_VSTAGGER_CPH") ? Grid::VerticalStaggering_CharneyPhillips
		: (getenv("TMX_REF_VSTAGGER") && std::string(getenv("TMX_REF_VSTAGGER")) == "lev") ? Grid::VerticalStaggering_Levels
		: (getenv("TMX_REF_VSTAGGER") && std::string(getenv("TMX_REF_VSTAGGER")) == "int") ? Grid::VerticalStaggering_Interfaces
		: Grid::VerticalStaggering_Lorenz);
	pGrid->InitializeDataLocal();
	model.SetGrid(pGrid, npatch);

	BaroclinicWaveJWTest::PerturbationType ePert =
		(pert == "exp") ? BaroclinicWaveJWTest::PerturbationType_Exp
		                : BaroclinicWaveJWTest::PerturbationType_None;
	if (fSW) {
		model.SetTestCase(new ShallowWaterTestCase2(2998.104995, 38.61068277, 0.0));   // SWTest2.cpp defaults
	} else if (fHS) {
		model.SetTestCase(new HeldSuarezTest(30000.0, 25.0, ztop));          // HeldSuarezTest.cpp:336-338 defaults
	} else if (fSC) {
		model.GetGrid()->SetReferenceLength(model.GetGrid()->GetReferenceLength() / planet);   // SupercellTest.cpp:285-286
		model.SetTestCase(new SupercellTest(ztop, planet));
	} else if (tcase == "schar") {
		// ScharMountainSphereTest.cpp defaults (:417-441): reduced planet X = 500, no rotation, Rayleigh layer above 20 km
		model.SetTestCase(new ScharMountainSphereTest(ztop, 500.0, 0.0, 45.0, 0.0, 250.0, 5000.0, 4000.0, 300.0, 20.0, 0.0, 20000.0, 25.0));
		model.GetGrid()->SetReferenceLength(0.5 * M_PI / 30.0 * 500.0);
	} else if (udiffS != 0.0 || udiffV != 0.0 || planet != 1.0) {
		model.SetTestCase(new JWSmallPlanetDiffusion(ztop, ePert, ntracers, planet, udiffS, udiffV));
		model.GetGrid()->SetReferenceLength(0.5 * M_PI / 30.0 * planet);   // as ScharMountainSphereTest.cpp:375 scales it
	} else if (ntracers > 0) {
		model.SetTestCase(new JWWithTracers(0.0, ztop, ePert, ntracers));
	} else {
		model.SetTestCase(new BaroclinicWaveJWTest(0.0, ztop, ePert));
	}

	// Model::Go init sequence (EvaluateGeometricTerms, component Initialize)
	model.Go();

	TmxdWriter w;
	if (out != "") w.open(out);
	w.iscalar("cfg/ne", ne);
	w.iscalar("cfg/np", 4);
	w.iscalar("cfg/levels", levels);
	w.iscalar("cfg/npatch", pGrid->GetActivePatchCount());
	w.iscalar("cfg/ninstances", model.GetComponentDataInstances());
	w.iscalar("cfg/ntracers", ntracers);
	w.iscalar("cfg/hypervis_order", hvorder);
	w.iscalar("cfg/fully_explicit", fExplicitVertical ? 1 : 0);
	w.iscalar("cfg/uniform_diffusion", pGrid->HasUniformDiffusion() ? 1 : 0);
	w.scalar("cfg/udiff_scalar", pGrid->GetScalarUniformDiffusionCoeff());
	w.scalar("cfg/udiff_vector", pGrid->GetVectorUniformDiffusionCoeff());
	w.scalar("cfg/ztop", pGrid->GetZtop());
	w.scalar("cfg/dt", dt);
	w.scalar("cfg/nu", dNu);
	if (geom && w.fp) {
		dumpGeometry(w, model, pGrid);
		dumpHaloTransforms(w, pGrid);
	}

	Time time = model.GetStartTime();
	TimestepScheme * pTS = model.GetTimestepScheme();

	// WorkflowProcess of the stock test, performed after every step like Model::Go does (Model.cpp:470-481)
	WorkflowProcess * pPhysics = NULL;
	if (physics == "hs") pPhysics = new HeldSuarezPhysics(model, timeDeltaT);
	else if (physics == "kessler") pPhysics = new KesslerPhysics(model, timeDeltaT);
	else if (physics != "none") { fprintf(stderr, "bad --physics\n"); return 2; }
	if (pPhysics) pPhysics->Initialize(time);
	w.iscalar("cfg/physics", (physics == "hs") ? 1 : ((physics == "kessler") ? 2 : 0));
	if (physics == "hs") {
		// inputs of HeldSuarezPhysics::Perform that are not part of the prognostic state (see --mode physics)
		const PhysicalConstants & phys = model.GetPhysicalConstants();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			const PatchBox & box = pPatch->GetPatchBox();
			const DataArray4D<double> & dE = pPatch->GetDataState(0, DataLocation_REdge);
			DataArray2D<double> ps(box.GetATotalWidth(), box.GetBTotalWidth());
			for (int i = 0; i < box.GetATotalWidth(); i++)
			for (int j = 0; j < box.GetBTotalWidth(); j++)
				ps[i][j] = phys.PressureFromRhoTheta(dE[4][i][j][0] * dE[2][i][j][0]);
			dump2(w, pname(pPatch->GetPatchIndex()) + "hs_surface_pressure", ps);
			dump2(w, pname(pPatch->GetPatchIndex()) + "hs_latitude", pPatch->GetLatitude());
		}
		w.scalar("phys/gamma", phys.GetGamma());
		w.scalar("phys/kappa", phys.GetKappa());
	}
	if (physics == "kessler") {
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			dump3(w, pname(pPatch->GetPatchIndex()) + "kessler_z_levels", pPatch->GetZLevels());
		}
	}

	printChecksums(pGrid, "initial", &w);
	dumpState(w, pGrid, "initial", 0);

	// optional real warm-up steps (gives W != 0)
	for (int s = 0; s < nwarm; s++) {
		pTS->Step(s == 0, false, time, dt);
		time += timeDeltaT;
		if (pPhysics) pPhysics->Perform(time);
	}
	if (nwarm > 0) {
		printChecksums(pGrid, "warm", &w);
		dumpState(w, pGrid, "warm", 0);
	}

	if (mode == "percall") {
		const double sdt = dt * 0.4358665215084590;
		// H explicit
		pGrid->CopyData(0, 1, DataType_State);
		pGrid->CopyData(0, 1, DataType_Tracers);
		pH->StepExplicit(0, 1, time, sdt);
		dumpState(w, pGrid, "h_explicit_in_after", 0);   // input instance after the prologue mutation
		dumpState(w, pGrid, "h_explicit", 1);
		// V explicit (continues on the same update instance, as the stepper does)
		pV->StepExplicit(0, 1, time, sdt);
		dumpState(w, pGrid, "v_explicit", 1);
		// DSS
		pGrid->PostProcessSubstage(1, DataType_State);
		pGrid->PostProcessSubstage(1, DataType_Tracers);
		dumpState(w, pGrid, "dss", 1);
		// V implicit
		pGrid->CopyData(1, 2, DataType_State);
		pGrid->CopyData(1, 2, DataType_Tracers);
		pV->StepImplicit(2, 2, time, sdt);
		dumpState(w, pGrid, "v_implicit", 2);
		// hyperviscosity (init 2 -> update 3, work 4)
		pH->StepAfterSubCycle(2, 3, 4, time, dt);
		dumpState(w, pGrid, "hypervis_work", 4);
		dumpState(w, pGrid, "hypervis", 3);
		// stage algebra: X5 <- 0.3*X5(=0) ... use explicit combo on instances 0..3
		DataArray1D<double> dCombo(7);
		dCombo[0] = 0.25; dCombo[1] = -0.5; dCombo[2] = 1.75; dCombo[3] = 0.5;
		pGrid->LinearCombineData(dCombo, 3, DataType_State);
		pGrid->LinearCombineData(dCombo, 3, DataType_Tracers);
		dumpState(w, pGrid, "lincomb", 3);

	} else if (mode == "physics") {
		// HeldSuarezPhysics::Perform (src/atm/HeldSuarezPhysics.cpp:60-301) on instance 0.
		// Its "surface pressure" is PressureFromRhoTheta(REdge[rho][0] * REdge[rhotheta][0]) of whatever sits in the
		// interface slots of rho and rho*theta -- with Lorenz staggering those are never updated after the test case
		// filled them, so the value is dumped as an input of the call.
		const PhysicalConstants & phys = model.GetPhysicalConstants();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			const PatchBox & box = pPatch->GetPatchBox();
			const DataArray4D<double> & dE = pPatch->GetDataState(0, DataLocation_REdge);
			DataArray2D<double> ps(box.GetATotalWidth(), box.GetBTotalWidth());
			for (int i = 0; i < box.GetATotalWidth(); i++)
			for (int j = 0; j < box.GetBTotalWidth(); j++)
				ps[i][j] = phys.PressureFromRhoTheta(dE[4][i][j][0] * dE[2][i][j][0]);
			dump2(w, pname(pPatch->GetPatchIndex()) + "hs_surface_pressure", ps);
			dump2(w, pname(pPatch->GetPatchIndex()) + "hs_latitude", pPatch->GetLatitude());
		}
		w.scalar("phys/gamma", phys.GetGamma());
		w.scalar("phys/kappa", phys.GetKappa());
		dumpState(w, pGrid, "hs_before", 0);
		HeldSuarezPhysics hs(model, timeDeltaT);
		hs.Perform(time);
		dumpState(w, pGrid, "hs_after", 0);

	} else if (mode == "kessler") {
		// SURVEY 8(f) item 1, second half: KesslerPhysics::Perform (test/dcmip2016/KesslerPhysics.cpp:83-285 ->
		// interface/kessler.f90) on instance 0; --physics kessler is implied.
		if (physics != "kessler") { fprintf(stderr, "--mode kessler needs --physics kessler\n"); return 2; }
		dumpState(w, pGrid, "kessler_before", 0);
		pPhysics->Perform(time);
		dumpState(w, pGrid, "kessler_after", 0);
		if (moisten > 0.0) {
			// Second call on a test INPUT of our own: the same state with the vapour scaled up and cloud / rain water
			// added in closed form, so that condensation, autoconversion, accretion, sedimentation (with sub-cycling)
			// and evaporation all act -- on the coarse grids of the fixtures the stock supercell never forms cloud.
			for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
				GridPatch * pPatch = pGrid->GetActivePatch(n);
				const PatchBox & box = pPatch->GetPatchBox();
				DataArray4D<double> & dT = pPatch->GetDataTracers(0);
				const DataArray4D<double> & dN = pPatch->GetDataState(0, DataLocation_Node);
				const DataArray2D<double> & dLon = pPatch->GetLongitude();
				const DataArray2D<double> & dLat = pPatch->GetLatitude();
				const DataArray3D<double> & dZ = pPatch->GetZLevels();
				for (int i = box.GetAInteriorBegin(); i < box.GetAInteriorEnd(); i++)
				for (int j = box.GetBInteriorBegin(); j < box.GetBInteriorEnd(); j++)
				for (int k = 0; k < pGrid->GetRElements(); k++) {
					const double s1 = 0.5 * (1.0 + sin(3.0 * dLon[i][j]) * cos(2.0 * dLat[i][j]));
					dT[0][i][j][k] *= 1.0 + (moisten - 1.0) * s1;
					dT[1][i][j][k] = dN[4][i][j][k] * ((dZ[i][j][k] < 9000.0) ? 0.003 * s1 * s1 : 0.0);
					dT[2][i][j][k] = dN[4][i][j][k] * ((dZ[i][j][k] < 12000.0 && s1 > 0.3) ? 0.006 * (s1 - 0.3) : 0.0);
				}
			}
			dumpState(w, pGrid, "kessler_moist_before", 0);
			pPhysics->Perform(time);
			dumpState(w, pGrid, "kessler_moist_after", 0);
		}
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			dump3(w, pname(pPatch->GetPatchIndex()) + "kessler_prect", pPatch->GetUserData2D());
		}

	} else if (mode == "interp") {
		// Grid::ReduceInterpolate (src/atm/Grid.cpp:867-992 -> GridPatchCSGLL::InterpolateData,
		// GridPatchCSGLL.cpp:1365-1780), the interpolation behind OutputManagerReference (OutputManagerReference.cpp:
		// 588-628), on instance 0 after the warm-up steps.  Dumped: the sample points exactly as the output manager
		// derives them (ConvertReferenceToPatchCoord), the per-point quantities a caller obtains from the reference's
		// own functions (element origin, Lagrange coefficients, RLL transform matrix, vertical operators), and the
		// outputs for the flag combinations.
		const int nlon = 12, nlat = 6, npts = nlon * nlat + 2;
		DataArray1D<double> dLon(npts), dLat(npts), dAlpha(npts), dBeta(npts);
		DataArray1D<int> iPatch(npts);
		for (int j = 0; j < nlat; j++) for (int i = 0; i < nlon; i++) {
			dLon[j * nlon + i] = (i + 0.37) * 2.0 * M_PI / nlon;
			dLat[j * nlon + i] = -0.5 * M_PI + (j + 0.41) * M_PI / nlat;
		}
		dLon[npts - 2] = 0.0; dLat[npts - 2] = 0.0;                     // centre of panel 0: a GLL node when ne is even
		dLon[npts - 1] = 0.25 * M_PI; dLat[npts - 1] = 0.1;             // on the edge between panels 0 and 1
		pGrid->ConvertReferenceToPatchCoord(dLon, dLat, dAlpha, dBeta, iPatch);
		const int nreta = 5;
		DataArray1D<double> dREta(nreta);
		const double reta_v[5] = { 0.0, 0.13, 0.5, 0.777, 1.0 };
		for (int k = 0; k < nreta; k++) dREta[k] = reta_v[k];
		std::vector<double> va(npts), vb(npts), vreta(nreta), ca(npts * 4), cb(npts * 4), mat(npts * 4);
		std::vector<int> vp(npts), via(npts), vib(npts);
		for (int q = 0; q < npts; q++) {
			va[q] = dAlpha[q]; vb[q] = dBeta[q]; vp[q] = iPatch[q];
			GridPatch * pPatch = NULL;
			for (int n = 0; n < pGrid->GetActivePatchCount(); n++)
				if (pGrid->GetActivePatch(n)->GetPatchIndex() == iPatch[q]) pPatch = pGrid->GetActivePatch(n);
			const PatchBox & box = pPatch->GetPatchBox();
			GridPatchGLL * pGLL = dynamic_cast<GridPatchGLL *>(pPatch);
			int iA = (int)((dAlpha[q] - pPatch->GetAEdge(box.GetAInteriorBegin())) / pGLL->GetElementDeltaA());
			int iB = (int)((dBeta[q] - pPatch->GetBEdge(box.GetBInteriorBegin())) / pGLL->GetElementDeltaB());
			if (iA < 0) iA = 0;
			if (iA >= box.GetAInteriorWidth() / 4) iA = box.GetAInteriorWidth() / 4 - 1;
			if (iB < 0) iB = 0;
			if (iB >= box.GetBInteriorWidth() / 4) iB = box.GetBInteriorWidth() / 4 - 1;
			iA = box.GetHaloElements() + iA * 4; iB = box.GetHaloElements() + iB * 4;
			via[q] = iA; vib[q] = iB;
			PolynomialInterp::LagrangianPolynomialCoeffs(4, &(pPatch->GetAEdges()[iA]), &ca[q * 4], dAlpha[q]);
			PolynomialInterp::LagrangianPolynomialCoeffs(4, &(pPatch->GetBEdges()[iB]), &cb[q * 4], dBeta[q]);
			// the covariant (alpha, beta) -> (lon, lat) map is linear: probe it with unit vectors, as for the halo transforms
			CubedSphereTrans::CoVecTransRLLFromABP(tan(dAlpha[q]), tan(dBeta[q]), box.GetPanel(), 1.0, 0.0, mat[q * 4 + 0], mat[q * 4 + 2]);
			CubedSphereTrans::CoVecTransRLLFromABP(tan(dAlpha[q]), tan(dBeta[q]), box.GetPanel(), 0.0, 1.0, mat[q * 4 + 1], mat[q * 4 + 3]);
		}
		for (int k = 0; k < nreta; k++) vreta[k] = dREta[k];
		w.rec("interp/alpha", 0, std::vector<size_t>(1, npts), &va[0]);
		w.rec("interp/beta", 0, std::vector<size_t>(1, npts), &vb[0]);
		w.rec("interp/patch", 1, std::vector<size_t>(1, npts), &vp[0]);
		w.rec("interp/node_a", 1, std::vector<size_t>(1, npts), &via[0]);
		w.rec("interp/node_b", 1, std::vector<size_t>(1, npts), &vib[0]);
		{ std::vector<size_t> d2(2); d2[0] = npts; d2[1] = 4;
		  w.rec("interp/coeff_a", 0, d2, &ca[0]); w.rec("interp/coeff_b", 0, d2, &cb[0]); w.rec("interp/rll_from_abp", 0, d2, &mat[0]); }
		w.rec("interp/reta", 0, std::vector<size_t>(1, nreta), &vreta[0]);
		w.scalar("interp/earth_radius", model.GetPhysicalConstants().GetEarthRadius());
		for (int src = 0; src < 2; src++) {
			LinearColumnInterpFEM op;
			op.Initialize(src ? LinearColumnInterpFEM::InterpSource_Interfaces : LinearColumnInterpFEM::InterpSource_Levels,
				1, pGrid->GetREtaLevels(), pGrid->GetREtaInterfaces(), dREta);
			dump2(w, src ? "interp/op_interfaces" : "interp/op_levels", op.GetCoeffs());
		}
		dumpState(w, pGrid, "interp_in", 0);
		const int nc = 5;
		for (int inc = 0; inc < 2; inc++) for (int prim = 0; prim < 2; prim++) {
			DataArray3D<double> out(nc, nreta, npts);
			pGrid->ReduceInterpolate(DataType_State, dREta, dAlpha, dBeta, iPatch, out, DataLocation_None, inc != 0, prim != 0);
			char nm[64]; snprintf(nm, 64, "interp/state_ref%d_prim%d", inc, prim);
			std::vector<size_t> d3(3); d3[0] = nc; d3[1] = nreta; d3[2] = npts;
			w.rec(nm, 0, d3, &out[0][0][0]);
		}
		{
			DataArray3D<double> out(nc, nreta, npts);
			pGrid->ReduceInterpolate(DataType_State, dREta, dAlpha, dBeta, iPatch, out, DataLocation_Node, true, false);
			std::vector<size_t> d3(3); d3[0] = nc; d3[1] = nreta; d3[2] = npts;
			w.rec("interp/state_nodes_only", 0, d3, &out[0][0][0]);
		}
		if (ntracers > 0) {
			DataArray3D<double> out(ntracers, nreta, npts);
			pGrid->ReduceInterpolate(DataType_Tracers, dREta, dAlpha, dBeta, iPatch, out, DataLocation_None, true);
			std::vector<size_t> d3(3); d3[0] = ntracers; d3[1] = nreta; d3[2] = npts;
			w.rec("interp/tracers", 0, d3, &out[0][0][0]);
		}

	} else if (mode == "steps") {
		for (int s = 0; s < nsteps; s++) {
			pTS->Step((s == 0) && (nwarm == 0), false, time, dt);
			time += timeDeltaT;
			if (pPhysics) pPhysics->Perform(time);
			if (((s + 1) % every == 0) || (s == nsteps - 1)) {
				char tag[32]; snprintf(tag, 32, "step%d", s + 1);
				printChecksums(pGrid, tag, &w);
				dumpState(w, pGrid, tag, 0);
				if (scheme.compare(0, 6, "strang") == 0) {
					// the Strang stepper carries instance 1 over to the next step (TimestepSchemeStrang.cpp:639-657)
					dumpState(w, pGrid, std::string(tag) + "_carry", 1);
				}
			}
		}

	} else if (mode == "time") {
		std::vector<double> t;
		for (int s = 0; s < nsteps; s++) {
			auto t0 = std::chrono::steady_clock::now();
			pTS->Step((s == 0) && (nwarm == 0), false, time, dt);
			auto t1 = std::chrono::steady_clock::now();
			time += timeDeltaT;
			t.push_back(std::chrono::duration<double>(t1 - t0).count());
		}
		double tmin = 1e30, tsum = 0;
		for (size_t i = 0; i < t.size(); i++) { tsum += t[i]; if (t[i] < tmin) tmin = t[i]; }
		printf("timing steps=%d mean_s=%.6f min_s=%.6f\n", nsteps, tsum / t.size(), tmin);
		printChecksums(pGrid, "final", &w);
	}

	w.close();

} catch (Exception & e) {
	std::cout << e.ToString() << std::endl;
	TempestDeinitialize();
	return 1;
}
	TempestDeinitialize();
	return 0;
}
