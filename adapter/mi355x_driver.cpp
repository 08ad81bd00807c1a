// adapter/mi355x_driver.cpp -- end-to-end drop-in check (test infrastructure).
//
// Runs the reference's own Model / Grid / BaroclinicWaveJWTest with the adapter plugins of
// TempestMI355X.h installed, and compares against the pure reference inside the same process:
//
//   1. build the model exactly as the reference test driver does (TempestInitialize.h:476-586);
//   2. `--warm` steps with the reference classes (leaves the balanced initial state);
//   3. K steps with the reference classes                      -> state R
//   4. restore, K steps through the adapter (--mode production: TimestepSchemeARS343_MI355X;
//      --mode parity: HorizontalDynamicsMI355X + VerticalDynamicsMI355X under the reference's own
//      TimestepSchemeARS343)                                   -> state G
//   5. print max |G - R| / max |R| per variable; exit status 0 iff below --tol.
//
// The reference test case is pulled in by #include of its own translation unit where it lies.

#define main tmx_unused_jw_main
#include "BaroclinicWaveJWTest.cpp"
#undef main
#define main tmx_unused_schar_main
#include "ScharMountainSphereTest.cpp"
#undef main
// the stock test cases of BASELINE configs 1, 4 and 5, each from its own translation unit
#define main tmx_unused_sw2_main
#include "SWTest2.cpp"
#undef main
#define main tmx_unused_heldsuarez_main
#include "HeldSuarezTest.cpp"
#undef main
#define main tmx_unused_supercell_main
#include "SupercellTest.cpp"
#undef main
#include "test_inputs.h"       // JWWithTracers: the JW state plus closed-form tracer densities (a test input of ours)

#include "TempestMI355X.h"
#include "HeldSuarezPhysics.h"
#include "KesslerPhysics.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// Plugins that forward either to the reference implementation or to the adapter.
class HSwitch : public HorizontalDynamics {
public:
	HorizontalDynamicsFEM m_ref;
	HorizontalDynamicsMI355X m_dev;
	bool m_fDevice;
	HSwitch(Model & model, double dNu) :
		HorizontalDynamics(model),
		m_ref(model, 4, 4, dNu, dNu, dNu, 0.0),
		m_dev(model, 4, 4, dNu, dNu, dNu),
		m_fDevice(false) { }
	virtual void Initialize() { static_cast<HorizontalDynamics &>(m_ref).Initialize(); }
	virtual void StepExplicit(int a, int b, const Time & t, double dt) {
		if (m_fDevice) m_dev.StepExplicit(a, b, t, dt); else static_cast<HorizontalDynamics &>(m_ref).StepExplicit(a, b, t, dt);
	}
	virtual void StepAfterSubCycle(int a, int b, int c, const Time & t, double dt) {
		if (m_fDevice) m_dev.StepAfterSubCycle(a, b, c, t, dt); else static_cast<HorizontalDynamics &>(m_ref).StepAfterSubCycle(a, b, c, t, dt);
	}
};

class VSwitch : public VerticalDynamics {
public:
	VerticalDynamicsFEM m_ref;
	VerticalDynamicsMI355X m_dev;
	bool m_fDevice;
	VSwitch(Model & model, bool fFullyExplicit) :
		VerticalDynamics(model),
		m_ref(model, 4, 1, 0, fFullyExplicit, true, false),
		m_dev(model, 4, 1, fFullyExplicit),
		m_fDevice(false) { }
	virtual void Initialize() { static_cast<VerticalDynamics &>(m_ref).Initialize(); }
	virtual void StepExplicit(int a, int b, const Time & t, double dt) {
		if (m_fDevice) m_dev.StepExplicit(a, b, t, dt); else static_cast<VerticalDynamics &>(m_ref).StepExplicit(a, b, t, dt);
	}
	virtual void StepImplicit(int a, int b, const Time & t, double dt) {
		if (m_fDevice) m_dev.StepImplicit(a, b, t, dt); else static_cast<VerticalDynamics &>(m_ref).StepImplicit(a, b, t, dt);
	}
	virtual void StepImplicitTermsExplicitly(int a, int b, const Time & t, double dt) {
		if (m_fDevice) m_dev.StepImplicitTermsExplicitly(a, b, t, dt); else static_cast<VerticalDynamics &>(m_ref).StepImplicitTermsExplicitly(a, b, t, dt);
	}
	virtual void FilterNegativeTracers(int a) {
		if (m_fDevice) m_dev.FilterNegativeTracers(a); else static_cast<VerticalDynamics &>(m_ref).FilterNegativeTracers(a);
	}
};

class TSwitch : public TimestepScheme {
public:
	TimestepScheme * m_pRef;      // the reference's own scheme
	TimestepSchemeMI355X * m_pDev;   // its device drop-in (production mode)
	bool m_fDevice;
	static bool StrangVariant(const std::string & scheme, TimestepSchemeStrang::ExplicitDiscretization & e) {
		if (scheme == "strang") e = TimestepSchemeStrang::KinnmarkGrayUllrich35;
		else if (scheme == "strang_fe") e = TimestepSchemeStrang::ForwardEuler;
		else if (scheme == "strang_rk4") e = TimestepSchemeStrang::RungeKutta4;
		else if (scheme == "strang_ssp3") e = TimestepSchemeStrang::RungeKuttaSSP3;
		else if (scheme == "strang_ssprk53") e = TimestepSchemeStrang::RungeKuttaSSPRK53;
		else return false;
		return true;
	}
	TSwitch(Model & model, const std::string & scheme, double dOffCentering) : TimestepScheme(model), m_pRef(NULL), m_pDev(NULL), m_fDevice(false) {
		TimestepSchemeStrang::ExplicitDiscretization e;
		if (StrangVariant(scheme, e)) {
			m_pRef = new TimestepSchemeStrang(model, dOffCentering, e);
			m_pDev = new TimestepSchemeStrang_MI355X(model, dOffCentering, e);
		}
		else if (scheme == "ark232") { m_pRef = new TimestepSchemeARK232(model); m_pDev = new TimestepSchemeARK232_MI355X(model); }
		else if (scheme == "ars232") { m_pRef = new TimestepSchemeARS232(model); m_pDev = new TimestepSchemeARS232_MI355X(model); }
		else if (scheme == "ars222") { m_pRef = new TimestepSchemeARS222(model); m_pDev = new TimestepSchemeARS222_MI355X(model); }
		else if (scheme == "ars443") { m_pRef = new TimestepSchemeARS443(model); m_pDev = new TimestepSchemeARS443_MI355X(model); }
		else if (scheme == "ars343") { m_pRef = new TimestepSchemeARS343(model); m_pDev = new TimestepSchemeARS343_MI355X(model); }
		else _EXCEPTIONT("unknown --scheme");
	}
	virtual ~TSwitch() { delete m_pRef; delete m_pDev; }
	virtual int GetComponentDataInstances() const { return m_pRef->GetComponentDataInstances(); }
	virtual int GetTracerDataInstances() const { return m_pRef->GetTracerDataInstances(); }
	virtual void Initialize() { m_pRef->Initialize(); }
	virtual void Step(bool f, bool l, const Time & t, double dt) {
		if (m_fDevice) m_pDev->Step(f, l, t, dt); else m_pRef->Step(f, l, t, dt);
	}
};

typedef std::vector< std::vector<double> > Snapshot;

// instance ix of every patch: node array, redge array, tracer array (possibly empty)
static void save(GridGLL * pGrid, Snapshot & s, int ix = 0) {
	s.clear();
	for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
		GridPatch * p = pGrid->GetActivePatch(n);
		for (int loc = 0; loc < 2; loc++) {
			DataArray4D<double> & d = p->GetDataState(ix, loc ? DataLocation_REdge : DataLocation_Node);
			s.push_back(std::vector<double>(&d[0][0][0][0], &d[0][0][0][0] + d.GetTotalSize()));
		}
		DataArray4D<double> & t = p->GetDataTracers(ix);
		if (t.GetSize(0) > 0) s.push_back(std::vector<double>(&t[0][0][0][0], &t[0][0][0][0] + t.GetTotalSize()));
		else s.push_back(std::vector<double>());
	}
}

static void restore(GridGLL * pGrid, const Snapshot & s, int ix = 0) {
	int q = 0;
	for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
		GridPatch * p = pGrid->GetActivePatch(n);
		for (int loc = 0; loc < 2; loc++) {
			DataArray4D<double> & d = p->GetDataState(ix, loc ? DataLocation_REdge : DataLocation_Node);
			memcpy(&d[0][0][0][0], &s[q][0], s[q].size() * sizeof(double));
			q++;
		}
		DataArray4D<double> & t = p->GetDataTracers(ix);
		if (t.GetSize(0) > 0) memcpy(&t[0][0][0][0], &s[q][0], s[q].size() * sizeof(double));
		q++;
	}
}

int main(int argc, char ** argv) {
	TempestInitialize(&argc, &argv);
	std::string mode = "production", tcase = "jw", scheme = "ars343";
	double offCentering = 0.0;
	int restartAt = -1, restartStatus = 0;
	double msDev = -1.0;
	int ne = 4, levels = 10, nsteps = 3, npatch = 6, nwarm = 1, ntracers = 0, physics = 0, hostsync = 1, modelgo = 0;
	double dt = 200.0, ztop = 30000.0, tol = 1.0e-10, nu = 1.0e15;
	bool fZtopGiven = false;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		const char * v = (i + 1 < argc) ? argv[i+1] : "";
		if (a == "--mode") { mode = v; i++; }
		else if (a == "--ne") { ne = atoi(v); i++; }
		else if (a == "--levels") { levels = atoi(v); i++; }
		else if (a == "--steps") { nsteps = atoi(v); i++; }
		else if (a == "--warm") { nwarm = atoi(v); i++; }
		else if (a == "--patches") { npatch = atoi(v); i++; }
		else if (a == "--dt") { dt = atof(v); i++; }
		else if (a == "--tol") { tol = atof(v); i++; }
		else if (a == "--case") { tcase = v; i++; }
		else if (a == "--restart") { restartAt = atoi(v); i++; }           // production mode: restart image after K steps
		else if (a == "--offcentering") { offCentering = atof(v); i++; }   // TimestepSchemeStrang constructor argument
		else if (a == "--scheme") { scheme = v; i++; }                // parity mode: ars343 (default), ark232, ars232
		else if (a == "--tracers") { ntracers = atoi(v); i++; }
		else if (a == "--hostsync") { hostsync = atoi(v); i++; }    // production mode: host synchronisation interval (steps)
		else if (a == "--modelgo") { modelgo = atoi(v); i++; }      // 1: the step loop as Model::Go runs it (last step flagged, NO manual flush before the host state is read)
		else if (a == "--physics") { physics = atoi(v); i++; }      // 1: Held-Suarez forcing, 2: Kessler microphysics, after every step
		else if (a == "--ztop") { ztop = atof(v); fZtopGiven = true; i++; }
		else if (a == "--nu") { nu = atof(v); i++; }
		else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
	}
	int status = 1;
try {
	AnnounceSetVerbosityLevel(0);
	// stock test cases: --case sw2 (BASELINE config 1: SWTest2, Strang), heldsuarez (config 5: HeldSuarezTest + forcing),
	// supercellreal (config 4: SupercellTest + Kessler, --explicitvertical, nu = 0)
	const bool fSW = (tcase == "sw2"), fHS = (tcase == "heldsuarez"), fSCReal = (tcase == "supercellreal");
	if (fSW) { levels = 1; if (scheme == "ars343") scheme = "strang"; }
	if (fHS) { if (!fZtopGiven) ztop = 35000.0; physics = 1; }
	if (fSCReal) { if (!fZtopGiven) ztop = 20000.0; ntracers = 3; physics = 2; nu = 0.0; }
	EquationSet eqn(fSW ? EquationSet::ShallowWaterEquations : EquationSet::PrimitiveNonhydrostaticEquations);
	if (fSCReal) { eqn.InsertTracer("RhoQv", "RhoQv"); eqn.InsertTracer("RhoQc", "RhoQc"); eqn.InsertTracer("RhoQr", "RhoQr"); }
	else for (int c = 0; c < ntracers; c++) { char nm[16]; snprintf(nm, 16, "RhoQ%d", c); eqn.InsertTracer(nm, nm); }
	UserDataMeta metaUserData;
	if (fSCReal) metaUserData.InsertDataItem2D("PRECT");
	Model model(eqn, metaUserData);
	const int isec = (int)dt, iusec = (int)((dt - isec) * 1.0e6 + 0.5);
	Time timeDeltaT(0, 0, 0, isec, iusec, Time::CalendarNoLeap, Time::TypeDelta);
	model.SetDeltaT(timeDeltaT);
	model.SetEndTime(model.GetStartTime());

	if (fSW && mode != "production") { fprintf(stderr, "--case sw2 runs in production mode\n"); return 2; }
	TSwitch * pT = new TSwitch(model, scheme, offCentering);
	HSwitch * pH = new HSwitch(model, nu);
	// --case supercell: the configuration of test/dcmip2016/SupercellTest.cpp (planet / 120, no rotation, uniform
	// diffusion 1500 / 500 m^2/s, --explicitvertical) without its Fortran physics, driven by the JW state
	const bool fSupercell = (tcase == "supercell");
	VSwitch * pV = fSW ? NULL : new VSwitch(model, fSupercell || fSCReal);
	model.SetTimestepScheme(pT);
	model.SetHorizontalDynamics(pH);
	if (fSW) model.SetVerticalDynamics(new VerticalDynamicsStub(model));      // TempestInitialize.h:362-365
	else model.SetVerticalDynamics(pV);

	GridCSGLL * pGrid = new GridCSGLL(model);
	pGrid->DefineParameters();
	pGrid->SetParameters(levels, (npatch < 6) ? 6 : npatch, ne, 4, 4, 1,
		Grid::VerticalDiscretization_FiniteElement, Grid::VerticalStaggering_Lorenz);
	pGrid->InitializeDataLocal();
	model.SetGrid(pGrid, npatch);
	if (fSW) {
		model.SetTestCase(new ShallowWaterTestCase2(2998.104995, 38.61068277, 0.0));      // SWTest2.cpp defaults
	} else if (fHS) {
		model.SetTestCase(new HeldSuarezTest(30000.0, 25.0, ztop));                         // HeldSuarezTest.cpp:336-338
	} else if (fSCReal) {
		model.GetGrid()->SetReferenceLength(model.GetGrid()->GetReferenceLength() / 120.0); // SupercellTest.cpp:285-286
		model.SetTestCase(new SupercellTest(ztop, 120.0));
	} else if (tcase == "schar") {
		// ScharMountainSphereTest.cpp defaults: reduced planet X = 500, no rotation, Rayleigh layer above 20 km
		model.SetTestCase(new ScharMountainSphereTest(ztop, 500.0, 0.0, 45.0, 0.0, 250.0, 5000.0, 4000.0, 300.0, 20.0, 0.0, 20000.0, 25.0));
		model.GetGrid()->SetReferenceLength(0.5 * M_PI / 30.0 * 500.0);
	} else if (fSupercell) {
		model.SetTestCase(new JWSmallPlanetDiffusion(ztop, BaroclinicWaveJWTest::PerturbationType_Exp, ntracers, 120.0, 1500.0, 500.0));
		model.GetGrid()->SetReferenceLength(0.5 * M_PI / 30.0 * 120.0);
	} else if (ntracers > 0) {
		model.SetTestCase(new JWWithTracers(0.0, ztop, BaroclinicWaveJWTest::PerturbationType_Exp, ntracers));
	} else {
		model.SetTestCase(new BaroclinicWaveJWTest(0.0, ztop, BaroclinicWaveJWTest::PerturbationType_Exp));
	}
	model.Go();     // initialisation only (end time == start time)

	// column physics as WorkflowProcess (Model::Go calls Perform after every step, Model.cpp:470-481)
	WorkflowProcess * pPhysRef = NULL, * pPhysDev = NULL;
	KesslerPhysicsMI355X * pKesDev = NULL;
	if (physics == 1) { pPhysRef = new HeldSuarezPhysics(model, timeDeltaT); pPhysDev = new HeldSuarezPhysicsMI355X(model, timeDeltaT); }
	if (physics == 2) { pPhysRef = new KesslerPhysics(model, timeDeltaT); pKesDev = new KesslerPhysicsMI355X(model, timeDeltaT); pPhysDev = pKesDev; }
	if (pPhysRef) pPhysRef->Initialize(model.GetStartTime());
	if (pPhysDev) pPhysDev->Initialize(model.GetStartTime());       // as Model::Go does for attached processes

	Time time = model.GetStartTime();
	for (int s = 0; s < nwarm; s++) { pT->Step(s == 0, false, time, dt); time += timeDeltaT; if (pPhysRef) pPhysRef->Perform(time); }
	Snapshot start, start1, ref, got;
	save(pGrid, start);
	save(pGrid, start1, 1);       // the Strang stepper carries instance 1 over to the next step
	std::vector< std::vector<double> > prect0, prectRef, prectDev;
	auto savePrect = [&](std::vector< std::vector<double> > & out) {
		out.clear();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			DataArray3D<double> & u = pGrid->GetActivePatch(n)->GetUserData2D();
			out.push_back((u.GetRows() > 0) ? std::vector<double>(&u[0][0][0], &u[0][0][0] + u.GetTotalSize()) : std::vector<double>());
		}
	};
	auto restorePrect = [&](const std::vector< std::vector<double> > & in) {
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			DataArray3D<double> & u = pGrid->GetActivePatch(n)->GetUserData2D();
			if (u.GetRows() > 0) memcpy(&u[0][0][0], &in[n][0], in[n].size() * sizeof(double));
		}
	};
	savePrect(prect0);
	const auto tRef0 = std::chrono::steady_clock::now();
	for (int s = 0; s < nsteps; s++) { pT->Step(false, false, time, dt); if (pPhysRef) pPhysRef->Perform(time); }
	const double msRef = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tRef0).count() / (nsteps > 0 ? nsteps : 1);
	save(pGrid, ref);
	savePrect(prectRef);

	restore(pGrid, start);
	restore(pGrid, start1, 1);
	restorePrect(prect0);
	MI355XEngine & engine = MI355XEngine::For(model);
	if (mode == "production") { pT->m_fDevice = true; engine.SetHostSyncInterval(hostsync); engine.MarkHostModified(); }
	else if (mode == "parity") { pH->m_fDevice = true; pV->m_fDevice = true; }
	else { fprintf(stderr, "bad mode\n"); return 2; }
	if (restartAt >= 0) {
		// --restart K (production mode): K steps on the device, the restart image written by the device into the active-state
		// containers (poisoned first, so that nothing but the image is used), the image read back into a cleared device
		// instance, and the remaining steps on the device again; then the same with the REFERENCE stepper continuing from
		// the image on the host.  Both must end where the uninterrupted reference run ends.
		if (mode != "production" || restartAt > nsteps) { fprintf(stderr, "--restart needs --mode production and K <= steps\n"); return 2; }
		for (int s = 0; s < restartAt; s++) { pT->Step(false, false, time, dt); if (pPhysDev) pPhysDev->Perform(time); }
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			const DataContainer & dc = pGrid->GetActivePatch(n)->GetDataContainerActiveState();
			memset(const_cast<unsigned char *>(dc.GetPointer()), 0xff, dc.GetTotalByteSize());
		}
		engine.WriteActiveStateImage();
		Snapshot image;
		save(pGrid, image);
		// (a) the reference continues from the image
		pT->m_fDevice = false;
		for (int s = restartAt; s < nsteps; s++) { pT->Step(false, false, time, dt); if (pPhysRef) pPhysRef->Perform(time); }
		Snapshot viaHost;
		save(pGrid, viaHost);
		double dh = 0.0;
		for (size_t a = 0; a < ref.size(); a++) {
			const PatchBox & box = pGrid->GetActivePatch((int)(a / 3))->GetPatchBox();
			const int nA = box.GetATotalWidth(), nB = box.GetBTotalWidth();
			const int Lk = (a % 3 == 1) ? levels + 1 : levels;
			for (size_t c = 0; c * nA * nB * Lk < ref[a].size(); c++)
				for (int i = 1; i < nA - 1; i++) for (int j = 1; j < nB - 1; j++) for (int k = 0; k < Lk; k++) {
					if (a % 3 != 2 && ((c == 3) != (a % 3 == 1))) continue;      // prognostic slots
					const size_t o = ((c * nA + i) * nB + j) * Lk + k;
					dh = std::max(dh, fabs(viaHost[a][o] - ref[a][o]));
				}
		}
		printf("restart image written by the device after %d steps, reference stepper continues from it: max abs diff vs the uninterrupted reference run %.3e\n", restartAt, dh);
		if (dh != 0.0) restartStatus = 1;
		// (b) the device continues from the image it reads back
		restore(pGrid, image);
		pT->m_fDevice = true;
		engine.ReadActiveStateImage();
		for (int s = restartAt; s < nsteps; s++) { pT->Step(false, false, time, dt); if (pPhysDev) pPhysDev->Perform(time); }
	} else {
		// one untimed step first: engine creation, geometry upload and the first launches are not part of a step
		Snapshot s0, s1; save(pGrid, s0); save(pGrid, s1, 1);
		pT->Step(false, false, time, dt); if (pPhysDev) pPhysDev->Perform(time);
		engine.Flush();
		if (pKesDev) pKesDev->FlushPrecipitation();      // empties the device accumulator of the untimed step
		restore(pGrid, s0); restore(pGrid, s1, 1); restorePrect(prect0);
		if (mode == "production") engine.MarkHostModified();
		const auto tDev0 = std::chrono::steady_clock::now();
		// --modelgo: Model::Go's own sequence (Model.cpp:420-509): Step(fLastStep on the last one), the WorkflowProcesses, then
		// the output managers read the host arrays -- nothing but the plugins themselves refreshes them
		for (int s = 0; s < nsteps; s++) { pT->Step(false, modelgo && s == nsteps - 1, time, dt); if (pPhysDev) pPhysDev->Perform(time); }
		if (!modelgo) engine.Flush();
		MI355XEngine::Check(tmx_sync(engine.Handle()));
		msDev = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tDev0).count() / (nsteps > 0 ? nsteps : 1);
	}
	// deferred host synchronisation (--hostsync > 1): what a driver does at the reference's host sync points
	// (OutputManager::ManageOutput / FinalOutput, ComputeErrorNorms: Model.cpp:484-509, :695)
	if (!modelgo) {
		engine.Flush();
		if (pKesDev && hostsync > 1) pKesDev->FlushPrecipitation();
	}
	save(pGrid, got);
	savePrect(prectDev);

	// compare prognostic slots on interior nodes
	const int L = levels;
	double dmax[5] = {0, 0, 0, 0, 0}, smax[5] = {0, 0, 0, 0, 0}, tdmax = 0.0, tsmax = 0.0;
	int q = 0;
	for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
		const PatchBox & box = pGrid->GetActivePatch(n)->GetPatchBox();
		const int nA = box.GetATotalWidth(), nB = box.GetBTotalWidth();
		// tracers (third snapshot entry of the patch)
		{
			const size_t per = (size_t)nA * nB * L;
			const std::vector<double> & tg = got[q + 2], & tr = ref[q + 2];
			for (size_t c = 0; c * per < tr.size(); c++)
				for (int i = 1; i < nA - 1; i++) for (int j = 1; j < nB - 1; j++) for (int k = 0; k < L; k++) {
					const size_t o = c * per + ((size_t)i * nB + j) * L + k;
					const double d = fabs(tg[o] - tr[o]), sv = fabs(tr[o]);
					if (d > tdmax) tdmax = d;
					if (sv > tsmax) tsmax = sv;
				}
		}
		for (int loc = 0; loc < 2; loc++, q++) {
			const int Lk = loc ? L + 1 : L;
			for (int c = 0; c < 5; c++) {
				if (fSW && (c >= 3 || loc == 1)) continue;      // shallow water: U, V, H on the one level
				if ((c == 3) != (loc == 1)) continue;
				for (int i = 1; i < nA - 1; i++) for (int j = 1; j < nB - 1; j++) for (int k = 0; k < Lk; k++) {
					const size_t o = (((size_t)c * nA + i) * nB + j) * Lk + k;
					const double d = fabs(got[q][o] - ref[q][o]), sv = fabs(ref[q][o]);
					if (d > dmax[c]) dmax[c] = d;
					if (sv > smax[c]) smax[c] = sv;
				}
			}
		}
		q++;     // skip the tracer entry
	}
	double worst = 0.0;
	printf("adapter %s ne%d L%d patches=%d steps=%d: max rel diff vs reference (U V rhotheta W rho):", mode.c_str(), ne, levels, npatch, nsteps);
	for (int c = 0; c < 5; c++) {
		const double r = (smax[c] > 0) ? dmax[c] / smax[c] : dmax[c];
		printf(" %.3e", r);
		if (r > worst) worst = r;
	}
	if (ntracers > 0) {
		const double r = (tsmax > 0) ? tdmax / tsmax : tdmax;
		printf("  tracers %.3e", r);
		if (r > worst) worst = r;
	}
	if (physics == 2) {
		double dm = 0.0, sm = 0.0;
		for (size_t n = 0; n < prectRef.size(); n++) for (size_t t = 0; t < prectRef[n].size(); t++) {
			dm = std::max(dm, fabs(prectDev[n][t] - prectRef[n][t])); sm = std::max(sm, fabs(prectRef[n][t]));
		}
		const double r = (sm > 0) ? dm / sm : dm;
		printf("  precipitation %.3e", r);
		if (r > worst) worst = r;
	}
	if (pGrid->HasRayleighFriction()) printf("  [Rayleigh friction active]");
	if (physics == 1) printf("  [Held-Suarez physics after every step]");
	if (physics == 2) printf("  [Kessler physics after every step]");
	printf("  scheme=%s case=%s hostsync=%d\n", scheme.c_str(), tcase.c_str(), hostsync);
	status = (worst < tol && restartStatus == 0) ? 0 : 1;
	if (msDev >= 0.0)
		printf("wall time per step incl. host synchronisation: reference %.2f ms, adapter (%s mode, hostsync %d) %.3f ms\n", msRef, mode.c_str(), hostsync, msDev);
	printf("closed-form metric (factors taken from the reference grid, verified bit for bit): %s\n",
		tmx_info(MI355XEngine::For(model).Handle(), TMX_INFO_METRIC_CLOSED_FORM) == 1 ? "active" : "refused, arrays streamed");
	// the output interpolation.  Host arrays and device now hold the same instance 0: the reference's
	// Grid::ReduceInterpolate on the host arrays against ReduceInterpolateMI355X on the device state, both flag sets
	// OutputManagerReference uses plus the primitive conversion.
	if (!fSW) {
		const int nlon = 16, nlat = 8, npts = nlon * nlat;
		DataArray1D<double> dLon(npts), dLat(npts), dAlpha(npts), dBeta(npts);
		DataArray1D<int> iPatch(npts);
		for (int j = 0; j < nlat; j++) for (int i = 0; i < nlon; i++) {
			dLon[j * nlon + i] = (i + 0.5) * 2.0 * M_PI / nlon;
			dLat[j * nlon + i] = -0.5 * M_PI + (j + 0.5) * M_PI / nlat;
		}
		pGrid->ConvertReferenceToPatchCoord(dLon, dLat, dAlpha, dBeta, iPatch);
		engine.MarkHostModified(); engine.EnsureDevice();      // parity mode leaves the device copy of instance 0 behind the host's
		ReduceInterpolateMI355X interp(model);
		double worstI = 0.0;
		for (int v = 0; v < 3; v++) {
			const bool fRef = (v != 1), fPrim = (v == 2);
			const DataArray1D<double> & dREta = pGrid->GetREtaLevels();
			DataArray3D<double> a(5, dREta.GetRows(), npts), b(5, dREta.GetRows(), npts);
			pGrid->ReduceInterpolate(DataType_State, dREta, dAlpha, dBeta, iPatch, a, DataLocation_None, fRef, fPrim);
			interp(DataType_State, dREta, dAlpha, dBeta, iPatch, b, DataLocation_None, fRef, fPrim);
			for (int c = 0; c < 5; c++) {
				double dm = 0.0, sm = 0.0;
				for (int k = 0; k < dREta.GetRows(); k++) for (int i = 0; i < npts; i++) {
					dm = std::max(dm, fabs(a[c][k][i] - b[c][k][i])); sm = std::max(sm, fabs(a[c][k][i]));
				}
				if (sm > 0.0) worstI = std::max(worstI, dm / sm);
			}
		}
		if (ntracers > 0) {
			const DataArray1D<double> & dREta = pGrid->GetREtaLevels();
			DataArray3D<double> a(ntracers, dREta.GetRows(), npts), b(ntracers, dREta.GetRows(), npts);
			pGrid->ReduceInterpolate(DataType_Tracers, dREta, dAlpha, dBeta, iPatch, a, DataLocation_None, true);
			interp(DataType_Tracers, dREta, dAlpha, dBeta, iPatch, b, DataLocation_None, true);
			for (int c = 0; c < ntracers; c++) {
				double dm = 0.0, sm = 0.0;
				for (int k = 0; k < dREta.GetRows(); k++) for (int i = 0; i < npts; i++) {
					dm = std::max(dm, fabs(a[c][k][i] - b[c][k][i])); sm = std::max(sm, fabs(a[c][k][i]));
				}
				if (sm > 0.0) worstI = std::max(worstI, dm / sm);
			}
		}
		printf("output interpolation (Grid::ReduceInterpolate vs device, %d points): max rel diff %.3e\n", npts, worstI);
		if (!(worstI < 1.0e-12)) status = 1;
	}
	printf("%s\n", status == 0 ? "ADAPTER_OK" : "ADAPTER_MISMATCH");
} catch (Exception & e) {
	std::cout << e.ToString() << std::endl;
	status = 3;
}
	TempestDeinitialize();
	return status;
}
