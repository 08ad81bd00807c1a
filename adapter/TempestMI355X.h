// adapter/TempestMI355X.h -- the reference-side binding of the MI355X engine.
//
// Three thin subclasses of Tempest's own plugin interfaces that forward to the C ABI
// (include/tempest_mi355x.h).  Compiles against the reference headers (src/atm, src/base) and
// contains no reference code; this is the file a Tempest maintainer adds.
//
//   TimestepScheme*_MI355X       production mode (ARS343, ARS232, ARS222, ARS443, ARK232, Strang): the whole step runs on
//                                the GPU; the host arrays (instance 0) are refreshed per MI355XEngine's host
//                                synchronisation policy (every step by default).
//   HeldSuarezPhysicsMI355X /    the column physics WorkflowProcesses on the device, sharing that policy.
//   KesslerPhysicsMI355X
//   HorizontalDynamicsMI355X /   parity mode: the reference's own TimestepScheme, Grid::CopyData,
//   VerticalDynamicsMI355X       LinearCombineData and ApplyDSS keep running on the host arrays; every
//                                call uploads the instances it reads and downloads the one it writes.
//
// Interfaces honoured: HorizontalDynamics (src/atm/HorizontalDynamics.h:34-175), VerticalDynamics
// (src/atm/VerticalDynamics.h:29-135), TimestepScheme (src/atm/TimestepScheme.h:32-125).
#ifndef TEMPEST_MI355X_ADAPTER_H
#define TEMPEST_MI355X_ADAPTER_H

#include "Model.h"
#include "GridGLL.h"
#include "GridCSGLL.h"
#include "GridPatchGLL.h"
#include "GridSpacing.h"
#include "CubedSphereTrans.h"
#include "HorizontalDynamics.h"
#include "VerticalDynamics.h"
#include "VerticalDynamicsFEM.h"
#include "LinearColumnOperatorFEM.h"
#include "PolynomialInterp.h"
#include "TimestepScheme.h"
#include "TimestepSchemeStrang.h"
#include "Exception.h"
#include "WorkflowProcess.h"

#include "tempest_mi355x.h"
#include "LinearAlgebra.h"

// the reference's own declaration of dgbsv_ (LinearAlgebra.h: non-const pointers, int result) behind the ABI's probe signature
static void TempestMI355X_dgbsv(const int * n, const int * kl, const int * ku, const int * nrhs, double * ab, const int * ldab, int * ipiv,
	double * b, const int * ldb, int * info) {
	int in = *n, ikl = *kl, iku = *ku, inrhs = *nrhs, ildab = *ldab, ildb = *ldb;
	dgbsv_(&in, &ikl, &iku, &inrhs, ab, &ildab, ipiv, b, &ildb, info);
}
#include "WorkflowProcess.h"

#include <cmath>
#include <map>
#include <vector>

///////////////////////////////////////////////////////////////////////////////

class MI355XEngine {
public:
	static MI355XEngine & For(Model & model) {
		static std::map<Model *, MI355XEngine *> s_map;
		std::map<Model *, MI355XEngine *>::iterator it = s_map.find(&model);
		if (it == s_map.end()) {
			it = s_map.insert(std::make_pair(&model, new MI355XEngine(model))).first;
		}
		return *(it->second);
	}

	void SetFullyExplicit(bool fFullyExplicit) {
		m_fFullyExplicit = fFullyExplicit;
	}

	void SetViscosity(double dNuScalar, double dNuDiv, double dNuVort, int nOrder) {
		m_dNuScalar = dNuScalar; m_dNuDiv = dNuDiv; m_dNuVort = dNuVort; m_nHypervisOrder = nOrder;
	}

	static void Check(int iRet) {
		if (iRet != TMX_OK) {
			_EXCEPTION1("tempest_mi355x: %s", tmx_last_error());
		}
	}

	tmx_engine * Handle() { Initialize(); return m_pEngine; }

	// ---- where the current copy of instance 0 lives (shared by the TimestepScheme and the WorkflowProcess plugins) ----
	//
	// Host synchronisation policy.  Interval 1 (default): instance 0 is uploaded before and downloaded after every device
	// operation on it, so any reference code may read or modify it on the host in between (output managers, host-side
	// WorkflowProcesses, error norms; Model.cpp:477-509, :695) -- at ne30 L30 that is ~23 ms of transfers around a 1.7 ms
	// step.  Interval n > 1: the state stays on the device; the host copy is refreshed every n-th step, on the last step
	// and on Flush(), and uploaded again only after MarkHostModified().
	void SetHostSyncInterval(int nSteps) { m_nHostSyncInterval = (nSteps < 1) ? 1 : nSteps; }
	int GetHostSyncInterval() const { return m_nHostSyncInterval; }

	///	Engine option by name (tmx_set_option, include/tempest_mi355x.h): recorded here, applied before tmx_finalize.  The
	///	engine reads no environment variable on its own; what a run used can be printed with tmx_options_report.
	void SetOption(const std::string & strName, double dValue) {
		if (m_pEngine != NULL) {
			Check(tmx_set_option(m_pEngine, strName.c_str(), dValue));
		} else {
			m_vecOptions.push_back(std::pair<std::string, double>(strName, dValue));
		}
	}

	///	Instances whose content survives from one Step to the next: instance 0, plus instance 1 for the Strang stepper
	///	(its carry-over, TimestepSchemeStrang.cpp:639-657).  They move together between host and device.
	void SetCarriedInstances(int nInstances) { m_nCarried = (nInstances < 1) ? 1 : nInstances; }

	///	The host copy of instance 0 was changed outside the plugins: upload it before the next device operation.
	void MarkHostModified() { m_fHostNewer = true; m_fDeviceNewer = false; }

	///	Make the device copy of instance 0 current (called by every plugin before it touches instance 0 on the device).
	void EnsureDevice() {
		if (m_fHostNewer || (m_nHostSyncInterval == 1)) {
			for (int ix = 0; ix < m_nCarried; ix++) Upload(ix);
			m_fHostNewer = false;
		}
	}

	///	A plugin changed instance 0 on the device.  fStep: a time step was completed (counts towards the interval).
	///	Model::Go runs Step, then the WorkflowProcesses (column physics), then the output managers (Model.cpp:470-509): when
	///	the Step just taken refreshed the host copy (interval reached, last step), a physics plugin that runs after it
	///	refreshes it again, so that output and error norms see the forcing as well.
	void DeviceModified(bool fStep, bool fForceSync) {
		m_fDeviceNewer = true;
		if (fStep) m_nStepsSinceSync++;
		const bool fFlush = (m_nHostSyncInterval == 1) || fForceSync || (m_nStepsSinceSync >= m_nHostSyncInterval)
			|| (!fStep && m_fLastStepFlushed);
		if (fStep) m_fLastStepFlushed = fFlush;
		if (fFlush) {
			Flush();
		}
	}

	///	Called at the end of every Flush() that refreshed the host copy (KesslerPhysicsMI355X: accumulated precipitation).
	void SetFlushHook(void (*pfnHook)(void *), void * pContext) { m_pfnFlushHook = pfnHook; m_pFlushHookContext = pContext; }

	///	Bring the host copy of instance 0 up to date now (output managers, ComputeErrorNorms, end of run).
	void Flush() {
		if (m_fDeviceNewer) {
			Initialize();
			Check(tmx_sync(m_pEngine));
			for (int ix = 0; ix < m_nCarried; ix++) Download(ix);
			m_fDeviceNewer = false;
			if (m_pfnFlushHook != NULL) {
				m_pfnFlushHook(m_pFlushHookContext);
			}
		}
		m_nStepsSinceSync = 0;
	}

	// GridPatch::GetDataState arrays -> device
	void Upload(int ixData) {
		Initialize();
		GridGLL * pGrid = Grid();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			Check(tmx_upload_state(m_pEngine, pPatch->GetPatchIndex(), ixData,
				&(pPatch->GetDataState(ixData, DataLocation_Node)[0][0][0][0]),
				&(pPatch->GetDataState(ixData, DataLocation_REdge)[0][0][0][0])));
			if (m_model.GetEquationSet().GetTracers() > 0) {
				Check(tmx_upload_tracers(m_pEngine, pPatch->GetPatchIndex(), ixData,
					&(pPatch->GetDataTracers(ixData)[0][0][0][0])));
			}
		}
	}

	// GridPatch::GetReferenceState / GetReferenceTracers -> device (uniform diffusion, output without the reference state)
	void SetReferenceState() {
		Initialize();
		GridGLL * pGrid = Grid();
		const bool fTracers = m_model.GetEquationSet().GetTracers() > 0;
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			Check(tmx_set_patch_reference_state(m_pEngine, pPatch->GetPatchIndex(),
				&(pPatch->GetReferenceState(DataLocation_Node)[0][0][0][0]),
				&(pPatch->GetReferenceState(DataLocation_REdge)[0][0][0][0]),
				fTracers ? &(pPatch->GetReferenceTracers()[0][0][0][0]) : NULL));
		}
	}

	// device -> GridPatch::GetDataState arrays (interior nodes; halo untouched)
	void Download(int ixData) {
		Initialize();
		GridGLL * pGrid = Grid();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			Check(tmx_download_state(m_pEngine, pPatch->GetPatchIndex(), ixData,
				&(pPatch->GetDataState(ixData, DataLocation_Node)[0][0][0][0]),
				&(pPatch->GetDataState(ixData, DataLocation_REdge)[0][0][0][0])));
			if (m_model.GetEquationSet().GetTracers() > 0) {
				Check(tmx_download_tracers(m_pEngine, pPatch->GetPatchIndex(), ixData,
					&(pPatch->GetDataTracers(ixData)[0][0][0][0])));
			}
		}
	}

	///	Restart write, device side: instance 0 of every active patch straight into the bytes of
	///	GridPatch::GetDataContainerActiveState(), which OutputManagerComposite::Output then writes unchanged
	///	(OutputManagerComposite.cpp:331-349).  One contiguous device-to-host copy per patch, no host transposition.
	void WriteActiveStateImage() {
		Initialize();
		Check(tmx_sync(m_pEngine));
		GridGLL * pGrid = Grid();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			const DataContainer & dc = pPatch->GetDataContainerActiveState();
			Check(tmx_pack_active_state(m_pEngine, pPatch->GetPatchIndex(), 0,
				const_cast<unsigned char *>(dc.GetPointer()), dc.GetTotalByteSize()));
		}
		m_fDeviceNewer = false; m_fHostNewer = false; m_nStepsSinceSync = 0;
	}

	///	Restart read: the active-state containers (as a restart file filled them) -> instance 0 on the device.
	void ReadActiveStateImage() {
		Initialize();
		GridGLL * pGrid = Grid();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			const DataContainer & dc = pPatch->GetDataContainerActiveState();
			Check(tmx_unpack_active_state(m_pEngine, pPatch->GetPatchIndex(), 0, dc.GetPointer(), dc.GetTotalByteSize()));
		}
		m_fDeviceNewer = false; m_fHostNewer = false;
	}

	~MI355XEngine() {
		if (m_pEngine != NULL) {
#ifdef TEMPEST_MPIOMP
			// peer-to-peer transport: the neighbour ranks write into this rank's block until they have finished
			int fFinalized = 0;
			MPI_Finalized(&fFinalized);
			if (m_fPeerToPeer && !fFinalized) {
				tmx_sync(m_pEngine);
				MPI_Barrier(MPI_COMM_WORLD);
			}
#endif
			tmx_destroy(m_pEngine);
		}
	}

private:
	MI355XEngine(Model & model) :
		m_model(model), m_pEngine(NULL), m_fPeerToPeer(false),
		m_dNuScalar(1.0e15), m_dNuDiv(1.0e15), m_dNuVort(1.0e15), m_nHypervisOrder(4), m_fFullyExplicit(false),
		m_nHostSyncInterval(1), m_nStepsSinceSync(0), m_fLastStepFlushed(false), m_pfnFlushHook(NULL), m_pFlushHookContext(NULL),
		m_fHostNewer(true), m_fDeviceNewer(false), m_nCarried(1)
	{ }

	GridGLL * Grid() {
		GridGLL * pGrid = dynamic_cast<GridGLL *>(m_model.GetGrid());
		if (pGrid == NULL) {
			_EXCEPTIONT("tempest_mi355x requires a GridGLL");
		}
		return pGrid;
	}

	// Called on first use, i.e. after Model::Go has evaluated the geometric terms.
	void Initialize() {
		if (m_pEngine != NULL) {
			return;
		}
		GridGLL * pGrid = Grid();
		if (pGrid->GetActivePatchCount() == 0) {
			_EXCEPTIONT("tempest_mi355x: no active patch on this rank");
		}
		const PhysicalConstants & phys = m_model.GetPhysicalConstants();
		const int nOrder = pGrid->GetHorizontalOrder();
		const int nRElements = pGrid->GetRElements();
		const int nPatches = pGrid->GetPatchCount();

		int nRank = 0;
		int nRanks = 1;
#ifdef TEMPEST_MPIOMP
		MPI_Comm_rank(MPI_COMM_WORLD, &nRank);
		MPI_Comm_size(MPI_COMM_WORLD, &nRanks);
#endif

		tmx_config cfg;
		cfg.abi_version = TMX_ABI_VERSION;
		cfg.horizontal_order = nOrder;
		cfg.vertical_order = pGrid->GetVerticalOrder();
		cfg.levels = nRElements;
		cfg.n_patches = nPatches;
		cfg.n_instances = m_model.GetComponentDataInstances();
		cfg.n_tracers = m_model.GetEquationSet().GetTracers();
		cfg.hypervis_order = m_nHypervisOrder;
		cfg.element_delta_a =
			dynamic_cast<GridPatchGLL *>(pGrid->GetActivePatch(0))->GetElementDeltaA();
		cfg.reference_length = pGrid->GetReferenceLength();
		cfg.grav = phys.GetG();
		cfg.Rd = phys.GetR();
		cfg.cp = phys.GetCp();
		cfg.p0 = phys.GetP0();
		cfg.nu_scalar = m_dNuScalar;
		cfg.nu_div = m_dNuDiv;
		cfg.nu_vort = m_dNuVort;
		cfg.rank = nRank;
		cfg.n_ranks = nRanks;
		cfg.device = -1;
#ifdef TEMPEST_MPIOMP
		if (nRanks > 1) {
			// one rank = one GPU: bind by the rank's index among the ranks of its node (a plain `mpiexec -np N` would
			// otherwise put every rank on device 0, and tmx_comm_init rejects two ranks on one device)
			MPI_Comm commNode;
			MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, nRank, MPI_INFO_NULL, &commNode);
			int nLocalRank = 0;
			MPI_Comm_rank(commNode, &nLocalRank);
			MPI_Comm_free(&commNode);
			const int nDevices = tmx_device_count();
			if (nDevices > 0) {
				cfg.device = nLocalRank % nDevices;
			}
		}
#endif
		cfg.equation_set =
			(m_model.GetEquationSet().GetType() == EquationSet::ShallowWaterEquations)
				? TMX_EQN_SHALLOW_WATER : TMX_EQN_PRIMITIVE_NONHYDROSTATIC;
		// --explicitvertical: either announced by VerticalDynamicsMI355X (parity mode) or read from the reference's own
		// VerticalDynamicsFEM when only the TimestepScheme is replaced (production mode)
		bool fFullyExplicit = m_fFullyExplicit;
		VerticalDynamicsFEM * pVFEM = dynamic_cast<VerticalDynamicsFEM *>(m_model.GetVerticalDynamics());
		if (pVFEM != NULL) {
			fFullyExplicit = pVFEM->IsFullyExplicit();
		}
		cfg.fully_explicit = fFullyExplicit ? 1 : 0;
		cfg.uniform_diffusion = pGrid->HasUniformDiffusion() ? 1 : 0;
		cfg.uniform_diffusion_scalar = pGrid->GetScalarUniformDiffusionCoeff();
		cfg.uniform_diffusion_vector = pGrid->GetVectorUniformDiffusionCoeff();
		cfg.ztop = pGrid->GetZtop();
		Check(tmx_create(&cfg, &m_pEngine));

		// 1-D spectral-element matrices and the ten column operators
		if (cfg.equation_set == TMX_EQN_SHALLOW_WATER) {
			Check(tmx_set_operators(m_pEngine,
				&(pGrid->GetDxBasis1D()[0][0]), &(pGrid->GetStiffness1D()[0][0]),
				NULL, NULL, NULL, NULL, NULL));
		} else {
			const LinearColumnOperator * ops[TMX_OP_COUNT] = {
				&(pGrid->GetOpInterpNodeToREdge()), &(pGrid->GetOpInterpREdgeToNode()),
				&(pGrid->GetOpDiffNodeToNode()), &(pGrid->GetOpDiffNodeToREdge()),
				&(pGrid->GetOpDiffREdgeToNode()), &(pGrid->GetOpDiffREdgeToREdge()),
				&(pGrid->GetOpDiffDiffNodeToNode()), &(pGrid->GetOpDiffDiffREdgeToREdge()),
				&(pGrid->GetOpPenaltyNodeToNode().GetLeftOp()), &(pGrid->GetOpPenaltyNodeToNode().GetRightOp())};
			const double * coeff[TMX_OP_COUNT];
			const int * ixb[TMX_OP_COUNT];
			const int * ixe[TMX_OP_COUNT];
			int nOut[TMX_OP_COUNT];
			int nIn[TMX_OP_COUNT];
			std::vector< std::vector<int> > vecB(TMX_OP_COUNT), vecE(TMX_OP_COUNT);
			for (int o = 0; o < TMX_OP_COUNT; o++) {
				const DataArray2D<double> & dC = ops[o]->GetCoeffs();
				nOut[o] = dC.GetRows();
				nIn[o] = dC.GetColumns();
				coeff[o] = &(dC[0][0]);
				for (int k = 0; k < nOut[o]; k++) {
					vecB[o].push_back(ops[o]->GetIxBegin()[k]);
					vecE[o].push_back(ops[o]->GetIxEnd()[k]);
				}
				ixb[o] = &(vecB[o][0]);
				ixe[o] = &(vecE[o][0]);
			}
			// dStiffness1D is stored (m,i) with S[m][i] used as dStiffness1D(i,s): same row-major layout
			Check(tmx_set_operators(m_pEngine,
				&(pGrid->GetDxBasis1D()[0][0]), &(pGrid->GetStiffness1D()[0][0]),
				coeff, ixb, ixe, nOut, nIn));
		}

		// Patches of the whole grid and their halo rings
		const int nGlobalNodes = nOrder * pGrid->GetABaseResolution();
		GridSpacingGaussLobattoRepeated glspacing(
			0.5 * M_PI / pGrid->GetABaseResolution(), -0.25 * M_PI, nOrder);

		for (int n = 0; n < nPatches; n++) {
			const PatchBox & box = pGrid->GetPatchBox(n);
			int ixNeighbor[8];
			const int iProbeA[8] = {
				box.GetAGlobalInteriorEnd(), box.GetAGlobalInteriorBegin(),
				box.GetAGlobalInteriorBegin()-1, box.GetAGlobalInteriorBegin(),
				box.GetAGlobalInteriorEnd(), box.GetAGlobalInteriorBegin()-1,
				box.GetAGlobalInteriorBegin()-1, box.GetAGlobalInteriorEnd()};
			const int iProbeB[8] = {
				box.GetBGlobalInteriorBegin(), box.GetBGlobalInteriorEnd(),
				box.GetBGlobalInteriorBegin(), box.GetBGlobalInteriorBegin()-1,
				box.GetBGlobalInteriorEnd(), box.GetBGlobalInteriorEnd(),
				box.GetBGlobalInteriorBegin()-1, box.GetBGlobalInteriorBegin()-1};
			for (int d = 0; d < 8; d++) {
				int ixA, ixB; bool f1, f2, f3;
				CubedSphereTrans::RelativeCoord(
					nGlobalNodes, box.GetPanel(), iProbeA[d], iProbeB[d],
					ixNeighbor[d], ixA, ixB, f1, f2, f3);
				if ((ixA == (-1)) && (ixB == (-1))) {
					ixNeighbor[d] = -1;
				}
			}
			Check(tmx_define_patch(m_pEngine, n, box.GetPanel(),
				box.GetAInteriorWidth() / nOrder, box.GetBInteriorWidth() / nOrder,
				pGrid->GetPatchProcessor(n), ixNeighbor));
			{
				// GridPatchGLL::InitializeCoordinateData (GridPatchGLL.cpp:67-75) for a patch this rank may not hold: the
				// difference of the first two element-edge coordinates of the patch, from the same GridSpacing
				const int iA0 = box.GetAGlobalInteriorBegin(), iB0 = box.GetBGlobalInteriorBegin();
				const double dDeltaA = glspacing.GetEdge(iA0 + nOrder) - glspacing.GetEdge(iA0);
				const double dDeltaB = glspacing.GetEdge(iB0 + nOrder) - glspacing.GetEdge(iB0);
				Check(tmx_set_patch_element_spacing(m_pEngine, n, dDeltaA, dDeltaB));
			}
		}

		for (int n = 0; n < nPatches; n++) {
			const PatchBox & box = pGrid->GetPatchBox(n);
			const int nA = box.GetATotalWidth();
			const int nB = box.GetBTotalWidth();
			std::vector<int> hi, hj, sp, si, sj, spanel;
			std::vector<double> trans;
			for (int i = 0; i < nA; i++) {
			for (int j = 0; j < nB; j++) {
				if ((i != 0) && (j != 0) && (i != nA-1) && (j != nB-1)) {
					continue;
				}
				const int gA = box.GetAGlobalBegin() + i;
				const int gB = box.GetBGlobalBegin() + j;
				int ixPanel, ixA, ixB; bool f1, f2, f3;
				CubedSphereTrans::RelativeCoord(
					nGlobalNodes, box.GetPanel(), gA, gB, ixPanel, ixA, ixB, f1, f2, f3);
				hi.push_back(i);
				hj.push_back(j);
				double dT[4] = {1.0, 0.0, 0.0, 1.0};
				int ixSrcPatch = -1;
				if (!((ixA == (-1)) && (ixB == (-1)))) {
					for (int m = 0; m < nPatches; m++) {
						if (pGrid->GetPatchBox(m).ContainsGlobalPoint(ixPanel, ixA, ixB)) {
							ixSrcPatch = m;
							break;
						}
					}
				}
				if (ixSrcPatch < 0) {
					sp.push_back(-1); si.push_back(0); sj.push_back(0); spanel.push_back(-1);
				} else {
					const PatchBox & boxSrc = pGrid->GetPatchBox(ixSrcPatch);
					sp.push_back(ixSrcPatch);
					si.push_back(ixA - boxSrc.GetAGlobalBegin());
					sj.push_back(ixB - boxSrc.GetBGlobalBegin());
					spanel.push_back(ixPanel);
					if (ixPanel != box.GetPanel()) {
						// the call GridPatchCSGLL::TransformHaloVelocities makes, probed with unit vectors
						const double dX = tan(glspacing.GetNode(gA));
						const double dY = tan(glspacing.GetNode(gB));
						double dA0 = 1.0, dB0 = 0.0, dA1 = 0.0, dB1 = 1.0;
						CubedSphereTrans::CoVecPanelTrans(ixPanel, box.GetPanel(), dA0, dB0, dX, dY);
						CubedSphereTrans::CoVecPanelTrans(ixPanel, box.GetPanel(), dA1, dB1, dX, dY);
						dT[0] = dA0; dT[1] = dA1; dT[2] = dB0; dT[3] = dB1;
					}
				}
				trans.insert(trans.end(), dT, dT + 4);
			}
			}
			Check(tmx_set_patch_halo(m_pEngine, n, (int)hi.size(),
				&hi[0], &hj[0], &sp[0], &si[0], &sj[0], &spanel[0], &trans[0]));
		}

		// Geometry of the patches this rank owns
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			tmx_patch_geometry g;
			g.jacobian2d = &(pPatch->GetJacobian2D()[0][0]);
			g.contra_metric_2d_a = &(pPatch->GetContraMetric2DA()[0][0][0]);
			g.contra_metric_2d_b = &(pPatch->GetContraMetric2DB()[0][0][0]);
			g.jacobian = &(pPatch->GetJacobian()[0][0][0]);
			g.jacobian_redge = &(pPatch->GetJacobianREdge()[0][0][0]);
			g.contra_metric_a = &(pPatch->GetContraMetricA()[0][0][0][0]);
			g.contra_metric_b = &(pPatch->GetContraMetricB()[0][0][0][0]);
			g.contra_metric_xi = &(pPatch->GetContraMetricXi()[0][0][0][0]);
			g.contra_metric_a_redge = &(pPatch->GetContraMetricAREdge()[0][0][0][0]);
			g.contra_metric_b_redge = &(pPatch->GetContraMetricBREdge()[0][0][0][0]);
			g.contra_metric_xi_redge = &(pPatch->GetContraMetricXiREdge()[0][0][0][0]);
			g.deriv_r_node = &(pPatch->GetDerivRNode()[0][0][0][0]);
			g.deriv_r_redge = &(pPatch->GetDerivRREdge()[0][0][0][0]);
			g.coriolis_f = &(pPatch->GetCoriolisF()[0][0]);
			g.topography = &(pPatch->GetTopography()[0][0]);
			g.element_area_node = &(pPatch->GetElementAreaNode()[0][0][0]);
			Check(tmx_set_patch_geometry(m_pEngine, pPatch->GetPatchIndex(), &g));

			// Rayleigh friction (HorizontalDynamicsFEM::ApplyRayleighFriction runs at the end of StepAfterSubCycle)
			if (pGrid->HasRayleighFriction() &&
			    m_model.GetEquationSet().GetType() != EquationSet::ShallowWaterEquations) {
				Check(tmx_set_patch_rayleigh(m_pEngine, pPatch->GetPatchIndex(),
					&(pPatch->GetRayleighStrength(DataLocation_Node)[0][0][0]),
					&(pPatch->GetRayleighStrength(DataLocation_REdge)[0][0][0]),
					&(pPatch->GetReferenceState(DataLocation_Node)[0][0][0][0]),
					&(pPatch->GetReferenceState(DataLocation_REdge)[0][0][0][0])));
			}

			// Factors of the closed-form 3-D metric (GridPatchCSGLL.cpp:370-568; m_dXNode = tan(m_dANode),
			// :212-219).  The engine verifies them against the arrays above bit for bit before using them.
			if (m_model.GetEquationSet().GetType() != EquationSet::ShallowWaterEquations) {
				const PatchBox & box = pPatch->GetPatchBox();
				std::vector<double> dXNode(box.GetATotalWidth()), dYNode(box.GetBTotalWidth());
				for (int i = 0; i < box.GetATotalWidth(); i++) dXNode[i] = tan(pPatch->GetANode(i));
				for (int j = 0; j < box.GetBTotalWidth(); j++) dYNode[j] = tan(pPatch->GetBNode(j));
				Check(tmx_set_patch_metric_factors(m_pEngine, pPatch->GetPatchIndex(), &dXNode[0], &dYNode[0],
					&(pPatch->GetTopographyDeriv()[0][0][0]), phys.GetEarthRadius(),
					&(pGrid->GetREtaLevels()[0]), &(pGrid->GetREtaInterfaces()[0])));
			}
		}
		// Which band LU does THIS build's LAPACK compute -- update a - l u as one rounding or two?  The engine holds both; asked here with the
		// very dgbsv_ that LAPACK::DGBSV calls (LinearAlgebra.h), so that the drop-in equals the CPU path of the build it is dropped into.
		// An explicit option "lu_fma" (SetOption / --tmx_option) wins.
		{
			bool fExplicit = false;
			for (size_t o = 0; o < m_vecOptions.size(); o++) if (m_vecOptions[o].first == "lu_fma") fExplicit = true;
			if (!fExplicit) {
				const int iFlavour = tmx_lu_flavour_from_dgbsv(m_pEngine, &TempestMI355X_dgbsv);
				if (iFlavour < 0) Announce("WARNING: %s", tmx_last_error());
				else Announce("MI355X engine: band LU updates %s (this build's dgbsv_)", iFlavour ? "fused" : "as multiply + subtract");
			}
		}
		for (size_t o = 0; o < m_vecOptions.size(); o++) {
			Check(tmx_set_option(m_pEngine, m_vecOptions[o].first.c_str(), m_vecOptions[o].second));
		}
		Check(tmx_finalize(m_pEngine));

		// reference state the uniform diffusion is taken relative to (Grid::HasUniformDiffusion)
		if (pGrid->HasUniformDiffusion()) {
			for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
				GridPatch * pPatch = pGrid->GetActivePatch(n);
				Check(tmx_set_patch_reference_state(m_pEngine, pPatch->GetPatchIndex(),
					&(pPatch->GetReferenceState(DataLocation_Node)[0][0][0][0]),
					&(pPatch->GetReferenceState(DataLocation_REdge)[0][0][0][0]),
					(cfg.n_tracers > 0) ? &(pPatch->GetReferenceTracers()[0][0][0][0]) : NULL));
			}
		}

#ifdef TEMPEST_MPIOMP
		const char * szHalo = getenv("TMX_HALO");
		if ((nRanks > 1) && (szHalo != NULL) && (std::string(szHalo) == "p2p")) {
			// peer-to-peer transport: every rank's blob to every rank, then map the neighbours (the ranks may share a device)
			const int nBlob = tmx_halo_p2p_blob_bytes(m_pEngine);
			std::vector<unsigned char> vecMine(nBlob), vecAll((size_t)nBlob * nRanks);
			// Every step of the set-up is agreed on by all ranks (a rank that throws on its own leaves the others in a
			// collective): export, gather, connect, each followed by an all-reduce of the status.  If any rank cannot map a
			// neighbour (another node, no peer access) ALL ranks fall back to RCCL send / recv.
			int iOk = (tmx_halo_p2p_export(m_pEngine, &(vecMine[0])) == TMX_OK) ? 1 : 0, iAll = 0;
			std::string strWhy = iOk ? "" : tmx_last_error();
			MPI_Allreduce(&iOk, &iAll, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
			// fConnected: THIS rank completed tmx_halo_p2p_connect (only then may it be switched back below: a rank whose export
			// succeeded while another rank's failed never connected, and tmx_halo_p2p_enable would refuse it on that rank alone)
			bool fConnected = false;
			if (iAll) {
				MPI_Allgather(&(vecMine[0]), nBlob, MPI_BYTE, &(vecAll[0]), nBlob, MPI_BYTE, MPI_COMM_WORLD);
				iOk = (tmx_halo_p2p_connect(m_pEngine, &(vecAll[0])) == TMX_OK) ? 1 : 0;
				fConnected = (iOk != 0);
				if (!iOk) strWhy = tmx_last_error();
				MPI_Allreduce(&iOk, &iAll, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
			}
			MPI_Barrier(MPI_COMM_WORLD);
			if (iAll) {
				m_fPeerToPeer = true;
			} else {
				if (!strWhy.empty()) {
					fprintf(stderr, "tempest_mi355x: rank %d: peer-to-peer halo not available (%s); all ranks use RCCL send/recv\n", nRank, strWhy.c_str());
				}
				InitRcclComm(nRank);
				if (fConnected) {
					Check(tmx_halo_p2p_enable(m_pEngine, 0));      // this rank did connect: switch it back
				}
			}

		} else if (nRanks > 1) {
			InitRcclComm(nRank);
		}
#endif
	}

#ifdef TEMPEST_MPIOMP
	///	RCCL communicator over MPI_COMM_WORLD (collective: every rank calls it; an error here is fatal on all ranks alike).
	void InitRcclComm(int nRank) {
		unsigned char id[128];
		int iOk = 1, iAll = 0;
		if (nRank == 0) {
			iOk = (tmx_comm_unique_id(id) == TMX_OK) ? 1 : 0;
		}
		MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
		if (iOk) {
			iOk = (tmx_comm_init(m_pEngine, id) == TMX_OK) ? 1 : 0;
		}
		std::string strWhy = iOk ? "" : tmx_last_error();
		MPI_Allreduce(&iOk, &iAll, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
		if (!iAll) {
			_EXCEPTION1("tempest_mi355x: RCCL communicator could not be created on every rank (%s)",
				strWhy.empty() ? "another rank failed" : strWhy.c_str());
		}
	}
#endif

private:
	Model & m_model;
	tmx_engine * m_pEngine;
	bool m_fPeerToPeer;
	std::vector< std::pair<std::string, double> > m_vecOptions;
	double m_dNuScalar, m_dNuDiv, m_dNuVort;
	int m_nHypervisOrder;
	bool m_fFullyExplicit;
	int m_nHostSyncInterval;
	int m_nStepsSinceSync;
	bool m_fLastStepFlushed;
	void (*m_pfnFlushHook)(void *);
	void * m_pFlushHookContext;
	bool m_fHostNewer;
	bool m_fDeviceNewer;
	int m_nCarried;
};

///////////////////////////////////////////////////////////////////////////////

///	<summary>
///		Drop-in for HorizontalDynamicsFEM (nonhydrostatic primitive equations).
///	</summary>
class HorizontalDynamicsMI355X : public HorizontalDynamics {
public:
	HorizontalDynamicsMI355X(
		Model & model, int nHorizontalOrder, int nHyperviscosityOrder,
		double dNuScalar, double dNuDiv, double dNuVort
	) : HorizontalDynamics(model), m_nHyperviscosityOrder(nHyperviscosityOrder) {
		MI355XEngine::For(model).SetViscosity(dNuScalar, dNuDiv, dNuVort, nHyperviscosityOrder);
	}

	virtual void StepExplicit(int iDataInitial, int iDataUpdate, const Time & time, double dDeltaT) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataInitial);
		eng.Upload(iDataUpdate);
		MI355XEngine::Check(tmx_h_step_explicit(eng.Handle(), iDataInitial, iDataUpdate, dDeltaT));
		eng.Download(iDataInitial);   // derived slots written by the reference's prologue
		eng.Download(iDataUpdate);
	}

	virtual void StepAfterSubCycle(
		int iDataInitial, int iDataUpdate, int iDataWorking, const Time & time, double dDeltaT
	) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataInitial);
		MI355XEngine::Check(tmx_h_step_after_subcycle(
			eng.Handle(), iDataInitial, iDataUpdate, iDataWorking, dDeltaT));
		eng.Download(iDataUpdate);
		eng.Download(iDataWorking);
	}

	// HorizontalDynamicsFEM.cpp:2574-2578 (may be asked before the grid exists: no engine call)
	virtual int GetSubStepAfterSubCycleCount() {
		return m_nHyperviscosityOrder / 2;
	}

	virtual int SubStepAfterSubCycle(int iDataInitial, int iDataUpdate, int iDataWorking, const Time & time, double dDeltaT, int iSubStep) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataInitial);
		if (iSubStep == 1) {
			eng.Upload(iDataWorking);
		}
		int iResult = -1;
		MI355XEngine::Check(tmx_h_substep_after_subcycle(eng.Handle(), iDataInitial, iDataUpdate, iDataWorking, dDeltaT, iSubStep, &iResult));
		eng.Download(iResult);
		return iResult;
	}

private:
	int m_nHyperviscosityOrder;
};

///	<summary>
///		Drop-in for VerticalDynamicsFEM (implicit mode).
///	</summary>
class VerticalDynamicsMI355X : public VerticalDynamics {
public:
	VerticalDynamicsMI355X(Model & model, int nHorizontalOrder, int nVerticalOrder, bool fFullyExplicit = false) :
		VerticalDynamics(model)
	{
		MI355XEngine::For(model).SetFullyExplicit(fFullyExplicit);
	}

	virtual void StepExplicit(int iDataInitial, int iDataUpdate, const Time & time, double dDeltaT) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataInitial);
		eng.Upload(iDataUpdate);
		MI355XEngine::Check(tmx_v_step_explicit(eng.Handle(), iDataInitial, iDataUpdate, dDeltaT));
		eng.Download(iDataUpdate);
	}

	virtual void StepImplicit(int iDataInitial, int iDataUpdate, const Time & time, double dDeltaT) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataInitial);
		if (iDataUpdate != iDataInitial) {
			eng.Upload(iDataUpdate);
		}
		MI355XEngine::Check(tmx_v_step_implicit(eng.Handle(), iDataInitial, iDataUpdate, dDeltaT));
		MI355XEngine::Check(tmx_sync(eng.Handle()));
		eng.Download(iDataUpdate);
	}

	// ARK232 only (TimestepSchemeARK232.cpp:167-225)
	virtual void StepImplicitTermsExplicitly(int iDataInitial, int iDataUpdate, const Time & time, double dDeltaT) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataInitial);
		if (iDataUpdate != iDataInitial) {
			eng.Upload(iDataUpdate);
		}
		MI355XEngine::Check(tmx_v_step_implicit_terms_explicitly(eng.Handle(), iDataInitial, iDataUpdate, dDeltaT));
		eng.Download(iDataUpdate);
	}

	// TimestepSchemeStrang.cpp:242, :480
	virtual void FilterNegativeTracers(int iDataUpdate) {
		if (m_model.GetEquationSet().GetTracers() == 0) {
			return;
		}
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.Upload(iDataUpdate);
		MI355XEngine::Check(tmx_v_filter_negative_tracers(eng.Handle(), iDataUpdate));
		eng.Download(iDataUpdate);
	}
};

///	<summary>
///		Drop-in for the reference's IMEX / split steppers: the whole TimestepScheme::Step on the device.  One class
///		per scheme id of the C ABI; the instance counts are the ones of the scheme it stands in for
///		(tmx_scheme_instances), because patch allocation uses them (GridPatch.cpp:334-374).
///	</summary>
class TimestepSchemeMI355X : public TimestepScheme {
public:
	TimestepSchemeMI355X(Model & model, int iScheme) :
		TimestepScheme(model), m_iScheme(iScheme)
	{
		if (tmx_scheme_instances(iScheme) <= 0) {
			_EXCEPTIONT("TimestepSchemeMI355X: unknown scheme id");
		}
	}

	virtual int GetComponentDataInstances() const { return tmx_scheme_instances(m_iScheme); }
	virtual int GetTracerDataInstances() const { return tmx_scheme_instances(m_iScheme); }

	///	Host synchronisation policy and hooks: see MI355XEngine.
	void SetHostSyncInterval(int nSteps) { MI355XEngine::For(m_model).SetHostSyncInterval(nSteps); }
	void MarkHostModified() { MI355XEngine::For(m_model).MarkHostModified(); }
	void Flush() { MI355XEngine::For(m_model).Flush(); }

	virtual void Step(bool fFirstStep, bool fLastStep, const Time & time, double dDeltaT) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		eng.SetCarriedInstances((m_iScheme >= TMX_SCHEME_STRANG) ? 2 : 1);      // every Strang variant carries instance 1
		eng.EnsureDevice();
		MI355XEngine::Check(tmx_step(eng.Handle(), m_iScheme, fFirstStep ? 1 : 0, fLastStep ? 1 : 0, dDeltaT));
		eng.DeviceModified(true, fLastStep);
	}

private:
	int m_iScheme;
};

#define TMX_DEFINE_SCHEME(Name, Id) \
	class Name : public TimestepSchemeMI355X { public: Name(Model & model) : TimestepSchemeMI355X(model, Id) { } }
TMX_DEFINE_SCHEME(TimestepSchemeARS343_MI355X, TMX_SCHEME_ARS343);     // src/atm/TimestepSchemeARS343.cpp
TMX_DEFINE_SCHEME(TimestepSchemeARS232_MI355X, TMX_SCHEME_ARS232);     // TimestepSchemeARS232.cpp
TMX_DEFINE_SCHEME(TimestepSchemeARS222_MI355X, TMX_SCHEME_ARS222);     // TimestepSchemeARS222.cpp
TMX_DEFINE_SCHEME(TimestepSchemeARS443_MI355X, TMX_SCHEME_ARS443);     // TimestepSchemeARS443.cpp
TMX_DEFINE_SCHEME(TimestepSchemeARK232_MI355X, TMX_SCHEME_ARK232);     // TimestepSchemeARK232.cpp
#undef TMX_DEFINE_SCHEME

///	<summary>
///		TimestepSchemeStrang with the constructor of the reference class (TimestepSchemeStrang.h:52-56): off-centering and
///		any of its explicit discretisations.
///	</summary>
class TimestepSchemeStrang_MI355X : public TimestepSchemeMI355X {
public:
	TimestepSchemeStrang_MI355X(
		Model & model,
		double dOffCentering = 0.0,
		TimestepSchemeStrang::ExplicitDiscretization eExplicitDiscretization = TimestepSchemeStrang::KinnmarkGrayUllrich35
	) :
		TimestepSchemeMI355X(model, SchemeId(eExplicitDiscretization)),
		m_dOffCentering(dOffCentering)
	{
		if ((dOffCentering < 0.0) || (dOffCentering > 1.0)) {
			_EXCEPTIONT("OffCentering parameter out of range [0,1]");
		}
	}

	static int SchemeId(TimestepSchemeStrang::ExplicitDiscretization e) {
		switch (e) {
			case TimestepSchemeStrang::ForwardEuler: return TMX_SCHEME_STRANG_FE;
			case TimestepSchemeStrang::RungeKutta4: return TMX_SCHEME_STRANG_RK4;
			case TimestepSchemeStrang::RungeKuttaSSP3: return TMX_SCHEME_STRANG_SSP3;
			case TimestepSchemeStrang::RungeKuttaSSPRK53: return TMX_SCHEME_STRANG_SSPRK53;
			default: return TMX_SCHEME_STRANG;
		}
	}

	virtual void Step(bool fFirstStep, bool fLastStep, const Time & time, double dDeltaT) {
		MI355XEngine::Check(tmx_set_strang_off_centering(MI355XEngine::For(m_model).Handle(), m_dOffCentering));
		TimestepSchemeMI355X::Step(fFirstStep, fLastStep, time, dDeltaT);
	}

private:
	double m_dOffCentering;
};

///////////////////////////////////////////////////////////////////////////////

///	<summary>
///		HeldSuarezPhysics (src/atm/HeldSuarezPhysics.h) on the device: a WorkflowProcess with the same constructor,
///		attached with Model::AttachWorkflowProcess exactly like the reference class.  The state stays on the device
///		between the time step and the forcing (MI355XEngine's host synchronisation policy decides when the host copy is
///		refreshed); the "surface pressure" of HeldSuarezPhysics.cpp:113-116 is formed on the device from the interface
///		entries the engine tracks through the stage algebra.
///	</summary>
class HeldSuarezPhysicsMI355X : public WorkflowProcess {
public:
	HeldSuarezPhysicsMI355X(Model & model, const Time & timeFrequency) :
		WorkflowProcess(model, timeFrequency), m_fInputsSet(false) { }

	///	Model::Go calls this after the geometric terms exist and before the first step (Model.cpp:343-366): the engine
	///	must know from the first step on that the surface slots are to be carried through the stage algebra.
	virtual void Initialize(const Time & timeStart) {
		WorkflowProcess::Initialize(timeStart);
		RegisterInputs();
	}

	void RegisterInputs() {
		if (m_fInputsSet) {
			return;
		}
		MI355XEngine & eng = MI355XEngine::For(m_model);
		tmx_engine * pEngine = eng.Handle();
		Grid * pGrid = m_model.GetGrid();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			MI355XEngine::Check(tmx_set_patch_physics_inputs(pEngine, pPatch->GetPatchIndex(),
				&(pPatch->GetLatitude()[0][0]), NULL));
		}
		m_fInputsSet = true;
		// the surface slots travel with the state: make the host copy current, then have it uploaded again
		eng.Flush();
		eng.MarkHostModified();
	}

	virtual void Perform(const Time & time) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		tmx_engine * pEngine = eng.Handle();
		RegisterInputs();      // late attachment: from here on (the step just taken ran without tracked surface slots)
		eng.EnsureDevice();
		MI355XEngine::Check(tmx_physics_held_suarez(pEngine, 0, m_timeFrequency.GetSeconds()));
		eng.DeviceModified(false, false);
		WorkflowProcess::Perform(time);
	}

private:
	bool m_fInputsSet;
};

///	<summary>
///		KesslerPhysics (test/dcmip2016/KesslerPhysics.h) on the device: same constructor, same attachment.  The
///		accumulated precipitation (UserData2D item 0, KesslerPhysics.cpp:234) is kept on the device and written to the
///		patch's UserData2D by FlushPrecipitation() (called whenever the host copy of the state is refreshed here).
///	</summary>
class KesslerPhysicsMI355X : public WorkflowProcess {
public:
	KesslerPhysicsMI355X(Model & model, const Time & timeFrequency) :
		WorkflowProcess(model, timeFrequency), m_fInputsSet(false) { }

	void FlushPrecipitation() {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		Grid * pGrid = m_model.GetGrid();
		for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
			GridPatch * pPatch = pGrid->GetActivePatch(n);
			DataArray3D<double> & dataUserData2D = pPatch->GetUserData2D();
			if (dataUserData2D.GetRows() == 0) {
				_EXCEPTIONT("Insufficient entries in UserData2D");
			}
			// the device accumulator is added to the host array and reset, so that the host array keeps the running total
			const PatchBox & box = pPatch->GetPatchBox();
			DataArray2D<double> dAcc(box.GetATotalWidth(), box.GetBTotalWidth());
			MI355XEngine::Check(tmx_download_precipitation(eng.Handle(), pPatch->GetPatchIndex(), &(dAcc[0][0]), 1));
			for (int i = box.GetAInteriorBegin(); i < box.GetAInteriorEnd(); i++)
			for (int j = box.GetBInteriorBegin(); j < box.GetBInteriorEnd(); j++)
				dataUserData2D[0][i][j] += dAcc[i][j];
		}
	}

	static void FlushHook(void * pThis) { static_cast<KesslerPhysicsMI355X *>(pThis)->FlushPrecipitation(); }

	virtual void Perform(const Time & time) {
		MI355XEngine & eng = MI355XEngine::For(m_model);
		tmx_engine * pEngine = eng.Handle();
		Grid * pGrid = m_model.GetGrid();
		if (!m_fInputsSet) {
			for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
				GridPatch * pPatch = pGrid->GetActivePatch(n);
				MI355XEngine::Check(tmx_set_patch_level_heights(pEngine, pPatch->GetPatchIndex(), &(pPatch->GetZLevels()[0][0][0])));
			}
			m_fInputsSet = true;
		}
		eng.SetFlushHook(&KesslerPhysicsMI355X::FlushHook, this);      // precipitation follows every refresh of the host copy
		eng.EnsureDevice();
		MI355XEngine::Check(tmx_physics_kessler(pEngine, 0, m_timeFrequency.GetSeconds()));
		eng.DeviceModified(false, false);
		WorkflowProcess::Perform(time);
	}

private:
	bool m_fInputsSet;
};

///////////////////////////////////////////////////////////////////////////////

///	<summary>
///		Grid::ReduceInterpolate (src/atm/Grid.h:390-400) on the device: what OutputManagerReference needs for an output
///		(OutputManagerReference.cpp:588-628) without bringing the state back to the host.  Same arguments as the
///		reference call; the per-point data come from the reference's own functions, exactly as
///		GridPatchCSGLL::InterpolateData (GridPatchCSGLL.cpp:1580-1622) derives them.  The plan is rebuilt when the
///		sample points or the vertical coordinate change.
///	</summary>
class ReduceInterpolateMI355X {
public:
	ReduceInterpolateMI355X(Model & model) : m_model(model), m_pPlan(NULL) { }

	~ReduceInterpolateMI355X() {
		if (m_pPlan != NULL) {
			tmx_interp_destroy(m_pPlan);
		}
	}

	void operator()(
		DataType eDataType,
		const DataArray1D<double> & dREta,
		const DataArray1D<double> & dAlpha,
		const DataArray1D<double> & dBeta,
		const DataArray1D<int> & iPatch,
		DataArray3D<double> & dInterpData,
		DataLocation eOnlyVariablesAt = DataLocation_None,
		bool fIncludeReferenceState = true,
		bool fConvertToPrimitive = true
	) {
		if ((eDataType != DataType_State) && (eDataType != DataType_Tracers)) {
			_EXCEPTIONT("ReduceInterpolateMI355X: only DataType_State and DataType_Tracers live on the device");
		}
		MI355XEngine & eng = MI355XEngine::For(m_model);
		tmx_engine * pEngine = eng.Handle();
		GridGLL * pGrid = dynamic_cast<GridGLL *>(m_model.GetGrid());
		const int nPoints = dAlpha.GetRows();
		const int nREta = dREta.GetRows();

		// (re)build the plan
		std::vector<double> vecKey;
		for (int k = 0; k < nREta; k++) vecKey.push_back(dREta[k]);
		for (int i = 0; i < nPoints; i++) { vecKey.push_back(dAlpha[i]); vecKey.push_back(dBeta[i]); vecKey.push_back(iPatch[i]); }
		if ((m_pPlan == NULL) || (vecKey != m_vecKey)) {
			if (m_pPlan != NULL) {
				tmx_interp_destroy(m_pPlan);
				m_pPlan = NULL;
			}
			const int nOrder = pGrid->GetHorizontalOrder();
			std::vector<int> vecPatch(nPoints), vecA(nPoints, 1), vecB(nPoints, 1);
			std::vector<double> vecCA(nPoints * nOrder, 0.0), vecCB(nPoints * nOrder, 0.0), vecM(nPoints * 4, 0.0);
			for (int i = 0; i < nPoints; i++) {
				vecPatch[i] = iPatch[i];
				GridPatch * pPatch = NULL;
				for (int n = 0; n < pGrid->GetActivePatchCount(); n++) {
					if (pGrid->GetActivePatch(n)->GetPatchIndex() == iPatch[i]) pPatch = pGrid->GetActivePatch(n);
				}
				if (pPatch == NULL) {
					continue;      // another rank's patch: this rank contributes zeros
				}
				const PatchBox & box = pPatch->GetPatchBox();
				GridPatchGLL * pGLL = dynamic_cast<GridPatchGLL *>(pPatch);
				int iA = (dAlpha[i] - pPatch->GetAEdge(box.GetAInteriorBegin())) / pGLL->GetElementDeltaA();
				int iB = (dBeta[i] - pPatch->GetBEdge(box.GetBInteriorBegin())) / pGLL->GetElementDeltaB();
				if (iA < 0) iA = 0;
				if (iA >= box.GetAInteriorWidth() / nOrder) iA = box.GetAInteriorWidth() / nOrder - 1;
				if (iB < 0) iB = 0;
				if (iB >= box.GetBInteriorWidth() / nOrder) iB = box.GetBInteriorWidth() / nOrder - 1;
				iA = box.GetHaloElements() + iA * nOrder;
				iB = box.GetHaloElements() + iB * nOrder;
				vecA[i] = iA; vecB[i] = iB;
				PolynomialInterp::LagrangianPolynomialCoeffs(nOrder, &(pPatch->GetAEdges()[iA]), &vecCA[i * nOrder], dAlpha[i]);
				PolynomialInterp::LagrangianPolynomialCoeffs(nOrder, &(pPatch->GetBEdges()[iB]), &vecCB[i * nOrder], dBeta[i]);
				CubedSphereTrans::CoVecTransRLLFromABP(tan(dAlpha[i]), tan(dBeta[i]), box.GetPanel(), 1.0, 0.0, vecM[i * 4 + 0], vecM[i * 4 + 2]);
				CubedSphereTrans::CoVecTransRLLFromABP(tan(dAlpha[i]), tan(dBeta[i]), box.GetPanel(), 0.0, 1.0, vecM[i * 4 + 1], vecM[i * 4 + 3]);
			}
			LinearColumnInterpFEM opLevels, opInterfaces;
			opLevels.Initialize(LinearColumnInterpFEM::InterpSource_Levels, pGrid->GetVerticalOrder(),
				pGrid->GetREtaLevels(), pGrid->GetREtaInterfaces(), dREta);
			opInterfaces.Initialize(LinearColumnInterpFEM::InterpSource_Interfaces, pGrid->GetVerticalOrder(),
				pGrid->GetREtaLevels(), pGrid->GetREtaInterfaces(), dREta);
			tmx_interp_points pts;
			pts.n_points = nPoints; pts.n_reta = nREta;
			pts.patch = &vecPatch[0]; pts.node_a = &vecA[0]; pts.node_b = &vecB[0];
			pts.coeff_a = &vecCA[0]; pts.coeff_b = &vecCB[0]; pts.rll_from_abp = &vecM[0];
			pts.op_levels = &(opLevels.GetCoeffs()[0][0]); pts.op_interfaces = &(opInterfaces.GetCoeffs()[0][0]);
			MI355XEngine::Check(tmx_interp_create(pEngine, &pts, &m_pPlan));
			m_vecKey = vecKey;
		}

		if (eDataType == DataType_State) {
			if (!fIncludeReferenceState) {
				eng.SetReferenceState();
			}
			const int iOnly = (eOnlyVariablesAt == DataLocation_Node) ? 1 : ((eOnlyVariablesAt == DataLocation_REdge) ? 2 : 0);
			MI355XEngine::Check(tmx_interp_state(pEngine, m_pPlan, 0, iOnly, fIncludeReferenceState ? 1 : 0,
				fConvertToPrimitive ? 1 : 0, m_model.GetPhysicalConstants().GetEarthRadius(), &(dInterpData[0][0][0])));
		} else {
			MI355XEngine::Check(tmx_interp_tracers(pEngine, m_pPlan, 0, &(dInterpData[0][0][0])));
		}
#ifdef TEMPEST_MPIOMP
		// as Grid::ReduceInterpolate (Grid.cpp:962-990): the ranks' partial arrays are summed on rank 0
		int nRank;
		MPI_Comm_rank(MPI_COMM_WORLD, &nRank);
		const int nCount = dInterpData.GetRows() * dInterpData.GetColumns() * dInterpData.GetSubColumns();
		if (nRank == 0) MPI_Reduce(MPI_IN_PLACE, &(dInterpData[0][0][0]), nCount, MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
		else MPI_Reduce(&(dInterpData[0][0][0]), NULL, nCount, MPI_DOUBLE, MPI_SUM, 0, MPI_COMM_WORLD);
#endif
	}

private:
	Model & m_model;
	tmx_interp * m_pPlan;
	std::vector<double> m_vecKey;
};

#endif
