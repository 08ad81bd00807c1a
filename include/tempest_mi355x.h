/* tempest_mi355x.h -- C ABI of the MI355X-native spectral-element dynamics engine.
 *
 * Drop-in boundary for ONE path of paullric/tempestmodel: the per-GridPatch RHS evaluation
 * (HorizontalDynamicsFEM / VerticalDynamicsFEM), the ARS343 IMEX stage loop and the DSS / halo
 * exchange.  Every entry point names the reference interface it stands in for (file:line under
 * the reference tree).  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * All functions return 0 on success and a negative TMX_ERR_* code on failure;
 * tmx_last_error() then returns a message (the reference convention is `throw Exception(...)`,
 * src/base/Exception.h:25-49 -- the host adapter turns a non-zero return into _EXCEPTIONT).
 * The library is not re-entrant: one engine <-> one host thread <-> one GPU, exactly as one
 * reference MPI rank is one thread.
 *
 * Interchange layout = the reference's own patch layout (src/atm/GridPatch.cpp:341-356,
 * src/base/DataArray4D.h:123-231): per patch, row-major
 *     node  [5][na][nb][L]      redge [5][na][nb][L+1]     na = 4*elems_a + 2, nb = 4*elems_b + 2
 * with a one-node halo ring and duplicated element-edge nodes; variable order U,V,rho*theta,W,rho
 * (src/atm/EquationSet.cpp:51-84).  The engine keeps its own element-major layout in HBM
 *  and converts on upload / download.
 *
 * Supported configuration (anything else -> TMX_ERR_UNSUPPORTED): horizontal order np = 4,
 * FE vertical order 1, Lorenz staggering (W on interfaces), FORMULATION_RHOTHETA_PI,
 * USE_DIRECTSOLVE + USE_JACOBIAN_DIAGONAL, upwinding on all variables, Clark-form implicit W
 * advection, hyperviscosity order 4, tracers without uniform diffusion (src/atm/Defines.h:23-80,
 * src/atm/VerticalDynamicsFEM.cpp:33-47).
 */
#ifndef TEMPEST_MI355X_H
#define TEMPEST_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

#define TMX_ABI_VERSION 5

#define TMX_OK                 0
#define TMX_ERR_INVALID       -1   /* bad argument / call order                      */
#define TMX_ERR_UNSUPPORTED   -2   /* configuration outside the supported path      */
#define TMX_ERR_DEVICE        -3   /* HIP runtime error                             */
#define TMX_ERR_SINGULAR      -4   /* column solve hit an exactly zero pivot        */
#define TMX_ERR_COMM          -5   /* RCCL error                                    */

typedef struct tmx_engine tmx_engine;

/* column operators, in the order of GridGLL::GetOp* (src/atm/GridGLL.h:299-355) */
enum {
	TMX_OP_INTERP_NODE_TO_REDGE = 0,
	TMX_OP_INTERP_REDGE_TO_NODE,
	TMX_OP_DIFF_NODE_TO_NODE,
	TMX_OP_DIFF_NODE_TO_REDGE,
	TMX_OP_DIFF_REDGE_TO_NODE,
	TMX_OP_DIFF_REDGE_TO_REDGE,
	TMX_OP_DIFFDIFF_NODE_TO_NODE,
	TMX_OP_DIFFDIFF_REDGE_TO_REDGE,
	TMX_OP_PENALTY_LEFT,
	TMX_OP_PENALTY_RIGHT,
	TMX_OP_COUNT
};

/* Scalar configuration.  Sources in the reference: Grid parameters (src/atm/Grid.h:626-816),
 * PhysicalConstants (src/atm/PhysicalConstants.h:120-135, read AFTER Model::SetTestCase),
 * HorizontalDynamicsFEM constructor arguments (src/atm/HorizontalDynamicsFEM.h:42-50),
 * TimestepScheme::GetComponentDataInstances (src/atm/TimestepScheme.h:54-60). */
typedef struct {
	int    abi_version;        /* TMX_ABI_VERSION                                              */
	int    horizontal_order;   /* np; must be 4                                                */
	int    vertical_order;     /* must be 1                                                    */
	int    levels;             /* Grid::GetRElements()                                         */
	int    n_patches;          /* total number of patches of the grid (all ranks)              */
	int    n_instances;        /* state instances per patch (ARS343: 7)                        */
	int    n_tracers;          /* EquationSet::GetTracers(); tracer densities on levels (nonhydrostatic set only) */
	int    hypervis_order;     /* 0, 2 (viscosity) or 4 (two-pass hyperviscosity): HorizontalDynamicsFEM ctor */
	double element_delta_a;    /* GridPatchGLL::GetElementDeltaA()                             */
	double reference_length;   /* Grid::GetReferenceLength()                                   */
	double grav, Rd, cp, p0;   /* PhysicalConstants::GetG/GetR/GetCp/GetP0                     */
	double nu_scalar, nu_div, nu_vort;
	int    rank, n_ranks;      /* this process / number of processes sharing the grid          */
	int    device;             /* HIP device ordinal, -1 = current, -2 = plan only (no GPU)    */
	int    equation_set;       /* EquationSet::Type (src/atm/EquationSet.h): TMX_EQN_*          */
	int    fully_explicit;     /* VerticalDynamicsFEM ctor fFullyExplicit (--explicitvertical, TempestInitialize.h:372-379):
	                              V.StepExplicit applies -dt F of the column, V.StepImplicit does nothing       */
	int    uniform_diffusion;  /* Grid::HasUniformDiffusion() (src/atm/Grid.cpp:399-415); requires fully_explicit and
	                              tmx_set_patch_reference_state on every owned patch                           */
	double uniform_diffusion_scalar, uniform_diffusion_vector;   /* Grid::GetScalar/VectorUniformDiffusionCoeff() [m^2/s] */
	double ztop;               /* Grid::GetZtop() (vertical uniform diffusion divides by ztop^2); unused otherwise  */
} tmx_config;

#define TMX_EQN_PRIMITIVE_NONHYDROSTATIC 0   /* 5 components U,V,rho*theta,W,rho                        */
#define TMX_EQN_SHALLOW_WATER            1   /* 3 components U,V,H on one level; node is [3][na][nb][1],
                                                redge is ignored; vertical dynamics are the reference's
                                                VerticalDynamicsStub (levels == 1, TempestInitialize.h:362) */

/* Geometry of one patch, reference-layout pointers (accessors src/atm/GridPatch.h:482-740).
 * Arrays are read during the call only. */
typedef struct {
	const double * jacobian2d;             /* [na][nb]          GetJacobian2D           */
	const double * contra_metric_2d_a;     /* [na][nb][2]       GetContraMetric2DA      */
	const double * contra_metric_2d_b;     /* [na][nb][2]       GetContraMetric2DB      */
	const double * jacobian;               /* [na][nb][L]       GetJacobian             */
	const double * jacobian_redge;         /* [na][nb][L+1]     GetJacobianREdge        */
	const double * contra_metric_a;        /* [na][nb][L][3]    GetContraMetricA        */
	const double * contra_metric_b;        /* [na][nb][L][3]    GetContraMetricB        */
	const double * contra_metric_xi;       /* [na][nb][L][3]    GetContraMetricXi       */
	const double * contra_metric_a_redge;  /* [na][nb][L+1][3]  GetContraMetricAREdge   */
	const double * contra_metric_b_redge;  /* [na][nb][L+1][3]  GetContraMetricBREdge   */
	const double * contra_metric_xi_redge; /* [na][nb][L+1][3]  GetContraMetricXiREdge  */
	const double * deriv_r_node;           /* [na][nb][L][3]    GetDerivRNode           */
	const double * deriv_r_redge;          /* [na][nb][L+1][3]  GetDerivRREdge          */
	const double * coriolis_f;             /* [na][nb]          GetCoriolisF            */
	const double * topography;             /* [na][nb]          GetTopography (shallow water only; may be NULL) */
	const double * element_area_node;      /* [na][nb][L]       GetElementAreaNode (tracer filters; NULL when n_tracers == 0) */
} tmx_patch_geometry;

/* ---- life cycle ---------------------------------------------------------------------------- */

/* Replaces the constructors + Initialize() of HorizontalDynamicsFEM / VerticalDynamicsFEM /
 * TimestepSchemeARS343 (src/atm/HorizontalDynamicsFEM.cpp:61-209,
 * src/atm/VerticalDynamicsFEM.cpp:95-435, src/atm/TimestepSchemeARS343.cpp:25-146). */
int tmx_create(const tmx_config * cfg, tmx_engine ** out);
void tmx_destroy(tmx_engine * e);
const char * tmx_last_error(void);

/* GridGLL::GetDxBasis1D / GetStiffness1D (src/atm/GridGLL.h:268-290) and the ten column operators
 * (LinearColumnOperator::GetCoeffs/GetIxBegin/GetIxEnd, src/atm/LinearColumnOperator.h).
 * dx_basis[s*4+i] = dDxBasis1D(s,i); stiffness[i*4+s] = dStiffness1D(i,s);
 * coeff[op] is dense [n_out][n_in] row-major. */
int tmx_set_operators(tmx_engine * e, const double * dx_basis, const double * stiffness,
	const double * const * coeff, const int * const * ix_begin, const int * const * ix_end,
	const int * n_out, const int * n_in);

/* One call per patch of the WHOLE grid, on every rank (PatchBox, src/atm/PatchBox.h; owner =
 * Grid::DistributePatches, src/atm/Grid.cpp:1038-1062; neighbor_panels = GridPatch::GetNeighborPanel
 * in Direction order Right,Top,Left,Bottom,TopRight,TopLeft,BottomLeft,BottomRight, -1 = none). */
int tmx_define_patch(tmx_engine * e, int patch, int panel, int elems_a, int elems_b, int owner_rank,
	const int * neighbor_panels);

/* Halo ring of one patch (every patch, every rank): ring node (halo_i, halo_j) mirrors interior node
 * (src_i, src_j) of patch src_patch (-1: no source, cube corner); trans[n][2][2] is the covector
 * panel transform applied to (U,V) arriving there.  Stands in for the ExchangeBuffer registry and
 * GridPatchCSGLL::TransformHaloVelocities (src/atm/Connectivity.cpp:797-924,
 * src/atm/GridPatchCSGLL.cpp:1783-1924, src/atm/CubedSphereTrans.h:1751-1823). */
/* GridPatchGLL::GetElementDeltaA() / GetElementDeltaB() of one patch (any patch of the grid, owned or not; before
 * tmx_finalize).  The reference forms them as the difference of two element-edge coordinates of the patch
 * (src/atm/GridPatchGLL.cpp:67-75): they differ from pi / (2 ne), and between patches, in the last bits, and every
 * horizontal derivative is scaled with the patch's own 1 / delta (HorizontalDynamicsFEM.cpp:837-838, 1013-1016, 1957-1958,
 * 1973 ...).  Optional: without the call tmx_config.element_delta_a serves both directions. */
int tmx_set_patch_element_spacing(tmx_engine * e, int patch, double delta_a, double delta_b);

int tmx_set_patch_halo(tmx_engine * e, int patch, int n, const int * halo_i, const int * halo_j,
	const int * src_patch, const int * src_i, const int * src_j, const int * src_panel,
	const double * trans);

/* Geometry of a patch owned by this rank (GridPatchCSGLL::EvaluateGeometricTerms output,
 * src/atm/GridPatchCSGLL.cpp:295-574). */
int tmx_set_patch_geometry(tmx_engine * e, int patch, const tmx_patch_geometry * g);

/* OPTIONAL, per owned patch, after tmx_set_patch_geometry: the factors of the closed form behind the 3-D
 * terrain-following metric (Gal-Chen coordinate, src/atm/GridPatchCSGLL.cpp:370-568):
 *   x_node[i] = tan(alpha_i), y_node[j] = tan(beta_j)  (GridPatchCSGLL::m_dXNode / m_dYNode = tan of GetANode / GetBNode),
 *   topography_deriv[i][j][2]                           (GridPatch::GetTopographyDeriv, after its start-up DSS),
 *   earth_radius                                        (PhysicalConstants::GetEarthRadius),
 *   reta_levels[L], reta_interfaces[L+1]                (Grid::GetREtaLevels / GetREtaInterfaces).
 * The engine re-evaluates contra_metric_{a,b,xi}[..][2], contra_metric_xi_redge and deriv_r_node[..][0..1] from
 * these with the reference's own expressions and compares them with the arrays of tmx_set_patch_geometry BIT FOR
 * BIT.  Only if every value of every owned patch agrees do the kernels evaluate the metric in registers instead of
 * streaming the 3-D arrays from HBM (and the arrays are then not kept on the device); otherwise the stored arrays
 * stay in use.  Results are identical either way; tmx_info(TMX_INFO_METRIC_CLOSED_FORM) tells which is active.
 * Returns TMX_OK also when the comparison fails (it is not an error). */
int tmx_set_patch_metric_factors(tmx_engine * e, int patch, const double * x_node, const double * y_node,
	const double * topography_deriv, double earth_radius, const double * reta_levels, const double * reta_interfaces);

/* OPTIONAL, per owned patch: Rayleigh friction (Grid::HasRayleighFriction(); test cases with a damping layer,
 * e.g. test/nonhydro_sphere/ScharMountainSphereTest.cpp).  strength_node [na][nb][L] / strength_redge [na][nb][L+1]
 * = GridPatch::GetRayleighStrength(DataLocation_Node / _REdge), ref_node [5][na][nb][L] / ref_redge [5][na][nb][L+1]
 * = GridPatch::GetReferenceState (src/atm/GridPatch.h, filled by GridPatchCSGLL::EvaluateTestCase,
 * src/atm/GridPatchCSGLL.cpp:663-683).  When set (on every owned patch or on none), tmx_h_step_after_subcycle ends
 * with HorizontalDynamicsFEM::ApplyRayleighFriction (src/atm/HorizontalDynamicsFEM.cpp:2420-2570, called at
 * :2719-2724 under APPLY_RAYLEIGH_WITH_HYPERVIS): ten sub-cycled backward-Euler relaxations of U, V, rho*theta and
 * W towards the reference state wherever the strength is non-zero.  Not available for the shallow-water set. */
int tmx_set_patch_rayleigh(tmx_engine * e, int patch, const double * strength_node, const double * strength_redge,
	const double * ref_node, const double * ref_redge);

/* Per owned patch, required when cfg.uniform_diffusion != 0: the reference state the uniform diffusion is taken
 * relative to.  ref_node [5][na][nb][L] / ref_redge [5][na][nb][L+1] = GridPatch::GetReferenceState(DataLocation_Node /
 * _REdge), ref_tracers [n_tracers][na][nb][L] = GridPatch::GetReferenceTracers() (NULL when n_tracers == 0); same
 * layout as tmx_upload_state.  Also accepted without uniform diffusion (tmx_interp_state can subtract it).  Used by: the uniform-diffusion extras of HorizontalDynamicsFEM::StepExplicit
 * (src/atm/HorizontalDynamicsFEM.cpp:1817-1859; tracer fluxes :1105-1148), the vertical diffusion of U,V in
 * VerticalDynamicsFEM::StepExplicit (src/atm/VerticalDynamicsFEM.cpp:1059-1105), of rho*theta and W in BuildF
 * (:2593-2635) and of the tracer mixing ratios in UpdateColumnTracers (:4117-4141). */
int tmx_set_patch_reference_state(tmx_engine * e, int patch, const double * ref_node, const double * ref_redge,
	const double * ref_tracers);

/* Builds the HBM layout, DSS groups, unique-column lists and exchange plan. */
int tmx_finalize(tmx_engine * e);

/* ---- state transfer (GridPatch::GetDataState, src/atm/GridPatch.h:806-870) ------------------ */

/* Reads the prognostic slots of interior nodes (U,V,rho*theta,rho from node; W from redge). */
int tmx_upload_state(tmx_engine * e, int patch, int instance, const double * node, const double * redge);
/* Writes interior nodes: prognostic slots, plus the derived slots the reference keeps beside them
 * (W on levels, U,V on interfaces: HorizontalDynamicsFEM.cpp:817-831).  Halo ring untouched. */
int tmx_download_state(tmx_engine * e, int patch, int instance, double * node, double * redge);

/* Tracer densities of one instance, GridPatch::GetDataTracers(instance): [n_tracers][na][nb][L] (src/atm/GridPatch.h:885-905).
 * Every operation of this interface that the reference applies to DataType_Tracers next to DataType_State (CopyData,
 * LinearCombineData, ZeroData, PostProcessSubstage/ApplyDSS, the tracer parts of StepExplicit / StepImplicit /
 * StepAfterSubCycle incl. FilterNegativeTracers) acts on them as well. */
int tmx_upload_tracers(tmx_engine * e, int patch, int instance, const double * tracers);
int tmx_download_tracers(tmx_engine * e, int patch, int instance, double * tracers);

/* ---- stage algebra (Grid::CopyData / ZeroData / LinearCombineData, src/atm/Grid.cpp:1585-1621) */
int tmx_copy_data(tmx_engine * e, int src, int dst);
int tmx_zero_data(tmx_engine * e, int ix);
int tmx_linear_combine_data(tmx_engine * e, const double * coeff, int n_coeff, int dst);

/* ---- dynamics ------------------------------------------------------------------------------ */

/* HorizontalDynamicsFEM::StepExplicit -> StepNonhydrostaticPrimitive or StepShallowWater
 * (src/atm/HorizontalDynamicsFEM.cpp:1787-1863, 701-1783, 321-647). */
int tmx_h_step_explicit(tmx_engine * e, int i_initial, int i_update, double dt);
/* VerticalDynamicsFEM::StepExplicit, implicit mode (src/atm/VerticalDynamicsFEM.cpp:616-1159). */
int tmx_v_step_explicit(tmx_engine * e, int i_initial, int i_update, double dt);
/* VerticalDynamicsFEM::StepImplicit (src/atm/VerticalDynamicsFEM.cpp:1230-1638): one Newton step
 * per unique column with the analytic band Jacobian and a dgbsv-equivalent pivoted band LU. */
int tmx_v_step_implicit(tmx_engine * e, int i_initial, int i_update, double dt);
/* GridGLL::PostProcessSubstage(State) = GridCSGLL::ApplyDSS incl. Grid::Exchange
 * (src/atm/GridGLL.cpp:571-583, src/atm/GridCSGLL.cpp:435-781, src/atm/Grid.cpp:627-685). */
int tmx_apply_dss(tmx_engine * e, int instance);
/* HorizontalDynamicsFEM::StepAfterSubCycle (src/atm/HorizontalDynamicsFEM.cpp:2637-2726). */
int tmx_h_step_after_subcycle(tmx_engine * e, int i_initial, int i_update, int i_working, double dt);

/* TimestepScheme::Step of the reference's IMEX / split steppers, the whole step resident on the device and
 * asynchronous on the engine's stream.  Each is the reference's own call sequence:
 *   ARS343  src/atm/TimestepSchemeARS343.cpp:150-236     ARS232  src/atm/TimestepSchemeARS232.cpp:52-150
 *   ARS222  src/atm/TimestepSchemeARS222.cpp:51-119      ARK232  src/atm/TimestepSchemeARK232.cpp:53-225
 *   STRANG  src/atm/TimestepSchemeStrang.cpp:450-674 with KinnmarkGrayUllrich35 (the default --timescheme of every
 *           reference test, TempestInitialize.h:142); STRANG_FE / _RK4 / _SSP3 / _SSPRK53 are the same step with the
 *           constructor's other ExplicitDiscretization values (TimestepSchemeStrang.h:39-45, .cpp:39-51,484-648)
 * first_step / last_step have the meaning of TimestepScheme::Step's fFirstStep / fLastStep.
 * The stage CopyData calls of these sequences are mostly not performed: a copy that is only read until it is next overwritten
 * reads its source (whole instance, or just U,V in front of the implicit column solve, which rewrites the rest).  This is
 * invisible at this interface: every other entry point first gives the instances it names storage of their own, so after
 * tmx_step every data instance holds what the reference's holds. */
enum { TMX_SCHEME_ARS343 = 0, TMX_SCHEME_ARS232, TMX_SCHEME_ARS222, TMX_SCHEME_ARS443, TMX_SCHEME_ARK232, TMX_SCHEME_STRANG,
       TMX_SCHEME_STRANG_FE, TMX_SCHEME_STRANG_RK4, TMX_SCHEME_STRANG_SSP3, TMX_SCHEME_STRANG_SSPRK53 };
/* dOffCentering of TimestepSchemeStrang's constructor (TimestepSchemeStrang.cpp:25-37,58-61,651-660), in [0, 1]; 0 by default. */
int tmx_set_strang_off_centering(tmx_engine * e, double off_centering);
int tmx_step(tmx_engine * e, int scheme, int first_step, int last_step, double dt);
/* TimestepScheme::GetComponentDataInstances of that scheme (-1: unknown). */
int tmx_scheme_instances(int scheme);
int tmx_step_ars343(tmx_engine * e, double dt);   /* = tmx_step(e, TMX_SCHEME_ARS343, 0, 0, dt) */

/* VerticalDynamicsFEM::StepImplicitTermsExplicitly (src/atm/VerticalDynamicsFEM.cpp:439-612): update -= dt F(initial)
 * on every column (used by ARK232). */
/* ---- restart image: the bytes of GridPatch::GetDataContainerActiveState() (src/atm/GridPatch.cpp:359-361,387),
 * which OutputManagerComposite::Output writes per patch (src/atm/OutputManagerComposite.cpp:331-349) and a restart reads back:
 *   8 bytes   patch index (DataArray1D<int> of one entry, padded to sizeof(size_t))
 *   node   [5][na][nb][L]       redge  [5][na][nb][L+1]       tracers [n_tracers][na][nb][L]
 * The transposition from the device layout is done ON the device, so the host side of a restart write is one contiguous
 * device-to-host copy per patch straight into the container (dst = dcActiveState.GetPointer()).  Interior nodes carry every
 * array the reference keeps there (the prognostic ones, W on levels and U,V on interfaces as HorizontalDynamicsFEM.cpp:817-831
 * leaves them, and the two interface-level-0 entries described at tmx_set_patch_physics_inputs when they are tracked).  The halo
 * ring and the interface copies of rho / rho*theta above level 0 are written as zeros: nothing on the path reads them before
 * the next exchange rewrites the ring, so a run restarted from the image continues bit for bit (tests/test_gpu_adapter.py,
 * --restart).  tmx_unpack_active_state is the inverse (restart read).  Bytes must equal tmx_active_state_bytes(e, patch). */
long long tmx_active_state_bytes(tmx_engine * e, int patch);
int tmx_pack_active_state(tmx_engine * e, int patch, int instance, void * dst, size_t dst_bytes);
int tmx_unpack_active_state(tmx_engine * e, int patch, int instance, const void * src, size_t src_bytes);

/* ---- output interpolation: Grid::ReduceInterpolate (src/atm/Grid.cpp:867-992) ->
 * GridPatchCSGLL::InterpolateData (src/atm/GridPatchCSGLL.cpp:1365-1780), what OutputManagerReference calls for every
 * output (src/atm/OutputManagerReference.cpp:588-628).  The state stays on the device; only the interpolated fields
 * [component][n_reta][n_points] come back.  The caller passes, per sample point, what it gets from the reference's own
 * functions (as for the halo transforms): patch index and (alpha, beta)-derived quantities
 *   node_a / node_b  first node of the containing element incl. the halo offset (iA, iB of GridPatchCSGLL.cpp:1588-1611)
 *   coeff_a/_b       PolynomialInterp::LagrangianPolynomialCoeffs over GetAEdges()[iA..iA+3] / GetBEdges()[iB..iB+3]
 *   rll_from_abp     2x2 matrix of CubedSphereTrans::CoVecTransRLLFromABP(tan alpha, tan beta, panel, ...) probed with
 *                    unit vectors, row-major (lon; lat) x (alpha, beta); may be NULL if convert_to_primitive is never used
 *   op_levels / op_interfaces   dense coefficients [n_reta][L] / [n_reta][L+1] of LinearColumnInterpFEM::Initialize(
 *                    InterpSource_Levels / _Interfaces, vertical order, REtaLevels, REtaInterfaces, dREta)
 * Points on patches of other ranks produce zeros (the reference sums the ranks' arrays with MPI_Reduce). */
typedef struct tmx_interp tmx_interp;
typedef struct {
	int n_points, n_reta;
	const int * patch; const int * node_a; const int * node_b;
	const double * coeff_a; const double * coeff_b;
	const double * rll_from_abp;
	const double * op_levels; const double * op_interfaces;
} tmx_interp_points;
int tmx_interp_create(tmx_engine * e, const tmx_interp_points * points, tmx_interp ** out);
void tmx_interp_destroy(tmx_interp * plan);
/* DataType_State: out[5][n_reta][n_points].  only_variables_at: 0 DataLocation_None (all), 1 levels only, 2 interfaces
 * only (skipped rows are zero); include_reference_state = 0 subtracts the interpolated reference state (needs
 * tmx_set_patch_reference_state); convert_to_primitive: W / d_xi R, (U,V) -> zonal / meridional velocity. */
int tmx_interp_state(tmx_engine * e, tmx_interp * plan, int instance, int only_variables_at, int include_reference_state,
	int convert_to_primitive, double earth_radius, double * out);
/* DataType_Tracers: out[n_tracers][n_reta][n_points]. */
int tmx_interp_tracers(tmx_engine * e, tmx_interp * plan, int instance, double * out);

/* HorizontalDynamics::GetSubStepAfterSubCycleCount() / SubStepAfterSubCycle(iInit,iUpd,iWork,time,dt,iSubStep)
 * (src/atm/HorizontalDynamics.h:128-147, HorizontalDynamicsFEM.cpp:2574-2633): the two halves of the hyperviscosity
 * step without the DSS calls, for callers that exchange between them (Model::SubStep).  Sub-step 0 leaves the
 * Laplacians of the initial instance in the working instance; sub-step 1 forms the update instance, filters the
 * tracers and applies Rayleigh friction.  *result_instance = the instance holding the output (as the reference's
 * return value).  Sub-step 0, tmx_apply_dss(work), sub-step 1, tmx_apply_dss(update) equals tmx_h_step_after_subcycle
 * bit for bit when there is no Rayleigh friction (which the full call applies after the last DSS). */
int tmx_h_substep_after_subcycle_count(tmx_engine * e);
int tmx_h_substep_after_subcycle(tmx_engine * e, int i_initial, int i_update, int i_working, double dt, int i_substep,
	int * result_instance);

int tmx_v_step_implicit_terms_explicitly(tmx_engine * e, int i_initial, int i_update, double dt);

/* VerticalDynamics::FilterNegativeTracers(iDataUpdate) (src/atm/VerticalDynamics.h:123-126; VerticalDynamicsFEM.cpp:
 * 4286-4347): per stored column and tracer, clip negative densities and rescale the positive ones so that the
 * column mass is unchanged.  Called by TimestepSchemeStrang after its carry-over combination
 * (TimestepSchemeStrang.cpp:242, :480); a no-op without tracers. */
int tmx_v_filter_negative_tracers(tmx_engine * e, int instance);

/* ---- column physics as a WorkflowProcess on the device ------------------------ */

/* Per owned patch: latitude [na][nb] (GridPatch::GetLatitude) and the "surface pressure" [na][nb] that
 * HeldSuarezPhysics::Perform forms as PhysicalConstants::PressureFromRhoTheta(REdge[rho][i][j][0] * REdge[rhotheta][i][j][0])
 * (src/atm/HeldSuarezPhysics.cpp:99-101).  With Lorenz staggering nothing on this path updates those interface
 * slots except Grid::CopyData / ZeroData / LinearCombineData, which run over whole arrays and so change them by one
 * rounding per combination.  surface_pressure == NULL (recommended, what the adapter passes): the engine carries those two
 * entries per column ("surface slots", taken from the redge array of tmx_upload_state and written back by
 * tmx_download_state) through the same stage algebra and forms the pressure from them, bit for bit as the reference;
 * surface_pressure != NULL pins the value the caller computed (constant in time).  All patches use one mode. */
int tmx_set_patch_physics_inputs(tmx_engine * e, int patch, const double * latitude, const double * surface_pressure);
/* HeldSuarezPhysics::Perform (src/atm/HeldSuarezPhysics.cpp:60-301) on `instance` with dDeltaT = dt: backward-Euler
 * boundary-layer friction on U,V and the relaxation of rho*theta towards the Held-Suarez equilibrium temperature.
 * Keeping the state on the device across Model::Go's physics call removes the download + upload per step that a
 * host-side WorkflowProcess forces (src/atm/Model.cpp:477-481). */
int tmx_physics_held_suarez(tmx_engine * e, int instance, double dt);

/* Kessler warm-rain microphysics of the DCMIP2016 supercell (BASELINE config 4): KesslerPhysics::Perform
 * (test/dcmip2016/KesslerPhysics.cpp:83-285) around SUBROUTINE KESSLER (test/dcmip2016/interface/kessler.f90:64-185),
 * a WorkflowProcess like the Held-Suarez forcing (Model.cpp:470-481).  Needs n_tracers >= 3 with tracers 0,1,2 =
 * RhoQv, RhoQc, RhoQr (SupercellTest.cpp:271-275).
 * tmx_set_patch_level_heights: per owned patch z_levels [na][nb][L] = GridPatch::GetZLevels() (KesslerPhysics.cpp:141).
 * tmx_physics_kessler: Perform on `instance` with dDeltaT = dt; updates rho, rho*theta and the three tracer densities and
 *   adds precl * dt to the accumulated precipitation (UserData2D item 0 "PRECT", KesslerPhysics.cpp:234).
 * tmx_download_precipitation: the accumulated precipitation of a patch into prect [na][nb] (interior nodes written);
 *   reset != 0 zeroes the device accumulator afterwards. */
int tmx_set_patch_level_heights(tmx_engine * e, int patch, const double * z_levels);
int tmx_physics_kessler(tmx_engine * e, int instance, double dt);
int tmx_download_precipitation(tmx_engine * e, int patch, double * prect, int reset);

/* Wait for the engine's stream; reports a deferred TMX_ERR_SINGULAR from column solves. */
int tmx_sync(tmx_engine * e);

/* ---- multi-GPU (one engine per rank) -------------------------------------------------------- */

/* Number of HIP devices visible to this process (0 if none): lets the caller bind rank -> device before tmx_create
 * (cfg.device); two ranks of one communicator on the same device make tmx_comm_init fail with TMX_ERR_DEVICE. */
int tmx_device_count(void);

/* 128-byte RCCL unique id, created on rank 0 and broadcast by the host (the reference's
 * MPI_COMM_WORLD, src/atm/TempestInitialize.h:715-727). */
int tmx_comm_unique_id(unsigned char id[128]);
int tmx_comm_init(tmx_engine * e, const unsigned char id[128]);

/* Peer-to-peer halo transport, an alternative to the RCCL send/recv of tmx_comm_init for the exchange of
 * Connectivity.cpp:928-1120 / Grid.cpp:627-685: every rank keeps its ghost columns in a block that its neighbour ranks map
 * through HIP IPC, the gather kernel of a rank writes its boundary columns straight into the neighbours' blocks (posted
 * writes over xGMI, no send buffer, no collective), and an arrival counter per neighbour orders the two sides.
 *   tmx_halo_p2p_blob_bytes: size of a rank's blob (the same on every rank of a layout);
 *   tmx_halo_p2p_export:     allocate the block, write this rank's blob;
 *   tmx_halo_p2p_connect:    blobs = the n_ranks blobs in rank order (gathered by the caller with whatever transport it has:
 *                            the reference's MPI_Allgather, torch.distributed); maps the neighbours and switches the exchange over.
 * Ranks are separate processes (a process cannot map its own export); they may share a device.  Before tmx_destroy the caller
 * lets all ranks finish (a barrier): neighbours write into this rank's block.  A neighbour whose message does not arrive within
 * TMX_P2P_TIMEOUT_S (default 600 s; 0 = wait for ever, like MPI_Wait) makes the next tmx_sync return TMX_ERR_COMM instead of
 * hanging the device; the exchanges queued behind the failed one return at once.  The engine's state is then invalid.
 *   tmx_halo_p2p_reset:      recovery after TMX_ERR_COMM.  Every rank returns from tmx_sync, barrier, every rank calls this
 *                            (arrival counters and exchange count back to zero), barrier, upload the state again, go on with
 *                            either transport. */
int tmx_halo_p2p_blob_bytes(tmx_engine * e);
int tmx_halo_p2p_export(tmx_engine * e, unsigned char * blob);
int tmx_halo_p2p_connect(tmx_engine * e, const unsigned char * blobs);
int tmx_halo_p2p_reset(tmx_engine * e);
/* With both transports set up (tmx_comm_init and tmx_halo_p2p_connect): on = 1 peer-to-peer, on = 0 RCCL.  Every rank
 * switches at the same point of its program (bench.py times one against the other at start-up and keeps the faster). */
int tmx_halo_p2p_enable(tmx_engine * e, int on);

/* ---- introspection for tests / bench --------------------------------------------------------- */

enum {
	TMX_INFO_LOCAL_COLUMNS = 0,   /* stored columns (16 per element) on this rank        */
	TMX_INFO_UNIQUE_COLUMNS,      /* columns solved by the implicit step                */
	TMX_INFO_DSS_GROUPS,          /* co-located node groups touching this rank          */
	TMX_INFO_LOCAL_ELEMENTS,
	TMX_INFO_GHOST_COLUMNS,       /* remote copies received per exchange                */
	TMX_INFO_HBM_BYTES,           /* device bytes allocated by the engine on this rank   */
	TMX_INFO_METRIC_CLOSED_FORM,  /* 1: 3-D metric evaluated in registers, 0: streamed  */
	TMX_INFO_EARLY_TILES,         /* 64-column tiles launched before the exchange starts (0: stages are not split) */
	TMX_INFO_LATE_TILES,          /* tiles updated while the exchange is in flight      */
	TMX_INFO_COMM_RANKS,          /* ncclCommCount of the communicator made by tmx_comm_init (0: none)  */
	TMX_INFO_SEND_COLUMNS,        /* columns this rank sends per exchange (all peers)     */
	TMX_INFO_HALO_TRANSPORT,      /* 0 none (one rank), 1 RCCL send/recv, 2 peer-to-peer writes, 3 in-process loopback (tests) */
	TMX_INFO_UNIQUE_LAYOUT,       /* 1: the node-unique state layout is set up (tmx_step uses it where the program allows) */
	TMX_INFO_UNIQUE_INSTANCES,    /* data instances currently held in node-unique form                              */
	TMX_INFO_UNIQUE_CONVERSIONS,  /* conversions between the two forms so far (0 inside a run of tmx_step calls)      */
	TMX_INFO_PARTIAL_SLOTS,       /* seam-node copies that go through the partial slots (tile and patch edges)         */
	TMX_INFO_UNIQUE_DSS_GROUPS,   /* node groups the DSS kernel of that layout finishes                               */
	TMX_INFO_PREFIX_STAGES,       /* explicit stages so far whose combination started from a partial sum an earlier stage stored */
	TMX_INFO_EXPERIMENTS_BUILD,   /* 1: the experiments flavour of the library (archived experiments and cross-check kernels compiled in) */
	TMX_INFO_MIXED_STEPS,         /* steps so far whose explicit stages read an element-major instance copy by copy beside node-unique ones ("unique_mixed") */
	TMX_INFO_COLUMN_KERNEL        /* the column-solve kernel of the last launch: 0 one wavefront per column group, 1 two-wavefront pair, 2 lane group, -1 none yet */
};
long long tmx_info(tmx_engine * e, int what);

/* Options by name (integers passed as double).  They change HOW the engine computes, never what: results are bit-identical under
 * every setting, with the one exception "contraction_mfma", whose description says so.  The library reads NO environment variable
 * on its own -- a stray TMX_* in a job script cannot change a run.  The list (tmx_options_report prints it with the values in use):
 *   "unique_layout"      -1 default (= 1), 0 off, 1 on: hold the state once per patch node inside tmx_step (node-unique slabs: 9/16 of
 *                        the bytes, the DSS reduced to the nodes on tile and patch edges) where every kernel of the configuration
 *                        has that form -- nonhydrostatic set, implicit vertical dynamics, no tracers / uniform diffusion (a Rayleigh
 *                        layer, applied per stored copy, leaves the state element-major at the end of a step and the next step reads it
 *                        copy by copy: "unique_mixed"); replaces nothing of the reference interface (GridPatch data stay as they are at
 *                        the boundary).  Before tmx_finalize.
 *   "unique_tile_shape"  elements of a wavefront: 0 = the element-major order, 1 x 4 strips along beta that wrap around patch rows
 *                        (default: measured fastest), 1 = 2 x 2 blocks, 2 = strips that stay inside a patch row.  Before tmx_finalize.
 *   "unique_xcd_order"   1 (default): every XCD sweeps a contiguous range of tiles.  Before tmx_finalize.
 *   "unique_mixed"       1 (default): an element-major instance a step reads before overwriting it -- a freshly uploaded state, or the model state
 *                        behind an operation the reference applies per copy with per-copy inputs (Held-Suarez forcing: HeldSuarezPhysics.cpp:60-301
 *                        evaluates every copy of a seam node with that copy's own latitude, which differs in the last bit) -- is read that
 *                        way, copy by copy, by the explicit stages, and the step still runs on the node-unique layout (bit-identical: the
 *                        stages compute per copy and average as before).  0: the round-4 rule (check that the copies agree, convert, or run
 *                        the whole step element-major).
 *   "contraction_mfma"   1: the 4 x 4 contractions of the fused explicit kernel as v_mfma_f64_4x4x4 -- NOT bit-exact (1e-15 per call,
 *                        W 1.5e-10 after 100 steps at ne15 L30: outside the 1e-10 gate) and measured no faster.  Before tmx_finalize.
 *   "vt_explicit_walk", "vx_walk", "vite_walk"   fully explicit vertical mode: the tracer column update, V.StepExplicit's U,V update and the explicitly
 *                        evaluated implicit terms as column walks with sliding register windows: -1000 (default) the number of segments per column chosen
 *                        from the grid size, -n = n segments, 0 = the level-parallel kernels they superseded (bit-identical, tested).
 *   "step_graph", "share_copies", "split_stage_off", "exchange_overlap_off", "p2p_timeout_s", "metric_stored", "xcd_vertical",
 *   "vi_*" (column solve kernel selection), "vt_*", "kessler_column", "hvis_pull", "hvis_block", "tracer_lincomb_pass", "udv_separate",
 *   "vx_fused": kernel selection / A-B switches, each with a bit-identity test; "debug_skip_exchange": timing aid with WRONG results
 *   at rank boundaries (tools/rank_share_timing.py only).
 *   "h_walk"            explicit stage on the node-unique layout: -1000 (default) a wavefront walks a column segment with a sliding register
 *                        window (tmx_k_hwalk.hip), segments per column from the grid size; -n = n segments; 0 = the level-parallel kernels.
 *   "hv_walk"           1 (default): the hyperviscosity pass of the node-unique layout as a walk (-n: n segments per column); 0 = the level-parallel kernel.
 *   "h_walk_udiff"      uniform-diffusion configurations: the element-major walk applies the horizontal uniform diffusion (HorizontalDynamicsFEM.cpp:1817-1859)
 *                        to its results in registers (1) and V.StepExplicit's U,V part (VerticalDynamicsFEM.cpp:984-1105) behind it (2, default); 0 = kernels of their own.
 *   "unique_prefix"     1 (default): an explicit stage also stores the leading partial sum of a later stage's combination (ARS343: the
 *                        fourth stage reads 3 instances instead of 7); 0: every stage reads all its terms.  Same bits.
 *   "lu_fma"            band LU of the column solves (LAPACK::DGBSV, src/base/LinearAlgebra.cpp:156-202): 1 (default) = updates a - l u fused
 *                        (a reference linked to OpenBLAS, or to MKL on its FMA code paths), 0 = multiply and subtract rounded separately
 *                        (a BLAS without fused multiply-adds).  tmx_lu_flavour_from_dgbsv (below) asks the caller's own LAPACK.
 * Archived experiments and cross-check kernels -- "unique_blocks", "unique_tile_shape" = 3, 4, "hvis_pull", "hvis_block", "vi_split_kernels", "vi_pair" = 0,
 * "vi_carry", "vi_sparse", "vi_split_back", "vi_back_sub", "vx_fused", "vt_explicit_v1", "debug_skip_exchange" -- are compiled into the
 * experiments flavour of the library only (libtempest_mi355x_exp.so, `make -C tempestmodel_amd/csrc exp`; tmx_info(TMX_INFO_EXPERIMENTS_BUILD)):
 * the production library holds neither the kernels nor the switch, and answers TMX_ERR_UNSUPPORTED to any value but the default. */
int tmx_set_option(tmx_engine * e, const char * name, double value);
int tmx_get_option(tmx_engine * e, const char * name, double * value);
/* Which band LU does the CALLER'S LAPACK compute (replaces nothing in the reference: it makes the drop-in safe by default)?  `dgbsv` is the
 * Fortran-ABI routine the caller's Tempest is linked to (dgbsv_: what LAPACK::DGBSV calls, src/base/LinearAlgebra.h).  A 33 x 33 system of
 * the column solve's band width is solved by it and, on the host, by LAPACK's algorithm with the update a - l u fused and unfused; returns 1
 * (fused), 0 (multiply + subtract) and sets the engine's "lu_fma" option accordingly (e may be NULL: probe only), -1 if the routine's bits
 * are neither (option untouched, tmx_last_error says so).  adapter/TempestMI355X.h calls it when the engine is created. */
typedef void (*tmx_dgbsv_fn)(const int * n, const int * kl, const int * ku, const int * nrhs, double * ab, const int * ldab, int * ipiv,
	double * b, const int * ldb, int * info);
int tmx_lu_flavour_from_dgbsv(tmx_engine * e, tmx_dgbsv_fn dgbsv);
/* "name=value" lines of every option as in use (defaults included) and, if tmx_options_from_environment applied any, the variables it
 * took; returns the buffer length needed (buf may be NULL). */
int tmx_options_report(tmx_engine * e, char * buf, int cap);
/* Test / bench plumbing only: turn the historical TMX_* environment variables into options (right after tmx_create).  Says what
 * it applied in ONE line on stderr (silenced by TMX_QUIET) and in tmx_options_report; returns how many, or TMX_ERR_UNSUPPORTED (< 0) if a
 * variable names a value this build refuses -- a variable that would have changed the run and cannot is an error, not a silent default.
 * The adapter does not call it. */
int tmx_options_from_environment(tmx_engine * e);

/* Host-side DSS / exchange plan, for tests of the multi-rank logic (works on an engine created with
 * cfg.device = -2, "plan only": no HIP call is made and no kernel can be launched).
 * what = 0: send list, quadruples (patch, i, j, dest_rank) in wire order
 * what = 1: receive list, quadruples (patch, i, j, src_rank) in ghost-buffer order
 * what = 2: DSS groups, (n, col0..col3) with col >= NS meaning ghost NS + index
 * what = 3: (NS, ncol, n_unique, n_groups, n_cross_panel_groups)
 * what = 4: per group, index into the cross-panel matrices or -1
 * what = 5: per group, 2 bits per member: the member's pairing in the reference's averaging order (groups of four,
 *           members stored as [m0, alpha partner, beta partner, diagonal]: 0 {01|23}, 1 {02|13}, 2 {03|12}; cube
 *           corners: 0 = (x + next) + previous, 1 = (x + previous) + next)
 * Returns the number of ints written (needed, when out == NULL), -1 on error. */
int tmx_plan_get(tmx_engine * e, int what, int * out, int cap);
/* [n_cross][4][4][4] doubles: [member m][partner q] the 2x2 matrix rotating q's (U,V) into the frame of m's patch. */
int tmx_plan_get_matrices(tmx_engine * e, double * out, int cap);
/* Host-side tables of the node-unique layout ("unique_layout") for a tile shape, built on the spot (plan-only engines too):
 * what = 0 sizes (unique nodes, padded, thread columns, partial slots, groups left to the DSS kernel, of them rank-local, early tiles,
 * late tiles); 1 element-major column of a thread; 2 its node; 3 store code; 4 averaging recipe; 5 / 6 / 7 sources, destinations and
 * member counts of the remaining groups; 8 node of a slot; 9 slots sent to other ranks; 10 representative column of a node;
 * 11 node of an element-major column; 12 per-tile node bases and pitches.  Returns the ints written (needed, when out == NULL), -1 on error. */
int tmx_debug_unique_tables(tmx_engine * e, int tile_shape, int what, int * out, int cap);
/* Does the stepper program of a scheme have a node-unique form?  Returns the number of units the node-unique interpreter runs it in
 * (a unit = the operations it fuses into one stage / solve / pass), 0 if some operation has none, -1 for an unknown scheme.  No device. */
int tmx_debug_program_unique(int scheme, int first_step, int last_step);
/* Host logic of "unique_mixed": 1 if only the explicit stages of the scheme's program read data instance `instance` before the program overwrites
 * it (so that it may stay element-major, read copy by copy, while the step runs on the node-unique layout), 0 if not, -1 unknown scheme. */
int tmx_debug_program_mixed(int scheme, int first_step, int last_step, int instance);
/* Host logic of the Rayleigh layer on the node-unique layout: 1 if only copies of StepAfterSubCycle's result follow it in the scheme's program
 * (the relaxed, element-major result is never read by a node-unique operation), 0 if the program stays element-major, -1 unknown scheme. */
int tmx_debug_program_rayleigh(int scheme, int first_step, int last_step);
/* Host logic for the CPU tests: the explicit stages of a scheme's program that can start from a partial sum an earlier stage
 * stores (node-unique layout, option "unique_prefix").  Returns their number; pairs[2 k], pairs[2 k + 1] (k < cap): operation
 * index of the producing and of the consuming stage; terms[k]: source instances the consuming stage no longer reads. */
int tmx_debug_program_prefix(int scheme, int first_step, int last_step, int * pairs, int * terms, int cap);
/* Diagnostic builds of the library (-DTMX_H_TIMING) only, zeros otherwise: shader cycles per wavefront of the fused explicit kernel by
 * phase, out[16 instantiations (combination terms)][7 phases + wavefront count] (tools/h_timing.py); the read clears the counters. */
int tmx_debug_h_timing(tmx_engine * e, unsigned long long * out);
/* The same for the column-segment walk that runs the explicit stage on the node-unique layout (option "h_walk"; tmx_k_hwalk.hip):
 * out[16][8], rows = combination terms, columns = prologue, loads issued, pointwise + Exner, contractions, updates, wait for the
 * combination's terms, seam exchange + stores, wavefront count. */
int tmx_debug_h_walk_timing(tmx_engine * e, unsigned long long * out);

/* Test hooks (tests/ only): n engines = ranks 0..n-1 of one grid inside ONE process on one device; the
 * RCCL transport is replaced by device-to-device copies in wire order, everything else (packing, ghost
 * indexing, DSS with remote members) is the production path. */
int tmx_debug_dss_loopback(tmx_engine ** engines, int n, int instance);
/* Same transport replacement for WHOLE production calls (tmx_step, tmx_h_step_after_subcycle, tmx_apply_dss ...):
 * registers the n engines as one loopback group; afterwards every exchange of a member packs, waits at a barrier for
 * the other members, copies its peers' send segments into its ghost buffer and waits again.  The caller drives each
 * member from its own host thread (one thread per engine, as one MPI rank is one thread in the reference).
 * n = 0 dissolves the group of engines[0]. */
int tmx_debug_loopback_group(tmx_engine ** engines, int n);
/* Grouped ncclSend/ncclRecv to self through the engine's communicator and stream. */
int tmx_debug_comm_selftest(tmx_engine * e);

/* Pivot statistics of the column solve (bench.py): enable != 0 starts counting, 0 stops; out (may be NULL) receives
 * {pivot steps in which all 64 columns of a wavefront chose the same pivot row, all pivot steps} counted so far. */
int tmx_debug_pivot_stats(tmx_engine * e, int enable, unsigned long long out[2]);

/* Host logic of the stepper programs, no device needed (tests/): the whole-instance copies a time scheme's program performs per
 * step and the ones it replaces by slot sharing (a CopyData in front of a V.StepImplicit that does nothing reads its source's
 * slot until it is overwritten: fully explicit vertical mode VerticalDynamicsFEM.cpp:1239-1242, VerticalDynamicsStub; with
 * implicit vertical dynamics the same for the U,V slabs, which the column solve does not touch), decided
 * exactly as tmx_step decides them.  mode 0: implicit vertical dynamics, 1: fully explicit, 2: shallow water; `steps`
 * consecutive steps, counts of the last one; share = 0 switches the sharing off (TMX_SHARE_COPIES=0).  The run also carries
 * symbolic contents (value ids per instance and slot, U,V and the rest apart) through the reference's semantics and through the
 * executed decisions, and returns -1000 - n when n reads through the instance maps would not have found the reference's value
 * (0: sound); share = 2 is the check's negative control: every share accepted and no copy-on-write. */
int tmx_debug_program_copies(int scheme, int mode, int steps, int share, int * copies, int * shared);

/* Per-kernel device time accumulated with HIP events on the engine's stream since the last reset.
 * kernel ids: TMX_K_* below.  Profiling is off by default (no events recorded). */
enum {
	TMX_K_H_EXPLICIT = 0, TMX_K_V_EXPLICIT, TMX_K_VI_ASSEMBLE, TMX_K_VI_SOLVE, TMX_K_DSS,
	TMX_K_HYPERVIS, TMX_K_LINCOMB, TMX_K_EXCHANGE, TMX_K_COUNT
};
int tmx_profile_enable(tmx_engine * e, int on);
int tmx_profile_get(tmx_engine * e, int kernel, double * total_ms, long long * launches);
int tmx_profile_reset(tmx_engine * e);

#ifdef __cplusplus
}
#endif
#endif /* TEMPEST_MI355X_H */
