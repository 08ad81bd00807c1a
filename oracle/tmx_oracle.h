/* oracle/tmx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference hot path (paullric/tempestmodel), used as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It is never
 * linked into, imported by, or called from the product library.
 *
 * Parity status: PINNED -- every routine is checked against dumps of the real reference
 * (oracle/_ref, built by oracle/Makefile from /root/reference) in tests/test_oracle_vs_reference.py
 * via the committed fixtures under tests/golden/.
 *
 * Data layout is the reference's own: per patch, per instance,
 *   node [5][na][nb][L]   and  redge [5][na][nb][L+1]      (k fastest, one-node halo ring,
 * duplicated element-edge nodes; src/atm/GridPatch.cpp:341-356, src/base/DataArray4D.h:123-231).
 */
#ifndef TMX_ORACLE_H
#define TMX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
	ORC_G_JAC2D = 0,   /* [na][nb]        */
	ORC_G_CM2A,        /* [na][nb][2]     */
	ORC_G_CM2B,        /* [na][nb][2]     */
	ORC_G_JAC,         /* [na][nb][L]     */
	ORC_G_JAC_E,       /* [na][nb][L+1]   */
	ORC_G_CMA,         /* [na][nb][L][3]  */
	ORC_G_CMB,
	ORC_G_CMXI,
	ORC_G_CMA_E,       /* [na][nb][L+1][3] */
	ORC_G_CMB_E,
	ORC_G_CMXI_E,
	ORC_G_DR_N,        /* [na][nb][L][3]  */
	ORC_G_DR_E,        /* [na][nb][L+1][3] */
	ORC_G_CORIOLIS,    /* [na][nb]        */
	ORC_G_TOPO,        /* [na][nb]  (shallow water only) */
	ORC_G_COUNT
};

enum {
	ORC_OP_INTERP_N2E = 0,
	ORC_OP_INTERP_E2N,
	ORC_OP_DIFF_N2N,
	ORC_OP_DIFF_N2E,
	ORC_OP_DIFF_E2N,
	ORC_OP_DIFF_E2E,
	ORC_OP_DIFFDIFF_N2N,
	ORC_OP_DIFFDIFF_E2E,
	ORC_OP_PENALTY_L,
	ORC_OP_PENALTY_R,
	ORC_OP_COUNT
};

typedef struct {
	int panel, na, nb, nea, neb;
	int neighbor_panels[8];            /* Right, Top, Left, Bottom, TR, TL, BL, BR; -1 = none */
	const double * g[ORC_G_COUNT];
	int nhalo;
	const int * hi; const int * hj;    /* ring node local indices                    */
	const int * hsp;                   /* source patch (-1 = none)                   */
	const int * hsi; const int * hsj;  /* source local indices                       */
	const int * hspanel;               /* source panel                               */
	const double * htrans;             /* [nhalo][2][2] covector transform           */
	/* Rayleigh friction (all NULL = none): strength [na][nb][L] / [na][nb][L+1] (GridPatch::GetRayleighStrength)
	 * and the reference state [5][na][nb][L] / [5][na][nb][L+1] (GridPatch::GetReferenceState)                */
	const double * ray_node; const double * ray_redge;
	const double * ref_node; const double * ref_redge;
	const double * ref_tracer;         /* [ntracers][na][nb][L] GridPatch::GetReferenceTracers (uniform diffusion of tracers; may be NULL) */
	const double * area_node;          /* [na][nb][L] GridPatch::GetElementAreaNode (tracer filters; NULL without tracers) */
	const double * lat;                /* [na][nb] GridPatch::GetLatitude (column physics; may be NULL) */
	const double * surf_p;             /* [na][nb] the "surface pressure" HeldSuarezPhysics::Perform forms from the interface slots (may be NULL) */
	double delta_a, delta_b;           /* GridPatchGLL::GetElementDeltaA / B of this patch; 0 = orc_grid.delta_a */
} orc_patch;

typedef struct {
	int np, L, npatch;
	double delta_a, ref_length;
	double grav, Rd, cp, p0;
	double dx[16], stiff[16];          /* [s][i] row-major, np = 4                   */
	const double * opc[ORC_OP_COUNT];  /* dense [nout][nin]                          */
	const int * opb[ORC_OP_COUNT];
	const int * ope[ORC_OP_COUNT];
	int opnin[ORC_OP_COUNT];
	int opnout[ORC_OP_COUNT];
	double nu_scalar, nu_div, nu_vort;
	const orc_patch * patches;
	int shallow_water;                 /* 1: ShallowWaterEquations (U,V,H; H kept in slot 2, one level)  */
	int ntracers;                      /* EquationSet::GetTracers(); tracer densities live on levels      */
	int hypervis_order;                /* 4 (two-pass hyperviscosity), 2 (viscosity) or 0 (none); HorizontalDynamicsFEM ctor */
	int fully_explicit;                /* VerticalDynamicsFEM ctor fFullyExplicit (--explicitvertical): StepExplicit applies -dt F, StepImplicit does nothing */
	int uniform_diffusion;             /* Grid::HasUniformDiffusion(); supported with fully_explicit only */
	double udiff_scalar, udiff_vector; /* Grid::GetScalarUniformDiffusionCoeff / GetVectorUniformDiffusionCoeff [m^2/s] */
	double ztop;                       /* Grid::GetZtop() */
} orc_grid;

/* state: node[inst*npatch + p], redge[inst*npatch + p] */
typedef struct {
	int ninst;
	double ** node;
	double ** redge;
	double ** tracer;                  /* tracer[inst*npatch + p] = [ntracers][na][nb][L] (GridPatch::GetDataTracers), NULL if none */
} orc_state;

/* Sample points of Grid::ReduceInterpolate (src/atm/Grid.cpp:867-992), with the per-point quantities the caller
 * derives from the reference's own functions: element origin node (iA, iB of GridPatchCSGLL.cpp:1588-1611), Lagrange
 * coefficients (PolynomialInterp::LagrangianPolynomialCoeffs), the 2x2 matrix of CubedSphereTrans::
 * CoVecTransRLLFromABP, and the dense coefficients of the two LinearColumnInterpFEM operators. */
typedef struct {
	int npts, nreta;
	const int * patch; const int * node_a; const int * node_b;
	const double * coeff_a; const double * coeff_b;     /* [npts][4] */
	const double * rll;                                 /* [npts][4] row-major (lon; lat) x (alpha, beta) */
	const double * op_levels;                           /* [nreta][L]   */
	const double * op_interfaces;                       /* [nreta][L+1] */
} orc_interp;

/* GridPatchCSGLL::InterpolateData (GridPatchCSGLL.cpp:1365-1780) summed over patches: out[5][nreta][npts] (State;
 * only_at: 0 all variables, 1 levels only, 2 interfaces only -- skipped rows stay zero) / out[ntracers][nreta][npts] */
void orc_interpolate_state(const orc_grid * g, const orc_state * s, int ix, const orc_interp * q, int only_at,
	int include_reference_state, int convert_to_primitive, double earth_radius, double * out);
void orc_interpolate_tracers(const orc_grid * g, const orc_state * s, int ix, const orc_interp * q, double * out);

void orc_copy(const orc_grid * g, orc_state * s, int src, int dst);
void orc_zero(const orc_grid * g, orc_state * s, int ix);
void orc_linear_combine(const orc_grid * g, orc_state * s, const double * coeff, int ncoeff, int dst);

void orc_h_step_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt);
void orc_v_step_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt);
int  orc_v_step_implicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt);
void orc_apply_dss(const orc_grid * g, orc_state * s, int ix);
void orc_h_step_after_subcycle(const orc_grid * g, orc_state * s, int iinit, int iupd, int iwork, double dt);
int  orc_step_ars343(const orc_grid * g, orc_state * s, double dt);
void orc_v_step_implicit_terms_explicitly(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt);
/* scheme: 0 ARS343, 1 ARS232, 2 ARS222, 3 ARS443, 4 ARK232, 5 Strang (KGU35), 6-9 Strang with ForwardEuler / RungeKutta4 /
 * RungeKuttaSSP3 / RungeKuttaSSPRK53 (TimestepSchemeStrang.h:39-45) */
int  orc_step(const orc_grid * g, orc_state * s, int scheme, int first, int last, double dt);
void orc_set_strang_off_centering(double oc);     /* TimestepSchemeStrang ctor dOffCentering, default 0 */

/* HeldSuarezPhysics::Perform (src/atm/HeldSuarezPhysics.cpp:60-301) on instance ix */
void orc_held_suarez(const orc_grid * g, orc_state * s, int ix, double dt);

/* SURVEY 8(f) item 1, second half: the Kessler microphysics of the DCMIP2016 supercell.  orc_kessler_column =
 * SUBROUTINE KESSLER (test/dcmip2016/interface/kessler.f90:64-185, mixed single/double arithmetic as compiled),
 * orc_kessler = KesslerPhysics::Perform (test/dcmip2016/KesslerPhysics.cpp:83-285) on instance ix with tracers
 * RhoQv, RhoQc, RhoQr; zlev[p] = [na][nb][L] level heights, prect[p] = [na][nb] accumulated precipitation or NULL. */
void orc_kessler_column(double * theta, double * qv, double * qc, double * qr, const double * rho, const double * pk,
	double dt, const double * z, int nz, double * precl_out);
void orc_kessler(const orc_grid * g, orc_state * s, int ix, double dt, const double * const * zlev, double * const * prect);

/* banded LU with partial pivoting, LAPACK dgbsv semantics (1 rhs); ab is [n][ldab] column-major band
 * storage with ldab = 2*kl+ku+1, entry (i,j) at ab[j*ldab + kl+ku+i-j].  Returns info. */
int orc_dgbsv(int n, int kl, int ku, double * ab, int ldab, int * ipiv, double * b);
/* 1 (default): the three update statements of the band LU are fused multiply-adds, as in MKL / OpenBLAS; 0: mul + sub */
void orc_set_lu_fma(int on);
int orc_get_lu_fma(void);

/* single-column implicit pieces, exposed for unit tests: builds F and the band Jacobian of one column */
void orc_column_f_jacobian(const orc_grid * g, const orc_patch * p, const double * node, const double * redge,
	int iA, int iB, double dt, double * x0, double * F, double * AB);

#ifdef __cplusplus
}
#endif
#endif
