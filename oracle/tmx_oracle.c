/* oracle/tmx_oracle.c -- TEST INFRASTRUCTURE ONLY (see tmx_oracle.h).
 *
 * CPU restatement, in plain C and in the reference's own data layout, of the hot path of
 * paullric/tempestmodel for its default configuration (FORMULATION_RHOTHETA_PI,
 * USE_DIRECTSOLVE + USE_JACOBIAN_DIAGONAL, Lorenz staggering, FE vertical order 1, upwinding on
 * all variables, Clark-form implicit W advection, 4th-order hyperviscosity; src/atm/Defines.h:23-80,
 * src/atm/VerticalDynamicsFEM.cpp:33-47).  Each function cites the reference lines it follows.
 * No reference source text is reproduced; loops are restated over flat arrays.
 */
#include "tmx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { UIx = 0, VIx = 1, PIx = 2, WIx = 3, RIx = 4 };

#define NP 4

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */

static size_t node_size(const orc_grid * g, const orc_patch * p) { return (size_t)5 * p->na * p->nb * g->L; }
static size_t redge_size(const orc_grid * g, const orc_patch * p) { return (size_t)5 * p->na * p->nb * (g->L + 1); }
static size_t tracer_size(const orc_grid * g, const orc_patch * p) { return (size_t)g->ntracers * p->na * p->nb * g->L; }
static int has_tracers(const orc_grid * g, const orc_state * s) { return g->ntracers > 0 && s->tracer != NULL; }

/* banded mat-vec row: out[k] = sum_{l in [b,e)} C[k][l] in[l]  (LinearColumnOperator.h:82-101) */
static double op_row(const orc_grid * g, int op, const double * in, int k) {
	const int nin = g->opnin[op];
	const double * c = g->opc[op] + (size_t)k * nin;
	double out = 0.0;
	for (int l = g->opb[op][k]; l < g->ope[op][k]; l++) out += c[l] * in[l];
	return out;
}

static void op_apply(const orc_grid * g, int op, const double * in, double * out) {
	for (int k = 0; k < g->opnout[op]; k++) out[k] = op_row(g, op, in, k);
}

/* Exner pressure from rho*theta (PhysicalConstants.h:397-399) */
static double exner_from_rhotheta(const orc_grid * g, double rt) {
	return g->cp * exp(g->Rd / (g->cp - g->Rd) * log(g->Rd / g->p0 * rt));
}

/* ------------------------------------------------------------------------------------------ */
/* stage algebra: Grid::CopyData / ZeroData / LinearCombineData  (GridPatch.cpp:1402-1553)    */

void orc_copy(const orc_grid * g, orc_state * s, int src, int dst) {
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		memcpy(s->node[dst * g->npatch + p], s->node[src * g->npatch + p], node_size(g, P) * sizeof(double));
		memcpy(s->redge[dst * g->npatch + p], s->redge[src * g->npatch + p], redge_size(g, P) * sizeof(double));
		if (has_tracers(g, s)) memcpy(s->tracer[dst * g->npatch + p], s->tracer[src * g->npatch + p], tracer_size(g, P) * sizeof(double));
	}
}

void orc_zero(const orc_grid * g, orc_state * s, int ix) {
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		memset(s->node[ix * g->npatch + p], 0, node_size(g, P) * sizeof(double));
		memset(s->redge[ix * g->npatch + p], 0, redge_size(g, P) * sizeof(double));
		if (has_tracers(g, s)) memset(s->tracer[ix * g->npatch + p], 0, tracer_size(g, P) * sizeof(double));
	}
}

void orc_linear_combine(const orc_grid * g, orc_state * s, const double * coeff, int ncoeff, int dst) {
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		for (int loc = 0; loc < (has_tracers(g, s) ? 3 : 2); loc++) {
			size_t n = (loc == 2) ? tracer_size(g, P) : (loc ? redge_size(g, P) : node_size(g, P));
			double * d = (loc == 2) ? s->tracer[dst * g->npatch + p] : (loc ? s->redge[dst * g->npatch + p] : s->node[dst * g->npatch + p]);
			/* premultiply the destination by its own coefficient (GridPatch.cpp:1461-1468) */
			if (coeff[dst] == 0.0) memset(d, 0, n * sizeof(double));
			else for (size_t i = 0; i < n; i++) d[i] *= coeff[dst];
			for (int m = 0; m < ncoeff; m++) {
				if (m == dst || coeff[m] == 0.0) continue;
				const double * a = (loc == 2) ? s->tracer[m * g->npatch + p] : (loc ? s->redge[m * g->npatch + p] : s->node[m * g->npatch + p]);
				const double c = coeff[m];
				for (size_t i = 0; i < n; i++) d[i] += c * a[i];
			}
		}
	}
}

/* ------------------------------------------------------------------------------------------ */
/* H explicit: HorizontalDynamicsFEM::StepNonhydrostaticPrimitive                              */
/* (HorizontalDynamicsFEM.cpp:701-1783)                                                        */

static void sw_step_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt);

/* HorizontalDynamicsFEM::FilterNegativeTracers (HorizontalDynamicsFEM.cpp:213-317, POSITIVE_DEFINITE_FILTER_TRACERS,
 * Defines.h:80): per element, level and tracer clip negatives to zero and rescale the rest to keep the element mass */
static void filter_negative_tracers_h(const orc_grid * g, orc_state * s, int iupd) {
	if (!has_tracers(g, s)) return;
	const int L = g->L;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		double * t = s->tracer[iupd * g->npatch + p];
		for (int a = 0; a < P->nea; a++)
		for (int b = 0; b < P->neb; b++)
		for (int c = 0; c < g->ntracers; c++)
		for (int k = 0; k < L; k++) {
			double total = 0.0, nonneg = 0.0;
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = a * NP + i + 1, iB = b * NP + j + 1;
				const double q = t[(((size_t)c * na + iA) * nb + iB) * L + k];
				const double pm = q * P->area_node[((size_t)iA * nb + iB) * L + k];
				total += pm;
				if (q >= 0.0) nonneg += pm;
			}
			const double r = total / nonneg;
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = a * NP + i + 1, iB = b * NP + j + 1;
				double * q = &t[(((size_t)c * na + iA) * nb + iB) * L + k];
				if (*q > 0.0) *q *= r; else *q = 0.0;
			}
		}
	}
}

/* element spacing of a patch: GridPatchGLL::GetElementDeltaA / B (GridPatchGLL.cpp:67-75: a difference of two edge
 * coordinates of the patch, so not exactly pi / (2 ne) and not the same on every patch); 0 in orc_patch = the grid's delta_a */
static inline double patch_da(const orc_grid * g, const orc_patch * P) { return (P->delta_a > 0.0) ? P->delta_a : g->delta_a; }
static inline double patch_db(const orc_grid * g, const orc_patch * P) { return (P->delta_b > 0.0) ? P->delta_b : g->delta_a; }

static void scalar_hyperdiff_c(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nu, int scale_locally, int comp, int remove_ref);
static void vector_hyperdiff_s(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nud, double nuv, int scale_locally, int from_ref);

void orc_h_step_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt) {
	if (g->shallow_water) { sw_step_explicit(g, s, iinit, iupd, dt); return; }
	const int L = g->L;
	const double * D = g->dx;      /* D[s*4+i]  = dDxBasis1D(s,i)   */
	const double * S = g->stiff;   /* S[i*4+s]  = dStiffness1D(i,s) */

	double * aux = (double *)malloc(sizeof(double) * 9 * NP * NP * L);
	double * fa_r = (double *)malloc(sizeof(double) * 4 * NP * NP * L);
	double * colbuf = (double *)malloc(sizeof(double) * (L + 1));
#define AUX(c,i,j,k) aux[(((size_t)(c) * NP + (i)) * NP + (j)) * L + (k)]
#define FLX(c,i,j,k) fa_r[(((size_t)(c) * NP + (i)) * NP + (j)) * L + (k)]
	enum { ConUa = 0, ConUb, ConUx, CovUx, KE, UCZA, UCZB, UCZX, EXN };
	enum { FAR = 0, FBR, FAP, FBP };

	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const double inv_da = 1.0 / patch_da(g, P), inv_db = 1.0 / patch_db(g, P);
		const int na = P->na, nb = P->nb;
		double * nin = s->node[iinit * g->npatch + p];
		double * ein = s->redge[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		double * eup = s->redge[iupd * g->npatch + p];
#define NI(c,i,j,k) nin[(((size_t)(c) * na + (i)) * nb + (j)) * L + (k)]
#define EI(c,i,j,k) ein[(((size_t)(c) * na + (i)) * nb + (j)) * (L + 1) + (k)]
#define NU(c,i,j,k) nup[(((size_t)(c) * na + (i)) * nb + (j)) * L + (k)]
#define EU(c,i,j,k) eup[(((size_t)(c) * na + (i)) * nb + (j)) * (L + 1) + (k)]
		const double * J2 = P->g[ORC_G_JAC2D];
		const double * Jn = P->g[ORC_G_JAC];
		const double * CA = P->g[ORC_G_CMA];
		const double * CB = P->g[ORC_G_CMB];
		const double * CX = P->g[ORC_G_CMXI];
		const double * CXe = P->g[ORC_G_CMXI_E];
		const double * DR = P->g[ORC_G_DR_N];
		const double * F = P->g[ORC_G_CORIOLIS];
#define M3(A,i,j,k) A[((size_t)(i) * nb + (j)) * L + (k)]
#define M4(A,i,j,k,c) A[(((size_t)(i) * nb + (j)) * L + (k)) * 3 + (c)]
#define M4E(A,i,j,k,c) A[(((size_t)(i) * nb + (j)) * (L + 1) + (k)) * 3 + (c)]

		/* prologue: W edge->node, U,V node->edge on the INPUT instance, interior nodes only
		 * (HorizontalDynamicsFEM.cpp:817-831, GridPatchGLL.cpp:80-143) */
		for (int i = 1; i < na - 1; i++)
		for (int j = 1; j < nb - 1; j++) {
			op_apply(g, ORC_OP_INTERP_E2N, &EI(WIx, i, j, 0), &NI(WIx, i, j, 0));
			op_apply(g, ORC_OP_INTERP_N2E, &NI(UIx, i, j, 0), &EI(UIx, i, j, 0));
			op_apply(g, ORC_OP_INTERP_N2E, &NI(VIx, i, j, 0), &EI(VIx, i, j, 0));
		}

		for (int a = 0; a < P->nea; a++)
		for (int b = 0; b < P->neb; b++) {
			const int eA = a * NP + 1, eB = b * NP + 1;

			/* (i) contravariant velocities, kinetic energy, Exner pressure (:876-963) */
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++)
			for (int k = 0; k < L; k++) {
				const int iA = eA + i, iB = eB + j;
				const double ua = NI(UIx, iA, iB, k), ub = NI(VIx, iA, iB, k), ux = NI(WIx, iA, iB, k);
				const double m0 = M4(CA, iA, iB, k, 0), m1 = M4(CA, iA, iB, k, 1), m2 = M4(CA, iA, iB, k, 2);
				const double m3 = M4(CB, iA, iB, k, 1), m4 = M4(CB, iA, iB, k, 2), m5 = M4(CX, iA, iB, k, 2);
				AUX(CovUx, i, j, k) = ux;
				AUX(ConUa, i, j, k) = m0 * ua + m1 * ub + m2 * ux;
				AUX(ConUb, i, j, k) = m1 * ua + m3 * ub + m4 * ux;
				AUX(ConUx, i, j, k) = m2 * ua + m4 * ub + m5 * ux;
				AUX(KE, i, j, k) = 0.5 * (AUX(ConUa, i, j, k) * ua + AUX(ConUb, i, j, k) * ub + AUX(ConUx, i, j, k) * ux);
				AUX(EXN, i, j, k) = exner_from_rhotheta(g, NI(PIx, iA, iB, k));
			}

			/* (ii) u x zeta (:966-1039) */
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++)
			for (int k = 0; k < L; k++) {
				const int iA = eA + i, iB = eB + j;
				const double dxua = op_row(g, ORC_OP_DIFF_N2N, &NI(UIx, iA, iB, 0), k);
				const double dxub = op_row(g, ORC_OP_DIFF_N2N, &NI(VIx, iA, iB, 0), k);
				double daub = 0.0, daux = 0.0, dbua = 0.0, dbux = 0.0;
				for (int q = 0; q < NP; q++) {
					daub += NI(VIx, eA + q, iB, k) * D[q * NP + i];
					daux += AUX(CovUx, q, j, k) * D[q * NP + i];
					dbua += NI(UIx, iA, eB + q, k) * D[q * NP + j];
					dbux += AUX(CovUx, i, q, k) * D[q * NP + j];
				}
				daub *= inv_da; daux *= inv_da; dbua *= inv_db; dbux *= inv_db;
				const double cua = AUX(ConUa, i, j, k), cub = AUX(ConUb, i, j, k), cux = AUX(ConUx, i, j, k);
				const double jza = dbux - dxub, jzb = dxua - daux, jzx = daub - dbua;
				AUX(UCZA, i, j, k) = cub * jzx - cux * jzb;
				AUX(UCZB, i, j, k) = cux * jza - cua * jzx;
				AUX(UCZX, i, j, k) = -cua * daux - cub * dbux;
			}

			/* (iii) fluxes (:1042-1100) */
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++)
			for (int k = 0; k < L; k++) {
				const int iA = eA + i, iB = eB + j;
				const double fa = M3(Jn, iA, iB, k) * AUX(ConUa, i, j, k);
				const double fb = M3(Jn, iA, iB, k) * AUX(ConUb, i, j, k);
				FLX(FAR, i, j, k) = fa * NI(RIx, iA, iB, k);
				FLX(FBR, i, j, k) = fb * NI(RIx, iA, iB, k);
				FLX(FAP, i, j, k) = fa * NI(PIx, iA, iB, k);
				FLX(FBP, i, j, k) = fb * NI(PIx, iA, iB, k);
			}

			/* (iv) pointwise update on levels (:1183-1421) */
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++)
			for (int k = 0; k < L; k++) {
				const int iA = eA + i, iB = eB + j;
				const double invJ = 1.0 / M3(Jn, iA, iB, k);
				const double cua = AUX(ConUa, i, j, k), cub = AUX(ConUb, i, j, k);
				double dake = 0, dbke = 0, dap = 0, dbp = 0, darf = 0, dapf = 0, dbrf = 0, dbpf = 0;
				for (int q = 0; q < NP; q++) {
					darf -= FLX(FAR, q, j, k) * S[i * NP + q];
					dapf -= FLX(FAP, q, j, k) * S[i * NP + q];
					dap += AUX(EXN, q, j, k) * D[q * NP + i];
					dake += AUX(KE, q, j, k) * D[q * NP + i];
				}
				for (int q = 0; q < NP; q++) {
					dbrf -= FLX(FBR, i, q, k) * S[j * NP + q];
					dbpf -= FLX(FBP, i, q, k) * S[j * NP + q];
					dbp += AUX(EXN, i, q, k) * D[q * NP + j];
					dbke += AUX(KE, i, q, k) * D[q * NP + j];
				}
				darf *= inv_da; dbrf *= inv_db; dapf *= inv_da; dbpf *= inv_db;
				dap *= inv_da; dbp *= inv_db; dake *= inv_da; dbke *= inv_db;
				double upa = 0.0, upb = 0.0;
				upa += AUX(UCZA, i, j, k);
				upb += AUX(UCZB, i, j, k);
				upa += F[iA * nb + iB] * J2[iA * nb + iB] * cub;
				upb -= F[iA * nb + iB] * J2[iA * nb + iB] * cua;
				const double pgfa = dap * NI(PIx, iA, iB, k) / NI(RIx, iA, iB, k);
				const double pgfb = dbp * NI(PIx, iA, iB, k) / NI(RIx, iA, iB, k);
				const double daphi = g->grav * M4(DR, iA, iB, k, 0);
				const double dbphi = g->grav * M4(DR, iA, iB, k, 1);
				upa -= (pgfa + dake + daphi);
				upb -= (pgfb + dbke + dbphi);
				NU(UIx, iA, iB, k) += dt * upa;
				NU(VIx, iA, iB, k) += dt * upb;
				NU(RIx, iA, iB, k) -= dt * invJ * (darf + dbrf);
				NU(PIx, iA, iB, k) -= dt * invJ * (dapf + dbpf);
			}

			/* tracers: flux form with the same mass-flux base J u^a, J u^b (:1090-1100, :1531-1553) */
			if (has_tracers(g, s)) {
				const double * tin = s->tracer[iinit * g->npatch + p];
				double * tup = s->tracer[iupd * g->npatch + p];
				double taf[NP][NP], tbf[NP][NP];
				for (int c = 0; c < g->ntracers; c++)
				for (int k = 0; k < L; k++) {
					for (int i = 0; i < NP; i++)
					for (int j = 0; j < NP; j++) {
						const int iA = eA + i, iB = eB + j;
						const double q = tin[(((size_t)c * na + iA) * nb + iB) * L + k];
						taf[i][j] = M3(Jn, iA, iB, k) * AUX(ConUa, i, j, k) * q;
						tbf[i][j] = M3(Jn, iA, iB, k) * AUX(ConUb, i, j, k) * q;
						if (g->uniform_diffusion) {
							/* uniform diffusion of the tracer mixing ratio (:1105-1148) */
							double daq = 0.0, dbq = 0.0;
							for (int m = 0; m < NP; m++) {
								daq += tin[(((size_t)c * na + eA + m) * nb + iB) * L + k] / NI(RIx, eA + m, iB, k) * D[m * NP + i];
								dbq += tin[(((size_t)c * na + iA) * nb + eB + m) * L + k] / NI(RIx, iA, eB + m, k) * D[m * NP + j];
							}
							daq *= inv_da; dbq *= inv_db;
							const double cdaq = M4(CA, iA, iB, k, 0) * daq + M4(CA, iA, iB, k, 1) * dbq;
							const double cdbq = M4(CA, iA, iB, k, 1) * daq + M4(CB, iA, iB, k, 1) * dbq;
							taf[i][j] -= g->udiff_scalar * M3(Jn, iA, iB, k) * NI(RIx, iA, iB, k) * cdaq;
							tbf[i][j] -= g->udiff_scalar * M3(Jn, iA, iB, k) * NI(RIx, iA, iB, k) * cdbq;
						}
					}
					for (int i = 0; i < NP; i++)
					for (int j = 0; j < NP; j++) {
						const int iA = eA + i, iB = eB + j;
						const double invJ = 1.0 / M3(Jn, iA, iB, k);
						double da = 0.0, db = 0.0;
						for (int q = 0; q < NP; q++) {
							da -= taf[q][j] * S[i * NP + q];
							db -= tbf[i][q] * S[j * NP + q];
						}
						da *= inv_da; db *= inv_db;
						tup[(((size_t)c * na + iA) * nb + iB) * L + k] -= dt * invJ * (da + db);
					}
				}
			}

			/* (v) W on interfaces (:1612-1659) */
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				const double u0 = op_row(g, ORC_OP_INTERP_N2E, &NU(UIx, iA, iB, 0), 0);
				const double v0 = op_row(g, ORC_OP_INTERP_N2E, &NU(VIx, iA, iB, 0), 0);
				EU(WIx, iA, iB, 0) = -(M4E(CXe, iA, iB, 0, 0) * u0 + M4E(CXe, iA, iB, 0, 1) * v0) / M4E(CXe, iA, iB, 0, 2);
				for (int k = 0; k < L; k++) colbuf[k] = AUX(UCZX, i, j, k);
				for (int k = 1; k < L; k++) {
					EU(WIx, iA, iB, k) += dt * op_row(g, ORC_OP_INTERP_N2E, colbuf, k);
				}
			}
		}
	}
	free(aux); free(fa_r); free(colbuf);
#undef AUX
#undef FLX
	if (g->uniform_diffusion) {
		/* HorizontalDynamicsFEM::StepExplicit, :1817-1859: uniform diffusion of (state - reference state) */
		vector_hyperdiff_s(g, s, iinit, iupd, dt, -g->udiff_vector, -g->udiff_vector, 0, 0);
		vector_hyperdiff_s(g, s, iinit, iupd, dt, g->udiff_vector, g->udiff_vector, 0, 1);
		scalar_hyperdiff_c(g, s, iinit, iupd, dt, g->udiff_scalar, 0, PIx, 1);
		scalar_hyperdiff_c(g, s, iinit, iupd, dt, g->udiff_vector, 0, WIx, 1);
	}
	filter_negative_tracers_h(g, s, iupd);      /* HorizontalDynamicsFEM::StepExplicit, :1862 */
}

/* ------------------------------------------------------------------------------------------ */
/* V explicit (implicit mode): VerticalDynamicsFEM::StepExplicit                               */
/* (VerticalDynamicsFEM.cpp:616-1159; active parts :816-828, :984-1023)                        */

static void penalty_apply(const orc_grid * g, const double * w, const double * in, double * out, double * buf) {
	/* LinearColumnDiscPenaltyFEM::Apply at vertical order 1 (LinearColumnOperatorFEM.cpp:1863-1888) */
	const int L = g->L;
	op_apply(g, ORC_OP_PENALTY_L, in, buf);
	for (int a = 0; a < L - 1; a++) out[a] += buf[a] * w[a];
	op_apply(g, ORC_OP_PENALTY_R, in, buf);
	for (int a = 1; a < L; a++) out[a] += buf[a] * w[a - 1];
}

static void v_step_fully_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt);

void orc_v_step_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt) {
	if (g->shallow_water) return;      /* VerticalDynamicsStub (TempestInitialize.h:362-365) */
	if (g->fully_explicit) { v_step_fully_explicit(g, s, iinit, iupd, dt); return; }
	const int L = g->L;
	double * xidot = (double *)malloc(sizeof(double) * (L + 1));
	double * w = (double *)malloc(sizeof(double) * (L + 1));
	double * buf = (double *)malloc(sizeof(double) * (L + 1));
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		double * nin = s->node[iinit * g->npatch + p];
		double * ein = s->redge[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		const double * CXe = P->g[ORC_G_CMXI_E];
		for (int i = 1; i < na - 1; i++)
		for (int j = 1; j < nb - 1; j++) {
			for (int k = 0; k <= L; k++) {
				xidot[k] = M4E(CXe, i, j, k, 0) * EI(UIx, i, j, k)
				         + M4E(CXe, i, j, k, 1) * EI(VIx, i, j, k)
				         + M4E(CXe, i, j, k, 2) * EI(WIx, i, j, k);
			}
			xidot[0] = 0.0; xidot[L] = 0.0;
			for (int a = 0; a < L - 1; a++) w[a] = dt * fabs(xidot[a + 1]);
			penalty_apply(g, w, &NI(UIx, i, j, 0), &NU(UIx, i, j, 0), buf);
			penalty_apply(g, w, &NI(VIx, i, j, 0), &NU(VIx, i, j, 0), buf);
		}
	}
	free(xidot); free(w); free(buf);
}

/* ------------------------------------------------------------------------------------------ */
/* DSS: Grid::Exchange + TransformHaloVelocities + GridCSGLL::ApplyDSS (State)                 */
/* (Grid.cpp:627-685, GridPatchCSGLL.cpp:1783-1924, GridCSGLL.cpp:435-781)                    */

static void avg_a(double * x, int nb, int Lk, int iA, int jb, int je) {
	for (int j = jb; j < je; j++)
	for (int k = 0; k < Lk; k++) {
		double * hi = &x[((size_t)iA * nb + j) * Lk + k];
		double * lo = &x[((size_t)(iA - 1) * nb + j) * Lk + k];
		*hi = 0.5 * (*hi + *lo);
		*lo = *hi;
	}
}
static void avg_b(double * x, int nb, int Lk, int iB, int ib, int ie) {
	for (int i = ib; i < ie; i++)
	for (int k = 0; k < Lk; k++) {
		double * hi = &x[((size_t)i * nb + iB) * Lk + k];
		double * lo = &x[((size_t)i * nb + iB - 1) * Lk + k];
		*hi = 0.5 * (*hi + *lo);
		*lo = *hi;
	}
}

static void dss_average_field(const orc_patch * P, double * x, int Lk) {
	const int na = P->na, nb = P->nb;
	const int TR = P->neighbor_panels[4], TL = P->neighbor_panels[5];
	const int BL = P->neighbor_panels[6], BR = P->neighbor_panels[7];
	int jb, je, ib, ie;
	for (int a = 1; a < P->nea; a++) avg_a(x, nb, Lk, a * NP + 1, 0, nb);
	jb = 0; je = nb; if (TL < 0) je -= 2; if (BL < 0) jb += 2;
	avg_a(x, nb, Lk, 1, jb, je);
	jb = 0; je = nb; if (TR < 0) je -= 2; if (BR < 0) jb += 2;
	avg_a(x, nb, Lk, P->nea * NP + 1, jb, je);
	for (int b = 1; b < P->neb; b++) avg_b(x, nb, Lk, b * NP + 1, 0, na);
	ib = 0; ie = na; if (BL < 0) ib += 2; if (BR < 0) ie -= 2;
	avg_b(x, nb, Lk, 1, ib, ie);
	ib = 0; ie = na; if (TL < 0) ib += 2; if (TR < 0) ie -= 2;
	avg_b(x, nb, Lk, P->neb * NP + 1, ib, ie);
#define X3(i,j,k) x[((size_t)(i) * nb + (j)) * Lk + (k)]
	const double third = 1.0 / 3.0;
	if (TR < 0) { int iA = na - 2, iB = nb - 2;
		for (int k = 0; k < Lk; k++) X3(iA, iB, k) = third * (X3(iA, iB, k) + X3(iA + 1, iB, k) + X3(iA, iB + 1, k)); }
	if (TL < 0) { int iA = 1, iB = nb - 2;
		for (int k = 0; k < Lk; k++) X3(iA, iB, k) = third * (X3(iA, iB, k) + X3(iA - 1, iB, k) + X3(iA, iB + 1, k)); }
	if (BL < 0) { int iA = 1, iB = 1;
		for (int k = 0; k < Lk; k++) X3(iA, iB, k) = third * (X3(iA, iB, k) + X3(iA - 1, iB, k) + X3(iA, iB - 1, k)); }
	if (BR < 0) { int iA = na - 2, iB = 1;
		for (int k = 0; k < Lk; k++) X3(iA, iB, k) = third * (X3(iA, iB, k) + X3(iA + 1, iB, k) + X3(iA, iB - 1, k)); }
#undef X3
}

static void dss_tracers(const orc_grid * g, orc_state * s, int ix);

void orc_apply_dss(const orc_grid * g, orc_state * s, int ix) {
	const int L = g->L;
	/* 1. exchange: pack all strips first (sources are interior nodes, never modified by unpacking) */
	size_t tot = 0;
	for (int p = 0; p < g->npatch; p++) tot += (size_t)g->patches[p].nhalo * (5 * L + 1);
	double * buf = (double *)malloc(sizeof(double) * tot);
	size_t off = 0;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		for (int m = 0; m < P->nhalo; m++, off += 5 * L + 1) {
			const int q = P->hsp[m];
			if (q < 0) continue;
			const orc_patch * Q = &g->patches[q];
			const double * qn = s->node[ix * g->npatch + q];
			const double * qe = s->redge[ix * g->npatch + q];
			const int si = P->hsi[m], sj = P->hsj[m];
			double * b = buf + off;
			/* Node array carries U,V,rhotheta,rho; REdge array carries W (Connectivity.cpp:422-457) */
			const int cn[4] = { UIx, VIx, PIx, RIx };
			for (int c = 0; c < 4; c++)
				for (int k = 0; k < L; k++)
					b[c * L + k] = qn[(((size_t)cn[c] * Q->na + si) * Q->nb + sj) * L + k];
			for (int k = 0; k <= L; k++)
				b[4 * L + k] = qe[(((size_t)WIx * Q->na + si) * Q->nb + sj) * (L + 1) + k];
			/* covector rotation of halo velocities from another panel */
			if (P->hspanel[m] != P->panel) {
				const double * T = P->htrans + 4 * (size_t)m;
				for (int k = 0; k < L; k++) {
					const double ua = b[k], ub = b[L + k];
					b[k] = T[0] * ua + T[1] * ub;
					b[L + k] = T[2] * ua + T[3] * ub;
				}
			}
		}
	}
	off = 0;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		double * pn = s->node[ix * g->npatch + p];
		double * pe = s->redge[ix * g->npatch + p];
		for (int m = 0; m < P->nhalo; m++, off += 5 * L + 1) {
			if (P->hsp[m] < 0) continue;
			const double * b = buf + off;
			const int i = P->hi[m], j = P->hj[m];
			const int cn[4] = { UIx, VIx, PIx, RIx };
			for (int c = 0; c < 4; c++)
				for (int k = 0; k < L; k++)
					pn[(((size_t)cn[c] * na + i) * nb + j) * L + k] = b[c * L + k];
			for (int k = 0; k <= L; k++)
				pe[(((size_t)WIx * na + i) * nb + j) * (L + 1) + k] = b[4 * L + k];
		}
	}
	free(buf);

	/* 2. pairwise averaging, alpha then beta, 1/3 rule at cube corners (GridCSGLL.cpp:560-777) */
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		for (int c = 0; c < 5; c++) {
			const int Lk = (c == WIx) ? L + 1 : L;
			double * x = (c == WIx)
				? s->redge[ix * g->npatch + p] + (size_t)c * na * nb * Lk
				: s->node[ix * g->npatch + p] + (size_t)c * na * nb * Lk;
			dss_average_field(P, x, Lk);
		}
	}
	dss_tracers(g, s, ix);
}

/* Grid::Exchange + ApplyDSS for DataType_Tracers: scalars on levels, no covector rotation */
static void dss_tracers(const orc_grid * g, orc_state * s, int ix) {
	if (!has_tracers(g, s)) return;
	const int L = g->L, nt = g->ntracers;
	size_t tot = 0;
	for (int p = 0; p < g->npatch; p++) tot += (size_t)g->patches[p].nhalo * nt * L;
	double * buf = (double *)malloc(sizeof(double) * (tot ? tot : 1));
	size_t off = 0;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		for (int m = 0; m < P->nhalo; m++, off += (size_t)nt * L) {
			const int q = P->hsp[m];
			if (q < 0) continue;
			const orc_patch * Q = &g->patches[q];
			const double * qt = s->tracer[ix * g->npatch + q];
			for (int c = 0; c < nt; c++)
				for (int k = 0; k < L; k++)
					buf[off + (size_t)c * L + k] = qt[(((size_t)c * Q->na + P->hsi[m]) * Q->nb + P->hsj[m]) * L + k];
		}
	}
	off = 0;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		double * pt = s->tracer[ix * g->npatch + p];
		for (int m = 0; m < P->nhalo; m++, off += (size_t)nt * L) {
			if (P->hsp[m] < 0) continue;
			for (int c = 0; c < nt; c++)
				for (int k = 0; k < L; k++)
					pt[(((size_t)c * P->na + P->hi[m]) * P->nb + P->hj[m]) * L + k] = buf[off + (size_t)c * L + k];
		}
	}
	free(buf);
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		for (int c = 0; c < nt; c++)
			dss_average_field(P, s->tracer[ix * g->npatch + p] + (size_t)c * P->na * P->nb * L, L);
	}
}

/* ------------------------------------------------------------------------------------------ */
/* banded LU with partial pivoting: LAPACK dgbsv = dgbtrf(unblocked dgbtf2) + dgbtrs            */
/* (third-party; call site src/base/LinearAlgebra.cpp:156-202 <- VerticalDynamicsFEM.cpp:1457)  */

/* Rounding of the update a - l*u: LAPACK's dgbtf2 / dgbtrs / dtbsv are built from the BLAS calls dger / dtbsv, and
 * every optimised BLAS on a CPU with FMA units (MKL, OpenBLAS) evaluates them with fused multiply-adds.  Measured
 * (tests/analysis/lapack_spread.py): the reference linked with MKL and with OpenBLAS's netlib LAPACK agree BIT FOR
 * BIT after 100 steps, and orc_dgbsv reproduces both bit for bit on the per-call fixtures only when its three
 * update statements are fused as well.  orc_lu_fma = 1 (default) is therefore the reference's arithmetic on any
 * FMA-capable host; 0 = separate multiply and subtract (a BLAS without FMA), kept for the sensitivity analysis. */
static int orc_lu_fma = 1;
void orc_set_lu_fma(int on) { orc_lu_fma = on; }
int orc_get_lu_fma(void) { return orc_lu_fma; }
static inline double lu_upd(double a, double l, double u) { return orc_lu_fma ? fma(-l, u, a) : a - l * u; }

int orc_dgbsv(int n, int kl, int ku, double * ab, int ldab, int * ipiv, double * b) {
	const int kv = ku + kl;
#define AB(i,j) ab[(size_t)(j) * ldab + (kv + (i) - (j))]   /* full-matrix (i,j), 0-based */
	int info = 0;
	/* zero the fill-in super-diagonals of the first columns */
	for (int j = ku + 1; j < (kv < n ? kv : n); j++)
		for (int i = kv - j; i < kl; i++) ab[(size_t)j * ldab + i] = 0.0;
	int ju = 0;
	for (int j = 0; j < n; j++) {
		if (j + kv < n) for (int i = 0; i < kl; i++) ab[(size_t)(j + kv) * ldab + i] = 0.0;
		const int km = (kl < n - 1 - j) ? kl : n - 1 - j;
		int jp = 0; double amax = fabs(AB(j, j));
		for (int i = 1; i <= km; i++) { double v = fabs(AB(j + i, j)); if (v > amax) { amax = v; jp = i; } }
		ipiv[j] = j + jp;
		if (AB(j + jp, j) != 0.0) {
			int t = j + ku + jp; if (t > n - 1) t = n - 1;
			if (t > ju) ju = t;
			if (jp != 0) for (int c = j; c <= ju; c++) { double tmp = AB(j + jp, c); AB(j + jp, c) = AB(j, c); AB(j, c) = tmp; }
			if (km > 0) {
				const double r = 1.0 / AB(j, j);
				for (int i = 1; i <= km; i++) AB(j + i, j) *= r;
				for (int c = j + 1; c <= ju; c++) {
					const double u = AB(j, c);
					if (u != 0.0) for (int i = 1; i <= km; i++) AB(j + i, c) = lu_upd(AB(j + i, c), AB(j + i, j), u);
				}
			}
		} else if (info == 0) info = j + 1;
	}
	if (info != 0) return info;
	/* forward: apply L and the interchanges */
	for (int j = 0; j < n - 1; j++) {
		const int lm = (kl < n - 1 - j) ? kl : n - 1 - j;
		const int l = ipiv[j];
		if (l != j) { double tmp = b[l]; b[l] = b[j]; b[j] = tmp; }
		for (int i = 1; i <= lm; i++) b[j + i] = lu_upd(b[j + i], b[j], AB(j + i, j));
	}
	/* backward: U x = y, U has kl+ku super-diagonals (column-oriented, as dtbsv) */
	for (int j = n - 1; j >= 0; j--) {
		if (b[j] != 0.0) {
			b[j] /= AB(j, j);
			const double t = b[j];
			int lo = j - kv; if (lo < 0) lo = 0;
			for (int i = j - 1; i >= lo; i--) b[i] = lu_upd(b[i], t, AB(i, j));
		}
	}
#undef AB
	return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* V implicit: SetupReferenceColumn + PrepareColumn + BuildF + BuildJacobianF + DGBSV          */
/* (VerticalDynamicsFEM.cpp:1643-1835, 1839-2179, 2183-2780, 2784-3187, 1438-1485)             */

enum { FP = 0, FW = 1, FR = 2, FTOT = 3, OFFD = 4, LDAB = 13 };
#define VECF(c,k) (FTOT * (k) + (c))
/* band index of d(row (c1,k1)) / d(col (c0,k0))  (VerticalDynamicsFEM.h:110-122) */
#define MATF(c0,k0,c1,k1) ((2 * OFFD + (FTOT * (k1) + (c1)) - (FTOT * (k0) + (c0))) + LDAB * (FTOT * (k0) + (c0)))

typedef struct {
	double *un, *vn, *ue, *ve, *dua, *dub;             /* U,V on nodes/edges, d/dxi on edges   */
	double *pn, *wn, *rn, *pe, *we, *re;               /* rhotheta, W, rho on nodes / edges     */
	double *exn, *dpe, *xdn, *xde, *ddw, *mf, *pf, *dmf, *dpf, *ke, *dke, *uw, *aux, *udd, *udr;
} colwork;

static double * dalloc(int n) { return (double *)calloc((size_t)n, sizeof(double)); }

static void colwork_init(colwork * w, int L) {
	double ** f = (double **)w;
	for (size_t i = 0; i < sizeof(colwork) / sizeof(double *); i++) f[i] = dalloc(L + 2);
}
static void colwork_free(colwork * w) {
	double ** f = (double **)w;
	for (size_t i = 0; i < sizeof(colwork) / sizeof(double *); i++) free(f[i]);
}

static void column_f_jacobian(const orc_grid * g, const orc_patch * P, const double * nin, const double * ein,
	int iA, int iB, double dt, double * x0, double * F, double * AB, colwork * w) {
	const int L = g->L, na = P->na, nb = P->nb;
	const int n = FTOT * (L + 1);
	const double * Jn = P->g[ORC_G_JAC], * Je = P->g[ORC_G_JAC_E];
	const double * CA = P->g[ORC_G_CMA], * CB = P->g[ORC_G_CMB], * CX = P->g[ORC_G_CMXI];
	const double * CAe = P->g[ORC_G_CMA_E], * CBe = P->g[ORC_G_CMB_E], * CXe = P->g[ORC_G_CMXI_E];
	const double * DRe = P->g[ORC_G_DR_E];
	(void)na;

	/* SetupReferenceColumn (:1643-1835) */
	for (int k = 0; k < L; k++) { w->un[k] = NI(UIx, iA, iB, k); w->vn[k] = NI(VIx, iA, iB, k); }
	op_apply(g, ORC_OP_INTERP_N2E, w->un, w->ue);
	op_apply(g, ORC_OP_INTERP_N2E, w->vn, w->ve);
	op_apply(g, ORC_OP_DIFF_N2E, w->un, w->dua);
	op_apply(g, ORC_OP_DIFF_N2E, w->vn, w->dub);
	memset(x0, 0, sizeof(double) * n);
	for (int k = 0; k < L; k++) { x0[VECF(FP, k)] = NI(PIx, iA, iB, k); x0[VECF(FR, k)] = NI(RIx, iA, iB, k); }
	for (int k = 0; k <= L; k++) x0[VECF(FW, k)] = EI(WIx, iA, iB, k);

	/* PrepareColumn (:1839-2179) */
	for (int k = 0; k < L; k++) { w->pn[k] = x0[VECF(FP, k)]; w->rn[k] = x0[VECF(FR, k)]; }
	for (int k = 0; k <= L; k++) w->we[k] = x0[VECF(FW, k)];
	op_apply(g, ORC_OP_INTERP_E2N, w->we, w->wn);
	op_apply(g, ORC_OP_INTERP_N2E, w->rn, w->re);
	op_apply(g, ORC_OP_INTERP_N2E, w->pn, w->pe);
	for (int k = 0; k < L; k++) w->exn[k] = exner_from_rhotheta(g, w->pn[k]);
	op_apply(g, ORC_OP_DIFF_N2E, w->exn, w->dpe);
	for (int k = 0; k < L; k++)
		w->xdn[k] = M4(CX, iA, iB, k, 0) * w->un[k] + M4(CX, iA, iB, k, 1) * w->vn[k] + M4(CX, iA, iB, k, 2) * w->wn[k];
	for (int k = 1; k < L; k++)
		w->xde[k] = M4E(CXe, iA, iB, k, 0) * w->ue[k] + M4E(CXe, iA, iB, k, 1) * w->ve[k] + M4E(CXe, iA, iB, k, 2) * w->we[k];
	w->xde[0] = 0.0; w->xde[L] = 0.0;
	op_apply(g, ORC_OP_DIFFDIFF_E2E, w->we, w->ddw);

	/* BuildF (:2183-2780) */
	memset(F, 0, sizeof(double) * n);
	w->mf[0] = 0.0; w->mf[L] = 0.0; w->pf[0] = 0.0; w->pf[L] = 0.0;
	for (int k = 1; k < L; k++) {
		w->mf[k] = Je[((size_t)iA * nb + iB) * (L + 1) + k] * w->re[k] * w->xde[k];
		w->pf[k] = Je[((size_t)iA * nb + iB) * (L + 1) + k] * w->pe[k] * w->xde[k];
	}
	op_apply(g, ORC_OP_DIFF_E2N, w->mf, w->dmf);
	op_apply(g, ORC_OP_DIFF_E2N, w->pf, w->dpf);
	for (int k = 0; k < L; k++) {
		const double invJ = 1.0 / M3(Jn, iA, iB, k);
		F[VECF(FR, k)] = w->dmf[k] * invJ;
		F[VECF(FP, k)] += w->dpf[k] * invJ;
	}
	for (int k = 0; k < L; k++) {
		const double ua = w->un[k], ub = w->vn[k], ux = w->wn[k];
		const double ca = M4(CA, iA, iB, k, 0) * ua + M4(CA, iA, iB, k, 1) * ub + M4(CA, iA, iB, k, 2) * ux;
		const double cb = M4(CB, iA, iB, k, 0) * ua + M4(CB, iA, iB, k, 1) * ub + M4(CB, iA, iB, k, 2) * ux;
		const double cx = M4(CX, iA, iB, k, 0) * ua + M4(CX, iA, iB, k, 1) * ub + M4(CX, iA, iB, k, 2) * ux;
		w->ke[k] = 0.5 * (ca * ua + cb * ub + cx * ux);
	}
	op_apply(g, ORC_OP_DIFF_N2E, w->ke, w->dke);
	for (int k = 1; k < L; k++) {
		const double pgf = w->dpe[k] * w->pe[k] / w->re[k];
		F[VECF(FW, k)] = pgf;
		F[VECF(FW, k)] += g->grav * M4E(DRe, iA, iB, k, 2);
		const double ua = w->ue[k], ub = w->ve[k], ux = w->we[k];
		const double ca = M4E(CAe, iA, iB, k, 0) * ua + M4E(CAe, iA, iB, k, 1) * ub + M4E(CAe, iA, iB, k, 2) * ux;
		const double cb = M4E(CBe, iA, iB, k, 0) * ua + M4E(CBe, iA, iB, k, 1) * ub + M4E(CBe, iA, iB, k, 2) * ux;
		const double curl = -ca * w->dua[k] - cb * w->dub[k];
		F[VECF(FW, k)] += (w->dke[k] + curl);
	}
	if (g->uniform_diffusion) {
		/* uniform diffusion of rho*theta and W in the vertical (PrepareColumn :2104-2160, BuildF :2593-2635):
		 * second xi-derivative of (column - reference column), W not diffused on the boundaries */
		const double cs = g->udiff_scalar / (g->ztop * g->ztop), cvv = g->udiff_vector / (g->ztop * g->ztop);
		for (int k = 0; k < L; k++) w->aux[k] = P->ref_node[(((size_t)PIx * na + iA) * nb + iB) * L + k];
		op_apply(g, ORC_OP_DIFFDIFF_N2N, w->pn, w->udd);
		op_apply(g, ORC_OP_DIFFDIFF_N2N, w->aux, w->udr);
		for (int k = 0; k < L; k++) w->udd[k] = w->udd[k] - w->udr[k];
		for (int k = 0; k < L; k++) F[VECF(FP, k)] -= cs * w->udd[k];
		for (int k = 0; k <= L; k++) w->aux[k] = P->ref_redge[(((size_t)WIx * na + iA) * nb + iB) * (L + 1) + k];
		op_apply(g, ORC_OP_DIFFDIFF_E2E, w->aux, w->udr);
		for (int k = 0; k <= L; k++) w->udd[k] = w->ddw[k] - w->udr[k];
		w->udd[0] = 0.0; w->udd[L] = 0.0;
		for (int k = 0; k <= L; k++) F[VECF(FW, k)] -= cvv * w->udd[k];
	}
	/* upwinding (:2640-2712): weights |xidot| at interfaces 1..L-1 */
	const double upc = 0.5 * (1.0 / (double)L);            /* m_dUpwindCoeff (:394-395) */
	for (int a = 0; a < L - 1; a++) w->uw[a] = fabs(w->xde[a + 1]);
	{	/* rhotheta (c=2) on levels */
		memset(w->aux, 0, sizeof(double) * (L + 1));
		double * buf = w->mf;   /* mf no longer needed */
		penalty_apply(g, w->uw, w->pn, w->aux, buf);
		for (int k = 0; k < L; k++) F[VECF(FP, k)] -= w->aux[k];
	}
	{	/* W (c=3) on interfaces */
		w->ddw[0] = 0.0; w->ddw[L] = 0.0;
		for (int k = 0; k <= L; k++) F[VECF(FW, k)] -= upc * fabs(w->xde[k]) * w->ddw[k];
	}
	{	/* rho (c=4) on levels */
		memset(w->aux, 0, sizeof(double) * (L + 1));
		double * buf = w->mf;
		penalty_apply(g, w->uw, w->rn, w->aux, buf);
		for (int k = 0; k < L; k++) F[VECF(FR, k)] -= w->aux[k];
	}
	F[VECF(FW, 0)] = 0.0;
	F[VECF(FW, L)] = 0.0;
	/* (x - x0)/dt vanishes: the Newton iterate is the initial column (:2776-2779) */

	/* BuildJacobianF_LOR_RhoTheta_Pi (:2977-3187) */
	memset(AB, 0, sizeof(double) * n * LDAB);
	const double * cDe2n = g->opc[ORC_OP_DIFF_E2N]; const int * bDe2n = g->opb[ORC_OP_DIFF_E2N], * eDe2n = g->ope[ORC_OP_DIFF_E2N];
	const double * cIn2e = g->opc[ORC_OP_INTERP_N2E]; const int * bIn2e = g->opb[ORC_OP_INTERP_N2E], * eIn2e = g->ope[ORC_OP_INTERP_N2E];
	const double * cDn2e = g->opc[ORC_OP_DIFF_N2E]; const int * bDn2e = g->opb[ORC_OP_DIFF_N2E], * eDn2e = g->ope[ORC_OP_DIFF_N2E];
	const double * cIe2n = g->opc[ORC_OP_INTERP_E2N]; const int * bIe2n = g->opb[ORC_OP_INTERP_E2N], * eIe2n = g->ope[ORC_OP_INTERP_E2N];
	const double cv = g->cp - g->Rd;
#define JE(k) Je[((size_t)iA * nb + iB) * (L + 1) + (k)]
	for (int k = 0; k < L; k++) {
		const double invJ = 1.0 / M3(Jn, iA, iB, k);
		for (int m = bDe2n[k]; m < eDe2n[k]; m++) {
			if (m != 0 && m != L) {
				const double c = cDe2n[k * (L + 1) + m] * JE(m) * invJ * M4E(CXe, iA, iB, m, 2);
				AB[MATF(FW, m, FP, k)] += c * w->pe[m];
				AB[MATF(FW, m, FR, k)] += c * w->re[m];
			}
			for (int q = bIn2e[m]; q < eIn2e[m]; q++) {
				const double c = cDe2n[k * (L + 1) + m] * JE(m) * invJ * cIn2e[m * L + q] * w->xde[m];
				AB[MATF(FR, q, FR, k)] += c;
				AB[MATF(FP, q, FP, k)] += c;
			}
		}
	}
	for (int k = 1; k < L; k++) {
		const double cA = w->pe[k] * g->Rd / (w->re[k] * cv);
		for (int m = bDn2e[k]; m < eDn2e[k]; m++)
			AB[MATF(FP, m, FW, k)] += cA * cDn2e[k * L + m] * w->exn[m] / w->pn[m];
		const double cB = 1.0 / (w->re[k] * w->re[k]) * w->dpe[k];
		for (int q = bIn2e[k]; q < eIn2e[k]; q++) {
			const double cC = cB * cIn2e[k * L + q];
			AB[MATF(FP, q, FW, k)] += cC * w->re[k];
			AB[MATF(FR, q, FW, k)] += -cC * w->pe[k];
		}
	}
	for (int k = 1; k < L; k++)
		for (int l = bDn2e[k]; l < eDn2e[k]; l++)
			for (int m = bIe2n[l]; m < eIe2n[l]; m++)
				AB[MATF(FW, m, FW, k)] += cIe2n[l * (L + 1) + m] * cDn2e[k * L + l] * w->xdn[l];

	/* BuildJacobianF_Diffusion (:2784-2973): upwinding terms */
	const double * cDDe = g->opc[ORC_OP_DIFFDIFF_E2E]; const int * bDDe = g->opb[ORC_OP_DIFFDIFF_E2E], * eDDe = g->ope[ORC_OP_DIFFDIFF_E2E];
	const double * cPL = g->opc[ORC_OP_PENALTY_L]; const int * bPL = g->opb[ORC_OP_PENALTY_L], * ePL = g->ope[ORC_OP_PENALTY_L];
	const double * cPR = g->opc[ORC_OP_PENALTY_R]; const int * bPR = g->opb[ORC_OP_PENALTY_R], * ePR = g->ope[ORC_OP_PENALTY_R];
	for (int c = 2; c < 5; c++) {
		const int fc = c - 2;
		if (c == WIx) {
			for (int k = 0; k <= L; k++) {
				double sw;
				if (w->xde[k] > 0.0) sw = M4E(CXe, iA, iB, k, 2);
				else if (w->xde[k] < 0.0) sw = -M4E(CXe, iA, iB, k, 2);
				else sw = 0.0;
				AB[MATF(FW, k, fc, k)] -= upc * sw * w->ddw[k];
			}
			for (int k = 0; k <= L; k++)
				for (int q = bDDe[k]; q < eDDe[k]; q++)
					AB[MATF(fc, q, fc, k)] -= upc * fabs(w->xde[k]) * cDDe[k * (L + 1) + q];
		} else {
			const double * xs = (c == PIx) ? w->pn : w->rn;
			for (int a = 1; a < L; a++) {
				const double wt = fabs(w->xde[a]);
				double sw;
				if (w->xde[a] > 0.0) sw = M4E(CXe, iA, iB, a, 2);
				else if (w->xde[a] < 0.0) sw = -M4E(CXe, iA, iB, a, 2);
				else sw = 0.0;
				const int kl = a - 1, kr = a;    /* left element = level a-1, right element = level a */
				for (int q = bPL[kl]; q < ePL[kl]; q++) AB[MATF(FW, a, fc, kl)] -= sw * cPL[kl * L + q] * xs[q];
				for (int q = bPR[kr]; q < ePR[kr]; q++) AB[MATF(FW, a, fc, kr)] -= sw * cPR[kr * L + q] * xs[q];
				for (int q = bPL[kl]; q < ePL[kl]; q++) AB[MATF(fc, q, fc, kl)] -= wt * cPL[kl * L + q];
				for (int q = bPR[kr]; q < ePR[kr]; q++) AB[MATF(fc, q, fc, kr)] -= wt * cPR[kr * L + q];
			}
		}
	}
	for (int k = 0; k <= L; k++) {
		AB[MATF(FP, k, FP, k)] += 1.0 / dt;
		AB[MATF(FW, k, FW, k)] += 1.0 / dt;
		AB[MATF(FR, k, FR, k)] += 1.0 / dt;
	}
#undef JE
}

void orc_column_f_jacobian(const orc_grid * g, const orc_patch * p, const double * node, const double * redge,
	int iA, int iB, double dt, double * x0, double * F, double * AB) {
	colwork w; colwork_init(&w, g->L);
	column_f_jacobian(g, p, node, redge, iA, iB, dt, x0, F, AB, &w);
	colwork_free(&w);
}

/* VerticalDynamicsFEM::UpdateColumnTracers (VerticalDynamicsFEM.cpp:3783-4282), implicit mode, vertical order 1,
 * upwinding on, no uniform diffusion: backward-Euler vertical transport of every tracer density with the xi_dot of
 * the INITIAL column in the (tridiagonal) matrix and upwind weights, the xi_dot of the UPDATED W in the mass flux.
 * x0 = initial column state, w->ue / w->ve = initial U,V on interfaces (both left by column_f_jacobian),
 * eup = updated REdge array of the patch. */
static void update_column_tracers(const orc_grid * g, const orc_patch * P, int iA, int iB, double dt,
	const double * x0, const colwork * w, const double * eup, const double * tin, double * tup)
{
	const int L = g->L, na = P->na, nb = P->nb, nt = g->ntracers;
	const double * Jn = P->g[ORC_G_JAC], * Je = P->g[ORC_G_JAC_E], * CXe = P->g[ORC_G_CMXI_E];
	const int ld = L;                       /* m_matTracersLUDF is [L][L]; band (kl = ku = 1) rows 0..3 of each column */
	double * A = dalloc(L * ld), * A2 = dalloc(L * ld);
	double * xd0 = dalloc(L + 1), * xd1 = dalloc(L + 1), * qn = dalloc(L + 1), * qe = dalloc(L + 2);
	double * mf = dalloc(L + 1), * dmf = dalloc(L + 1), * Fq = dalloc(L + 1), * ww = dalloc(L + 1), * aux = dalloc(L + 1), * buf = dalloc(L + 1);
	int * ipiv = (int *)malloc(sizeof(int) * L);
#define TMAT(n_, k_) A[(2 + (k_) - (n_)) + ld * (n_)]      /* TracerMatFIx(n, k): d F_k / d q_n (VerticalDynamicsFEM.h:128-137) */
	const double * Cdiff = g->opc[ORC_OP_DIFF_E2N], * Cint = g->opc[ORC_OP_INTERP_N2E];
	const double * Cpl = g->opc[ORC_OP_PENALTY_L], * Cpr = g->opc[ORC_OP_PENALTY_R];
	const int ndiff = g->opnin[ORC_OP_DIFF_E2N], nint = g->opnin[ORC_OP_INTERP_N2E];
	const int npl = g->opnin[ORC_OP_PENALTY_L], npr = g->opnin[ORC_OP_PENALTY_R];
	/* xi_dot of the initial column on interfaces (:3943-3957) */
	for (int k = 1; k < L; k++)
		xd0[k] = M4E(CXe, iA, iB, k, 0) * w->ue[k] + M4E(CXe, iA, iB, k, 1) * w->ve[k] + M4E(CXe, iA, iB, k, 2) * x0[VECF(FW, k)];
	xd0[0] = 0.0; xd0[L] = 0.0;
	/* dRhoQ_k / dRhoQ_n (:3959-3976); off-diagonal terms only with implicit advection (:3910-3912) */
	if (!g->fully_explicit)
	for (int k = 0; k < L; k++)
		for (int m = g->opb[ORC_OP_DIFF_E2N][k]; m < g->ope[ORC_OP_DIFF_E2N][k]; m++)
			for (int n = g->opb[ORC_OP_INTERP_N2E][m]; n < g->ope[ORC_OP_INTERP_N2E][m]; n++)
				TMAT(n, k) += Cdiff[(size_t)k * ndiff + m] * Je[((size_t)iA * nb + iB) * (L + 1) + m]
					/ Jn[((size_t)iA * nb + iB) * L + k] * Cint[(size_t)m * nint + n] * xd0[m];
	/* upwinding in the Jacobian (:3978-4016) */
	if (!g->fully_explicit)
	for (int a = 1; a < L; a++) {
		const double weight = fabs(xd0[a]);
		for (int k = a - 1; k < a; k++)
			for (int n = g->opb[ORC_OP_PENALTY_L][k]; n < g->ope[ORC_OP_PENALTY_L][k]; n++)
				TMAT(n, k) -= weight * Cpl[(size_t)k * npl + n];
		for (int k = a; k < a + 1; k++)
			for (int n = g->opb[ORC_OP_PENALTY_R][k]; n < g->ope[ORC_OP_PENALTY_R][k]; n++)
				TMAT(n, k) -= weight * Cpr[(size_t)k * npr + n];
	}
	for (int k = 0; k < L; k++) TMAT(k, k) += 1.0 / dt;
	/* xi_dot with the UPDATED W, fully explicit: with the initial W (:4047-4086) */
	for (int k = 1; k < L; k++)
		xd1[k] = M4E(CXe, iA, iB, k, 0) * w->ue[k] + M4E(CXe, iA, iB, k, 1) * w->ve[k]
			+ M4E(CXe, iA, iB, k, 2) * (g->fully_explicit ? x0[VECF(FW, k)] : eup[(((size_t)WIx * na + iA) * nb + iB) * (L + 1) + k]);
	xd1[0] = 0.0; xd1[L] = 0.0;
	for (int c = 0; c < nt; c++) {
		for (int k = 0; k < L; k++) qn[k] = tin[(((size_t)c * na + iA) * nb + iB) * L + k];
		op_apply(g, ORC_OP_INTERP_N2E, qn, qe);
		for (int k = 0; k <= L; k++) mf[k] = Je[((size_t)iA * nb + iB) * (L + 1) + k] * qe[k] * xd1[k];
		if (g->uniform_diffusion) {
			/* uniform diffusion of (mixing ratio - reference mixing ratio) (:4117-4141) */
			for (int k = 0; k < L; k++) {
				aux[k] = qn[k] / w->rn[k];
				aux[k] -= P->ref_tracer[(((size_t)c * na + iA) * nb + iB) * L + k] / P->ref_node[(((size_t)RIx * na + iA) * nb + iB) * L + k];
			}
			op_apply(g, ORC_OP_DIFF_N2E, aux, buf);
			for (int k = 1; k < L; k++) mf[k] -= g->udiff_scalar * w->re[k] * buf[k];
		}
		mf[0] = 0.0; mf[L] = 0.0;
		op_apply(g, ORC_OP_DIFF_E2N, mf, dmf);
		for (int k = 0; k < L; k++) Fq[k] = dmf[k] / Jn[((size_t)iA * nb + iB) * L + k];
		/* upwinding with the initial xi_dot (:4153-4181); fully explicit: m_dXiDotREdge, the same numbers */
		for (int a = 0; a < L - 1; a++) ww[a] = fabs(g->fully_explicit ? xd1[a + 1] : xd0[a + 1]);
		for (int k = 0; k < L; k++) aux[k] = 0.0;
		penalty_apply(g, ww, qn, aux, buf);
		for (int k = 0; k < L; k++) Fq[k] -= aux[k];
		/* implicit velocity correction (:4183-4233) */
		if (!g->fully_explicit)
		for (int a = 1; a < L; a++) {
			double sw;
			if (xd0[a] > 0.0) sw = 1.0 * M4E(CXe, iA, iB, a, 2);
			else if (xd0[a] < 0.0) sw = -1.0 * M4E(CXe, iA, iB, a, 2);
			else sw = 0.0;
			const double jump = sw * (eup[(((size_t)WIx * na + iA) * nb + iB) * (L + 1) + a] - x0[VECF(FW, a)]);
			for (int k = a - 1; k < a; k++)
				for (int n = g->opb[ORC_OP_PENALTY_L][k]; n < g->ope[ORC_OP_PENALTY_L][k]; n++)
					Fq[k] -= Cpl[(size_t)k * npl + n] * qn[n] * jump;
			for (int k = a; k < a + 1; k++)
				for (int n = g->opb[ORC_OP_PENALTY_R][k]; n < g->ope[ORC_OP_PENALTY_R][k]; n++)
					Fq[k] -= Cpr[(size_t)k * npr + n] * qn[n] * jump;
		}
		/* DGBTRF once + DGBTRS per tracer == dgbsv on a fresh copy of the same matrix (:4024-4268) */
		memcpy(A2, A, sizeof(double) * L * ld);
		orc_dgbsv(L, 1, 1, A2, ld, ipiv, Fq);
		for (int k = 0; k < L; k++) tup[(((size_t)c * na + iA) * nb + iB) * L + k] -= Fq[k];
	}
#undef TMAT
	free(A); free(A2); free(xd0); free(xd1); free(qn); free(qe); free(mf); free(dmf); free(Fq); free(ww); free(aux); free(buf); free(ipiv);
}

/* VerticalDynamicsFEM::FilterNegativeTracers (VerticalDynamicsFEM.cpp:4286-4347): per column */
static void filter_negative_tracers_v(const orc_grid * g, orc_state * s, int iupd) {
	if (!has_tracers(g, s)) return;
	const int L = g->L;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		double * t = s->tracer[iupd * g->npatch + p];
		for (int i = 1; i < na - 1; i++)
		for (int j = 1; j < nb - 1; j++)
		for (int c = 0; c < g->ntracers; c++) {
			double total = 0.0, nonneg = 0.0;
			for (int k = 0; k < L; k++) {
				const double q = t[(((size_t)c * na + i) * nb + j) * L + k];
				const double pm = q * P->area_node[((size_t)i * nb + j) * L + k];
				total += pm;
				if (q >= 0.0) nonneg += pm;
			}
			const double r = total / nonneg;
			for (int k = 0; k < L; k++) {
				double * q = &t[(((size_t)c * na + i) * nb + j) * L + k];
				if (*q > 0.0) *q *= r; else *q = 0.0;
			}
		}
	}
}

int orc_v_step_implicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt) {
	if (g->fully_explicit) return 0;      /* VerticalDynamicsFEM::StepImplicit, :1239-1242 */
	if (g->shallow_water) return 0;    /* VerticalDynamicsStub */
	const int L = g->L;
	const int n = FTOT * (L + 1);
	double * x0 = dalloc(n), * F = dalloc(n), * AB = dalloc(n * LDAB);
	int * ipiv = (int *)malloc(sizeof(int) * n);
	colwork w; colwork_init(&w, L);
	int info = 0;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		const double * nin = s->node[iinit * g->npatch + p];
		const double * ein = s->redge[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		double * eup = s->redge[iupd * g->npatch + p];
		/* unique columns only (:1315-1337) */
		for (int a = 0; a < P->nea; a++)
		for (int b = 0; b < P->neb; b++) {
			const int iend = (a == P->nea - 1) ? NP : NP - 1;
			const int jend = (b == P->neb - 1) ? NP : NP - 1;
			for (int i = 0; i < iend; i++)
			for (int j = 0; j < jend; j++) {
				const int iA = 1 + a * NP + i, iB = 1 + b * NP + j;
				column_f_jacobian(g, P, nin, ein, iA, iB, dt, x0, F, AB, &w);
				int r = orc_dgbsv(n, OFFD, OFFD, AB, LDAB, ipiv, F);
				if (r != 0 && info == 0) info = r;
				/* one Newton step x = x0 - J^{-1} F (:1483-1485) and scatter (:1488-1529) */
				for (int k = 0; k < L; k++) {
					NU(PIx, iA, iB, k) = x0[VECF(FP, k)] - F[VECF(FP, k)];
					NU(RIx, iA, iB, k) = x0[VECF(FR, k)] - F[VECF(FR, k)];
				}
				for (int k = 0; k <= L; k++) EU(WIx, iA, iB, k) = x0[VECF(FW, k)] - F[VECF(FW, k)];
				if (has_tracers(g, s))
					update_column_tracers(g, P, iA, iB, dt, x0, &w, eup,
						s->tracer[iinit * g->npatch + p], s->tracer[iupd * g->npatch + p]);
			}
		}
		/* duplicate element-seam nodes inside the patch receive P, W, Rho (not U,V) (:1543-1633) */
		for (int a = 1; a < P->nea; a++) {
			const int iA = a * NP;     /* = interior begin + a*np - 1 */
			for (int b = 0; b < P->neb; b++) {
				const int jend = (b == P->neb - 1) ? NP : NP - 1;
				for (int j = 0; j < jend; j++) {
					const int iB = 1 + b * NP + j;
					for (int k = 0; k < L; k++) {
						NU(PIx, iA, iB, k) = NU(PIx, iA + 1, iB, k);
						NU(WIx, iA, iB, k) = NU(WIx, iA + 1, iB, k);
						NU(RIx, iA, iB, k) = NU(RIx, iA + 1, iB, k);
					}
					for (int k = 0; k <= L; k++) {
						EU(PIx, iA, iB, k) = EU(PIx, iA + 1, iB, k);
						EU(WIx, iA, iB, k) = EU(WIx, iA + 1, iB, k);
						EU(RIx, iA, iB, k) = EU(RIx, iA + 1, iB, k);
					}
					if (has_tracers(g, s)) {
						double * tup = s->tracer[iupd * g->npatch + p];
						for (int c = 0; c < g->ntracers; c++)
							for (int k = 0; k < L; k++)
								tup[(((size_t)c * na + iA) * nb + iB) * L + k] = tup[(((size_t)c * na + iA + 1) * nb + iB) * L + k];
					}
				}
			}
		}
		for (int b = 1; b < P->neb; b++)
		for (int i = 1; i < na - 1; i++) {
			const int iB = b * NP;
			for (int k = 0; k < L; k++) {
				NU(PIx, i, iB, k) = NU(PIx, i, iB + 1, k);
				NU(WIx, i, iB, k) = NU(WIx, i, iB + 1, k);
				NU(RIx, i, iB, k) = NU(RIx, i, iB + 1, k);
			}
			for (int k = 0; k <= L; k++) {
				EU(PIx, i, iB, k) = EU(PIx, i, iB + 1, k);
				EU(WIx, i, iB, k) = EU(WIx, i, iB + 1, k);
				EU(RIx, i, iB, k) = EU(RIx, i, iB + 1, k);
			}
			if (has_tracers(g, s)) {
				double * tup = s->tracer[iupd * g->npatch + p];
				for (int c = 0; c < g->ntracers; c++)
					for (int k = 0; k < L; k++)
						tup[(((size_t)c * na + i) * nb + iB) * L + k] = tup[(((size_t)c * na + i) * nb + iB + 1) * L + k];
			}
		}
	}
	colwork_free(&w);
	free(x0); free(F); free(AB); free(ipiv);
	filter_negative_tracers_v(g, s, iupd);      /* VerticalDynamicsFEM::StepImplicit, :1637 */
	return info;
}

/* ------------------------------------------------------------------------------------------ */
/* hyperviscosity: ApplyScalarHyperdiffusion, ApplyVectorHyperdiffusion (+ComputeCurlAndDiv),   */
/* StepAfterSubCycle (HorizontalDynamicsFEM.cpp:1867-2203, 2207-2414, 2637-2726;                */
/* GridPatchCSGLL.cpp:1132-1305)                                                               */

/* comp < 0: components 2..4; comp >= 2: that component only (iComponent); remove_ref: psi = initial - reference
 * state (fRemoveRefState, :2056-2066) */
static void scalar_hyperdiff_c(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nu, int scale_locally,
	int comp, int remove_ref) {
	const int L = g->L;
	const double * D = g->dx, * S = g->stiff;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const double inv_da = 1.0 / patch_da(g, P), inv_db = 1.0 / patch_db(g, P);
		double local_nu = nu;
		if (scale_locally && g->ref_length != 0.0) local_nu *= pow(patch_da(g, P) / g->ref_length, 3.2);
		const int na = P->na, nb = P->nb;
		const double * C2A = P->g[ORC_G_CM2A], * C2B = P->g[ORC_G_CM2B];
		for (int c = 2; c < 5; c++) {
			if (comp >= 0 && c != comp) continue;
			const int Lk = (c == WIx) ? L + 1 : L;
			const double * Jc = (c == WIx) ? P->g[ORC_G_JAC_E] : P->g[ORC_G_JAC];
			const double * xin = ((c == WIx) ? s->redge[iinit * g->npatch + p] : s->node[iinit * g->npatch + p]) + (size_t)c * na * nb * Lk;
			const double * xrf = remove_ref ? (((c == WIx) ? P->ref_redge : P->ref_node) + (size_t)c * na * nb * Lk) : NULL;
			double * xup = ((c == WIx) ? s->redge[iupd * g->npatch + p] : s->node[iupd * g->npatch + p]) + (size_t)c * na * nb * Lk;
#define XI(i,j,k) xin[((size_t)(i) * nb + (j)) * Lk + (k)]
#define XU(i,j,k) xup[((size_t)(i) * nb + (j)) * Lk + (k)]
#define JC(i,j,k) Jc[((size_t)(i) * nb + (j)) * Lk + (k)]
			for (int a = 0; a < P->nea; a++)
			for (int b = 0; b < P->neb; b++)
			for (int k = 0; k < Lk; k++) {
				const int eA = a * NP + 1, eB = b * NP + 1;
				double ga[NP][NP], gb[NP][NP], psi[NP][NP];
				for (int i = 0; i < NP; i++)
				for (int j = 0; j < NP; j++) {
					psi[i][j] = XI(eA + i, eB + j, k);
					if (xrf) psi[i][j] -= xrf[((size_t)(eA + i) * nb + (eB + j)) * Lk + k];
				}
				for (int i = 0; i < NP; i++)
				for (int j = 0; j < NP; j++) {
					const int iA = eA + i, iB = eB + j;
					double da = 0.0, db = 0.0;
					for (int q = 0; q < NP; q++) {
						da += psi[q][j] * D[q * NP + i];
						db += psi[i][q] * D[q * NP + j];
					}
					da *= inv_da; db *= inv_db;
					ga[i][j] = JC(iA, iB, k) * (C2A[(iA * nb + iB) * 2 + 0] * da + C2A[(iA * nb + iB) * 2 + 1] * db);
					gb[i][j] = JC(iA, iB, k) * (C2B[(iA * nb + iB) * 2 + 0] * da + C2B[(iA * nb + iB) * 2 + 1] * db);
				}
				for (int i = 0; i < NP; i++)
				for (int j = 0; j < NP; j++) {
					const int iA = eA + i, iB = eB + j;
					const double invJ = 1.0 / JC(iA, iB, k);
					double ua = 0.0, ub = 0.0;
					for (int q = 0; q < NP; q++) {
						ua += ga[q][j] * S[i * NP + q];
						ub += gb[i][q] * S[j * NP + q];
					}
					ua *= inv_da; ub *= inv_db;
					XU(iA, iB, k) -= dt * invJ * local_nu * (ua + ub);
				}
			}
#undef XI
#undef XU
#undef JC
		}
	}
}

static void scalar_hyperdiff(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nu, int scale_locally) {
	scalar_hyperdiff_c(g, s, iinit, iupd, dt, nu, scale_locally, -1, 0);
}

/* ApplyScalarHyperdiffusion, tracer branch (iType == 1, :1999-2035): the same Laplacian on every tracer density */
static void tracer_hyperdiff(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nu, int scale_locally) {
	if (!has_tracers(g, s)) return;
	const int L = g->L;
	const double * D = g->dx, * S = g->stiff;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const double inv_da = 1.0 / patch_da(g, P), inv_db = 1.0 / patch_db(g, P);
		double local_nu = nu;
		if (scale_locally && g->ref_length != 0.0) local_nu *= pow(patch_da(g, P) / g->ref_length, 3.2);
		const int na = P->na, nb = P->nb, Lk = L;
		const double * C2A = P->g[ORC_G_CM2A], * C2B = P->g[ORC_G_CM2B];
		const double * Jc = P->g[ORC_G_JAC];
		for (int c = 0; c < g->ntracers; c++) {
			const double * xin = s->tracer[iinit * g->npatch + p] + (size_t)c * na * nb * Lk;
			double * xup = s->tracer[iupd * g->npatch + p] + (size_t)c * na * nb * Lk;
#define XI(i,j,k) xin[((size_t)(i) * nb + (j)) * Lk + (k)]
#define XU(i,j,k) xup[((size_t)(i) * nb + (j)) * Lk + (k)]
#define JC(i,j,k) Jc[((size_t)(i) * nb + (j)) * Lk + (k)]
			for (int a = 0; a < P->nea; a++)
			for (int b = 0; b < P->neb; b++)
			for (int k = 0; k < Lk; k++) {
				const int eA = a * NP + 1, eB = b * NP + 1;
				double ga[NP][NP], gb[NP][NP];
				for (int i = 0; i < NP; i++)
				for (int j = 0; j < NP; j++) {
					const int iA = eA + i, iB = eB + j;
					double da = 0.0, db = 0.0;
					for (int q = 0; q < NP; q++) {
						da += XI(eA + q, iB, k) * D[q * NP + i];
						db += XI(iA, eB + q, k) * D[q * NP + j];
					}
					da *= inv_da; db *= inv_db;
					ga[i][j] = JC(iA, iB, k) * (C2A[(iA * nb + iB) * 2 + 0] * da + C2A[(iA * nb + iB) * 2 + 1] * db);
					gb[i][j] = JC(iA, iB, k) * (C2B[(iA * nb + iB) * 2 + 0] * da + C2B[(iA * nb + iB) * 2 + 1] * db);
				}
				for (int i = 0; i < NP; i++)
				for (int j = 0; j < NP; j++) {
					const int iA = eA + i, iB = eB + j;
					const double invJ = 1.0 / JC(iA, iB, k);
					double ua = 0.0, ub = 0.0;
					for (int q = 0; q < NP; q++) {
						ua += ga[q][j] * S[i * NP + q];
						ub += gb[i][q] * S[j * NP + q];
					}
					ua *= inv_da; ub *= inv_db;
					XU(iA, iB, k) -= dt * invJ * local_nu * (ua + ub);
				}
			}
#undef XI
#undef XU
#undef JC
		}
	}
}

/* from_ref: iDataInitial == DATA_INDEX_REFERENCE, the source is the reference state (:2249-2253, :2304-2308) */
static void vector_hyperdiff_s(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nud, double nuv, int scale_locally, int from_ref) {
	const int L = g->L;
	const double * D = g->dx, * S = g->stiff;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const double inv_da = 1.0 / patch_da(g, P), inv_db = 1.0 / patch_db(g, P);
		double lnd = nud, lnv = nuv;
		if (scale_locally && g->ref_length != 0.0) {
			lnd = lnd * pow(patch_da(g, P) / g->ref_length, 3.2);
			lnv = lnv * pow(patch_da(g, P) / g->ref_length, 3.2);
		}
		const int na = P->na, nb = P->nb;
		const double * J2 = P->g[ORC_G_JAC2D];
		const double * C2A = P->g[ORC_G_CM2A], * C2B = P->g[ORC_G_CM2B];
		const double * nin = from_ref ? P->ref_node : s->node[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		for (int a = 0; a < P->nea; a++)
		for (int b = 0; b < P->neb; b++)
		for (int k = 0; k < L; k++) {
			const int eA = a * NP + 1, eB = b * NP + 1;
			double cua[NP][NP], cub[NP][NP], dv[NP][NP], cl[NP][NP];
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				cua[i][j] = C2A[(iA * nb + iB) * 2 + 0] * NI(UIx, iA, iB, k) + C2A[(iA * nb + iB) * 2 + 1] * NI(VIx, iA, iB, k);
				cub[i][j] = C2B[(iA * nb + iB) * 2 + 0] * NI(UIx, iA, iB, k) + C2B[(iA * nb + iB) * 2 + 1] * NI(VIx, iA, iB, k);
			}
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				double daub = 0, dbua = 0, dajua = 0, dbjub = 0;
				for (int q = 0; q < NP; q++) {
					daub += NI(VIx, eA + q, iB, k) * D[q * NP + i];
					dbua += NI(UIx, iA, eB + q, k) * D[q * NP + j];
					dajua += J2[(eA + q) * nb + iB] * cua[q][j] * D[q * NP + i];
					dbjub += J2[iA * nb + eB + q] * cub[i][q] * D[q * NP + j];
				}
				daub *= inv_da; dbua *= inv_db; dajua *= inv_da; dbjub *= inv_db;
				const double invJ2 = 1.0 / J2[iA * nb + iB];
				dv[i][j] = (dajua + dbjub) * invJ2;
				cl[i][j] = (daub - dbua) * invJ2;
			}
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				double dad = 0, dbd = 0, dac = 0, dbc = 0;
				for (int q = 0; q < NP; q++) {
					dad -= S[i * NP + q] * dv[q][j];
					dbd -= S[j * NP + q] * dv[i][q];
					dac -= S[i * NP + q] * cl[q][j];
					dbc -= S[j * NP + q] * cl[i][q];
				}
				dad *= inv_da; dbd *= inv_db; dac *= inv_da; dbc *= inv_db;
				const double upa = lnd * dad - lnv * J2[iA * nb + iB] * (C2B[(iA * nb + iB) * 2 + 0] * dac + C2B[(iA * nb + iB) * 2 + 1] * dbc);
				const double upb = lnd * dbd + lnv * J2[iA * nb + iB] * (C2A[(iA * nb + iB) * 2 + 0] * dac + C2A[(iA * nb + iB) * 2 + 1] * dbc);
				NU(UIx, iA, iB, k) -= dt * upa;
				NU(VIx, iA, iB, k) -= dt * upb;
			}
		}
	}
}

static void vector_hyperdiff(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt, double nud, double nuv, int scale_locally) {
	vector_hyperdiff_s(g, s, iinit, iupd, dt, nud, nuv, scale_locally, 0);
}

/* HorizontalDynamicsFEM::ApplyRayleighFriction (HorizontalDynamicsFEM.cpp:2420-2570): where the strength is
 * non-zero, ten sub-cycled backward-Euler relaxations of U, V, rho*theta (levels) and W (interfaces) towards
 * the reference state; rho is not touched (nEffectiveC, :2455-2461).  Interior nodes only. */
static void rayleigh_friction(const orc_grid * g, orc_state * s, int iupd, double dt) {
	const int L = g->L;
	const int ncycles = 10;
	const double factor = 1.0 / ncycles;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		if (!P->ray_node) continue;
		const int na = P->na, nb = P->nb;
		double * nup = s->node[iupd * g->npatch + p];
		double * eup = s->redge[iupd * g->npatch + p];
		for (int i = 1; i < na - 1; i++)
		for (int j = 1; j < nb - 1; j++) {
			for (int k = 0; k < L; k++) {
				const double nu = P->ray_node[((size_t)i * nb + j) * L + k];
				if (nu == 0.0) continue;
				for (int c = 0; c < 4; c++) {
					if (c == 3) continue;                  /* W lives on interfaces */
					const size_t o = (((size_t)c * na + i) * nb + j) * L + k;
					for (int si = 0; si < ncycles; si++) {
						const double f = 1.0 / (1.0 + factor * dt * nu);
						nup[o] = f * nup[o] + (1.0 - f) * P->ref_node[o];
					}
				}
			}
			for (int k = 0; k <= L; k++) {
				const double nu = P->ray_redge[((size_t)i * nb + j) * (L + 1) + k];
				if (nu == 0.0) continue;
				const size_t o = (((size_t)3 * na + i) * nb + j) * (L + 1) + k;
				for (int si = 0; si < ncycles; si++) {
					const double f = 1.0 / (1.0 + factor * dt * nu);
					eup[o] = f * eup[o] + (1.0 - f) * P->ref_redge[o];
				}
			}
		}
	}
}

void orc_h_step_after_subcycle(const orc_grid * g, orc_state * s, int iinit, int iupd, int iwork, double dt) {
	orc_copy(g, s, iinit, iupd);
	if (g->nu_scalar == 0.0 && g->nu_div == 0.0 && g->nu_vort == 0.0) {
		/* no hyperdiffusion */
	} else if (g->hypervis_order == 0) {
	} else if (g->hypervis_order == 2) {
		/* viscosity (HorizontalDynamicsFEM.cpp:2672-2686): one pass, coefficients not scaled with the grid spacing */
		scalar_hyperdiff(g, s, iinit, iupd, dt, g->nu_scalar, 0);
		tracer_hyperdiff(g, s, iinit, iupd, dt, g->nu_scalar, 0);
		vector_hyperdiff(g, s, iinit, iupd, -dt, g->nu_div, g->nu_vort, 0);
		filter_negative_tracers_h(g, s, iupd);
		orc_apply_dss(g, s, iupd);
	} else {
		orc_zero(g, s, iwork);
		scalar_hyperdiff(g, s, iinit, iwork, 1.0, 1.0, 0);
		tracer_hyperdiff(g, s, iinit, iwork, 1.0, 1.0, 0);
		vector_hyperdiff(g, s, iinit, iwork, 1.0, 1.0, 1.0, 0);
		orc_apply_dss(g, s, iwork);
		scalar_hyperdiff(g, s, iwork, iupd, -dt, g->nu_scalar, 1);
		tracer_hyperdiff(g, s, iwork, iupd, -dt, g->nu_scalar, 1);
		vector_hyperdiff(g, s, iwork, iupd, -dt, g->nu_div, g->nu_vort, 1);
		filter_negative_tracers_h(g, s, iupd);       /* HorizontalDynamicsFEM.cpp:2707 */
		orc_apply_dss(g, s, iupd);
	}
	/* APPLY_RAYLEIGH_WITH_HYPERVIS (Defines.h:70, HorizontalDynamicsFEM.cpp:2719-2724) */
	if (!g->shallow_water) rayleigh_friction(g, s, iupd, dt);
}

/* ------------------------------------------------------------------------------------------ */
/* Column physics: HeldSuarezPhysics::Perform (src/atm/HeldSuarezPhysics.cpp:60-301), Lorenz staggering   */
/* (rho*theta on levels).  Boundary-layer friction on U,V by backward Euler, Newtonian relaxation of       */
/* rho*theta towards the Held-Suarez equilibrium temperature with the "Ullrich update" (:239-247).         */
/* Quirks of the reference kept as they are: the friction's sigma uses PressureFromRhoTheta(rho * rhotheta) */
/* (:107-110), and the surface pressure comes from the interface slots of rho and rho*theta (:113-116),    */
/* scratch entries with this staggering that only the whole-array stage algebra (orc_copy / orc_zero /      */
/* orc_linear_combine, as Grid::CopyData ...) touches -- read here exactly as the reference reads them.     */
/* orc_patch.surf_p, when not NULL, pins the value instead (tests of the forcing in isolation).             */

void orc_held_suarez(const orc_grid * g, orc_state * s, int ix, double dt) {
	const int L = g->L;
	const double gamma = g->cp / (g->cp - g->Rd), kappa = g->Rd / g->cp;
	const double pscal = g->p0 * pow(g->Rd / g->p0, gamma);
	const double bsig = 0.7, kf = 1.0 / 86400.0, ka = (1.0 / 40.0) / 86400.0, ks = (1.0 / 4.0) / 86400.0;
	const double dty = 60.0, dthz = 10.0, tmin = 200.0, tmax = 315.0;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		double * nn = s->node[ix * g->npatch + p];
		const double * ee = s->redge[ix * g->npatch + p];
#define NN(c,i,j,k) nn[(((size_t)(c) * na + (i)) * nb + (j)) * L + (k)]
#define EE(c,i,j,k) ee[(((size_t)(c) * na + (i)) * nb + (j)) * (L + 1) + (k)]
		for (int i = 1; i < na - 1; i++)
		for (int j = 1; j < nb - 1; j++) {
			const double ps = P->surf_p ? P->surf_p[(size_t)i * nb + j]
				: pscal * exp(log(EE(RIx, i, j, 0) * EE(PIx, i, j, 0)) * gamma);
			for (int k = 0; k < L; k++) {
				const double pr = pscal * exp(log(NN(RIx, i, j, k) * NN(PIx, i, j, k)) * gamma);
				const double sigma = pr / ps;
				double bs = (sigma - bsig) / (1.0 - bsig);
				if (bs < 0.0) bs = 0.0;
				NN(UIx, i, j, k) /= (1.0 + kf * bs * dt);
				NN(VIx, i, j, k) /= (1.0 + kf * bs * dt);
			}
			for (int k = 0; k < L; k++) {
				const double pr = pscal * exp(log(NN(PIx, i, j, k)) * gamma);
				const double sigma = pr / ps;
				double bs = (sigma - bsig) / (1.0 - bsig);
				if (bs < 0.0) bs = 0.0;
				const double T = pr / (NN(RIx, i, j, k) * g->Rd);
				const double lat = P->lat[(size_t)i * nb + j];
				const double sl = sin(lat), cl = cos(lat);
				const double c4 = cl * cl * cl * cl;
				const double kt = ka + (ks - ka) * bs * c4;
				double teq = tmax - dty * sl * sl - dthz * log(pr / g->p0) * cl * cl;
				teq *= pow(pr / g->p0, kappa);
				if (teq < tmin) teq = tmin;
				const double ddh = -kt / gamma * (1.0 + (gamma - 1.0) * teq / T);
				const double dh = -kt / gamma * (1.0 - teq / T);
				NN(PIx, i, j, k) *= 1.0 + dt / (1.0 - dt * ddh) * dh;
			}
		}
#undef NN
#undef EE
	}
}

/* ------------------------------------------------------------------------------------------ */
/* Column physics: Kessler warm-rain microphysics of the DCMIP2016 supercell (BASELINE config 4).           */
/* orc_kessler_column restates SUBROUTINE KESSLER (test/dcmip2016/interface/kessler.f90:64-185) as the       */
/* reference's build compiles it: the makefiles pass no default-real-8 flag (mk/defs.make:35: F90FLAGS += -O3), */
/* so the locals r, rhalf, velqr, sed, pc (kessler.f90:93) are SINGLE precision, real literals without a d0   */
/* exponent (.001, 0.1364, .875, .2046, .525, .5, 1.) are single precision values promoted to double, and the  */
/* specific intrinsics AMAX1 / AMIN1 convert their arguments to single precision and return single precision  */
/* (verified on the object amdflang -O3 produces: tests/test_oracle_vs_reference.py::test_kessler_column_*,    */
/* bit for bit on the committed vectors of tests/golden/kessler_columns.npz).  MAX / DIM are generic: double.  */

static inline double f4(double x) { return (double)(float)x; }
static inline double amax1_2(double a, double b) { const float x = (float)a, y = (float)b; return (double)((x > y) ? x : y); }
static inline double amin1_3(double a, double b, double c) {
	const float x = (float)a, y = (float)b, z = (float)c;
	const float m = (x < y) ? x : y;
	return (double)((m < z) ? m : z);
}

void orc_kessler_column(double * theta, double * qv, double * qc, double * qr, const double * rho, const double * pk,
	double dt, const double * z, int nz, double * precl_out)
{
	float * r = (float *)malloc(sizeof(float) * 5 * nz);
	float * rhalf = r + nz, * velqr = r + 2 * nz, * sed = r + 3 * nz, * pc = r + 4 * nz;
	const double f2x = 17.27;
	const double f5 = 237.3 * f2x * 2500000.0 / 1003.0;
	const double xk = .2875, psl = 1000.0, rhoqr = 1000.0;
	const double e1364 = (double)0.1364f, e875 = (double)0.875f, e2046 = (double)0.2046f, e525 = (double)0.525f;
	const double c001 = (double)0.001f, half = (double)0.5f;
	const double inv_xk = (double)1.0f / xk;
	for (int k = 0; k < nz; k++) {                                   /* :107-115 */
		r[k] = (float)(0.001 * rho[k]);
		rhalf[k] = (float)sqrt(rho[0] / rho[k]);
		pc[k] = (float)(3.8 / (pow(pk[k], inv_xk) * psl));
		velqr[k] = (float)(36.34 * pow(qr[k] * (double)r[k], e1364) * (double)rhalf[k]);
	}
	double dt_max = dt;                                              /* :118-123 */
	for (int k = 0; k < nz - 1; k++)
		if (velqr[k] != 0.0f) {
			const double c = 0.8 * (z[k + 1] - z[k]) / (double)velqr[k];
			dt_max = (dt_max < c) ? dt_max : c;
		}
	const int rainsplit = (int)ceil(dt / dt_max);                    /* :126-127 */
	const double dt0 = dt / (double)rainsplit;
	double precl = 0.0;
	for (int nt = 1; nt <= rainsplit; nt++) {
		precl = precl + rho[0] * qr[0] * (double)velqr[0] / rhoqr;   /* :135 */
		for (int k = 0; k < nz - 1; k++)                             /* :138-141 */
			sed[k] = (float)(dt0 * ((double)r[k + 1] * qr[k + 1] * (double)velqr[k + 1] - (double)r[k] * qr[k] * (double)velqr[k])
				/ ((double)r[k] * (z[k + 1] - z[k])));
		sed[nz - 1] = (float)(-dt0 * qr[nz - 1] * (double)velqr[nz - 1] / (half * (z[nz - 1] - z[nz - 2])));
		for (int k = 0; k < nz; k++) {                               /* :144-165 */
			const double qrprod = qc[k] - (qc[k] - dt0 * amax1_2(c001 * (qc[k] - .001), 0.0)) / (1.0 + dt0 * 2.2 * pow(qr[k], e875));
			qc[k] = amax1_2(qc[k] - qrprod, 0.0);
			qr[k] = amax1_2(qr[k] + qrprod + (double)sed[k], 0.0);
			const double pt = pk[k] * theta[k];
			const double qvs = (double)pc[k] * exp(f2x * (pt - 273.0) / (pt - 36.0));
			const double pt36 = pt - 36.0;
			const double prod = (qv[k] - qvs) / (1.0 + qvs * f5 / (pt36 * pt36));
			const double rq = (double)r[k] * qr[k];
			const double dimv = (qvs - qv[k] > 0.0) ? qvs - qv[k] : 0.0;
			const double ern = amin1_3(dt0 * (((1.6 + 124.9 * pow(rq, e2046)) * pow(rq, e525))
					/ (2550000.0 * (double)pc[k] / (3.8 * qvs) + 540000.0)) * (dimv / ((double)r[k] * qvs)),
				amax1_2(-prod - qc[k], 0.0), qr[k]);
			const double mx = (prod > -qc[k]) ? prod : -qc[k];
			theta[k] = theta[k] + 2500000.0 / (1003.0 * pk[k]) * (amax1_2(prod, -qc[k]) - ern);
			qv[k] = amax1_2(qv[k] - mx + ern, 0.0);
			qc[k] = qc[k] + mx;
			qr[k] = qr[k] - ern;
		}
		if (nt != rainsplit)                                         /* :168-172 */
			for (int k = 0; k < nz; k++)
				velqr[k] = (float)(36.34 * pow(qr[k] * (double)r[k], e1364) * (double)rhalf[k]);
	}
	*precl_out = precl / (double)rainsplit;
	free(r);
}

/* KesslerPhysics::Perform (test/dcmip2016/KesslerPhysics.cpp:83-285), Lorenz staggering / FORMULATION_RHOTHETA_PI:
 * rho*theta on levels (:156-158, :265-268).  tracers 0,1,2 = RhoQv, RhoQc, RhoQr.  zlev[p] = GridPatch::GetZLevels()
 * [na][nb][L]; prect[p] (may be NULL) = UserData2D item 0 [na][nb], accumulated precipitation (:234). */
void orc_kessler(const orc_grid * g, orc_state * s, int ix, double dt, const double * const * zlev, double * const * prect) {
	const int L = g->L;
	const double gamma = g->cp / (g->cp - g->Rd);
	const double pscal = g->p0 * pow(g->Rd / g->p0, gamma);
	double * buf = (double *)malloc(sizeof(double) * 8 * L);
	double * qv = buf, * qc = buf + L, * qr = buf + 2 * L, * rho = buf + 3 * L, * zc = buf + 4 * L, * pk = buf + 5 * L,
		* th = buf + 6 * L, * thv = buf + 7 * L;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		double * nn = s->node[ix * g->npatch + p];
		double * tr = s->tracer[ix * g->npatch + p];
#define NN(c,i,j,k) nn[(((size_t)(c) * na + (i)) * nb + (j)) * L + (k)]
#define TR(c,i,j,k) tr[(((size_t)(c) * na + (i)) * nb + (j)) * L + (k)]
		for (int i = 1; i < na - 1; i++)
		for (int j = 1; j < nb - 1; j++) {
			for (int k = 0; k < L; k++) thv[k] = NN(PIx, i, j, k) / NN(RIx, i, j, k);
			for (int k = 0; k < L; k++) {
				const double dRho = NN(RIx, i, j, k);
				const double dRhoD = dRho - TR(0, i, j, k) - TR(1, i, j, k) - TR(2, i, j, k);
				const double pr = pscal * exp(log(dRho * thv[k]) * gamma);
				const double tv = pr / (dRho * g->Rd);
				qv[k] = TR(0, i, j, k) / NN(RIx, i, j, k); if (qv[k] < 0.0) qv[k] = 0.0;
				qc[k] = TR(1, i, j, k) / NN(RIx, i, j, k); if (qc[k] < 0.0) qc[k] = 0.0;
				qr[k] = TR(2, i, j, k) / NN(RIx, i, j, k); if (qr[k] < 0.0) qr[k] = 0.0;
				th[k] = thv[k] / (1.0 + 0.61 * qv[k]);
				rho[k] = dRhoD;
				zc[k] = zlev[p][((size_t)i * nb + j) * L + k];
				pk[k] = tv / thv[k];
			}
			double precl = 0.0;
			orc_kessler_column(th, qv, qc, qr, rho, pk, dt, zc, L, &precl);
			if (prect && prect[p]) prect[p][(size_t)i * nb + j] += precl * dt;
			for (int k = 0; k < L; k++) {
				NN(RIx, i, j, k) = rho[k] / (1.0 - qv[k] - qc[k] - qr[k]);
				TR(0, i, j, k) = qv[k] * NN(RIx, i, j, k);
				TR(1, i, j, k) = qc[k] * NN(RIx, i, j, k);
				TR(2, i, j, k) = qr[k] * NN(RIx, i, j, k);
			}
			for (int k = 0; k < L; k++)
				NN(PIx, i, j, k) = NN(RIx, i, j, k) * th[k] * (1.0 + 0.61 * qv[k]);
		}
#undef NN
#undef TR
	}
	free(buf);
}

/* ------------------------------------------------------------------------------------------ */
/* ARS(3,4,3) IMEX step in U-form: TimestepSchemeARS343 (TimestepSchemeARS343.cpp:25-236)      */

static void ars343_tableau(double * u2, double * u3, double * u4, double * dexp, double * dimp) {
	const double gm = 0.4358665215084590;
	const double b1 = -1.5 * gm * gm + 4.0 * gm - 0.25;
	const double b2 = 1.5 * gm * gm - 5.0 * gm + 1.25;
	const double a42 = 0.5529291480359398, a43 = 0.5529291480359398;
	const double a31 = (1.0 - 4.5 * gm + 1.5 * gm * gm) * a42 + (2.75 - 10.5 * gm + 3.75 * gm * gm) * a43
		- 3.5 + 13 * gm - 4.5 * gm * gm;
	const double a32 = (-1.0 + 4.5 * gm - 1.5 * gm * gm) * a42 + (-2.75 + 10.5 * gm - 3.75 * gm * gm) * a43
		+ 4.0 - 12.5 * gm + 4.5 * gm * gm;
	const double a41 = 1.0 - a42 - a43;
	const double I[4][4] = { { gm, 0, 0, 0 }, { 0.5 * (1.0 - gm), gm, 0, 0 }, { b1, b2, gm, 0 }, { b1, b2, gm, 0 } };
	const double E[4][4] = { { gm, 0, 0, 0 }, { a31, a32, 0, 0 }, { a41, a42, a43, 0 }, { 0, b1, b2, gm } };
	for (int i = 0; i < 4; i++) { dexp[i] = E[i][i]; dimp[i] = I[i][i]; }
	for (int i = 0; i < 7; i++) u2[i] = u3[i] = u4[i] = 0.0;
	u2[0] = 1.0 - E[1][0] / E[0][0];
	u2[1] = E[1][0] / E[0][0] - I[1][0] / I[0][0];
	u2[2] = I[1][0] / I[0][0];
	u3[0] = 1.0 - E[2][0] / E[0][0];
	u3[1] = E[2][0] / E[0][0] - I[2][0] / I[0][0];
	u3[2] = I[2][0] / I[0][0];
	u3[3] = E[2][1] / E[1][1] - I[2][1] / I[1][1];
	u3[4] = I[2][1] / I[1][1];
	const double u37 = -E[2][1] / E[1][1];
	u4[0] = 1.0 - E[3][0] / E[0][0];
	u4[1] = E[3][0] / E[0][0] - I[3][0] / I[0][0];
	u4[2] = I[3][0] / I[0][0];
	u4[3] = E[3][1] / E[1][1] - I[3][1] / I[1][1];
	u4[4] = I[3][1] / I[1][1];
	u4[5] = E[3][2] / E[2][2] - I[3][2] / I[2][2];
	u4[6] = I[3][2] / I[2][2];
	const double u47 = -E[3][1] / E[1][1];
	const double u48 = -E[3][2] / E[2][2];
	u3[0] += u37 * u2[0]; u3[1] += u37 * u2[1]; u3[2] += u37 * u2[2];
	u4[0] += u47 * u2[0] + u48 * u3[0];
	u4[1] += u47 * u2[1] + u48 * u3[1];
	u4[2] += u47 * u2[2] + u48 * u3[2];
	u4[3] += u48 * u3[3];
	u4[4] += u48 * u3[4];
}

int orc_step_ars343(const orc_grid * g, orc_state * s, double dt) {
	double u2[7], u3[7], u4[7], de[4], di[4];
	int info = 0, r;
	ars343_tableau(u2, u3, u4, de, di);
	orc_copy(g, s, 0, 1);
	orc_h_step_explicit(g, s, 0, 1, de[0] * dt);
	orc_v_step_explicit(g, s, 0, 1, de[0] * dt);
	orc_apply_dss(g, s, 1);
	orc_copy(g, s, 1, 2);
	r = orc_v_step_implicit(g, s, 2, 2, di[0] * dt); if (r && !info) info = r;

	orc_linear_combine(g, s, u2, 7, 3);
	orc_h_step_explicit(g, s, 2, 3, de[1] * dt);
	orc_v_step_explicit(g, s, 2, 3, de[1] * dt);
	orc_apply_dss(g, s, 3);
	orc_copy(g, s, 3, 4);
	r = orc_v_step_implicit(g, s, 4, 4, di[1] * dt); if (r && !info) info = r;

	orc_linear_combine(g, s, u3, 7, 5);
	orc_h_step_explicit(g, s, 4, 5, de[2] * dt);
	orc_v_step_explicit(g, s, 4, 5, de[2] * dt);
	orc_apply_dss(g, s, 5);
	orc_copy(g, s, 5, 6);
	r = orc_v_step_implicit(g, s, 6, 6, di[2] * dt); if (r && !info) info = r;

	orc_linear_combine(g, s, u4, 7, 1);
	orc_h_step_explicit(g, s, 6, 1, de[3] * dt);
	orc_v_step_explicit(g, s, 6, 1, de[3] * dt);
	orc_apply_dss(g, s, 1);

	orc_copy(g, s, 1, 0);
	orc_h_step_after_subcycle(g, s, 1, 0, 2, dt);
	return info;
}

/* ------------------------------------------------------------------------------------------ */
/* VerticalDynamicsFEM::StepImplicitTermsExplicitly (VerticalDynamicsFEM.cpp:439-612):         */
/* update -= dt * F(initial) on every interior column                                          */

void orc_v_step_implicit_terms_explicitly(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt) {
	const int L = g->L;
	const int n = FTOT * (L + 1);
	double * x0 = dalloc(n), * F = dalloc(n), * AB = dalloc(n * LDAB);
	colwork w; colwork_init(&w, L);
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		const double * nin = s->node[iinit * g->npatch + p];
		const double * ein = s->redge[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		double * eup = s->redge[iupd * g->npatch + p];
		for (int iA = 1; iA < na - 1; iA++)
		for (int iB = 1; iB < nb - 1; iB++) {
			column_f_jacobian(g, P, nin, ein, iA, iB, dt, x0, F, AB, &w);
			for (int k = 0; k < L; k++) {
				NU(PIx, iA, iB, k) -= dt * F[VECF(FP, k)];
				NU(RIx, iA, iB, k) -= dt * F[VECF(FR, k)];
			}
			for (int k = 0; k <= L; k++) EU(WIx, iA, iB, k) -= dt * F[VECF(FW, k)];
			if (has_tracers(g, s))      /* UpdateColumnTracers(dt, initial, update, ...), VerticalDynamicsFEM.cpp:600-608 */
				update_column_tracers(g, P, iA, iB, dt, x0, &w, eup, s->tracer[iinit * g->npatch + p], s->tracer[iupd * g->npatch + p]);
		}
	}
	colwork_free(&w);
	free(x0); free(F); free(AB);
}

/* VerticalDynamicsFEM::StepExplicit with fFullyExplicit (--explicitvertical; :745-812, :984-1105): per column
 * Evaluate (= PrepareColumn + BuildF at the initial column) and rho*theta, W, rho -= dt F; tracers by
 * UpdateColumnTracers in its explicit branch; then the upwind penalty of U, V with dt |xi_dot| of that column and,
 * with uniform diffusion, the vertical diffusion of (U, V) - (U, V)_ref. */
static void v_step_fully_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt) {
	const int L = g->L;
	const int n = FTOT * (L + 1);
	double * x0 = dalloc(n), * F = dalloc(n), * AB = dalloc(n * LDAB);
	double * wts = dalloc(L + 1), * buf = dalloc(L + 2), * dd = dalloc(L + 1), * ddr = dalloc(L + 1), * col = dalloc(L + 1);
	colwork w; colwork_init(&w, L);
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		const double * nin = s->node[iinit * g->npatch + p];
		const double * ein = s->redge[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		double * eup = s->redge[iupd * g->npatch + p];
		for (int iA = 1; iA < na - 1; iA++)
		for (int iB = 1; iB < nb - 1; iB++) {
			column_f_jacobian(g, P, nin, ein, iA, iB, dt, x0, F, AB, &w);
			for (int k = 0; k < L; k++) NU(PIx, iA, iB, k) -= dt * F[VECF(FP, k)];
			for (int k = 0; k <= L; k++) EU(WIx, iA, iB, k) -= dt * F[VECF(FW, k)];
			for (int k = 0; k < L; k++) NU(RIx, iA, iB, k) -= dt * F[VECF(FR, k)];
			if (has_tracers(g, s))
				update_column_tracers(g, P, iA, iB, dt, x0, &w, eup, s->tracer[iinit * g->npatch + p], s->tracer[iupd * g->npatch + p]);
			for (int a = 0; a < L - 1; a++) wts[a] = dt * fabs(w.xde[a + 1]);
			penalty_apply(g, wts, &NI(UIx, iA, iB, 0), &NU(UIx, iA, iB, 0), buf);
			penalty_apply(g, wts, &NI(VIx, iA, iB, 0), &NU(VIx, iA, iB, 0), buf);
			if (g->uniform_diffusion) {
				const double cf = g->udiff_vector / (g->ztop * g->ztop);
				for (int c = UIx; c <= VIx; c++) {
					op_apply(g, ORC_OP_DIFFDIFF_N2N, c == UIx ? w.un : w.vn, dd);
					for (int k = 0; k < L; k++) col[k] = P->ref_node[(((size_t)c * na + iA) * nb + iB) * L + k];
					op_apply(g, ORC_OP_DIFFDIFF_N2N, col, ddr);
					for (int k = 0; k < L; k++) NU(c, iA, iB, k) += dt * cf * (dd[k] - ddr[k]);
				}
			}
		}
	}
	colwork_free(&w);
	free(x0); free(F); free(AB); free(wts); free(buf); free(dd); free(ddr); free(col);
}

/* ------------------------------------------------------------------------------------------ */
/* the other steppers of the path, call for call as the reference issues them                   */

static void hvd(const orc_grid * g, orc_state * s, int i, int u, double c) {
	orc_h_step_explicit(g, s, i, u, c);
	orc_v_step_explicit(g, s, i, u, c);
	orc_apply_dss(g, s, u);
}

static double g_strang_offc = 0.0;      /* TimestepSchemeStrang constructor argument dOffCentering */
void orc_set_strang_off_centering(double oc) { g_strang_offc = oc; }

/* scheme: 1 ARS232 (TimestepSchemeARS232.cpp:52-150), 2 ARS222 (TimestepSchemeARS222.cpp:51-119),
 * 3 ARS443 (TimestepSchemeARS443.cpp), 4 ARK232 (TimestepSchemeARK232.cpp:53-225),
 * 5 Strang/KGU35 (TimestepSchemeStrang.cpp:450-674), 0 ARS343 */
int orc_step(const orc_grid * g, orc_state * s, int scheme, int first, int last, double dt) {
	int info = 0, r;
#define VI(i, u, c) do { r = orc_v_step_implicit(g, s, i, u, c); if (r && !info) info = r; } while (0)
	if (scheme == 0) return orc_step_ars343(g, s, dt);
	if (scheme == 1) {
		const double gm = 1.0 - 1.0 / sqrt(2.0), dl = -(2.0 * sqrt(2.0)) / 3.0;
		const double I[3][3] = { { gm, 0, 0 }, { 1.0 - gm, gm, 0 }, { 1.0 - gm, gm, 0 } };
		const double E[3][3] = { { gm, 0, 0 }, { dl, 1.0 - dl, 0 }, { 0, 1.0 - gm, gm } };
		double u2[6] = { 1.0 - E[1][0] / E[0][0], E[1][0] / E[0][0] - I[1][0] / I[0][0], I[1][0] / I[0][0], 0, 0, 0 };
		double u3[7] = { 1.0 - E[2][0] / E[0][0], E[2][0] / E[0][0] - I[2][0] / I[0][0], I[2][0] / I[0][0],
			E[2][1] / E[1][1] - I[2][1] / I[1][1], I[2][1] / I[1][1], -E[2][1] / E[1][1], 0 };
		orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, E[0][0] * dt);
		orc_copy(g, s, 1, 2); VI(2, 2, I[0][0] * dt);
		orc_linear_combine(g, s, u2, 6, 5); orc_copy(g, s, 5, 3); hvd(g, s, 2, 3, E[1][1] * dt);
		orc_copy(g, s, 3, 4); VI(4, 4, I[1][1] * dt);
		orc_linear_combine(g, s, u3, 7, 6); hvd(g, s, 4, 6, E[2][2] * dt);
		orc_copy(g, s, 6, 2); orc_h_step_after_subcycle(g, s, 2, 1, 6, dt); orc_copy(g, s, 1, 0);
		return info;
	}
	if (scheme == 2) {
		const double gm = 1.0 - 0.5 * sqrt(2.0), dl = 1.0 - 1.0 / (2.0 * gm);
		const double I[2][2] = { { gm, 0 }, { 1.0 - gm, gm } };
		const double E[2][2] = { { gm, 0 }, { dl, 1.0 - dl } };
		double u2[4] = { 1.0 - E[1][0] / E[0][0], E[1][0] / E[0][0] - I[1][0] / I[0][0], I[1][0] / I[0][0], 0 };
		orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, E[0][0] * dt);
		orc_copy(g, s, 1, 2); VI(2, 2, I[0][0] * dt); orc_apply_dss(g, s, 2);
		orc_linear_combine(g, s, u2, 4, 3); hvd(g, s, 2, 3, E[1][1] * dt);
		VI(3, 3, I[1][1] * dt); orc_apply_dss(g, s, 3);
		orc_copy(g, s, 3, 2); orc_h_step_after_subcycle(g, s, 2, 1, 3, dt); orc_copy(g, s, 1, 0);
		return info;
	}
	if (scheme == 3) {
		const double I[4][4] = { { 1./2., 0, 0, 0 }, { 1./6., 1./2., 0, 0 }, { -1./2., 1./2., 1./2., 0 }, { 3./2., -3./2., 1./2., 1./2. } };
		const double E[4][4] = { { 1./2., 0, 0, 0 }, { 11./18., 1./18., 0, 0 }, { 5./6., -5./6., 1./2., 0 }, { 1./4., 7./4., 3./4., -7./4. } };
		double u2[8] = { 0 }, u3[9] = { 0 }, u4[10] = { 0 };
		u2[0] = 1.0 - E[1][0] / E[0][0]; u2[1] = E[1][0] / E[0][0] - I[1][0] / I[0][0]; u2[2] = I[1][0] / I[0][0];
		u3[0] = 1.0 - E[2][0] / E[0][0]; u3[1] = E[2][0] / E[0][0] - I[2][0] / I[0][0]; u3[2] = I[2][0] / I[0][0];
		u3[3] = E[2][1] / E[1][1] - I[2][1] / I[1][1]; u3[4] = I[2][1] / I[1][1]; u3[7] = -E[2][1] / E[1][1];
		u4[0] = 1.0 - E[3][0] / E[0][0]; u4[1] = E[3][0] / E[0][0] - I[3][0] / I[0][0]; u4[2] = I[3][0] / I[0][0];
		u4[3] = E[3][1] / E[1][1] - I[3][1] / I[1][1]; u4[4] = I[3][1] / I[1][1];
		u4[5] = E[3][2] / E[2][2] - I[3][2] / I[2][2]; u4[6] = I[3][2] / I[2][2];
		u4[7] = -E[3][1] / E[1][1]; u4[8] = -E[3][2] / E[2][2];
		orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, E[0][0] * dt);
		orc_copy(g, s, 1, 2); VI(2, 2, I[0][0] * dt); orc_apply_dss(g, s, 2);
		orc_linear_combine(g, s, u2, 8, 7); orc_copy(g, s, 7, 3); hvd(g, s, 2, 3, E[1][1] * dt);
		orc_copy(g, s, 3, 4); VI(4, 4, I[1][1] * dt); orc_apply_dss(g, s, 4);
		orc_linear_combine(g, s, u3, 9, 8); orc_copy(g, s, 8, 5); hvd(g, s, 4, 5, E[2][2] * dt);
		orc_copy(g, s, 5, 6); VI(6, 6, I[2][2] * dt); orc_apply_dss(g, s, 6);
		orc_linear_combine(g, s, u4, 10, 9); hvd(g, s, 6, 9, E[3][3] * dt);
		VI(9, 9, I[3][3] * dt); orc_apply_dss(g, s, 9);
		orc_copy(g, s, 9, 2); orc_h_step_after_subcycle(g, s, 2, 1, 9, dt); orc_copy(g, s, 1, 0);
		return info;
	}
	if (scheme == 4) {
		const double gm = 1.0 - 1.0 / sqrt(2.0), dl = 1.0 / (2.0 * sqrt(2.0));
		const double al = 1.0 / 6.0 * (3.0 + 2.0 * sqrt(2.0));
		const double I[3][3] = { { gm, gm, 0 }, { dl, dl, gm }, { dl, dl, gm } };
		const double E[3][3] = { { 2.0 * gm, 0, 0 }, { 1.0 - al, al, 0 }, { dl, dl, gm } };
		double u2[7] = { 1.0 - E[1][0] / E[0][0], E[1][0] / E[0][0] - I[1][0] / I[0][0],
			I[1][0] / I[0][0] - I[1][1] / I[0][1], I[1][1] / I[0][1], 0, 0, 0 };
		double u3[8] = { 1.0 - E[2][0] / E[0][0], E[2][0] / E[0][0] - I[2][0] / I[0][0],
			I[2][0] / I[0][0] - I[2][1] / I[0][1], I[2][1] / I[0][1],
			E[2][1] / E[1][1] - I[2][2] / I[1][2], I[2][2] / I[1][2], -E[2][1] / E[1][1], 0 };
		for (int n = 0; n < 2; n++) {
			orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, E[0][0] * dt / 2);
			if (n < 1) orc_copy(g, s, 1, 0);
		}
		orc_copy(g, s, 1, 2); orc_v_step_implicit_terms_explicitly(g, s, 1, 2, I[0][0] * dt); orc_apply_dss(g, s, 2);
		orc_copy(g, s, 2, 3); VI(3, 3, I[0][1] * dt); orc_apply_dss(g, s, 3);
		orc_linear_combine(g, s, u2, 7, 6); orc_copy(g, s, 6, 4); hvd(g, s, 3, 4, E[1][1] * dt);
		orc_copy(g, s, 4, 5); VI(5, 5, I[1][2] * dt); orc_apply_dss(g, s, 5);
		orc_linear_combine(g, s, u3, 8, 7); hvd(g, s, 5, 7, E[2][2] * dt);
		orc_copy(g, s, 7, 2); orc_h_step_after_subcycle(g, s, 7, 1, 3, dt); orc_copy(g, s, 1, 0);
		return info;
	}
	if (scheme >= 5 && scheme <= 9) {
		/* TimestepSchemeStrang::Step (TimestepSchemeStrang.cpp:450-674): 5 KinnmarkGrayUllrich35, 6 ForwardEuler, 7 RungeKutta4,
		 * 8 RungeKuttaSSP3, 9 RungeKuttaSSPRK53 (:484-648, combinations :53-117); off-centering of the constructor */
		const double half = 0.5 * dt, oc = g_strang_offc;
		const double carry[2] = { 1.0, 1.0 }, offc[2] = { (2.0 - oc) / 2.0, oc / 2.0 }, fin[2] = { 1.0, -1.0 };
		if (first) VI(0, 0, half);
		else { orc_linear_combine(g, s, carry, 2, 0); filter_negative_tracers_v(g, s, 0); }      /* TimestepSchemeStrang.cpp:476-481 */
		if (scheme == 6) {
			orc_copy(g, s, 0, 4); hvd(g, s, 0, 4, dt);
		} else if (scheme == 7) {
			const double rk4[5] = { -1.0 / 3.0, 1.0 / 3.0, 2.0 / 3.0, 1.0 / 3.0, 0.0 };
			orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, half);
			orc_copy(g, s, 0, 2); hvd(g, s, 1, 2, half);
			orc_copy(g, s, 0, 3); hvd(g, s, 2, 3, dt);
			orc_linear_combine(g, s, rk4, 5, 4); hvd(g, s, 3, 4, dt / 6.0);
		} else if (scheme == 8) {
			const double ca[3] = { 3.0 / 4.0, 1.0 / 4.0, 0.0 }, cb[5] = { 1.0 / 3.0, 0.0, 2.0 / 3.0, 0.0, 0.0 };
			orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, dt);
			orc_linear_combine(g, s, ca, 3, 2); hvd(g, s, 1, 2, 0.25 * dt);
			orc_linear_combine(g, s, cb, 5, 4); hvd(g, s, 2, 4, (2.0 / 3.0) * dt);
		} else if (scheme == 9) {
			const double ca[4] = { 0.355909775063327, 0.0, 0.644090224936674, 0.0 };
			const double cb[4] = { 0.367933791638137, 0.0, 0.0, 0.632066208361863 };
			const double cc[5] = { 0.762406163401431, 0.0, 0.237593836598569, 0.0, 0.0 };
			orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, 0.377268915331368 * dt);
			orc_copy(g, s, 1, 2); hvd(g, s, 1, 2, 0.377268915331368 * dt);
			orc_linear_combine(g, s, ca, 4, 3); hvd(g, s, 2, 3, 0.242995220537396 * dt);
			orc_linear_combine(g, s, cb, 4, 0); hvd(g, s, 3, 0, 0.238458932846290 * dt);
			orc_linear_combine(g, s, cc, 5, 4); hvd(g, s, 0, 4, 0.287632146308408 * dt);
		} else {
			const double kgu[5] = { -1.0 / 4.0, 5.0 / 4.0, 0.0, 0.0, 0.0 };
			orc_copy(g, s, 0, 1); hvd(g, s, 0, 1, dt / 5.0);
			orc_copy(g, s, 0, 2); hvd(g, s, 1, 2, dt / 5.0);
			orc_copy(g, s, 0, 3); hvd(g, s, 2, 3, dt / 3.0);
			orc_copy(g, s, 0, 2); hvd(g, s, 3, 2, 2.0 * dt / 3.0);
			orc_linear_combine(g, s, kgu, 5, 4); hvd(g, s, 2, 4, 3.0 * dt / 4.0);
		}
		orc_copy(g, s, 4, 1); orc_h_step_after_subcycle(g, s, 4, 1, 2, dt);
		orc_copy(g, s, 1, 0); VI(0, 0, 0.5 * (1.0 + oc) * dt);
		orc_linear_combine(g, s, offc, 2, 0);
		if (!last) orc_linear_combine(g, s, fin, 2, 1);
		return info;
	}
#undef VI
	return -1;
}


/* ------------------------------------------------------------------------------------------ */
/* HorizontalDynamicsFEM::StepShallowWater (HorizontalDynamicsFEM.cpp:321-647); H lives in slot 2 */

static void sw_step_explicit(const orc_grid * g, orc_state * s, int iinit, int iupd, double dt) {
	const int L = 1;
	const double * D = g->dx, * S = g->stiff;
	for (int p = 0; p < g->npatch; p++) {
		const orc_patch * P = &g->patches[p];
		const double inv_da = 1.0 / patch_da(g, P), inv_db = 1.0 / patch_db(g, P);
		const int na = P->na, nb = P->nb;
		const double * nin = s->node[iinit * g->npatch + p];
		double * nup = s->node[iupd * g->npatch + p];
		const double * J2 = P->g[ORC_G_JAC2D], * C2A = P->g[ORC_G_CM2A], * C2B = P->g[ORC_G_CM2B];
		const double * F = P->g[ORC_G_CORIOLIS], * ZS = P->g[ORC_G_TOPO];
		for (int a = 0; a < P->nea; a++)
		for (int b = 0; b < P->neb; b++) {
			const int eA = a * NP + 1, eB = b * NP + 1, k = 0;
			double cua[NP][NP], cub[NP][NP], ke[NP][NP], fa[NP][NP], fb[NP][NP];
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				const double ua = NI(UIx, iA, iB, k), ub = NI(VIx, iA, iB, k);
				cua[i][j] = C2A[(iA * nb + iB) * 2 + 0] * ua + C2A[(iA * nb + iB) * 2 + 1] * ub;
				cub[i][j] = C2B[(iA * nb + iB) * 2 + 0] * ua + C2B[(iA * nb + iB) * 2 + 1] * ub;
				ke[i][j] = 0.5 * (cua[i][j] * ua + cub[i][j] * ub);
				ke[i][j] += g->grav * NI(PIx, iA, iB, k);
			}
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				const double ba = J2[iA * nb + iB] * cua[i][j], bb = J2[iA * nb + iB] * cub[i][j];
				fa[i][j] = ba * (NI(PIx, iA, iB, k) - ZS[iA * nb + iB]);
				fb[i][j] = bb * (NI(PIx, iA, iB, k) - ZS[iA * nb + iB]);
			}
			for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++) {
				const int iA = eA + i, iB = eB + j;
				const double invJ2 = 1.0 / J2[iA * nb + iB];
				double daub = 0, dbua = 0, dake = 0, dbke = 0, dafa = 0, dbfb = 0;
				for (int q = 0; q < NP; q++) {
					dafa -= fa[q][j] * S[i * NP + q];
					daub += NI(VIx, eA + q, iB, k) * D[q * NP + i];
					dake += ke[q][j] * D[q * NP + i];
				}
				for (int q = 0; q < NP; q++) {
					dbfb -= fb[i][q] * S[j * NP + q];
					dbua += NI(UIx, iA, eB + q, k) * D[q * NP + j];
					dbke += ke[i][q] * D[q * NP + j];
				}
				dafa *= inv_da; daub *= inv_da; dake *= inv_da;
				dbfb *= inv_db; dbua *= inv_db; dbke *= inv_db;
				double upa = 0.0, upb = 0.0;
				const double zeta = (daub - dbua);
				const double ucza = cub[i][j] * zeta, uczb = -cua[i][j] * zeta;
				upa += F[iA * nb + iB] * J2[iA * nb + iB] * cub[i][j];
				upb -= F[iA * nb + iB] * J2[iA * nb + iB] * cua[i][j];
				upa += -dake + ucza;
				upb += -dbke + uczb;
				NU(UIx, iA, iB, k) += dt * upa;
				NU(VIx, iA, iB, k) += dt * upb;
				NU(PIx, iA, iB, k) -= dt * invJ2 * (dafa + dbfb);
			}
		}
	}
}

/* ------------------------------------------------------------------------------------------ */
/* Output interpolation: Grid::ReduceInterpolate -> GridPatchCSGLL::InterpolateData              */
/* (Grid.cpp:867-992, GridPatchCSGLL.cpp:1365-1780); SURVEY 8(f) item 4                          */

static void interp_column_apply(const double * C, int nout, int nin, const double * in, double * out) {
	/* LinearColumnOperator::Apply: out[k] = sum over the row's [begin, end) of C[k][l] in[l]; outside it C is zero */
	for (int k = 0; k < nout; k++) {
		double a = 0.0;
		for (int l = 0; l < nin; l++) if (C[(size_t)k * nin + l] != 0.0) a += C[(size_t)k * nin + l] * in[l];
		out[k] = a;
	}
}

static void interpolate_field(const orc_grid * g, const orc_interp * q, int pt, const double * x, const double * xref, int nlev,
	const double * divide_by, const double * op, double * out_col, double * col)
{
	const orc_patch * P = &g->patches[q->patch[pt]];
	const int nb = P->nb, iA = q->node_a[pt], iB = q->node_b[pt];
	const double * ca = q->coeff_a + (size_t)pt * NP, * cb = q->coeff_b + (size_t)pt * NP;
	for (int k = 0; k < nlev; k++) {
		col[k] = 0.0;
		for (int m = 0; m < NP; m++)
		for (int n = 0; n < NP; n++) {
			const double v = x[((size_t)(iA + m) * nb + (iB + n)) * nlev + k];
			if (divide_by) col[k] += ca[m] * cb[n] * v / divide_by[(((size_t)iA * nb + iB) * nlev + k) * 3 + 2];
			else col[k] += ca[m] * cb[n] * v;
		}
		if (xref)
			for (int m = 0; m < NP; m++)
			for (int n = 0; n < NP; n++)
				col[k] -= ca[m] * cb[n] * xref[((size_t)(iA + m) * nb + (iB + n)) * nlev + k];
	}
	interp_column_apply(op, q->nreta, nlev, col, out_col);
}

void orc_interpolate_state(const orc_grid * g, const orc_state * s, int ix, const orc_interp * q, int only_at,
	int include_reference_state, int convert_to_primitive, double earth_radius, double * out)
{
	const int L = g->L, npts = q->npts, nreta = q->nreta;
	double * col = dalloc(L + 2), * oc = dalloc(nreta);
	memset(out, 0, sizeof(double) * 5 * nreta * npts);
	for (int pt = 0; pt < npts; pt++) {
		const int p = q->patch[pt];
		const orc_patch * P = &g->patches[p];
		const int na = P->na, nb = P->nb;
		for (int c = 0; c < 5; c++) {
			const int edge = (c == WIx);
			if ((only_at == 1 && edge) || (only_at == 2 && !edge)) continue;
			const int nlev = edge ? L + 1 : L;
			const double * x = (edge ? s->redge[ix * g->npatch + p] : s->node[ix * g->npatch + p]) + (size_t)c * na * nb * nlev;
			const double * xr = include_reference_state ? NULL : ((edge ? P->ref_redge : P->ref_node) + (size_t)c * na * nb * nlev);
			const double * dv = (c == WIx && convert_to_primitive) ? P->g[ORC_G_DR_E] : NULL;
			interpolate_field(g, q, pt, x, xr, nlev, dv, edge ? q->op_interfaces : q->op_levels, oc, col);
			for (int k = 0; k < nreta; k++) out[((size_t)c * nreta + k) * npts + pt] = oc[k];
		}
		if (convert_to_primitive)
			for (int k = 0; k < nreta; k++) {
				const double ua = out[((size_t)UIx * nreta + k) * npts + pt] / earth_radius;
				const double ub = out[((size_t)VIx * nreta + k) * npts + pt] / earth_radius;
				const double * M = q->rll + (size_t)pt * 4;
				out[((size_t)UIx * nreta + k) * npts + pt] = M[0] * ua + M[1] * ub;
				out[((size_t)VIx * nreta + k) * npts + pt] = M[2] * ua + M[3] * ub;
			}
	}
	free(col); free(oc);
}

void orc_interpolate_tracers(const orc_grid * g, const orc_state * s, int ix, const orc_interp * q, double * out) {
	const int L = g->L, npts = q->npts, nreta = q->nreta;
	double * col = dalloc(L + 2), * oc = dalloc(nreta);
	memset(out, 0, sizeof(double) * g->ntracers * nreta * npts);
	for (int pt = 0; pt < npts; pt++) {
		const int p = q->patch[pt];
		const orc_patch * P = &g->patches[p];
		for (int c = 0; c < g->ntracers; c++) {
			const double * x = s->tracer[ix * g->npatch + p] + (size_t)c * P->na * P->nb * L;
			interpolate_field(g, q, pt, x, NULL, L, NULL, q->op_levels, oc, col);
			for (int k = 0; k < nreta; k++) out[((size_t)c * nreta + k) * npts + pt] = oc[k];
		}
	}
	free(col); free(oc);
}
