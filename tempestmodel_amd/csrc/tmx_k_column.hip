// tmx_k_column.hip -- hand-written CDNA4 (gfx950) kernels of the spectral-element hot path (one translation unit per operator family;
// layout, addressing and shared helpers: tmx_device.h).
//
// HBM layout: every field is a stack of "slabs" of NS doubles, one slab per
// (variable, level); inside a slab the index is the column  col = element*16 + i*4 + j
// (i = alpha node, j = beta node of the 4x4 GLL element).  One wavefront = 64 consecutive
// columns = 4 whole elements at one level, so every global access of a wavefront is one
// contiguous 512-byte segment and the np x np contractions stay inside a 16-lane group.
//
// All kernels are HBM-bound fp64 stencil work: one thread per (column, level),
// horizontal contractions through LDS, vertical stencils through neighbouring slabs (served by
// L2 / Infinity Cache).  Compiled with -ffp-contract=off so the arithmetic follows the operation
// order of the reference (which is built without FMA contraction).
//
// Reference behaviour restated (file:line under the reference tree) is cited per kernel.
#include "tmx_device.h"

// Both flavours of the band LU live in ONE library: this translation unit is compiled twice -- as it stands (LU updates fused, what OpenBLAS
// and MKL's FMA code paths compute) and with -DTMX_LU_NOFMA (multiply and subtract rounded separately, a BLAS without fused multiply-adds:
// tmx_device.h, LU_UPD) -- into a namespace each; tmx_lu_select.hip dispatches by the engine's option "lu_fma" (tmx_set_option, any time).
#ifdef TMX_LU_NOFMA
namespace lu_nofma {
#else
namespace lu_fma {
#endif

#if TMX_EXP      // the two-kernel cross-check path (option "vi_split_kernels"): in the experiments flavour of the library only
// ---------------------------------------------------------------------------------------------
// V implicit, part 1: residual F and analytic band Jacobian of one block row (level) of one column
// (VerticalDynamicsFEM::SetupReferenceColumn / PrepareColumn / BuildF / BuildJacobianF_LOR_RhoTheta_Pi /
//  BuildJacobianF_Diffusion, src/atm/VerticalDynamicsFEM.cpp:1643-1835, 1839-2179, 2183-2780,
//  2977-3187, 2784-2973).  Unknown ordering (rho*theta_k, W_k, rho_k) interleaved per level
// (VerticalDynamicsFEM.h:97-104).  Row r keeps its 9 band entries AB[r][d], d = col - r + 4.

__global__ __launch_bounds__(256) void k_vi_assemble(KParams p, const double * __restrict__ xin, double dt,
	int nunique, int NUS, const int * __restrict__ ucol, double * __restrict__ AB, double * __restrict__ RHS)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int u = blockIdx.x * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (u >= nunique || k > L) return;
	const int col = ucol ? ucol[u] : u;      // (null: node-unique layout, a unique column is its own state column)
	const double invdt = 1.0 / dt;
	const double upc = 0.5 * (1.0 / (double)L);             // m_dUpwindCoeff (:394-395)
	const double cv = p.cp - p.Rd;

	double rowP[TMX_BW], rowW[TMX_BW], rowR[TMX_BW];
#pragma unroll
	for (int d = 0; d < TMX_BW; d++) { rowP[d] = 0.0; rowW[d] = 0.0; rowR[d] = 0.0; }
	double fP = 0.0, fW = 0.0, fR = 0.0;
	// band offset of column (c', k') seen from row (c, k): d = 3(k'-k) + (c'-c) + 4
#define DOFF(cr, cc, dk) (3 * (dk) + ((cc) - (cr)) + 4)

	auto ldn = [&](int slabbase, int l) -> double {      // node slab, clamped level
		const int lc = l < 0 ? 0 : (l > L - 1 ? L - 1 : l);
		return xin[(size_t)(slabbase + lc) * NS + col];
	};
	auto ldw = [&](int l) -> double {
		const int lc = l < 0 ? 0 : (l > L ? L : l);
		return xin[(size_t)TMX_SLAB_W(L, lc) * NS + col];
	};
	const MetCol mcol = met_col(p, col);
	auto cxe = [&](int f, int m) -> double {
		double e0, e1, e2;
		metric_edge(p, mcol, m, col, e0, e1, e2);
		return f == G3E_CX0 ? e0 : (f == G3E_CX1 ? e1 : e2);
	};
	const double c2a0 = p.g2d[G2_C2A0 * NS + col], c2a1 = p.g2d[G2_C2A1 * NS + col], c2b1 = p.g2d[G2_C2B1 * NS + col];
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col], drx = p.g2d[G2_DRX * NS + col];

	// interface quantities at interface m: interpolated state and xi_dot (PrepareColumn :2056-2069)
	auto interp_n2e = [&](int slabbase, int m) -> double {
		double v = 0.0;
#pragma unroll
		for (int off = -2; off <= 1; off++) {
			const int l = m + off;
			if (l < 0 || l >= L) continue;
			v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * xin[(size_t)(slabbase + l) * NS + col];
		}
		return v;
	};
	auto xidot_e = [&](int m) -> double {
		if (m <= 0 || m >= L) return 0.0;
		const double ue = interp_n2e(TMX_SLAB_U(L, 0), m), ve = interp_n2e(TMX_SLAB_V(L, 0), m);
		return cxe(G3E_CX0, m) * ue + cxe(G3E_CX1, m) * ve + cxe(G3E_CX2, m) * ldw(m);
	};

	if (k < L) {
		// ---------------- rows (rho*theta_k) and (rho_k) ----------------
		const double invJ = 1.0 / jn;
		const double pm = ldn(TMX_SLAB_T(L, 0), k - 1), p0 = ldn(TMX_SLAB_T(L, 0), k), pp = ldn(TMX_SLAB_T(L, 0), k + 1);
		const double rm = ldn(TMX_SLAB_R(L, 0), k - 1), r0 = ldn(TMX_SLAB_R(L, 0), k), rp = ldn(TMX_SLAB_R(L, 0), k + 1);
		const double xd0 = xidot_e(k), xd1 = xidot_e(k + 1);
		const double pe0 = interp_n2e(TMX_SLAB_T(L, 0), k), pe1 = interp_n2e(TMX_SLAB_T(L, 0), k + 1);
		const double re0 = interp_n2e(TMX_SLAB_R(L, 0), k), re1 = interp_n2e(TMX_SLAB_R(L, 0), k + 1);
		// fluxes on interfaces, zero at the boundaries (BuildF :2215-2226, :2295-2306)
		const double mf0 = (k >= 1) ? je * re0 * xd0 : 0.0, mf1 = (k + 1 <= L - 1) ? je * re1 * xd1 : 0.0;
		const double pf0 = (k >= 1) ? je * pe0 * xd0 : 0.0, pf1 = (k + 1 <= L - 1) ? je * pe1 * xd1 : 0.0;
		const double de0 = OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 0), de1 = OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 1);
		double dmf = 0.0; dmf += de0 * mf0; dmf += de1 * mf1;
		double dpf = 0.0; dpf += de0 * pf0; dpf += de1 * pf1;
		fR = dmf * invJ;
		fP += dpf * invJ;
		// upwind penalty on levels (BuildF :2640-2712)
		const double wlo = fabs(xd0), whi = fabs(xd1);      // weights of interfaces k and k+1
		const double pl0 = OPC(TMX_OP_PENALTY_LEFT, k, 0), pl1 = OPC(TMX_OP_PENALTY_LEFT, k, 1);
		const double pr0 = OPC(TMX_OP_PENALTY_RIGHT, k, -1), pr1 = OPC(TMX_OP_PENALTY_RIGHT, k, 0);
		{
			double a = 0.0;
			if (k < L - 1) { double b = 0.0; b += pl0 * p0; b += pl1 * pp; a += b * whi; }
			if (k > 0) { double b = 0.0; b += pr0 * pm; b += pr1 * p0; a += b * wlo; }
			fP -= a;
			a = 0.0;
			if (k < L - 1) { double b = 0.0; b += pl0 * r0; b += pl1 * rp; a += b * whi; }
			if (k > 0) { double b = 0.0; b += pr0 * rm; b += pr1 * r0; a += b * wlo; }
			fR -= a;
		}
		// Jacobian: conservative flux terms (:3069-3103)
#pragma unroll
		for (int mm = 0; mm < 2; mm++) {
			const int m = k + mm;
			const double dcoef = mm ? de1 : de0;
			const double xd = mm ? xd1 : xd0;
			if (m != 0 && m != L) {
				const double c = dcoef * je * invJ * cxe(G3E_CX2, m);
				rowP[DOFF(0, 1, mm)] += c * (mm ? pe1 : pe0);
				rowR[DOFF(2, 1, mm)] += c * (mm ? re1 : re0);
			}
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int qn = m + off;
				if (qn < 0 || qn >= L) continue;
				const double ci = OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off);
				if (ci == 0.0) continue;
				const int dk = qn - k;
				if (dk < -1 || dk > 1) continue;
				const double c = dcoef * je * invJ * ci * xd;
				rowR[DOFF(2, 2, dk)] += c;
				rowP[DOFF(0, 0, dk)] += c;
			}
		}
		// Jacobian: upwinding on levels (BuildJacobianF_Diffusion :2905-2970); interface a = k acts
		// through the right operator, a = k+1 through the left operator
		if (k >= 1) {
			const double sw = (xd0 > 0.0) ? cxe(G3E_CX2, k) : ((xd0 < 0.0) ? -cxe(G3E_CX2, k) : 0.0);
			rowP[DOFF(0, 1, 0)] -= sw * pr0 * pm; rowP[DOFF(0, 1, 0)] -= sw * pr1 * p0;
			rowP[DOFF(0, 0, -1)] -= wlo * pr0;    rowP[DOFF(0, 0, 0)] -= wlo * pr1;
			rowR[DOFF(2, 1, 0)] -= sw * pr0 * rm; rowR[DOFF(2, 1, 0)] -= sw * pr1 * r0;
			rowR[DOFF(2, 2, -1)] -= wlo * pr0;    rowR[DOFF(2, 2, 0)] -= wlo * pr1;
		}
		if (k + 1 <= L - 1) {
			const double sw = (xd1 > 0.0) ? cxe(G3E_CX2, k + 1) : ((xd1 < 0.0) ? -cxe(G3E_CX2, k + 1) : 0.0);
			rowP[DOFF(0, 1, 1)] -= sw * pl0 * p0; rowP[DOFF(0, 1, 1)] -= sw * pl1 * pp;
			rowP[DOFF(0, 0, 0)] -= whi * pl0;     rowP[DOFF(0, 0, 1)] -= whi * pl1;
			rowR[DOFF(2, 1, 1)] -= sw * pl0 * r0; rowR[DOFF(2, 1, 1)] -= sw * pl1 * rp;
			rowR[DOFF(2, 2, 0)] -= whi * pl0;     rowR[DOFF(2, 2, 1)] -= whi * pl1;
		}
	}
	if (k >= 1 && k <= L - 1) {
		// ---------------- row (W_k), interior interfaces ----------------
		const double pm = ldn(TMX_SLAB_T(L, 0), k - 1), p0 = ldn(TMX_SLAB_T(L, 0), k);
		const double exm = exner_from_rhotheta(p, pm), ex0 = exner_from_rhotheta(p, p0);
		const double pe = interp_n2e(TMX_SLAB_T(L, 0), k), re = interp_n2e(TMX_SLAB_R(L, 0), k);
		const double dnm = OPC(TMX_OP_DIFF_NODE_TO_REDGE, k, -1), dn0 = OPC(TMX_OP_DIFF_NODE_TO_REDGE, k, 0);
		double dpe = 0.0; dpe += dnm * exm; dpe += dn0 * ex0;
		const double unm = ldn(TMX_SLAB_U(L, 0), k - 1), un0 = ldn(TMX_SLAB_U(L, 0), k);
		const double vnm = ldn(TMX_SLAB_V(L, 0), k - 1), vn0 = ldn(TMX_SLAB_V(L, 0), k);
		const double wem = ldw(k - 1), we0 = ldw(k), wep = ldw(k + 1);
		// W on levels k-1, k (InterpolateREdgeToNode)
		double wnm = 0.0; wnm += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 0) * wem; wnm += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 1) * we0;
		double wn0 = 0.0; wn0 += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * we0; wn0 += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * wep;
		// node metric at k-1, k
		double ca2m, cb2m, cx2m, ca20, cb20, cx20;
		metric_node3(p, mcol, k - 1, col, ca2m, cb2m, cx2m);
		metric_node3(p, mcol, k, col, ca20, cb20, cx20);
		// xi_dot on levels (:2049-2054) and kinetic energy on levels (BuildF :2455-2478)
		const double xdnm = ca2m * unm + cb2m * vnm + cx2m * wnm;
		const double xdn0 = ca20 * un0 + cb20 * vn0 + cx20 * wn0;
		double kem, ke0;
		{
			const double ca = c2a0 * unm + c2a1 * vnm + ca2m * wnm, cb = c2a1 * unm + c2b1 * vnm + cb2m * wnm;
			kem = 0.5 * (ca * unm + cb * vnm + xdnm * wnm);
		}
		{
			const double ca = c2a0 * un0 + c2a1 * vn0 + ca20 * wn0, cb = c2a1 * un0 + c2b1 * vn0 + cb20 * wn0;
			ke0 = 0.5 * (ca * un0 + cb * vn0 + xdn0 * wn0);
		}
		double dke = 0.0; dke += dnm * kem; dke += dn0 * ke0;
		double dua = 0.0; dua += dnm * unm; dua += dn0 * un0;
		double dub = 0.0; dub += dnm * vnm; dub += dn0 * vn0;
		// interface values
		const double ue = interp_n2e(TMX_SLAB_U(L, 0), k), ve = interp_n2e(TMX_SLAB_V(L, 0), k);
		const double cx0e = cxe(G3E_CX0, k), cx1e = cxe(G3E_CX1, k), cx2e = cxe(G3E_CX2, k);
		const double xde = cx0e * ue + cx1e * ve + cx2e * we0;
		// F_W (BuildF :2527-2572)
		fW = dpe * pe / re;
		fW += p.grav * drx;
		{
			const double ca = c2a0 * ue + c2a1 * ve + cx0e * we0;
			const double cb = c2a1 * ue + c2b1 * ve + cx1e * we0;
			const double curl = -ca * dua - cb * dub;
			fW += (dke + curl);
		}
		// second derivative of W for the upwind term (PrepareColumn :2091-2102, BuildF :2680-2691)
		const double ddm = OPC(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, -1), dd0 = OPC(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, 0), ddp = OPC(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, 1);
		double ddw = 0.0; ddw += ddm * wem; ddw += dd0 * we0; ddw += ddp * wep;
		fW -= upc * fabs(xde) * ddw;
		// Jacobian of the W row (:3106-3152)
		const double cA = pe * p.Rd / (re * cv);
		rowW[DOFF(1, 0, -1)] += cA * dnm * exm / pm;
		rowW[DOFF(1, 0, 0)] += cA * dn0 * ex0 / p0;
		const double cB = 1.0 / (re * re) * dpe;
		const double im = OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1), i0 = OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0);
		{
			const double cC = cB * im;
			rowW[DOFF(1, 0, -1)] += cC * re;
			rowW[DOFF(1, 2, -1)] += -cC * pe;
		}
		{
			const double cC = cB * i0;
			rowW[DOFF(1, 0, 0)] += cC * re;
			rowW[DOFF(1, 2, 0)] += -cC * pe;
		}
		// Clark-form dW_k/dW_m (:3140-3152): l = k-1 -> m in {k-1,k}; l = k -> m in {k,k+1}
		rowW[DOFF(1, 1, -1)] += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 0) * dnm * xdnm;
		rowW[DOFF(1, 1, 0)] += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 1) * dnm * xdnm;
		rowW[DOFF(1, 1, 0)] += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * dn0 * xdn0;
		rowW[DOFF(1, 1, 1)] += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * dn0 * xdn0;
		// upwinding on interfaces (BuildJacobianF_Diffusion :2873-2902)
		const double sw = (xde > 0.0) ? cx2e : ((xde < 0.0) ? -cx2e : 0.0);
		rowW[DOFF(1, 1, 0)] -= upc * sw * ddw;
		rowW[DOFF(1, 1, -1)] -= upc * fabs(xde) * ddm;
		rowW[DOFF(1, 1, 0)] -= upc * fabs(xde) * dd0;
		rowW[DOFF(1, 1, 1)] -= upc * fabs(xde) * ddp;
	}
	// identity / dt on the whole diagonal (:3172-3176); rows W_0, W_L and the unused level-L slots
	// of rho*theta and rho are pure identity rows with F = 0
	rowP[4] += invdt; rowW[4] += invdt; rowR[4] += invdt;

	const int r0 = TMX_FTOT * k;
#pragma unroll
	for (int d = 0; d < TMX_BW; d++) {
		AB[((size_t)(r0 + 0) * TMX_BW + d) * NUS + u] = rowP[d];
		AB[((size_t)(r0 + 1) * TMX_BW + d) * NUS + u] = rowW[d];
		AB[((size_t)(r0 + 2) * TMX_BW + d) * NUS + u] = rowR[d];
	}
	RHS[(size_t)(r0 + 0) * NUS + u] = fP;
	RHS[(size_t)(r0 + 1) * NUS + u] = fW;
	RHS[(size_t)(r0 + 2) * NUS + u] = fR;
#undef DOFF
}

void tmxk_vi_assemble(tmx_engine * e, const KParams & p, const double * xin, double dt) {
	if (e->nunique == 0) return;
	dim3 blk(64, 4), grd((e->nunique + 63) / 64, (p.L + 1 + 3) / 4);
	hipLaunchKernelGGL(k_vi_assemble, grd, blk, 0, e->stream, p, xin, dt, e->nunique, e->NUS,
		(const int *)e->d_ucol, e->d_ab, e->d_rhs);
}

// ---------------------------------------------------------------------------------------------
// V implicit, part 2: banded LU with partial pivoting + solve, one column per lane, the active
// window (5 rows x 9 columns) in registers.  Same elimination order, pivot choice (first maximum)
// and substitution order as LAPACK dgbtf2 / dgbtrs / dtbsv, which the reference calls through
// LAPACK::DGBSV (src/base/LinearAlgebra.cpp:156-202 <- src/atm/VerticalDynamicsFEM.cpp:1457-1459).
// Then x = x0 - delta (:1483-1485) is scattered to the column and to the duplicate element-seam
// nodes of the same patch (:1488-1633; rho*theta, W, rho only -- U,V are not touched).

__global__ __launch_bounds__(64) void k_vi_solve(KParams p, const double * xin, double * xup,
	int nunique, int NUS, const int * __restrict__ ucol, const int * __restrict__ udep,
	double * __restrict__ AB, double * __restrict__ RHS, int * __restrict__ flag)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int u = blockIdx.x * 64 + threadIdx.x;
	if (u >= nunique) return;
	const int n = TMX_FTOT * (L + 1);

	double R[5][TMX_BW], b[5];
#pragma unroll
	for (int r = 0; r < 5; r++) {
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) {
			const int d = c - r + 4;           // window column c = matrix column c (j = 0)
			R[r][c] = (d >= 0 && d < TMX_BW && r < n) ? AB[((size_t)r * TMX_BW + d) * NUS + u] : 0.0;
		}
		b[r] = (r < n) ? RHS[(size_t)r * NUS + u] : 0.0;
	}
	bool singular = false;
	for (int jcol = 0; jcol < n; jcol++) {
		// pivot: first row of maximum |entry| in the leading window column (idamax)
		int pv = 0;
		double amax = fabs(R[0][0]);
#pragma unroll
		for (int r = 1; r < 5; r++) {
			const double v = fabs(R[r][0]);
			if (v > amax) { amax = v; pv = r; }
		}
#pragma unroll
		for (int r = 1; r < 5; r++) {
			const bool sw = (pv == r);
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) {
				const double a0 = R[0][c], ar = R[r][c];
				R[0][c] = sw ? ar : a0;
				R[r][c] = sw ? a0 : ar;
			}
			const double b0 = b[0], br = b[r];
			b[0] = sw ? br : b0;
			b[r] = sw ? b0 : br;
		}
		if (R[0][0] == 0.0) singular = true;
		// row jcol of U and of the forward-substituted right-hand side
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) AB[((size_t)jcol * TMX_BW + c) * NUS + u] = R[0][c];
		RHS[(size_t)jcol * NUS + u] = b[0];
		const double rcp = 1.0 / R[0][0];
#pragma unroll
		for (int r = 1; r < 5; r++) {
			const double l = R[r][0] * rcp;
#pragma unroll
			for (int c = 1; c < TMX_BW; c++) R[r][c] = LU_UPD(R[r][c], l, R[0][c]);
			b[r] = LU_UPD(b[r], b[0], l);
		}
		// slide the window: drop the pivot row and the pivot column, take in matrix row jcol + 5
#pragma unroll
		for (int r = 0; r < 4; r++) {
#pragma unroll
			for (int c = 0; c < TMX_BW - 1; c++) R[r][c] = R[r + 1][c + 1];
			R[r][TMX_BW - 1] = 0.0;
			b[r] = b[r + 1];
		}
		const int rn = jcol + 5;
		if (rn < n) {
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) R[4][c] = AB[((size_t)rn * TMX_BW + c) * NUS + u];
			b[4] = RHS[(size_t)rn * NUS + u];
		} else {
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) R[4][c] = 0.0;
			b[4] = 0.0;
		}
	}
	if (singular) atomicOr(flag, TMX_FLAG_SINGULAR);

	// back substitution in dtbsv order, scatter x0 - delta
	const int col = ucol ? ucol[u] : u;      // (null: node-unique layout, a unique column is its own state column)
	int dep[3];
#pragma unroll
	for (int t = 0; t < 3; t++) dep[t] = udep ? udep[u * 3 + t] : -1;
	double xw[TMX_BW - 1];
#pragma unroll
	for (int c = 0; c < TMX_BW - 1; c++) xw[c] = 0.0;
	for (int r = n - 1; r >= 0; r--) {
		double acc = RHS[(size_t)r * NUS + u];
		double ur[TMX_BW];
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) ur[c] = AB[((size_t)r * TMX_BW + c) * NUS + u];
#pragma unroll
		for (int c = TMX_BW - 1; c >= 1; c--) acc = LU_UPD(acc, xw[c - 1], ur[c]);
		const double xr = acc / ur[0];
#pragma unroll
		for (int c = TMX_BW - 2; c >= 1; c--) xw[c] = xw[c - 1];
		xw[0] = xr;
		const int k = r / TMX_FTOT, cc = r - k * TMX_FTOT;
		if (cc != 1 && k == L) continue;       // unused level-L slots of rho*theta / rho
		const int slab = (cc == 0) ? TMX_SLAB_T(L, k) : ((cc == 1) ? TMX_SLAB_W(L, k) : TMX_SLAB_R(L, k));
		const double val = xin[(size_t)slab * NS + col] - xr;
		xup[(size_t)slab * NS + col] = val;
#pragma unroll
		for (int t = 0; t < 3; t++) if (dep[t] >= 0) xup[(size_t)slab * NS + dep[t]] = val;
	}
}

void tmxk_vi_solve(tmx_engine * e, const KParams & p, const double * xin, double * xup) {
	if (e->nunique == 0) return;
	hipLaunchKernelGGL(k_vi_solve, dim3((e->nunique + 63) / 64), dim3(64), 0, e->stream, p, xin, xup,
		e->nunique, e->NUS, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_rhs, e->d_flag);
}
#endif      // TMX_EXP

// ---------------------------------------------------------------------------------------------
// V implicit, fused: the same residual / Jacobian / pivoted band LU as k_vi_assemble + k_vi_solve,
// but each block row (level) is assembled in registers right before it enters the LU window, so the
// band matrix never goes to HBM (only the U rows needed by the back substitution do).  The state of
// the three levels a block row touches lives in a sliding register window; the loads of the next
// level are issued one block (three pivot steps) ahead of their first use.
// Arithmetic is statement-for-statement that of the split kernels: results are bit-identical.

struct NodeLev { double un, vn, pn, rn, ca2, cb2, cx2; };
struct EdgeLev { double we, ce0, ce1, ce2; };

// Loads are unconditional with the level clamped into range: a fixed number of loads per block row keeps the
// compiler's vmcnt bookkeeping exact, so a wait for the previous block row's prefetch does not also wait for
// the loads just issued.  Out-of-range levels (-1 below block row 0, L and L+1 past the top) therefore carry
// the values of the nearest level; compute_block only touches them under its k >= 1 / k + 1 <= L - 1 guards or
// with zero operator coefficients, so results do not depend on them (checked bitwise against the split kernels).
template <int MM = 0>
__device__ __forceinline__ NodeLev load_node(const KParams & p, const MetCol & mc, const double * xin, int l, int col, const double * etal = nullptr) {
	NodeLev n;
	const int L = p.L;
	const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l);
	const size_t NS = (size_t)p.NS;
	n.un = xin[TMX_SLAB_U(L, lc) * NS + col]; n.vn = xin[TMX_SLAB_V(L, lc) * NS + col];
	n.pn = xin[TMX_SLAB_T(L, lc) * NS + col]; n.rn = xin[TMX_SLAB_R(L, lc) * NS + col];
	metric_node3<MM>(p, mc, lc, col, n.ca2, n.cb2, n.cx2, etal);
	return n;
}
template <int MM = 0>
__device__ __forceinline__ EdgeLev load_edge(const KParams & p, const MetCol & mc, const double * xin, int l, int col, const double * etal = nullptr) {
	EdgeLev e;
	const int L = p.L;
	const int lc = l < 0 ? 0 : (l > L ? L : l);
	const size_t NS = (size_t)p.NS;
	e.we = xin[TMX_SLAB_W(L, lc) * NS + col];
	metric_edge<MM>(p, mc, lc, col, e.ce0, e.ce1, e.ce2, etal);
	return e;
}

struct ColConst { double c2a0, c2a1, c2b1, jn, je, drx, invdt, upc, cv; };

// Values one block row shares with the next (level k quantities become level k-1 quantities, interface
// k+1 quantities become interface k quantities).  Carrying them over instead of recomputing is the same
// arithmetic on the same operands, so results stay bit-identical to the stand-alone evaluation.
struct BlkCarry {
	double ex, wn, xdn, ke;            // node k: Exner pressure, W on the level, xi_dot, kinetic energy
	double ue, ve, xd, pe, re;         // interface k+1: U,V, xi_dot, rho*theta, rho interpolated
};

// block row k from node levels A = k-1, B = k, C = k+1 and interfaces a = k-1, b = k, c = k+1
// column-operator coefficients from the LDS copy (ds_read: no vector-memory round trip, no vmcnt wait behind the U-row stores)
#define OPCL(op, k, off) opsl[(((op) * (p.L + 1)) + (k)) * TMX_OPW + ((off) + 2)]
// UD: the uniform-diffusion terms of BuildF (VerticalDynamicsFEM.cpp:2593-2635), udP = K_s / ztop^2 * second derivative
// of (rho*theta - reference) on level k, udW = K_v / ztop^2 * the same for W on interface k, subtracted where the
// reference subtracts them (before the upwinding terms); only the fully explicit mode evaluates them.
// sign(x) * v as in the reference's upwinding terms ((x > 0) ? v : ((x < 0) ? -v : 0)), written as a chain of two selects: the
// nested conditional became exec-masked branches, which cut the block row's assembly into short basic blocks
__device__ __forceinline__ double signed_by(double x, double v) {
	double r = 0.0;
	r = (x < 0.0) ? -v : r;
	r = (x > 0.0) ? v : r;
	return r;
}

// INTERIOR: the caller guarantees 1 <= k <= L - 2 and arguments of the Exner function in the main range of exp / log (positive,
// normal, finite): every level-boundary condition below is then a compile-time `true` and exp(log()) is evaluated without
// branches (exner_from_rhotheta_bf), so that the whole block row is ONE basic block whose dependency chains the compiler can
// interleave -- with the dozen short blocks of the general form a lone assembly wavefront ran at a quarter of its issue rate.
// Same statements on the same operands either way.
template <bool CARRY, bool UD = false, bool INTERIOR = false>
__device__ __forceinline__ void compute_block(const KParams & p, const double * opsl, const ColConst & cc, int k,
	const NodeLev & A, const NodeLev & B, const NodeLev & C, const EdgeLev & ea, const EdgeLev & eb, const EdgeLev & ec,
	double * rowP, double * rowW, double * rowR, double & fP, double & fW, double & fR, BlkCarry & cy,
	double udP = 0.0, double udW = 0.0, const double * rmtab = nullptr, long long * tseg = nullptr)
{
#ifdef TMX_PAIR_TIMING
	long long ts_ = __builtin_readcyclecounter();
#define CBSTAMP(i) do { if (tseg) { const long long t1_ = __builtin_readcyclecounter(); tseg[i] += t1_ - ts_; ts_ = t1_; } } while (0)
#else
#define CBSTAMP(i)
#endif
	const int L = p.L;
#pragma unroll
	for (int d = 0; d < TMX_BW; d++) { rowP[d] = 0.0; rowW[d] = 0.0; rowR[d] = 0.0; }
	fP = 0.0; fW = 0.0; fR = 0.0;
#define DOFF(cr, cc_, dk) (3 * (dk) + ((cc_) - (cr)) + 4)
	// interior interpolation stencils (offsets -1, 0), verified by tmx_set_operators
	const int kib = (INTERIOR || k < L) ? k : L;
	const double ib_m = OPCL(TMX_OP_INTERP_NODE_TO_REDGE, kib, -1), ib_0 = OPCL(TMX_OP_INTERP_NODE_TO_REDGE, kib, 0);

	// ---- interface k: interpolated U,V,rho*theta,rho and xi_dot (PrepareColumn :2056-2069) ----
	double ue_b = 0.0, ve_b = 0.0, xd0 = 0.0, pe0 = 0.0, re0 = 0.0;
	if (INTERIOR || (k >= 1 && k <= L - 1)) {
		if (CARRY) { ue_b = cy.ue; ve_b = cy.ve; xd0 = cy.xd; pe0 = cy.pe; re0 = cy.re; }
		else {
			ue_b += ib_m * A.un; ue_b += ib_0 * B.un; ve_b += ib_m * A.vn; ve_b += ib_0 * B.vn;
			xd0 = eb.ce0 * ue_b + eb.ce1 * ve_b + eb.ce2 * eb.we;
			pe0 += ib_m * A.pn; pe0 += ib_0 * B.pn; re0 += ib_m * A.rn; re0 += ib_0 * B.rn;
		}
	}
	// ---- interface k+1 ----
	double ue_c = 0.0, ve_c = 0.0, xd1 = 0.0, pe1 = 0.0, re1 = 0.0;
	double ic_m = 0.0, ic_0 = 0.0;
	if (INTERIOR || k + 1 <= L - 1) {
		ic_m = OPCL(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, -1); ic_0 = OPCL(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, 0);
		ue_c += ic_m * B.un; ue_c += ic_0 * C.un; ve_c += ic_m * B.vn; ve_c += ic_0 * C.vn;
		xd1 = ec.ce0 * ue_c + ec.ce1 * ve_c + ec.ce2 * ec.we;
		pe1 += ic_m * B.pn; pe1 += ic_0 * C.pn; re1 += ic_m * B.rn; re1 += ic_0 * C.rn;
	}

	CBSTAMP(0);
	if (INTERIOR || k < L) {
		const double invJ = 1.0 / cc.jn;
		const double pm = A.pn, p0 = B.pn, pp = C.pn, rm = A.rn, r0 = B.rn, rp = C.rn;
		const double mf0 = (INTERIOR || k >= 1) ? cc.je * re0 * xd0 : 0.0, mf1 = (INTERIOR || k + 1 <= L - 1) ? cc.je * re1 * xd1 : 0.0;
		const double pf0 = (INTERIOR || k >= 1) ? cc.je * pe0 * xd0 : 0.0, pf1 = (INTERIOR || k + 1 <= L - 1) ? cc.je * pe1 * xd1 : 0.0;
		const double de0 = OPCL(TMX_OP_DIFF_REDGE_TO_NODE, k, 0), de1 = OPCL(TMX_OP_DIFF_REDGE_TO_NODE, k, 1);
		double dmf = 0.0; dmf += de0 * mf0; dmf += de1 * mf1;
		double dpf = 0.0; dpf += de0 * pf0; dpf += de1 * pf1;
		fR = dmf * invJ;
		fP += dpf * invJ;
		if (UD) fP -= udP;
		const double wlo = fabs(xd0), whi = fabs(xd1);
		const double pl0 = OPCL(TMX_OP_PENALTY_LEFT, k, 0), pl1 = OPCL(TMX_OP_PENALTY_LEFT, k, 1);
		const double pr0 = OPCL(TMX_OP_PENALTY_RIGHT, k, -1), pr1 = OPCL(TMX_OP_PENALTY_RIGHT, k, 0);
		{
			double a = 0.0;
			if (INTERIOR || k < L - 1) { double b = 0.0; b += pl0 * p0; b += pl1 * pp; a += b * whi; }
			if (INTERIOR || k > 0) { double b = 0.0; b += pr0 * pm; b += pr1 * p0; a += b * wlo; }
			fP -= a;
			a = 0.0;
			if (INTERIOR || k < L - 1) { double b = 0.0; b += pl0 * r0; b += pl1 * rp; a += b * whi; }
			if (INTERIOR || k > 0) { double b = 0.0; b += pr0 * rm; b += pr1 * r0; a += b * wlo; }
			fR -= a;
		}
		// conservative flux terms: m = k (interface b), m = k+1 (interface c)
		if (INTERIOR || k >= 1) {       // m = k, neither 0 nor L
			const double c = de0 * cc.je * invJ * eb.ce2;
			rowP[DOFF(0, 1, 0)] += c * pe0;
			rowR[DOFF(2, 1, 0)] += c * re0;
			const double cm = de0 * cc.je * invJ * ib_m * xd0, c0 = de0 * cc.je * invJ * ib_0 * xd0;
			rowR[DOFF(2, 2, -1)] += cm; rowP[DOFF(0, 0, -1)] += cm;
			rowR[DOFF(2, 2, 0)] += c0;  rowP[DOFF(0, 0, 0)] += c0;
		}
		if (INTERIOR || k + 1 <= L - 1) {   // m = k+1
			const double c = de1 * cc.je * invJ * ec.ce2;
			rowP[DOFF(0, 1, 1)] += c * pe1;
			rowR[DOFF(2, 1, 1)] += c * re1;
			const double cm = de1 * cc.je * invJ * ic_m * xd1, c0 = de1 * cc.je * invJ * ic_0 * xd1;
			rowR[DOFF(2, 2, 0)] += cm; rowP[DOFF(0, 0, 0)] += cm;
			rowR[DOFF(2, 2, 1)] += c0; rowP[DOFF(0, 0, 1)] += c0;
		}
		if (INTERIOR || k >= 1) {
			const double sw = signed_by(xd0, eb.ce2);
			rowP[DOFF(0, 1, 0)] -= sw * pr0 * pm; rowP[DOFF(0, 1, 0)] -= sw * pr1 * p0;
			rowP[DOFF(0, 0, -1)] -= wlo * pr0;    rowP[DOFF(0, 0, 0)] -= wlo * pr1;
			rowR[DOFF(2, 1, 0)] -= sw * pr0 * rm; rowR[DOFF(2, 1, 0)] -= sw * pr1 * r0;
			rowR[DOFF(2, 2, -1)] -= wlo * pr0;    rowR[DOFF(2, 2, 0)] -= wlo * pr1;
		}
		if (INTERIOR || k + 1 <= L - 1) {
			const double sw = signed_by(xd1, ec.ce2);
			rowP[DOFF(0, 1, 1)] -= sw * pl0 * p0; rowP[DOFF(0, 1, 1)] -= sw * pl1 * pp;
			rowP[DOFF(0, 0, 0)] -= whi * pl0;     rowP[DOFF(0, 0, 1)] -= whi * pl1;
			rowR[DOFF(2, 1, 1)] -= sw * pl0 * r0; rowR[DOFF(2, 1, 1)] -= sw * pl1 * rp;
			rowR[DOFF(2, 2, 0)] -= whi * pl0;     rowR[DOFF(2, 2, 1)] -= whi * pl1;
		}
	}

	CBSTAMP(1);
	// ---- node k quantities the W rows of this block and of the next one use ----
	double ex0 = 0.0, wn0 = 0.0, xdn0 = 0.0, ke0 = 0.0;
	if (INTERIOR || (k <= L - 1 && (CARRY || k >= 1))) {
		ex0 = INTERIOR ? exner_from_rhotheta_bf(p, B.pn, rmtab) : (rmtab ? exner_from_rhotheta_lds(p, B.pn, rmtab) : exner_from_rhotheta(p, B.pn));
		wn0 += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * eb.we; wn0 += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * ec.we;
		xdn0 = B.ca2 * B.un + B.cb2 * B.vn + B.cx2 * wn0;
		const double ca = cc.c2a0 * B.un + cc.c2a1 * B.vn + B.ca2 * wn0, cb = cc.c2a1 * B.un + cc.c2b1 * B.vn + B.cb2 * wn0;
		ke0 = 0.5 * (ca * B.un + cb * B.vn + xdn0 * wn0);
	}
	CBSTAMP(2);
	if (INTERIOR || (k >= 1 && k <= L - 1)) {
		const double pm = A.pn, p0 = B.pn;
		double exm, wnm, xdnm, kem;
		if (CARRY) { exm = cy.ex; wnm = cy.wn; xdnm = cy.xdn; kem = cy.ke; }
		else {
			exm = rmtab ? exner_from_rhotheta_lds(p, pm, rmtab) : exner_from_rhotheta(p, pm);
			wnm = 0.0; wnm += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 0) * ea.we; wnm += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 1) * eb.we;
			xdnm = A.ca2 * A.un + A.cb2 * A.vn + A.cx2 * wnm;
			const double ca = cc.c2a0 * A.un + cc.c2a1 * A.vn + A.ca2 * wnm, cb = cc.c2a1 * A.un + cc.c2b1 * A.vn + A.cb2 * wnm;
			kem = 0.5 * (ca * A.un + cb * A.vn + xdnm * wnm);
		}
		(void)wnm;
		const double pe = pe0, re = re0;
		const double dnm = OPCL(TMX_OP_DIFF_NODE_TO_REDGE, k, -1), dn0 = OPCL(TMX_OP_DIFF_NODE_TO_REDGE, k, 0);
		double dpe = 0.0; dpe += dnm * exm; dpe += dn0 * ex0;
		const double unm = A.un, un0 = B.un, vnm = A.vn, vn0 = B.vn;
		const double wem = ea.we, we0 = eb.we, wep = ec.we;
		double dke = 0.0; dke += dnm * kem; dke += dn0 * ke0;
		double dua = 0.0; dua += dnm * unm; dua += dn0 * un0;
		double dub = 0.0; dub += dnm * vnm; dub += dn0 * vn0;
		const double ue = ue_b, ve = ve_b;
		const double cx0e = eb.ce0, cx1e = eb.ce1, cx2e = eb.ce2;
		const double xde = xd0;
		fW = dpe * pe / re;
		fW += p.grav * cc.drx;
		{
			const double ca = cc.c2a0 * ue + cc.c2a1 * ve + cx0e * we0;
			const double cb = cc.c2a1 * ue + cc.c2b1 * ve + cx1e * we0;
			const double curl = -ca * dua - cb * dub;
			fW += (dke + curl);
		}
		if (UD) fW -= udW;
		const double ddm = OPCL(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, -1), dd0 = OPCL(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, 0), ddp = OPCL(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, 1);
		double ddw = 0.0; ddw += ddm * wem; ddw += dd0 * we0; ddw += ddp * wep;
		fW -= cc.upc * fabs(xde) * ddw;
		const double cA = pe * p.Rd / (re * cc.cv);
		rowW[DOFF(1, 0, -1)] += cA * dnm * exm / pm;
		rowW[DOFF(1, 0, 0)] += cA * dn0 * ex0 / p0;
		const double cB = 1.0 / (re * re) * dpe;
		{
			const double cC = cB * ib_m;
			rowW[DOFF(1, 0, -1)] += cC * re;
			rowW[DOFF(1, 2, -1)] += -cC * pe;
		}
		{
			const double cC = cB * ib_0;
			rowW[DOFF(1, 0, 0)] += cC * re;
			rowW[DOFF(1, 2, 0)] += -cC * pe;
		}
		rowW[DOFF(1, 1, -1)] += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 0) * dnm * xdnm;
		rowW[DOFF(1, 1, 0)] += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k - 1, 1) * dnm * xdnm;
		rowW[DOFF(1, 1, 0)] += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * dn0 * xdn0;
		rowW[DOFF(1, 1, 1)] += OPCL(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * dn0 * xdn0;
		const double sw = signed_by(xde, cx2e);
		rowW[DOFF(1, 1, 0)] -= cc.upc * sw * ddw;
		rowW[DOFF(1, 1, -1)] -= cc.upc * fabs(xde) * ddm;
		rowW[DOFF(1, 1, 0)] -= cc.upc * fabs(xde) * dd0;
		rowW[DOFF(1, 1, 1)] -= cc.upc * fabs(xde) * ddp;
	}
	CBSTAMP(3);
	rowP[4] += cc.invdt; rowW[4] += cc.invdt; rowR[4] += cc.invdt;
	if (CARRY) {
		cy.ex = ex0; cy.wn = wn0; cy.xdn = xdn0; cy.ke = ke0;
		cy.ue = ue_c; cy.ve = ve_c; cy.xd = xd1; cy.pe = pe1; cy.re = re1;
	}
#undef DOFF
#undef CBSTAMP
}

#undef OPCL

// one elimination step on the register window (dgbtf2 + forward substitution of dgbtrs), then slide.
// U rows go to the HBM scratch for the back substitution.  Entries 4..8 of a row are fill-in that exists only
// for some pivot choices (measured on the baroclinic wave: 9 of the 15 such entries per block row are exactly
// zero in every column, because rho*theta / W / rho rows pivot 4 / 3 / 2 rows down almost everywhere).  With
// SPARSE an entry that is zero in all 64 lanes is not written, and `umask` (LDS, one word per row) tells the back
// substitution which ones to read back; skipped entries are exact zeros, so the solve is unchanged.  This cuts
// the kernel's HBM traffic by about a fifth (measured 286 vs 320 us per call at ne30 L30).
// (Tried and dropped: detecting a wave-uniform pivot row with a ballot and swapping registers at compile time
// instead of 160 v_cndmask per step, and skipping the update of all-zero fill-in columns -- neither changed the
// time, the kernel is bound by HBM traffic at ne30 and by dependent-issue latency on small grids.)
// Tail of a pivot step shared by both paths: store the U row (row 0 of the window after the interchange, given as
// piv / pb), sparse fill-in entries only where some lane is non-zero.
// U rows and transformed right-hand side of the fused / pair kernels: one contiguous stream per wavefront,
// [wavefront][matrix row][9 entries + rhs][64 lanes], so that the forward loop appends and the back substitution
// reads 5 KB rows back to back instead of ten 512-byte pieces 389 KB apart.  `AB` below is the lane's pointer into
// its wavefront's stream (vi_stream_base), NUS / RHS / u are unused by these two helpers.
#define TMX_UW (TMX_BW + 1)
// Layout of a U row in the stream (TMX_USTREAM16, default): five 16-byte slots per lane, [row][slot][64 lanes][2] -- one
// global_store_dwordx4 / global_load_dwordx4 moves two entries.  Both phases of the solve are bound by the ISSUE of vector-memory
// instructions, not by bytes (measured: the producer's five prefetch loads per block row waited 1 000 - 2 800 cycles to
// issue behind the elimination's ten 8-byte stores per pivot; the back substitution slowed by 30 % when the same bytes were
// moved by twice as many instructions), so the stream is written and read in half as many, twice as wide instructions.
// Which two entries share a slot depends on the KIND of matrix row (j mod 3: rho*theta, W, rho), chosen so that the
// fill-in pattern of the usual pivot order packs without holes: entries 0..3 and the right-hand side (index 9) always
// exist, of the fill-in entries 4..8 a rho*theta row normally has {4,5,7}, a W row {4,6}, a rho row {5}.  Slots 0..2 are
// always stored; slot 3 / 4 only if one of its two entries is non-zero in some lane (mask bits 3 / 4).  Any other pivot
// order is still exact, it merely stores a few zeros.
#ifndef TMX_USTREAM16
#define TMX_USTREAM16 1
#endif
__device__ constexpr int TMX_UPAIR[3][5][2] = {
	{ { 0, 9 }, { 1, 2 }, { 3, 4 }, { 5, 7 }, { 6, 8 } },      // rho*theta rows
	{ { 0, 9 }, { 1, 2 }, { 3, 4 }, { 6, 5 }, { 7, 8 } },      // W rows
	{ { 0, 9 }, { 1, 2 }, { 3, 5 }, { 4, 6 }, { 7, 8 } } };    // rho rows
#define TMX_USLOT (TMX_USTREAM16 ? 2 : 1)      // doubles per lane and slot
// TMX_USTREAM_NT: the U-row stream is written once and read once, 5 KB per column and direction -- bit 0: the elimination's stores,
// bit 1: the back substitution's loads carry the non-temporal hint.  Measured (round 5, ne30 L30, same box, two alternations): the
// column solve itself does not move (0.476-0.492 ms per step with any of 0..3), but the stream no longer evicts the state from the
// L2 / Infinity Cache: the explicit kernels that follow go from 0.578 to 0.552 (stores), 0.549 (loads), 0.535 ms per step (both),
// the step from 1.394 to 1.350 ms.  Default: both -- except in the small-grid variant of the pair kernel (NTS below; profiles/r05_stream_hint_small_grids.txt).
#ifndef TMX_USTREAM_NT
#define TMX_USTREAM_NT 3
#endif
typedef double tmx_dbl2 __attribute__((ext_vector_type(2)));
// NTS: this instantiation's stream carries the hint at all (the small-grid variant of the pair kernel does not: at ne15 the whole stream, 66 MB,
// stays in the Infinity Cache between the phases and the hint costs 2 % of the step; at ne20, 125 MB, it already pays)
#define TMX_NTS_DEFAULT (TMX_USTREAM_NT != 0)
template <bool NTS>
__device__ __forceinline__ void ustream_store2(double * q, double a, double b) {
	tmx_dbl2 v; v.x = a; v.y = b;
	if (NTS && (TMX_USTREAM_NT & 1)) __builtin_nontemporal_store(v, (tmx_dbl2 *)q);
	else *(tmx_dbl2 *)q = v;
}
template <bool NTS>
__device__ __forceinline__ double2 ustream_load2(const double * q) {
	tmx_dbl2 v;
	if (NTS && (TMX_USTREAM_NT & 2)) v = __builtin_nontemporal_load((const tmx_dbl2 *)q);
	else v = *(const tmx_dbl2 *)q;
	return make_double2(v.x, v.y);
}
// Order of the rows in the scratch (TMX_USTREAM_ROWMAJOR, default): [matrix row][column group][slot][lane] -- the column groups
// advance through their rows at about the same pace, so at any time the chip writes (elimination) or reads (back substitution)
// one contiguous run of a few megabytes instead of one 5 KB piece in each of ~800 chunks 476 KB apart.  0: [group][row]....
// NUS below is the number of stream columns allocated (64 per column group).
#ifndef TMX_USTREAM_ROWMAJOR
#define TMX_USTREAM_ROWMAJOR 1
#endif
#define TMX_RS(NUS) (TMX_USTREAM_ROWMAJOR ? (size_t)(NUS) * TMX_UW : (size_t)TMX_UW * 64)      // doubles from one row of a group to its next
__device__ __forceinline__ double * vi_stream_base(double * AB, int n, int u) {
	return AB + ((size_t)(u >> 6) * (TMX_USTREAM_ROWMAJOR ? 1 : n) * TMX_UW) * 64 + (u & 63) * TMX_USLOT;
}
// the same with the column group and the lane given separately (groups of fewer than 64 columns, k_vi_pair)
__device__ __forceinline__ double * vi_stream_base_g(double * AB, int n, int grp, int lane) {
	return AB + ((size_t)grp * (TMX_USTREAM_ROWMAJOR ? 1 : n) * TMX_UW) * 64 + lane * TMX_USLOT;
}

// PRED: the sparse stores of the fill-in entries are predicated instead of branched around: the condition "some lane holds a
// non-zero" is wave-uniform, so the compiler would put an s_cbranch_vccz in front of every one of the stores (a VALU
// compare, the wait for VCC and a branch on the elimination's critical chain).  Folding an opaque per-lane `1` into the
// condition makes it divergent for the compiler: the store then runs under an EXEC mask that is all ones or all zeros (an
// instruction with an empty mask issues and does nothing), no branch.
// TYPE: kind of the matrix row jcol (jcol mod 3), a compile-time constant at every call site.
template <bool SPARSE, bool PRED, int TYPE, bool NTS = TMX_NTS_DEFAULT>
__device__ __forceinline__ void lu_store_urow(const double (&piv)[TMX_BW], double pb, double * __restrict__ AB, double * __restrict__ RHS,
	int jcol, int NUS, int u, int * umask)
{
#if TMX_USTREAM16
	auto val = [&](int e) -> double { return e == 9 ? pb : piv[e]; };
	double * row = AB + (size_t)jcol * TMX_RS(NUS);
#pragma unroll
	for (int sl = 0; sl < 3; sl++)
		ustream_store2<NTS>(row + sl * 128, val(TMX_UPAIR[TYPE][sl][0]), val(TMX_UPAIR[TYPE][sl][1]));
	int mask = 0;
	int one = 1;
	if (PRED) asm volatile("" : "+v"(one));
#pragma unroll
	for (int sl = 3; sl < 5; sl++) {
		const double a = val(TMX_UPAIR[TYPE][sl][0]), b = val(TMX_UPAIR[TYPE][sl][1]);
		const bool any = !SPARSE || __builtin_amdgcn_ballot_w64(a != 0.0 || b != 0.0) != 0;
		mask |= any ? (1 << sl) : 0;
		if (PRED ? (any && one) : any) ustream_store2<NTS>(row + sl * 128, a, b);
	}
	if (SPARSE ? (threadIdx.x == 0) : true) umask[jcol] = mask;      // dense stores (also the lane-divergent tails, where lane 0 may be masked out): every active lane writes the same word
#else
	if (SPARSE && PRED) {
#pragma unroll
		for (int c = 0; c < 4; c++) AB[(size_t)jcol * TMX_RS(NUS) + c * 64] = piv[c];
		AB[(size_t)jcol * TMX_RS(NUS) + TMX_BW * 64] = pb;
		int one = 1;
		asm volatile("" : "+v"(one));
		int mask = 0;
#pragma unroll
		for (int c = 4; c < TMX_BW; c++) {
			const bool any = __builtin_amdgcn_ballot_w64(piv[c] != 0.0) != 0;
			mask |= any ? (1 << c) : 0;
			if (any && one) AB[(size_t)jcol * TMX_RS(NUS) + c * 64] = piv[c];
		}
		if (threadIdx.x == 0) umask[jcol] = mask;
		return;
	}
#pragma unroll
	for (int c = 0; c < 4; c++) AB[(size_t)jcol * TMX_RS(NUS) + c * 64] = piv[c];
	int mask = 0;
#pragma unroll
	for (int c = 4; c < TMX_BW; c++) {
		if (!SPARSE || __builtin_amdgcn_ballot_w64(piv[c] != 0.0) != 0) {
			AB[(size_t)jcol * TMX_RS(NUS) + c * 64] = piv[c];
			mask |= 1 << c;
		}
	}
	if (SPARSE ? (threadIdx.x == 0) : true) umask[jcol] = mask;      // dense stores (also the lane-divergent tails, where lane 0 may be masked out): every active lane writes the same word
	AB[(size_t)jcol * TMX_RS(NUS) + TMX_BW * 64] = pb;
#endif
}

// The LDS ring between the two wavefronts of k_vi_pair holds a block row of the column Jacobian WITHOUT its structural zeros:
// compute_block only ever touches the band entries d = 1,4,5,7,8 of a rho*theta row, d = 0,1,2,3,4,5,7 of a W row and
// d = 1,3,4,6,7 of a rho row (DOFF of its statements; every other entry keeps the +0.0 it was initialised with), i.e. 17 of the
// 27 entries, plus the three residuals: 20 doubles per column instead of 30.  A third fewer LDS writes and reads per block row,
// and room for a THREE-deep ring in the LDS that held two dense block rows (TMX_RING_DEPTH).  TMX_RSLOT[kind][d] = slot of band
// entry d of the row of that kind, -1 = structural zero (the consumer takes the literal 0.0 the dense ring carried).
#define TMX_RING_NQ 20
__device__ constexpr int TMX_RSLOT[3][TMX_BW] = {
	{ -1, 0, -1, -1, 1, 2, -1, 3, 4 },          // rho*theta row
	{ 5, 6, 7, 8, 9, 10, -1, 11, -1 },          // W row
	{ -1, 12, -1, 13, 14, -1, 15, 16, -1 } };   // rho row
#define TMX_RSLOT_RHS 17                        // + kind
// entry c of the incoming matrix row: INK < 0: a dense row of stride IS (registers, zeros); INK = kind: the ring's block row at `in`
template <int INK, int IS>
__device__ __forceinline__ double lu_in_entry(const double * in, int c) {
	if (INK < 0) return in[c * IS];
	return (TMX_RSLOT[INK < 0 ? 0 : INK][c] >= 0) ? in[TMX_RSLOT[INK < 0 ? 0 : INK][c] * 64] : 0.0;
}

// Pivot step when every lane of the wavefront picked window row PV (the normal case: neighbouring columns have
// nearly identical matrices).  The interchange is a compile-time renaming and the rank-1 update writes straight
// into the slid position -- new row r-1, column c-1  <-  (row that sits at r after the swap)[c] - l_r * pivot[c] --
// so neither the 160 v_cndmask of the per-lane interchange nor the 36 moves of the slide are executed.  Only the
// pivot row and old row 0 are copied aside, because their slots are overwritten before their last use.
// Same multiplications and subtractions on the same operands as the generic path: bit-identical.
template <bool SPARSE, int PV, int IS, int TYPE, int INK = -1, bool NTS = TMX_NTS_DEFAULT>
__device__ __forceinline__ void lu_pivot_tail_uniform(double (&R)[5][TMX_BW], double (&b)[5], bool & singular,
	double * __restrict__ AB, double * __restrict__ RHS, int jcol, int NUS, int u,
	const double * inrow, double inb, int * umask)
{
	// The last window column of rows 0..3 is zero before every step (the slide clears it; only the incoming row 4 brings an
	// entry there): written as literals, so that those four zeros do not occupy registers -- they were spilled, and every
	// reload in this tail sat behind an s_waitcnt vmcnt(0) that also drained the U-row stores of the previous pivot.
	constexpr int Z = TMX_BW - 1;
	double piv[TMX_BW], t0[TMX_BW];
#pragma unroll
	for (int c = 0; c < TMX_BW; c++) { piv[c] = (c == Z && PV != 4) ? 0.0 : R[PV][c]; t0[c] = (c == Z) ? 0.0 : R[0][c]; }
	const double pb = b[PV], tb0 = b[0];
	if (piv[0] == 0.0) singular = true;
	// The reciprocal of the pivot and the four multipliers are started BEFORE the U-row stores: the IEEE division is a chain
	// of a dozen dependent instructions, and the (conditional) stores -- which nothing waits for -- cover its latency.  The
	// first entry of the incoming row is what the NEXT pivot search waits for: its LDS read is issued up front as well.
	const double in0 = lu_in_entry<INK, IS>(inrow, 0);
	const double rcp = 1.0 / piv[0];
	double l[5];
#pragma unroll
	for (int r = 1; r < 5; r++) l[r] = ((r == PV) ? t0[0] : R[r][0]) * rcp;
	lu_store_urow<SPARSE, TMX_STORE_PRED, TYPE, NTS>(piv, pb, AB, RHS, jcol, NUS, u, umask);
#pragma unroll
	for (int r = 1; r < 5; r++) {
		// after the interchange position r holds old row r, except position PV, which holds old row 0
#pragma unroll
		for (int c = 1; c < TMX_BW; c++) {
			const double src = (r == PV) ? t0[c] : ((c == Z && r != 4) ? 0.0 : R[r][c]);
			R[r - 1][c - 1] = LU_UPD(src, l[r], piv[c]);
		}
		R[r - 1][TMX_BW - 1] = 0.0;
		b[r - 1] = LU_UPD((r == PV) ? tb0 : b[r], pb, l[r]);
	}
	R[4][0] = in0;
#pragma unroll
	for (int c = 1; c < TMX_BW; c++) R[4][c] = lu_in_entry<INK, IS>(inrow, c);      // IS: stride of the incoming row (1 registers, 64 LDS ring)
	b[4] = inb;
}

// One elimination step.  UNI: try the wave-uniform specialisations first (a ballot checks that all lanes agree on
// the pivot row); lanes that disagree -- and kernels without the register budget for five specialised tails -- take
// the per-lane select path below.
template <int TYPE, bool SPARSE, bool UNI = false, int IS = 1, int EXPECT = -1, int INK = -1, bool NTS = TMX_NTS_DEFAULT>
__device__ __forceinline__ void lu_pivot_step(double (&R)[5][TMX_BW], double (&b)[5], bool & singular,
	double * __restrict__ AB, double * __restrict__ RHS, int jcol, int NUS, int u,
	const double * inrow, double inb, int * umask, int * n_uniform = nullptr)
{
	int pv = 0;
	if (UNI && TMX_SEARCH_TREE) {
		// first maximum of |column 0| (dgbtf2's idamax) as a tree: rows 0/1 and 2/3 side by side; strict comparisons keep the
		// lower row on ties, so the result is the sequential first maximum
		const double a0 = fabs(R[0][0]), a1 = fabs(R[1][0]), a2 = fabs(R[2][0]), a3 = fabs(R[3][0]), a4 = fabs(R[4][0]);
		const bool g1 = a1 > a0, g3 = a3 > a2;
		const double m01 = g1 ? a1 : a0, m23 = g3 ? a3 : a2;
		const int i01 = g1 ? 1 : 0, i23 = g3 ? 3 : 2;
		const bool gh = m23 > m01;
		const double m03 = gh ? m23 : m01;
		const int i03 = gh ? i23 : i01;
		pv = (a4 > m03) ? 4 : i03;
	} else {
		double amax = fabs(R[0][0]);
#pragma unroll
		for (int r = 1; r < 5; r++) {
			const double v = fabs(R[r][0]);
			if (v > amax) { amax = v; pv = r; }
		}
	}
	if (UNI) {
		const int pv0 = __builtin_amdgcn_readfirstlane(pv);
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(pv != pv0) == 0, 1)) {      // (the hint keeps the register allocator's split points out of this path)
			if (n_uniform) (*n_uniform)++;      // wave-uniform counter (scalar add): pivot steps that took the renaming path
			// the usual pivot row of this kind of matrix row first (rho*theta / W / rho rows pivot 4 / 3 / 2 rows down almost
			// everywhere): one compare and branch instead of the switch's two or three
			if (EXPECT >= 0 && __builtin_expect(pv0 == EXPECT, 1)) { lu_pivot_tail_uniform<SPARSE, (EXPECT >= 0 ? EXPECT : 0), IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return; }
			switch (pv0) {
				case 0: lu_pivot_tail_uniform<SPARSE, 0, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				case 1: lu_pivot_tail_uniform<SPARSE, 1, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				case 2: lu_pivot_tail_uniform<SPARSE, 2, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				case 3: lu_pivot_tail_uniform<SPARSE, 3, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				default: lu_pivot_tail_uniform<SPARSE, 4, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
			}
		}
#ifndef TMX_VI_SELECT_PATH
		// The lanes disagree (rare: neighbouring columns have nearly the same matrix): the same five tails, one pivot row at
		// a time under the lanes that chose it.  A lane executes exactly the tail of its own pivot row, i.e. the statements
		// of the per-lane select path below on the same operands; every U-row entry is stored (a fill-in entry that is zero
		// in the lanes of one tail need not be zero in the others, so the all-lanes ballot of the sparse stores does not
		// apply).  Unlike the select path -- whose 160 selects per step cost the kernel 30 spilled registers, reloaded in
		// the uniform tails behind an s_waitcnt vmcnt(0) that also drained the U-row stores -- this adds no live values.
		if (pv == 0) lu_pivot_tail_uniform<false, 0, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 1) lu_pivot_tail_uniform<false, 1, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 2) lu_pivot_tail_uniform<false, 2, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 3) lu_pivot_tail_uniform<false, 3, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 4) lu_pivot_tail_uniform<false, 4, IS, TYPE, INK, NTS>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		return;
#endif
	}
#pragma unroll
	for (int r = 1; r < 5; r++) {
		const bool sw = (pv == r);
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) {
			const double a0 = R[0][c], ar = R[r][c];
			R[0][c] = sw ? ar : a0;
			R[r][c] = sw ? a0 : ar;
		}
		const double b0 = b[0], br = b[r];
		b[0] = sw ? br : b0;
		b[r] = sw ? b0 : br;
	}
	if (R[0][0] == 0.0) singular = true;
	lu_store_urow<SPARSE, false, TYPE, NTS>(R[0], b[0], AB, RHS, jcol, NUS, u, umask);
	const double rcp = 1.0 / R[0][0];
#pragma unroll
	for (int r = 1; r < 5; r++) {
		const double l = R[r][0] * rcp;
#pragma unroll
		for (int c = 1; c < TMX_BW; c++) R[r][c] = LU_UPD(R[r][c], l, R[0][c]);
		b[r] = LU_UPD(b[r], b[0], l);
	}
#pragma unroll
	for (int r = 0; r < 4; r++) {
#pragma unroll
		for (int c = 0; c < TMX_BW - 1; c++) R[r][c] = R[r + 1][c + 1];
		R[r][TMX_BW - 1] = 0.0;
		b[r] = b[r + 1];
	}
#pragma unroll
	for (int c = 0; c < TMX_BW; c++) R[4][c] = lu_in_entry<INK, IS>(inrow, c);      // IS: stride of the incoming row (1 registers, 64 LDS ring)
	b[4] = inb;
}

// Back substitution (dtbsv order) over the U rows and transformed right-hand side the elimination left in the
// HBM scratch, result x0 - delta scattered to the column and its in-patch duplicates.  Rows are consumed three per
// loop trip (rho, W, rho*theta of one level) from three register sets; the loads that refill a set are issued right
// after it is consumed, i.e. three rows ahead of their use, without register-to-register rotation.  Everything the
// trip computes and loads sits in ONE basic block with a FIXED number of loads per row: a fill-in entry that the
// elimination did not store (umask) is read from a zero page behind RHS through a scalar-selected base instead of
// being skipped by a branch, the level-L slots that have no slab are peeled off into the first trip, and the result
// stores (the only conditional memory operations: duplicates exist for some lanes only) come last.  With
// branches between the rows the compiler could neither overlap one row's division with the next rows' partial sums
// nor count the loads in flight (it then waits for nearly all of them at every use: 1 180 cycles per row).
// n = 3(L+1).
struct URow { double u[TMX_BW]; double y, x0; };

template <int COMP, bool NTS = TMX_NTS_DEFAULT>
__device__ __forceinline__ void vi_load_urow(URow & d, int row, int L, size_t NS, int u, int col, int NUS,
	const double * __restrict__ AB, const double * __restrict__ RHS, const double * xin, const int * umask, const double * zp)
{
	const int rr = row < 0 ? 0 : row;
	const int mk = __builtin_amdgcn_readfirstlane(umask[rr]);
#if TMX_USTREAM16
	const double * rb = AB + (size_t)rr * TMX_RS(NUS);
	double2 v[5];
#pragma unroll
	for (int sl = 0; sl < 3; sl++) v[sl] = ustream_load2<NTS>(rb + sl * 128);
#pragma unroll
	for (int sl = 3; sl < 5; sl++) {
		const double * bp = (mk & (1 << sl)) ? rb + sl * 128 : zp;
		v[sl] = ustream_load2<NTS>(bp);
	}
#pragma unroll
	for (int sl = 0; sl < 5; sl++) {
		const int ea = TMX_UPAIR[COMP][sl][0], eb = TMX_UPAIR[COMP][sl][1];
		if (ea == 9) d.y = v[sl].x; else d.u[ea] = v[sl].x;
		if (eb == 9) d.y = v[sl].y; else d.u[eb] = v[sl].y;
	}
#else
	const double * rb = AB + (size_t)rr * TMX_RS(NUS);
#pragma unroll
	for (int c = 0; c < 4; c++) d.u[c] = rb[c * 64];
#pragma unroll
	for (int c = 4; c < TMX_BW; c++) {
		const double * bp = (mk & (1 << c)) ? rb + c * 64 : zp;
		d.u[c] = bp[0];
	}
	d.y = rb[TMX_BW * 64];
#endif
	const int k = rr / TMX_FTOT;
	const int kc = (COMP == 1 || k < L) ? k : L - 1;      // level-L slots of rho*theta / rho have no slab: value unused
	const int slab = (COMP == 0) ? TMX_SLAB_T(L, kc) : ((COMP == 1) ? TMX_SLAB_W(L, kc) : TMX_SLAB_R(L, kc));
	d.x0 = xin[(size_t)slab * NS + col];
}

__device__ __forceinline__ double vi_back_row(const URow & s, double (&xw)[TMX_BW - 1]) {
	double acc = s.y;
#pragma unroll
	for (int c = TMX_BW - 1; c >= 1; c--) acc = LU_UPD(acc, xw[c - 1], s.u[c]);
	const double xr = acc / s.u[0];
#pragma unroll
	for (int c = TMX_BW - 2; c >= 1; c--) xw[c] = xw[c - 1];
	xw[0] = xr;
	return xr;
}

template <bool NTS = TMX_NTS_DEFAULT>
__device__ __forceinline__ void vi_back_substitute(int L, size_t NS, int n, int u, int col, int NUS,
	const double * __restrict__ AB, const double * __restrict__ RHS, const double * xin, double * xup,
	const int * __restrict__ udep, const int * umask, int zlane = -1)
{
	int dep[3];
#pragma unroll
	for (int t = 0; t < 3; t++) dep[t] = udep ? udep[u * 3 + t] : -1;
	double xw[TMX_BW - 1];
#pragma unroll
	for (int c = 0; c < TMX_BW - 1; c++) xw[c] = 0.0;
	// zero page behind the streams (one slot per lane); AB is the lane's stream pointer (vi_stream_base)
	const double * zp = RHS + ((zlane >= 0) ? zlane : (u & 63)) * TMX_USLOT;
	// two trips (six rows) of register sets: a set is refilled for the trip after next right after it is consumed,
	// so five rows = 55 loads are in flight (vmcnt holds 63) -- the back substitution of a full grid is HBM-bound
	URow A, B, C, D, E, F;      // A,D: rows 3k+2 (rho); B,E: 3k+1 (W); C,F: 3k (rho*theta)
#define LOADROW(S, COMP, row) vi_load_urow<COMP, NTS>(S, row, L, NS, u, col, NUS, AB, RHS, xin, umask, zp)
	LOADROW(A, 2, n - 1); LOADROW(B, 1, n - 2); LOADROW(C, 0, n - 3);
	LOADROW(D, 2, n - 4); LOADROW(E, 1, n - 5); LOADROW(F, 0, n - 6);
	{
		// level L: only W has a slab
		vi_back_row(A, xw);
		LOADROW(A, 2, n - 7);
		const double vb = B.x0 - vi_back_row(B, xw);
		LOADROW(B, 1, n - 8);
		vi_back_row(C, xw);
		LOADROW(C, 0, n - 9);
		const size_t sW = (size_t)TMX_SLAB_W(L, L) * NS;
		xup[sW + col] = vb;
#pragma unroll
		for (int t = 0; t < 3; t++) if (dep[t] >= 0) xup[sW + dep[t]] = vb;
	}
	// one level: rows 3k+2, 3k+1, 3k from the sets (X, Y, Z), which are refilled with the rows of level k - 2
#define TRIP(X, Y, Z, k) do { \
		const int k_ = (k); \
		const double va = X.x0 - vi_back_row(X, xw); \
		LOADROW(X, 2, 3 * k_ - 4); \
		const double vb = Y.x0 - vi_back_row(Y, xw); \
		LOADROW(Y, 1, 3 * k_ - 5); \
		const double vc = Z.x0 - vi_back_row(Z, xw); \
		LOADROW(Z, 0, 3 * k_ - 6); \
		const size_t sR = (size_t)TMX_SLAB_R(L, k_) * NS, sW = (size_t)TMX_SLAB_W(L, k_) * NS, sT = (size_t)TMX_SLAB_T(L, k_) * NS; \
		xup[sR + col] = va; xup[sW + col] = vb; xup[sT + col] = vc; \
		_Pragma("unroll") for (int t = 0; t < 3; t++) \
			if (dep[t] >= 0) { xup[sR + dep[t]] = va; xup[sW + dep[t]] = vb; xup[sT + dep[t]] = vc; } \
	} while (0)
	int k = L - 1;
	for (; k >= 1; k -= 2) {
		TRIP(D, E, F, k);
		TRIP(A, B, C, k - 1);
	}
	if (k == 0) TRIP(D, E, F, 0);
#undef TRIP
#undef LOADROW
}

template <bool CARRY, bool SPARSE, bool CLOSED, bool UNI = false>
__global__ __launch_bounds__(64) void k_vi_fused(KParams p, const double * xin, double * xup, double dt,
	int nunique, int NUS, const int * __restrict__ ucol, const int * __restrict__ udep,
	double * __restrict__ AB, double * __restrict__ RHS, int * __restrict__ flag)
{
	// column-operator coefficients in LDS: read with ds_read (lgkmcnt), so that they neither need
	// vector-memory round trips nor force vmcnt(0) waits that would drain the prefetches and stores
	extern __shared__ double opsl_mem[];
	double * opsl = opsl_mem;
	const int L = p.L;
	constexpr int MM = CLOSED ? 1 : 2;
	double * etal = opsl_mem + TMX_OP_COUNT * (L + 1) * TMX_OPW;             // [2L+1] 1 - eta (closed-form metric), else unused
	int * umask = (int *)(etal + 2 * L + 1);                                 // [3(L+1)] which fill-in entries of a U row exist
	for (int t = threadIdx.x; t < TMX_OP_COUNT * (L + 1) * TMX_OPW; t += 64) opsl[t] = p.ops[t];
	if (CLOSED) for (int t = threadIdx.x; t < 2 * L + 1; t += 64) etal[t] = p.eta[t];
	__syncthreads();
	const size_t NS = (size_t)p.NS;
	const int u = blockIdx.x * 64 + threadIdx.x;
	if (u >= nunique) return;
	const int n = TMX_FTOT * (L + 1);
	const int col = ucol ? ucol[u] : u;      // (null: node-unique layout, a unique column is its own state column)
	RHS = AB + (size_t)n * TMX_UW * NUS;      // zero page behind the streams (vi_back_substitute)
	AB = vi_stream_base(AB, n, u);            // this lane's slot in its wavefront's U-row stream
	ColConst cc;
	cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
	cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
	cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
	const MetCol mcol = met_col(p, col);

	// level window for block k: nodes k-1,k,k+1 and interfaces k-1,k,k+1
	NodeLev nA = load_node<MM>(p, mcol, xin, -1, col, etal), nB = load_node<MM>(p, mcol, xin, 0, col, etal), nC = load_node<MM>(p, mcol, xin, 1, col, etal);
	EdgeLev eA = load_edge<MM>(p, mcol, xin, -1, col, etal), eB = load_edge<MM>(p, mcol, xin, 0, col, etal), eC = load_edge<MM>(p, mcol, xin, 1, col, etal);
	NodeLev nN = load_node<MM>(p, mcol, xin, 2, col, etal);
	EdgeLev eN = load_edge<MM>(p, mcol, xin, 2, col, etal);

	double R[5][TMX_BW], b[5];
	double rP[TMX_BW], rW[TMX_BW], rR[TMX_BW], fP, fW, fR;
	double sP[TMX_BW], sW[TMX_BW], sR[TMX_BW], gP, gW, gR;
	// blocks 0 and 1 -> matrix rows 0..5
	BlkCarry cy = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	compute_block<CARRY>(p, opsl, cc, 0, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy);
	nA = nB; nB = nC; nC = nN; eA = eB; eB = eC; eC = eN;
	nN = load_node<MM>(p, mcol, xin, 3, col, etal); eN = load_edge<MM>(p, mcol, xin, 3, col, etal);
	compute_block<CARRY>(p, opsl, cc, 1, nA, nB, nC, eA, eB, eC, sP, sW, sR, gP, gW, gR, cy);
	// window row r holds matrix columns 0..8; band entry d of row r sits at column r - 4 + d
#pragma unroll
	for (int c = 0; c < TMX_BW; c++) {
		R[0][c] = (c + 4 < TMX_BW) ? rP[c + 4] : 0.0;         // row 0: d = c + 4
		R[1][c] = (c + 3 < TMX_BW) ? rW[c + 3] : 0.0;         // row 1: d = c + 3
		R[2][c] = (c + 2 < TMX_BW) ? rR[c + 2] : 0.0;
		R[3][c] = (c + 1 < TMX_BW) ? sP[c + 1] : 0.0;
		R[4][c] = sW[c];
	}
	b[0] = fP; b[1] = fW; b[2] = fR; b[3] = gP; b[4] = gW;
	bool singular = false;
	lu_pivot_step<0, SPARSE, UNI>(R, b, singular, AB, RHS, 0, NUS, u, sR, gR, umask);      // pivot 0 takes in row 5
	// Drain the prologue's loads here, once: otherwise the per-column constants loaded before the loop count as
	// "possibly still in flight" at the loop header and the compiler waits vmcnt(0) -- i.e. for all U-row
	// stores of the previous block row -- at their first use in EVERY iteration.
	__builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#ifdef TMX_VI_TIMING
	long long t_asm = 0, t_lu = 0, t_all0 = __builtin_readcyclecounter();
#define TMARK(acc) do { const long long t1_ = __builtin_readcyclecounter(); acc += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define TMARK(acc)
#endif
	for (int m = 2; m <= L; m++) {
#ifdef TMX_VI_TIMING
		long long t0_ = __builtin_readcyclecounter();
#endif
		nA = nB; nB = nC; nC = nN; eA = eB; eB = eC; eC = eN;
		nN = load_node<MM>(p, mcol, xin, m + 2, col, etal); eN = load_edge<MM>(p, mcol, xin, m + 2, col, etal);   // used two blocks from now
		compute_block<CARRY>(p, opsl, cc, m, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy);
		TMARK(t_asm);
		lu_pivot_step<1, SPARSE, UNI>(R, b, singular, AB, RHS, 3 * m - 5, NUS, u, rP, fP, umask);
		lu_pivot_step<2, SPARSE, UNI>(R, b, singular, AB, RHS, 3 * m - 4, NUS, u, rW, fW, umask);
		lu_pivot_step<0, SPARSE, UNI>(R, b, singular, AB, RHS, 3 * m - 3, NUS, u, rR, fR, umask);
		TMARK(t_lu);
	}
#ifdef TMX_VI_TIMING
	const long long t_fwd = __builtin_readcyclecounter() - t_all0;
#endif
	{
		double z[TMX_BW];
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) z[c] = 0.0;
		// the last five rows: n is a multiple of three, so their kinds are W, rho, rho*theta, W, rho
		lu_pivot_step<1, SPARSE, UNI>(R, b, singular, AB, RHS, n - 5, NUS, u, z, 0.0, umask);
		lu_pivot_step<2, SPARSE, UNI>(R, b, singular, AB, RHS, n - 4, NUS, u, z, 0.0, umask);
		lu_pivot_step<0, SPARSE, UNI>(R, b, singular, AB, RHS, n - 3, NUS, u, z, 0.0, umask);
		lu_pivot_step<1, SPARSE, UNI>(R, b, singular, AB, RHS, n - 2, NUS, u, z, 0.0, umask);
		lu_pivot_step<2, SPARSE, UNI>(R, b, singular, AB, RHS, n - 1, NUS, u, z, 0.0, umask);
	}
	if (singular) atomicOr(flag, TMX_FLAG_SINGULAR);

	vi_back_substitute(L, NS, n, u, col, NUS, AB, RHS, xin, xup, udep, umask);
#ifdef TMX_VI_TIMING
	if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 7))
		printf("vi timing block %d: forward loop %lld cycles (assembly %lld, LU %lld), total incl. back substitution %lld\n",
			(int)blockIdx.x, t_fwd, t_asm, t_lu, (long long)(__builtin_readcyclecounter() - t_all0));
#endif
}

// usual pivot rows (window row index) of the W / rho / rho*theta rows of the column Jacobian, tried first by the pair kernel
#ifndef TMX_EXPECT_T
#define TMX_EXPECT_T 4
#define TMX_EXPECT_W 3
#define TMX_EXPECT_R 2
#endif
// Two-wavefront variant of k_vi_fused: the block rows of a column are independent of one another, only the
// elimination is sequential.  Wavefront 0 of a pair assembles block rows (compute_block, the same code) into an LDS ring,
// wavefront 1 runs the three pivot steps of each block row out of it.  The dependent chain of a column shrinks from
// assembly + LU to max(assembly, LU) (cycle counts in DESIGN.md), which is what bounds the kernel on small per-GPU grids.
// Same statements on the same operands: bit-identical to k_vi_fused.
// Round 5: the two roles no longer meet at a workgroup barrier after every block row.  The ring is TMX_RING_DEPTH block rows deep
// (sparse block rows, TMX_RSLOT above: three fit where two dense ones did) and the roles hand over through two LDS counters per
// pair -- `produced` (block rows published) and `consumed` -- so the assembly runs up to TMX_RING_DEPTH block rows ahead and a pair
// pays max(sum assembly, sum elimination) instead of the sum over block rows of max(assembly, elimination); with the barrier both
// roles spent 28 % of the forward phase waiting for each other (profiles/r04_pair_timing.txt).  Ordering: LDS executes a
// wavefront's instructions in order; the writer still drains lgkmcnt before it raises its counter, the reader polls the counter
// with a volatile read (s_sleep between polls) and a compiler barrier keeps the ring accesses behind it.  No vmcnt wait anywhere.
// BACK = false: the kernel ends with the elimination and leaves the fill-in masks of its U rows in `gmask` ([column group][3(L+1)]);
// k_vi_back does the back substitution as a launch of its own (one wavefront per workgroup, so that the HBM-bound half of the
// solve is spread evenly over the CUs whatever the residency of the pairs was).
#ifndef TMX_RING_DEPTH
#define TMX_RING_DEPTH 3
#endif
// (the counters are accessed through explicit LDS pointers: a volatile access through a generic pointer compiles to a FLAT load followed
// by s_waitcnt vmcnt(0), i.e. a wait for every U-row store in flight)
typedef __attribute__((address_space(3))) int tmx_lds_int;
// The poll loop is ONE asm statement: as a loop in the source it sat inside the block-row loop, and the register allocator answered the
// nested loop with 34 spilled registers in the elimination (reloaded behind s_waitcnt vmcnt, i.e. behind the U-row stores).
__device__ __forceinline__ void ring_wait(const int * counter, int need) {
	const unsigned addr = (unsigned)(size_t)(const tmx_lds_int *)counter;
	int v, sv;
	asm volatile("Lring_poll_%=:\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_cmp_ge_i32 %1, %3\n\t"
		"s_cbranch_scc1 Lring_go_%=\n\ts_sleep 1\n\ts_branch Lring_poll_%=\nLring_go_%=:"
		: "=&v"(v), "=&s"(sv) : "v"(addr), "s"(need) : "memory", "scc");
}
__device__ __forceinline__ void ring_post(int * counter, int value, int lane) {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wavefront's ring writes (producer) / ring reads (consumer) are done
	if (lane == 0) *(volatile tmx_lds_int *)counter = value;
}
// NPROD = 2 (with PAIRS = 1, closed-form metric; grids that leave SIMDs idle): TWO assembly wavefronts per elimination wavefront.  The block
// rows of a column are independent of one another, so wavefront q in {0, 1} assembles the block rows t = q (mod 2) -- without the carry
// between consecutive rows (compute_block<false>: the same statements on the same operands, one more Exner evaluation per row) -- into the
// same ring, and the elimination wavefront takes the rows in order from whichever producer made them (one `produced` counter each).  The
// forward phase of a column group then costs the elimination chain, not the longer assembly chain.
// RDT: block rows the ring holds (three where the LDS leaves room for two workgroups per CU with them, else two: from about 38 levels on the
// operator tables and three-deep rings of two pairs pass 80 KB per workgroup -- rounds 1-4 ran such grids, BASELINE config 5's 60 levels among
// them, with ONE workgroup per CU)
template <bool SPARSE, bool CLOSED, int PAIRS, bool BACK = true, int NPROD = 1, int RDT = TMX_RING_DEPTH>
__global__ __launch_bounds__(PAIRS * 64 * (1 + NPROD)) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vi_pair(KParams p, const double * xin, double * xup, double dt,
	int nunique, int NUS, const int * __restrict__ ucol, const int * __restrict__ udep,
	double * __restrict__ AB, double * __restrict__ RHS, int * __restrict__ flag, int cpw, unsigned long long * __restrict__ stats, int * __restrict__ gmask = nullptr,
	int stagger = 0)
{
	extern __shared__ double opsl_mem[];
	double * opsl = opsl_mem;
	const int L = p.L;
	constexpr int MM = CLOSED ? 1 : 2;
	constexpr int NQ = TMX_RING_NQ;                                          // doubles per block row and column in the ring (sparse: TMX_RSLOT)
	constexpr int RD = (NPROD == 2) ? RDT + 1 : RDT;      // (two rows are being written at a time)
	constexpr int NTHR = PAIRS * 64 * (1 + NPROD);
	static_assert(NPROD == 1 || (NPROD == 2 && PAIRS == 1 && CLOSED), "two producers: one column group per workgroup, closed-form metric");
	constexpr bool NTSK = TMX_NTS_DEFAULT && NPROD != 2;      // the small-grid variant's stream stays in the caches (ustream_store2)
	// Staggered start (stagger = window in units of 10 ns, 0 = off): workgroup i begins i / gridDim.x of the window late.  All
	// workgroups are resident at once and would otherwise run in lock-step -- every wavefront in its issue- and latency-bound
	// elimination first, then every wavefront in its HBM-bound back substitution; spread out, the back substitutions of the
	// early workgroups stream while the late ones still eliminate.  A sleeping wavefront issues nothing.
	if (stagger > 0) {
		const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
		const unsigned long long wait = (unsigned long long)stagger * blockIdx.x / gridDim.x;
		while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
	}
	double * etal = opsl_mem + TMX_OP_COUNT * (L + 1) * TMX_OPW;             // [2L+1]
	// PAIRS producer/consumer pairs per workgroup (2 pairs = 4 wavefronts = one per SIMD of a CU)
	// Wavefronts w and w + 4 of a workgroup share a SIMD (a workgroup's wavefronts go to the SIMDs in cyclic order).  With three
	// pairs (six wavefronts: one workgroup per CU, every CU carries the same load) the two doubly loaded SIMDs get
	// consumer + consumer and producer + consumer, never producer + producer (the assembly is the issue-heavy role):
	// wavefronts 0..5 = C0 P0 P1 P2 C1 C2.
	const int lane = threadIdx.x, wv = WAVE_UNIFORM(threadIdx.y);
	// (Measured and not kept, round 4: the roles in the other order -- consumer first -- for the workgroups of the second dispatch round, or
	// for every other workgroup, so that two workgroups sharing a CU would put a producer and a consumer on every SIMD: column solve
	// 0.500 / 0.486 against 0.480 ms per step.)
	// Round 6: WHICH of a CU's two workgroups takes the other order is no longer guessed from blockIdx (it helped only in the runs where the dispatcher
	// really put workgroups i and i + 256 on one CU: profiles/r06_column_roles_ab.txt) -- a workgroup claims one of two order slots of the CU it finds
	// itself on (hardware id registers; the slots live behind the engine's flag word) and releases it at its end.
#ifndef TMX_VI_CU_ORDER
#define TMX_VI_CU_ORDER 1
#endif
	constexpr bool CUORD = TMX_VI_CU_ORDER && PAIRS == 2 && NPROD == 1;
	__shared__ int s_order;
	const int role0 = (NPROD == 2) ? (wv == 2 ? 1 : 0) : ((PAIRS == 3) ? ((0x31 >> wv) & 1) : (wv & 1)), pair = (NPROD == 2) ? 0 : ((PAIRS == 3) ? ((0x990 >> (2 * wv)) & 3) : (wv >> 1));
	int * cu_slot = nullptr;
	if (CUORD) {
		const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_ID (CU, SH, SE in bits 8..15), XCC_ID
		cu_slot = flag + 64 + (((xcc & 15u) << 8) | ((hw >> 8) & 0xffu));
		if (wv == 0 && lane == 0) {
			int order = 0;
			if (atomicOr(cu_slot, 1) & 1) { order = 1; atomicOr(cu_slot, 2); }      // the first order is taken on this CU: the other one
			s_order = order;
		}
	}
	double * ring = etal + 2 * L + 1 + (size_t)pair * RD * NQ * 64;          // [PAIRS][RD][NQ][64]
	double * rmtab = etal + 2 * L + 1 + (size_t)PAIRS * RD * NQ * 64;       // [512] exp / log tables (exner_from_rhotheta_lds)
	int * umask = (int *)(rmtab + TMX_RMTAB_DOUBLES) + pair * TMX_FTOT * (L + 1);   // [PAIRS][3(L+1)]
	int * rcount = (int *)(rmtab + TMX_RMTAB_DOUBLES) + PAIRS * TMX_FTOT * (L + 1) + 4 * pair;   // [PAIRS]{produced, consumed, produced by the second producer, -}
	if (role0 == 1 && lane == 0) { rcount[0] = 0; rcount[1] = 0; rcount[2] = 0; }
	for (int t = wv * 64 + lane; t < TMX_OP_COUNT * (L + 1) * TMX_OPW; t += NTHR) opsl[t] = p.ops[t];
	if (CLOSED) for (int t = wv * 64 + lane; t < 2 * L + 1; t += NTHR) etal[t] = p.eta[t];
	tmx_rm_tables_to_lds(rmtab, wv * 64 + lane, NTHR);
	__syncthreads();
	const int order = CUORD ? WAVE_UNIFORM(s_order) : 0;
	const int role = role0 ^ order;      // (order 1: elimination first, assembly second -- with the CU's other workgroup an assembly and an elimination wavefront on every SIMD)
	const size_t NS = (size_t)p.NS;
	// cpw columns per wavefront (64, or fewer so that the column groups fill every SIMD of the chip evenly: tmxk_vi_fused).
	// Lanes beyond the group's columns do not return -- the block-row loops below are full of workgroup barriers -- they
	// redo the group's last column (the last column of the grid in a trailing, empty group): same values to the same
	// addresses of the state, a stream slot of their own.
	const int grp = blockIdx.x * PAIRS + pair;
	const int ulast = min(nunique, (grp + 1) * cpw) - 1;
	const int u = (grp * cpw < nunique) ? min(grp * cpw + min(lane, cpw - 1), ulast) : nunique - 1;
	const int n = TMX_FTOT * (L + 1);
	const int col = ucol ? ucol[u] : u;      // (null: node-unique layout, a unique column is its own state column)
	RHS = AB + (size_t)n * TMX_UW * NUS;      // zero page behind the streams (vi_back_substitute)
	AB = vi_stream_base_g(AB, n, grp, lane);  // this lane's slot in its wavefront's U-row stream

	if (role == 0 && NPROD == 2) {
		// ---- one of two producers: block rows q, q + 2, ... without the carry; the two levels that enter the window per row are loaded two rows ahead ----
		const int q = wv;
		ColConst cc;
		cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
		cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
		cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
		const MetCol mcol = met_col(p, col);
		struct LevRaw2 { double un, vn, pn, rn, we; };
		auto load_raw2 = [&](int l) {
			LevRaw2 r;
			const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l), le = l < 0 ? 0 : (l > L ? L : l);
			r.un = xin[TMX_SLAB_U(L, lc) * NS + col]; r.vn = xin[TMX_SLAB_V(L, lc) * NS + col];
			r.pn = xin[TMX_SLAB_T(L, lc) * NS + col]; r.rn = xin[TMX_SLAB_R(L, lc) * NS + col];
			r.we = xin[TMX_SLAB_W(L, le) * NS + col];
			return r;
		};
		auto enter2 = [&](const LevRaw2 & r, int l, NodeLev & n, EdgeLev & e) {
			const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l), le = l < 0 ? 0 : (l > L ? L : l);
			n.un = r.un; n.vn = r.vn; n.pn = r.pn; n.rn = r.rn; e.we = r.we;
			metric_node3<MM>(p, mcol, lc, col, n.ca2, n.cb2, n.cx2, etal);
			metric_edge<MM>(p, mcol, le, col, e.ce0, e.ce1, e.ce2, etal);
		};
		NodeLev nA = load_node<MM>(p, mcol, xin, q - 1, col, etal), nB = load_node<MM>(p, mcol, xin, q, col, etal), nC = load_node<MM>(p, mcol, xin, q + 1, col, etal);
		EdgeLev eA = load_edge<MM>(p, mcol, xin, q - 1, col, etal), eB = load_edge<MM>(p, mcol, xin, q, col, etal), eC = load_edge<MM>(p, mcol, xin, q + 1, col, etal);
		LevRaw2 r1 = load_raw2(q + 2), r2 = load_raw2(q + 3);
		double rP[TMX_BW], rW[TMX_BW], rR[TMX_BW], fP, fW, fR;
		BlkCarry cy = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		for (int t = q; t <= L; t += 2) {
			const LevRaw2 n1 = load_raw2(t + 4), n2 = load_raw2(t + 5);
			compute_block<false>(p, opsl, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, 0.0, 0.0, rmtab);
			if (t >= RD) ring_wait(rcount + 1, t - RD + 1);
			double * qd = ring + (size_t)(t % RD) * NQ * 64 + lane;
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) {
				if (TMX_RSLOT[0][c] >= 0) qd[TMX_RSLOT[0][c] * 64] = rP[c];
				if (TMX_RSLOT[1][c] >= 0) qd[TMX_RSLOT[1][c] * 64] = rW[c];
				if (TMX_RSLOT[2][c] >= 0) qd[TMX_RSLOT[2][c] * 64] = rR[c];
			}
			qd[(TMX_RSLOT_RHS + 0) * 64] = fP; qd[(TMX_RSLOT_RHS + 1) * 64] = fW; qd[(TMX_RSLOT_RHS + 2) * 64] = fR;
			ring_post(rcount + 2 * q, (t >> 1) + 1, lane);
			nA = nC; eA = eC;
			enter2(r1, t + 2, nB, eB); enter2(r2, t + 3, nC, eC);
			r1 = n1; r2 = n2;
		}
		return;
	}
	if (role == 0) {
		// ---- producer: assemble block rows 0..L ----
		// (round 5 A/B, same box, two alternations: issue priority for this role too, for neither, or for this one alone moves the step by
		// less than +-0.5 % -- 1.331 / 1.334 / 1.325 / 1.338 ms -- although the assembly is now the slower role: left as it was)
		ColConst cc;
		cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
		cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
		cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
		const MetCol mcol = met_col(p, col);
		// Level window of block row t: nodes / interfaces t-1, t, t+1 in (A, B, C).  Under load the memory latency of this kernel is
		// longer than one block row (measured: with the loads of level t+2 moved into place at the top of the next iteration the
		// producer stalled 1 000 - 2 700 cycles per block row on its prefetch), and vmcnt retires in order, so a wait for the newest
		// load cannot be hidden behind older ones.  DEEP: block rows go in PAIRS; the raw values of the levels t+3 and t+4 are loaded
		// at the top of the pair's first row (staging sets Y0, Y1) and first touched at the end of its second row, two block rows
		// later -- so NOTHING in flight at any point where the compiler touches a loaded value is younger than two rows.  (Round 5
		// loaded one level at the top of every row: the set loaded by a pair's second row was live across the loop's back-edge, the
		// register allocator moved it there -- a move of registers whose loads are in flight is a wait for them -- and every second
		// row waited for loads one row old; a conditional second row made that an s_waitcnt vmcnt(0).)  The staging sets hold the
		// five loaded values of a level only; the closed-form metric of the level is evaluated when it enters the window
		// (stored-metric engines, whose levels carry six more loaded values, keep the one-row scheme: DEEP = false).
		constexpr bool DEEP = CLOSED;
		struct LevRaw { double un, vn, pn, rn, we; };
		auto load_raw = [&](int l) {
			LevRaw r;
			const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l), le = l < 0 ? 0 : (l > L ? L : l);
			r.un = xin[TMX_SLAB_U(L, lc) * NS + col]; r.vn = xin[TMX_SLAB_V(L, lc) * NS + col];
			r.pn = xin[TMX_SLAB_T(L, lc) * NS + col]; r.rn = xin[TMX_SLAB_R(L, lc) * NS + col];
			r.we = xin[TMX_SLAB_W(L, le) * NS + col];
			return r;
		};
		auto enter = [&](const LevRaw & r, int l, NodeLev & n, EdgeLev & e) {      // level l becomes the window's C
			const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l), le = l < 0 ? 0 : (l > L ? L : l);
			n.un = r.un; n.vn = r.vn; n.pn = r.pn; n.rn = r.rn; e.we = r.we;
			metric_node3<MM>(p, mcol, lc, col, n.ca2, n.cb2, n.cx2, etal);
			metric_edge<MM>(p, mcol, le, col, e.ce0, e.ce1, e.ce2, etal);
		};
		NodeLev nA = load_node<MM>(p, mcol, xin, -1, col, etal), nB = load_node<MM>(p, mcol, xin, 0, col, etal), nC = load_node<MM>(p, mcol, xin, 1, col, etal);
		EdgeLev eA = load_edge<MM>(p, mcol, xin, -1, col, etal), eB = load_edge<MM>(p, mcol, xin, 0, col, etal), eC = load_edge<MM>(p, mcol, xin, 1, col, etal);
		LevRaw rN1 = load_raw(2), rY0 = rN1, rY1 = rN1;
		NodeLev nN = load_node<MM>(p, mcol, xin, 2, col, etal);      // DEEP = false: level t+2, loaded one block row ahead
		EdgeLev eN = load_edge<MM>(p, mcol, xin, 2, col, etal);
		BlkCarry cy = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		double rP[TMX_BW], rW[TMX_BW], rR[TMX_BW], fP, fW, fR;
#ifdef TMX_PAIR_TIMING
		long long tp_work = 0, tp_wait = 0, tp_mark = __builtin_readcyclecounter();
		long long tp_seg[6] = { 0, 0, 0, 0, 0, 0 };
		const long long tp_start = tp_mark;
#define PMARK(acc) do { const long long t1_ = __builtin_readcyclecounter(); acc += t1_ - tp_mark; tp_mark = t1_; } while (0)
#else
#define PMARK(acc)
#endif
		// phase: 0 = first row of a pair (loads the levels that this pair's second row and the next pair's first row take in), 1 = second row,
		// 2 = a lone last row
		auto block_row = [&](int t, int phase) {
			if (DEEP && phase == 0) { rY0 = load_raw(t + 3); rY1 = load_raw(t + 4); }
#ifdef TMX_PAIR_TIMING
			{
			PMARK(tp_seg[4]);      // prefetch issue
			compute_block<true>(p, opsl, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, 0.0, 0.0, rmtab, tp_seg);
			tp_mark = __builtin_readcyclecounter();
			}
#else
			// Interior block rows whose Exner arguments are ordinary numbers in every lane can take the branch-free instantiation
			// (one basic block of 350 vector instructions per block row).  Measured: SLOWER -- the assembly alone 118 instead of
			// 98 us per launch, the kernel +3 % -- so it stays an experiment (-DTMX_INTERIOR_ROWS).
			compute_block<true>(p, opsl, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, 0.0, 0.0, rmtab);
#endif
			PMARK(tp_work);
			// the slot of block row t held block row t - RD: free once the consumer has taken that one in (evaluated first, waited for here)
			if (t >= RD) ring_wait(rcount + 1, t - RD + 1);
			PMARK(tp_wait);
			double * q = ring + (size_t)(t % RD) * NQ * 64 + lane;
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) {
				if (TMX_RSLOT[0][c] >= 0) q[TMX_RSLOT[0][c] * 64] = rP[c];
				if (TMX_RSLOT[1][c] >= 0) q[TMX_RSLOT[1][c] * 64] = rW[c];
				if (TMX_RSLOT[2][c] >= 0) q[TMX_RSLOT[2][c] * 64] = rR[c];
			}
			q[(TMX_RSLOT_RHS + 0) * 64] = fP; q[(TMX_RSLOT_RHS + 1) * 64] = fW; q[(TMX_RSLOT_RHS + 2) * 64] = fR;
			ring_post(rcount, t + 1, lane);
			nA = nB; nB = nC; eA = eB; eB = eC;
			if (DEEP) {
				// first row: level t+2 from N1 (here since the last pair); second row: level t+2 = the pair's first load, and N1 <- its second
				if (phase != 1) enter(rN1, t + 2, nC, eC);
				else { enter(rY0, t + 2, nC, eC); rN1 = rY1; }
			}
			else { nC = nN; eC = eN; nN = load_node<MM>(p, mcol, xin, t + 3, col, etal); eN = load_edge<MM>(p, mcol, xin, t + 3, col, etal); }
#ifdef TMX_PAIR_TIMING
			PMARK(tp_seg[5]);      // ring writes + window shift (DEEP: touches loads issued two block rows earlier)
#endif
		};
		int t = 0;
		for (; t + 1 <= L; t += 2) {
			block_row(t, 0);
			block_row(t + 1, 1);
		}
		if (t <= L) block_row(t, 2);
#ifdef TMX_PAIR_TIMING
		if (stats && lane == 0) {
			unsigned long long * q = stats + 2 + 8 * ((size_t)(blockIdx.x * PAIRS + pair) * 2 + role);
			q[0] = 1 + role; q[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
			q[5] = (unsigned long long)((long long)__builtin_readcyclecounter() - tp_start); q[4] = tp_wait; q[3] = q[5] - q[4]; q[6] = 0;      // work = everything but the waits for a free ring slot
			(void)tp_work;
			// segments of the assembly: interface interpolation | flux rows (P, R) | Exner + node quantities | W row | shift + prefetch issue
			if (blockIdx.x % 16 == 0) printf("pairtiming producer wg %d pair %d segments: interp %lld flux %lld exner %lld wrow %lld prefetch issue %lld window shift %lld ring writes %lld\n",
				(int)blockIdx.x, pair, tp_seg[0], tp_seg[1], tp_seg[2], tp_seg[3], tp_seg[4], tp_seg[5], tp_work);
		}
#endif
		return;
	}

	// ---- consumer: pivoted band elimination on the 5x9 register window ----
	// Two wavefronts per SIMD: the elimination is the critical role, the assembly has slack (101 k against 146 k cycles
	// per column), so the consumer takes issue priority wherever it shares a SIMD (same-box A/B at ne30, three alternations
	// each: 202 -> 185 us per launch on one box, 213 -> 206 us on another; -DTMX_NO_SETPRIO builds without it).
	__builtin_amdgcn_s_setprio(3);
	// The incoming block row stays in the LDS ring until the pivot step that takes it in (stride-64 reads inside
	// lu_pivot_step) instead of being copied to 30 registers first: the consumer then fits the register budget of
	// two wavefronts per SIMD, which is what lets the pair kernel run on grids with more pairs than CUs x 2.
	double R[5][TMX_BW], b[5];
	bool singular = false;
	int nuni = 0;
#pragma unroll
	for (int r = 0; r < 5; r++) {
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) R[r][c] = 0.0;
		b[r] = 0.0;
	}
#ifdef TMX_PAIR_TIMING
	long long tp_work = 0, tp_wait = 0, tp_mark = __builtin_readcyclecounter();
	const long long tp_start = tp_mark;
#endif
	for (int m = 0; m <= L; m++) {
		PMARK(tp_work);
		if (NPROD == 2) ring_wait(rcount + 2 * (m & 1), (m >> 1) + 1);      // block row m is in the ring: made by producer m mod 2 as its (m / 2)-th
		else ring_wait(rcount, m + 1);      // block row m is in the ring
		PMARK(tp_wait);
		const double * q = ring + (size_t)(m % RD) * NQ * 64 + lane;
		// band entry d of the block row's rho*theta / W / rho row (structural zeros are literals), and the three residuals
#define QE(kind, d) lu_in_entry<kind, 64>(q, d)
#define QF(kind) q[(TMX_RSLOT_RHS + (kind)) * 64]
		if (m == 0) {
			// window row r holds matrix columns 0..8; band entry d of row r sits at column r - 4 + d
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) {
				R[0][c] = (c + 4 < TMX_BW) ? QE(0, c + 4 < TMX_BW ? c + 4 : 0) : 0.0;
				R[1][c] = (c + 3 < TMX_BW) ? QE(1, c + 3 < TMX_BW ? c + 3 : 0) : 0.0;
				R[2][c] = (c + 2 < TMX_BW) ? QE(2, c + 2 < TMX_BW ? c + 2 : 0) : 0.0;
			}
			b[0] = QF(0); b[1] = QF(1); b[2] = QF(2);
		} else if (m == 1) {
#pragma unroll
			for (int c = 0; c < TMX_BW; c++) {
				R[3][c] = (c + 1 < TMX_BW) ? QE(0, c + 1 < TMX_BW ? c + 1 : 0) : 0.0;
				R[4][c] = QE(1, c);
			}
			b[3] = QF(0); b[4] = QF(1);
			lu_pivot_step<0, SPARSE, true, 64, -1, 2, NTSK>(R, b, singular, AB, RHS, 0, NUS, u, q, QF(2), umask, &nuni);      // pivot 0 takes in row 5
		} else
		{
			// matrix rows 3m-5 (W of level m-2), 3m-4 (rho), 3m-3 (rho*theta of level m-1)
			lu_pivot_step<1, SPARSE, true, 64, TMX_EXPECT_W, 0, NTSK>(R, b, singular, AB, RHS, 3 * m - 5, NUS, u, q, QF(0), umask, &nuni);
			lu_pivot_step<2, SPARSE, true, 64, TMX_EXPECT_R, 1, NTSK>(R, b, singular, AB, RHS, 3 * m - 4, NUS, u, q, QF(1), umask, &nuni);
			lu_pivot_step<0, SPARSE, true, 64, TMX_EXPECT_T, 2, NTSK>(R, b, singular, AB, RHS, 3 * m - 3, NUS, u, q, QF(2), umask, &nuni);
		}
#undef QE
#undef QF
		ring_post(rcount + 1, m + 1, lane);      // the slot may be refilled
	}
	PMARK(tp_work);
	{
		double z[TMX_BW];
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) z[c] = 0.0;
		// the last five rows: n is a multiple of three, so their kinds are W, rho, rho*theta, W, rho
		lu_pivot_step<1, SPARSE, true, 1, -1, -1, NTSK>(R, b, singular, AB, RHS, n - 5, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<2, SPARSE, true, 1, -1, -1, NTSK>(R, b, singular, AB, RHS, n - 4, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<0, SPARSE, true, 1, -1, -1, NTSK>(R, b, singular, AB, RHS, n - 3, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<1, SPARSE, true, 1, -1, -1, NTSK>(R, b, singular, AB, RHS, n - 2, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<2, SPARSE, true, 1, -1, -1, NTSK>(R, b, singular, AB, RHS, n - 1, NUS, u, z, 0.0, umask, &nuni);
	}
#ifdef TMX_PAIR_TIMING
	const long long tp_fwd = __builtin_readcyclecounter();
#endif
	if (singular) atomicOr(flag, TMX_FLAG_SINGULAR);
	// statistics for bench.py (stats = {pivot steps on the wave-uniform path, all pivot steps}, per wavefront)
	if (stats && lane == 0) { atomicAdd(&stats[0], (unsigned long long)nuni); atomicAdd(&stats[1], (unsigned long long)n); }
	if (BACK) vi_back_substitute<NTSK>(L, NS, n, u, col, NUS, AB, RHS, xin, xup, udep, umask, lane);
	else for (int j = lane; j < n; j += 64) gmask[(size_t)grp * n + j] = umask[j];      // this wavefront's own LDS writes: no barrier needed
	if (CUORD && pair == 0 && lane == 0) atomicAnd(cu_slot, ~(1 << order));      // the order slot is free again (the workgroup's other wavefronts are about to end too)
#ifdef TMX_PAIR_TIMING
	if (stats && lane == 0) {
		unsigned long long * q = stats + 2 + 8 * ((size_t)(blockIdx.x * PAIRS + pair) * 2 + role);
		q[0] = 1 + role; q[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
		q[3] = tp_work; q[4] = tp_wait; q[5] = (unsigned long long)(tp_fwd - tp_start); q[6] = (unsigned long long)((long long)__builtin_readcyclecounter() - tp_fwd);
	}
#endif
#undef PMARK
}

// Back substitution of k_vi_pair<..., BACK = false>: one wavefront = one column group, same statements as the in-kernel call.
template <int NSUB>
__global__ __launch_bounds__(64 * NSUB) void k_vi_back(KParams p, const double * xin, double * xup, int nunique, int NUS,
	const int * __restrict__ ucol, const int * __restrict__ udep, double * __restrict__ AB, int cpw, const int * __restrict__ gmask)
{
	extern __shared__ int umask_mem[];
	// NSUB wavefronts share a column group: wavefront w serves the group's lanes [w * 64 / NSUB, (w + 1) * 64 / NSUB) with as
	// many active lanes (more wavefronts in flight for the same bytes: the phase is bound by the memory system)
	const int L = p.L, grp = blockIdx.x;
	const int n = TMX_FTOT * (L + 1);
	for (int j = threadIdx.x; j < n; j += 64 * NSUB) umask_mem[j] = gmask[(size_t)grp * n + j];
	__syncthreads();
	if ((int)(threadIdx.x & 63) >= 64 / NSUB) return;
	const int lane = (threadIdx.x >> 6) * (64 / NSUB) + (threadIdx.x & 63);
	const int ulast = min(nunique, (grp + 1) * cpw) - 1;
	const int u = (grp * cpw < nunique) ? min(grp * cpw + min(lane, cpw - 1), ulast) : nunique - 1;
	const int col = ucol ? ucol[u] : u;      // (null: node-unique layout, a unique column is its own state column)
	const double * RHS = AB + (size_t)n * TMX_UW * NUS;      // zero page behind the streams
	vi_back_substitute(L, (size_t)p.NS, n, u, col, NUS, vi_stream_base_g(AB, n, grp, lane), RHS, xin, xup, udep, umask_mem, lane);
}

// ---------------------------------------------------------------------------------------------
// V implicit for FEW columns per GPU (the per-rank share of a strong-scaling run: ne30 over 8 ranks leaves 6 075 unique
// columns for 1 024 SIMDs): one column per 16-lane group, four columns per wavefront.  The one-lane-per-column kernels
// above are bound there by the dependent chain of one column (93 pivot steps of ~270 instructions each, plus the
// assembly of 31 block rows one after the other) while most of the chip idles; here the chain is cut across lanes:
//   * assembly: lane c of a group evaluates block row t0 + c (compute_block without the carry-over: same arithmetic on the
//     same operands), 16 block rows at a time into an LDS ring -- two rounds instead of 31 sequential evaluations;
//   * elimination: lane c <= 8 holds COLUMN c of the 5 x 9 pivot window, lane 15 the right-hand side.  Lane 0 finds the
//     pivot row (first maximum) and the multipliers; pivot index and multipliers reach the group by DPP row broadcast,
//     the row interchange is four selects per lane, the rank-1 update four fused multiply-adds per lane, and the
//     window slides one column to the left by a DPP row shift (lanes 12-15 are masked out of the shift, so the
//     right-hand side stays put and the idle lanes 9-14 keep feeding zeros into column 8);
//   * the U rows go to the same HBM scratch as in the other kernels (one contiguous stream per wavefront,
//     [row][10][4 columns]); the back substitution runs on lane 0 of each group in dtbsv order.
// Every multiplication, fused update and division is the one of lu_pivot_step / vi_back_row: results are bit-identical
// to k_vi_pair (test_fused_column_kernels_are_bit_identical_to_split_kernels).  Per wavefront-instruction this kernel
// serves 4 columns instead of 64, so it only pays while the chip has SIMDs to spare: tmxk_vi_fused picks it by size.
template <int SRC> __device__ __forceinline__ double grp_bcast(double v) {
	const long long b = __double_as_longlong(v);
	// every lane is written (row and bank masks 0xf), so the "old" operand is never used: passing the value itself spares the
	// zero-initialisation of the destination the compiler emits for a constant
	const int l0 = (int)(b & 0xffffffffll), h0 = (int)(b >> 32);
	const int lo = __builtin_amdgcn_update_dpp(l0, l0, 0x150 + SRC, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(h0, h0, 0x150 + SRC, 0xf, 0xf, false);
	return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// lanes 0..11 of every 16-lane row take the value of the next lane, lanes 12..15 keep their own
__device__ __forceinline__ double grp_shift_left(double v) {
	const long long b = __double_as_longlong(v);
	const int l0 = (int)(b & 0xffffffffll), h0 = (int)(b >> 32);
	const int lo = __builtin_amdgcn_update_dpp(l0, l0, 0x101, 0xf, 0x7, false);
	const int hi = __builtin_amdgcn_update_dpp(h0, h0, 0x101, 0xf, 0x7, false);
	return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <bool CLOSED>
__global__ __launch_bounds__(64) void k_vi_group(KParams p, const double * xin, double * xup, double dt,
	int nunique, const int * __restrict__ ucol, const int * __restrict__ udep, double * __restrict__ AB, int * __restrict__ flag)
{
	constexpr int MM = CLOSED ? 1 : 2;
	constexpr int NQ = 3 * TMX_BW + 3;              // doubles per block row: three band rows and their right-hand sides
	__shared__ double blk[4][16][NQ];               // ring of 16 block rows per column
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
	const int n = TMX_FTOT * (L + 1);
	const int uraw = blockIdx.x * 4 + g;
	const bool valid = uraw < nunique;
	const int u = valid ? uraw : nunique - 1;       // a trailing, empty group redoes the last column (no stores)
	const int col = ucol ? ucol[u] : u;      // (null: node-unique layout, a unique column is its own state column)
	double * stream = AB + (size_t)blockIdx.x * n * 64 + lane;   // [row][4 columns][16 lanes]: entry c of U row j at (j * 64 + g * 16 + c); lane 15 = rhs

	ColConst cc;
	cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
	cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
	cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
	const MetCol mcol = met_col(p, col);

	// block rows [t0, t0 + 16) into the ring, one per lane
	auto assemble = [&](int t0) {
		const int t = t0 + c;
		if (t <= L) {
			const NodeLev nA = load_node<MM>(p, mcol, xin, t - 1, col, p.eta), nB = load_node<MM>(p, mcol, xin, t, col, p.eta),
				nC = load_node<MM>(p, mcol, xin, t + 1, col, p.eta);
			const EdgeLev eA = load_edge<MM>(p, mcol, xin, t - 1, col, p.eta), eB = load_edge<MM>(p, mcol, xin, t, col, p.eta),
				eC = load_edge<MM>(p, mcol, xin, t + 1, col, p.eta);
			double rP[TMX_BW], rW[TMX_BW], rR[TMX_BW], fP, fW, fR;
			BlkCarry cy = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
			compute_block<false>(p, p.ops, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy);
			double * q = &blk[g][c][0];
#pragma unroll
			for (int d = 0; d < TMX_BW; d++) { q[d] = rP[d]; q[TMX_BW + d] = rW[d]; q[2 * TMX_BW + d] = rR[d]; }
			q[3 * TMX_BW] = fP; q[3 * TMX_BW + 1] = fW; q[3 * TMX_BW + 2] = fR;
		}
	};
	// band entry d of matrix row r / its right-hand side, from the ring
	auto band = [&](int r, int d) -> double { const int t = r / 3; return blk[g][t & 15][(r - 3 * t) * TMX_BW + d]; };
	auto rhs = [&](int r) -> double { const int t = r / 3; return blk[g][t & 15][3 * TMX_BW + (r - 3 * t)]; };

#ifdef TMX_VI_TIMING
	const long long tg0 = __builtin_readcyclecounter();
#endif
	assemble(0);
	__syncthreads();      // one wavefront per workgroup: orders the ring writes before the other lanes' reads
#ifdef TMX_VI_TIMING
	const long long tg1 = __builtin_readcyclecounter();
#endif
	// window column c (lanes 0..8: matrix column j + c; lane 15: right-hand side; lanes 9..14: zeros), rows j .. j + 4
	double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0;
	if (c <= 8) {
		// row r, window column c <-> band entry d = c - r + 4 (inside the band for c <= r + 4)
		w0 = (c <= 4) ? band(0, c + 4) : 0.0;
		w1 = (c <= 5) ? band(1, c + 3) : 0.0;
		w2 = (c <= 6) ? band(2, c + 2) : 0.0;
		w3 = (c <= 7) ? band(3, c + 1) : 0.0;
		w4 = band(4, c);
	} else if (c == 15) { w0 = rhs(0); w1 = rhs(1); w2 = rhs(2); w3 = rhs(3); w4 = rhs(4); }
	bool singular = false;
	const bool carries = (c <= 8) || (c == 15);      // lanes that hold a window column or the right-hand side
	const int in_mul = (c <= 8) ? TMX_BW : 1, in_off = (c <= 8) ? c : 3 * TMX_BW;      // slot of this lane's entry of an incoming row
	// The ring holds 16 block rows = 48 matrix rows; pivot j takes in row j + 5.  The loop runs in segments between two
	// refills of the ring, so that the segment loop itself contains no vector-memory LOAD: with the refill inside it the
	// compiler put an s_waitcnt vmcnt(0) at the loop head, which also waits for the previous pivot's U-row store.
	for (int j = 0; j < n; ) {
	if (j > 0 && j + 5 < n) { assemble((j + 5) / 3); __syncthreads(); }      // row j + 5 = 48 m comes up: rows below it are in the window
	const int jend = min(n, ((j + 5) / 48 + 1) * 48 - 5);
	for (; j < jend; j++) {
		const int rin = j + 5;
		// incoming row: one LDS read per lane, no branch (lanes 9..14 read the rhs slot and drop it); the value is only
		// needed at the end of the step, so the read's latency is off the pivot chain
		const int rc = (rin < n) ? rin : n - 1;
		const int tin = rc / 3;
		const double vin = blk[g][tin & 15][(rc - 3 * tin) * in_mul + in_off];
		// pivot search on column 0 (lane 0 of the group): first maximum, as dgbtf2's idamax
		// (as a tree: rows 0/1 and 2/3 are compared side by side; strict comparisons keep the lower row on ties, so the result
		// is the sequential first maximum)
		int pv;
		{
			const double a0 = fabs(w0), a1 = fabs(w1), a2 = fabs(w2), a3 = fabs(w3), a4 = fabs(w4);
			const bool g1 = a1 > a0, g3 = a3 > a2;
			const double m01 = g1 ? a1 : a0, m23 = g3 ? a3 : a2;
			const int i01 = g1 ? 1 : 0, i23 = g3 ? 3 : 2;
			const bool gh = m23 > m01;
			const double m03 = gh ? m23 : m01;
			const int i03 = gh ? i23 : i01;
			pv = (a4 > m03) ? 4 : i03;
		}
		pv = __builtin_amdgcn_update_dpp(pv, pv, 0x150, 0xf, 0xf, false);
		// interchange: the pivot row's entry of this column, old row 0 into the pivot row's place
		// (a chain of selects, not a nested conditional: the compiler turned the latter into divergent branches, five
		// exec-mask round trips per pivot step)
		double pr = w0;
		pr = (pv == 1) ? w1 : pr; pr = (pv == 2) ? w2 : pr; pr = (pv == 3) ? w3 : pr; pr = (pv == 4) ? w4 : pr;
		const double t0 = w0;
		w1 = (pv == 1) ? t0 : w1; w2 = (pv == 2) ? t0 : w2; w3 = (pv == 3) ? t0 : w3; w4 = (pv == 4) ? t0 : w4;
		// U row j: entries 0..8 from lanes 0..8, transformed right-hand side from lane 15 (lanes 9..14 write their zeros:
		// one unconditional, fully coalesced 512-byte store per row)
		stream[(size_t)j * 64] = pr;
		// multipliers from lane 0
		if (c == 0 && pr == 0.0) singular = true;
		const double rcp = 1.0 / pr;
		const double l1 = grp_bcast<0>(w1 * rcp), l2 = grp_bcast<0>(w2 * rcp), l3 = grp_bcast<0>(w3 * rcp), l4 = grp_bcast<0>(w4 * rcp);
		// rank-1 update (and forward substitution on lane 15), then slide: rows up, columns one lane to the left
		w0 = grp_shift_left(LU_UPD(w1, l1, pr));
		w1 = grp_shift_left(LU_UPD(w2, l2, pr));
		w2 = grp_shift_left(LU_UPD(w3, l3, pr));
		w3 = grp_shift_left(LU_UPD(w4, l4, pr));
		w4 = (carries && rin < n) ? vin : 0.0;
	}
	}
	if (singular) atomicOr(flag, TMX_FLAG_SINGULAR);
#ifdef TMX_VI_TIMING
	const long long tg2 = __builtin_readcyclecounter();
#endif

	// Back substitution (dtbsv order).  Every lane loads its own entry of a U row (one coalesced load per row, four rows
	// in flight, no other vector-memory operation in the loop so that the waits are counted exactly); the row is gathered
	// to every lane of the group by DPP broadcasts and all lanes carry the same sequential chain (vi_back_row).  The
	// solution goes to LDS; afterwards the 16 lanes of a group form x0 - delta for the rows ROW = lane, lane + 16, ... and
	// store them (column and in-patch duplicates) in parallel.
	{
		double * xsol = &blk[g][0][0];          // the ring is free now: n <= 16 * NQ doubles per column
		double xw[TMX_BW - 1];
#pragma unroll
		for (int q = 0; q < TMX_BW - 1; q++) xw[q] = 0.0;
		auto fetch = [&](int j) -> double { return stream[(size_t)(j < 0 ? 0 : j) * 64]; };
		// TMX_GRP_PF rows in flight per lane (4, 16 and 32 measure the same: the elimination loop, not this one, is the chain
		// that bounds the kernel on the small grids it serves)
#ifndef TMX_GRP_PF
#define TMX_GRP_PF 4
#endif
		double f[TMX_GRP_PF];
#pragma unroll
		for (int q = 0; q < TMX_GRP_PF; q++) f[q] = fetch(n - 1 - q);
		auto row = [&](int j, double mine) {
			URow s_;
			s_.u[0] = grp_bcast<0>(mine); s_.u[1] = grp_bcast<1>(mine); s_.u[2] = grp_bcast<2>(mine); s_.u[3] = grp_bcast<3>(mine);
			s_.u[4] = grp_bcast<4>(mine); s_.u[5] = grp_bcast<5>(mine); s_.u[6] = grp_bcast<6>(mine); s_.u[7] = grp_bcast<7>(mine);
			s_.u[8] = grp_bcast<8>(mine); s_.y = grp_bcast<15>(mine);
			const double xr = vi_back_row(s_, xw);
			if (c == 0) xsol[j] = xr;
		};
		int j = n - 1;
		for (; j >= TMX_GRP_PF - 1; j -= TMX_GRP_PF) {
#pragma unroll
			for (int q = 0; q < TMX_GRP_PF; q++) { row(j - q, f[q]); f[q] = fetch(j - q - TMX_GRP_PF); }
		}
#pragma unroll
		for (int q = 0; q < TMX_GRP_PF; q++) if (j - q >= 0) row(j - q, f[q]);
		__syncthreads();
		if (valid) {
			int dep[3];
#pragma unroll
			for (int t = 0; t < 3; t++) dep[t] = udep ? udep[u * 3 + t] : -1;
			for (int r = c; r < n; r += 16) {
				const int k = r / TMX_FTOT, comp = r - k * TMX_FTOT;
				if (comp != 1 && k >= L) continue;                 // level-L slots of rho*theta / rho have no slab
				const size_t so = (size_t)((comp == 0) ? TMX_SLAB_T(L, k) : ((comp == 1) ? TMX_SLAB_W(L, k) : TMX_SLAB_R(L, k))) * NS;
				const double val = xin[so + col] - xsol[r];
				xup[so + col] = val;
#pragma unroll
				for (int t = 0; t < 3; t++) if (dep[t] >= 0) xup[so + dep[t]] = val;
			}
		}
	}
#ifdef TMX_VI_TIMING
	if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 700))
		printf("vi group timing block %d: first assembly round %lld cycles, elimination (+ later rounds) %lld, back substitution %lld\n",
			(int)blockIdx.x, tg1 - tg0, tg2 - tg1, (long long)__builtin_readcyclecounter() - tg2);
#endif
}

void tmxk_vi_fused(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) {
	if (e->nunique == 0) return;
	// node-unique layout (e->vi_unique, set by the caller together with tmxu_params_columns): unique column u IS state column u, no in-patch copies to fill
	const int * ucolp = e->vi_unique ? nullptr : (const int *)e->d_ucol, * udepp = e->vi_unique ? nullptr : (const int *)e->d_udep;
	const size_t lds = (size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW * sizeof(double) + (size_t)(2 * p.L + 1) * sizeof(double) + (size_t)TMX_FTOT * (p.L + 1) * sizeof(int);
	const bool sparse = e->opt_vi_sparse != 0;      // A/B switch (option "vi_sparse"), default on
	// The pair kernel shortens the per-column dependent chain to max(assembly, elimination) and is built for two
	// wavefronts per SIMD (256 registers): the 1024 SIMDs hold 1024 pairs at once.  Measured per launch: ne30 on one
	// GPU (760 column groups) 197-211 us against 229 us for the one-wavefront kernel; ne60 (3038 groups, three rounds)
	// 840 us against 899 us; 190 groups 95 us with one pair per workgroup (every workgroup alone on a CU), 104 us with
	// two; 380 groups 125 us with two pairs per workgroup, 131 us with one.  So it is the production kernel at every
	// size; the one-wavefront kernel stays as TMX_VI_PAIR=0 (bit-identical, tested).
	// few columns for the chip (strong-scaling share): one column per 16-lane group
	// (the options that select it may be set after tmx_finalize sized the scratch for the pair kernel's 10 doubles per row and column: the
	// lane-group kernel streams 16, and is only launched where its stream fits -- otherwise the pair kernel serves the share)
	const bool group_fits = (size_t)((e->nunique + 3) / 4) * 64 <= (size_t)TMX_UW * e->vi_stream_cols;
	if (group_fits && (e->vi_group == 1 || (e->vi_group < 0 && e->nunique <= e->vi_group_max))) {
		const int nwg = (e->nunique + 3) / 4;
		e->vi_kernel_launched = 2;
		if (p.closed) hipLaunchKernelGGL((k_vi_group<true>), dim3(nwg), dim3(64), 0, e->stream, p, xin, xup, dt, e->nunique, ucolp, udepp, e->d_ab, e->d_flag);
		else hipLaunchKernelGGL((k_vi_group<false>), dim3(nwg), dim3(64), 0, e->stream, p, xin, xup, dt, e->nunique, ucolp, udepp, e->d_ab, e->d_flag);
		return;
	}
	const bool use_pair = !TMX_EXP || (e->vi_pair == 1) || (e->vi_pair < 0 && e->vi_carry);
	e->vi_kernel_launched = use_pair ? 1 : 0;      // (tmx_info(TMX_INFO_COLUMN_KERNEL): which kernel really ran, whatever the options asked for)
	const int cpw = use_pair ? e->vi_cpw : 64;
	const int ngrp = (e->nunique + cpw - 1) / cpw;
	if (use_pair) {
		const size_t lds_common = (size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW * sizeof(double) + (size_t)(2 * p.L + 1) * sizeof(double) + TMX_RMTAB_DOUBLES * sizeof(double);
		const int pairs = (e->vi_pair_wg > 0) ? e->vi_pair_wg : ((ngrp <= 256) ? 1 : 2);
		// two assembly wavefronts per column group where every wavefront still has a SIMD to itself (option "vi_producers": 0 auto, 1, 2)
		const bool two_prod = p.closed && pairs == 1 && cpw == 64 && (e->vi_producers == 2 || (e->vi_producers == 0 && ngrp * 3 <= 1024));
		auto lds_pair_of = [&](int depth) { return (size_t)depth * TMX_RING_NQ * 64 * sizeof(double) + (size_t)TMX_FTOT * (p.L + 1) * sizeof(int) + 4 * sizeof(int); };      // ring, fill-in masks, hand-over counters
		// ring depth: three block rows, or two where only that lets a second workgroup onto every CU (160 KB of LDS; option "vi_ring_depth": 0 auto, 2, 3)
		const size_t cu_lds = 160 * 1024;
		int rdt = TMX_RING_DEPTH;
		if (e->vi_ring_depth == 2 || e->vi_ring_depth == 3) rdt = e->vi_ring_depth;
		else if (!two_prod && 2 * (lds_common + pairs * lds_pair_of(3)) > cu_lds && 2 * (lds_common + pairs * lds_pair_of(2)) <= cu_lds) rdt = 2;
		if (TMX_EXP && e->vi_split_back != 0) rdt = TMX_RING_DEPTH;      // (the separate-back-substitution cross-check is instantiated with the default depth)
		const size_t lds_pair = lds_pair_of(two_prod ? TMX_RING_DEPTH + 1 : rdt);
		// back substitution as a launch of its own (TMX_VI_SPLIT_BACK=0: inside the pair kernel); the masks go through d_rhs,
		// which only the split cross-check kernels use otherwise
		const bool split_back = TMX_EXP && e->vi_split_back != 0;      // (experiments flavour only)
		const int stagger = e->vi_stagger;
		const int ngrp_launched = ((ngrp + pairs - 1) / pairs) * pairs;
		const bool split = split_back && (size_t)ngrp_launched * TMX_FTOT * (p.L + 1) * sizeof(int) <= (size_t)TMX_FTOT * (p.L + 1) * e->NUS * sizeof(double);
#define LAUNCH_PAIR(CL_, NP_) do { if (TMX_EXP && split) hipLaunchKernelGGL((k_vi_pair<true, CL_, NP_, TMX_EXP == 0>), dim3((ngrp + NP_ - 1) / NP_), dim3(64, 2 * NP_), lds_common + NP_ * lds_pair, \
			e->stream, p, xin, xup, dt, e->nunique, e->vi_stream_cols, ucolp, udepp, e->d_ab, e->d_rhs, e->d_flag, cpw, e->d_pivot_stats, (int *)e->d_rhs, 0); \
		else if (rdt == 2) hipLaunchKernelGGL((k_vi_pair<true, CL_, NP_, true, 1, 2>), dim3((ngrp + NP_ - 1) / NP_), dim3(64, 2 * NP_), lds_common + NP_ * lds_pair, \
			e->stream, p, xin, xup, dt, e->nunique, e->vi_stream_cols, ucolp, udepp, e->d_ab, e->d_rhs, e->d_flag, cpw, e->d_pivot_stats, (int *)nullptr, stagger); \
		else hipLaunchKernelGGL((k_vi_pair<true, CL_, NP_, true, 1, 3>), dim3((ngrp + NP_ - 1) / NP_), dim3(64, 2 * NP_), lds_common + NP_ * lds_pair, \
			e->stream, p, xin, xup, dt, e->nunique, e->vi_stream_cols, ucolp, udepp, e->d_ab, e->d_rhs, e->d_flag, cpw, e->d_pivot_stats, (int *)nullptr, stagger); } while (0)
		if (two_prod && !(TMX_EXP && split))
			hipLaunchKernelGGL((k_vi_pair<true, true, 1, true, 2>), dim3(ngrp), dim3(64, 3), lds_common + lds_pair, e->stream, p, xin, xup, dt, e->nunique, e->vi_stream_cols,
				ucolp, udepp, e->d_ab, e->d_rhs, e->d_flag, cpw, e->d_pivot_stats, (int *)nullptr, stagger);
		else
		if (pairs == 1) { if (p.closed) LAUNCH_PAIR(true, 1); else LAUNCH_PAIR(false, 1); }
		else { if (p.closed) LAUNCH_PAIR(true, 2); else LAUNCH_PAIR(false, 2); }
#undef LAUNCH_PAIR
#if TMX_EXP
		if (split) {
			const int nsub = e->vi_back_sub;
#define LAUNCH_BACK(NS_) hipLaunchKernelGGL(k_vi_back<NS_>, dim3(ngrp), dim3(64 * NS_), (size_t)TMX_FTOT * (p.L + 1) * sizeof(int), e->stream, p, xin, xup, e->nunique, e->vi_stream_cols, \
				ucolp, udepp, e->d_ab, cpw, (const int *)e->d_rhs)
			if (nsub == 2) LAUNCH_BACK(2); else if (nsub == 4) LAUNCH_BACK(4); else LAUNCH_BACK(1);
#undef LAUNCH_BACK
		}
#endif
		return;
	}
#if TMX_EXP      // the one-wavefront kernel (options "vi_pair" = 0, "vi_carry" = 0, "vi_sparse" = 0)
#define LAUNCH_VI(C_, S_) do { if (p.closed) \
		hipLaunchKernelGGL((k_vi_fused<C_, S_, true, false>), dim3((e->nunique + 63) / 64), dim3(64), lds, e->stream, p, xin, xup, dt, \
			e->nunique, e->vi_stream_cols, ucolp, udepp, e->d_ab, e->d_rhs, e->d_flag); \
	else \
		hipLaunchKernelGGL((k_vi_fused<C_, S_, false, false>), dim3((e->nunique + 63) / 64), dim3(64), lds, e->stream, p, xin, xup, dt, \
			e->nunique, e->vi_stream_cols, ucolp, udepp, e->d_ab, e->d_rhs, e->d_flag); } while (0)
	if (!e->vi_carry) LAUNCH_VI(false, true);
	else if (sparse) LAUNCH_VI(true, true);
	else LAUNCH_VI(true, false);
#undef LAUNCH_VI
#else
	(void)lds; (void)sparse;
#endif
}


// ---------------------------------------------------------------------------------------------
// VerticalDynamicsFEM::StepImplicitTermsExplicitly (src/atm/VerticalDynamicsFEM.cpp:439-612):
// update -= dt * F(initial) for rho*theta, W, rho on EVERY column (F = BuildF of the initial column).

// UD (fully explicit mode with uniform diffusion): F additionally holds the vertical diffusion of rho*theta and W
// relative to the reference column xref (cs = K_s / ztop^2, cw = K_v / ztop^2).
// UVX: the thread of (column, level k < L) also applies V.StepExplicit's update of U,V (v_explicit_point: penalty and, with UD,
// the vertical diffusion), which reads the same U,V,W columns and metric rows -- one launch less per stage and the operands
// found in the cache
template <bool UD, bool UVX>
// KT_VC levels per workgroup.  A thread reads the levels k-2 .. k+2 of its column, so the level tiles of a column overlap by
// four levels and every value is fetched 2.7 times over at the L2's memory side (PMC, ne30 L40: 514 MB of reads per launch for
// 190 MB of operands: the level-neighbour tiles are 1 350 workgroups apart in dispatch order, on another XCD's L2).  Taller
// tiles make the neighbours wavefronts of one workgroup, and lose: config 4's step takes 5.12 / 5.21 / 5.34 ms with 4 / 8 / 16
// levels per workgroup (1 024-thread workgroups leave one per CU).  4 stays.
#ifndef KT_VC
#define KT_VC 4
#endif
__global__ __launch_bounds__(64 * KT_VC) void k_vi_terms_explicit(KParams p, const double * __restrict__ xin,
	double * __restrict__ xup, double dt, const double * __restrict__ xref, double cs, double cw, int ntile, int xmode)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (L + 1 + KT_VC - 1) / KT_VC, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + threadIdx.x;
	const int k = by * KT_VC + WAVE_UNIFORM(threadIdx.y);
	if (col >= p.ncol || k > L) return;
	ColConst cc;
	cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
	cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
	cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
	const MetCol mcol = met_col(p, col);
	// the values to update, loaded with the operands (at the point of use their latency would follow the whole evaluation)
	const double upW = xup[TMX_SLAB_W(L, k) * NS + col];
	const double upT = (k < L) ? xup[TMX_SLAB_T(L, k) * NS + col] : 0.0, upR = (k < L) ? xup[TMX_SLAB_R(L, k) * NS + col] : 0.0;
	const NodeLev nA = load_node(p, mcol, xin, k - 1, col), nB = load_node(p, mcol, xin, k, col), nC = load_node(p, mcol, xin, k + 1, col);
	const EdgeLev eA = load_edge(p, mcol, xin, k - 1, col), eB = load_edge(p, mcol, xin, k, col), eC = load_edge(p, mcol, xin, k + 1, col);
	double rP[TMX_BW], rW[TMX_BW], rR[TMX_BW], fP, fW, fR;
	BlkCarry cy;
	double udP = 0.0, udW = 0.0;
	if (UD) {
		// PrepareColumn :2104-2160: DiffDiff of the column minus DiffDiff of the reference column; W not on the boundaries
		if (k < L) {
			double dd = 0.0, ddr = 0.0;
#pragma unroll
			for (int off = -2; off <= 2; off++) {
				const int l = k + off;
				if (l < 0 || l >= L) continue;
				const double c = OPC(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, off);
				dd += c * xin[(size_t)TMX_SLAB_T(L, l) * NS + col];
				ddr += c * xref[(size_t)TMX_SLAB_T(L, l) * NS + col];
			}
			udP = cs * (dd - ddr);
		}
		if (k > 0 && k < L) {
			double dd = 0.0, ddr = 0.0;
#pragma unroll
			for (int off = -2; off <= 2; off++) {
				const int l = k + off;
				if (l < 0 || l > L) continue;
				const double c = OPC(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, off);
				dd += c * xin[(size_t)TMX_SLAB_W(L, l) * NS + col];
				ddr += c * xref[(size_t)TMX_SLAB_W(L, l) * NS + col];
			}
			udW = cw * (dd - ddr);
		}
	}
	compute_block<false, UD>(p, p.ops, cc, k, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, udP, udW);
	if (k < L) {
		xup[TMX_SLAB_T(L, k) * NS + col] = upT - dt * fP;
		xup[TMX_SLAB_R(L, k) * NS + col] = upR - dt * fR;
	}
	xup[TMX_SLAB_W(L, k) * NS + col] = upW - dt * fW;
	if (UVX && k < L) v_explicit_point<UD>(p, xin, xup, dt, xref, cw, col, k);
}

// The same update by a thread that walks a column (or one of `nseg` segments of it) block row by block row, as the column solve's
// assembly does: the three node levels and three interfaces a block row touches in a sliding register window, the quantities a
// block row shares with the next carried over (compute_block<CARRY>: one evaluation of the Exner function per level instead of
// two, every state value loaded once instead of three to five times), the next levels loaded two iterations ahead, operator
// coefficients and the 1 - eta table in LDS.  The uniform-diffusion stencils take the state's five levels from the window (the
// value that has just left it, the three in it, the one about to enter) and keep sliding windows of the reference's.  A segment that
// does not start at the bottom evaluates the block row below it first, only to fill the carry.  Same statements on the same
// operands as k_vi_terms_explicit: bit-identical (tested).
#define OPCL(op, k, off) opsl[(((op) * (p.L + 1)) + (k)) * TMX_OPW + ((off) + 2)]
template <bool UD, bool CLOSED>
__global__ __launch_bounds__(128) void k_vi_terms_explicit_slide(KParams p, const double * __restrict__ xin,
	double * __restrict__ xup, double dt, const double * __restrict__ xref, double cs, double cw, int ntile, int xmode, int nseg)
{
	extern __shared__ double opsl_mem[];
	double * opsl = opsl_mem;
	const int L = p.L;
	constexpr int MM = CLOSED ? 1 : 2;
	double * etal = opsl_mem + TMX_OP_COUNT * (L + 1) * TMX_OPW;
	// the exp / log tables of the Exner function in LDS too: read from global memory each lookup is a vector-memory load behind an
	// s_waitcnt vmcnt(0), i.e. two waits per block row for every level prefetch in flight (the column solve's assembly: the same)
	double * rmtab = etal + 2 * L + 1;
	{
		const int tid = threadIdx.y * 64 + threadIdx.x;
		for (int t = tid; t < TMX_OP_COUNT * (L + 1) * TMX_OPW; t += 128) opsl[t] = p.ops[t];
		if (CLOSED) for (int t = tid; t < 2 * L + 1; t += 128) etal[t] = p.eta[t];
		tmx_rm_tables_to_lds(rmtab, tid, 128);
		__syncthreads();
	}
	const size_t NS = (size_t)p.NS;
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (nseg + 1) / 2, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + threadIdx.x;
	if (col >= p.ncol) return;
	const int sg = WAVE_UNIFORM(by * 2 + (int)threadIdx.y);
	const int seg = (L + 1 + nseg - 1) / nseg, k0 = sg * seg, k1 = min(L + 1, k0 + seg);      // block rows k0 .. k1 - 1 of 0 .. L
	if (sg >= nseg || k0 >= k1) return;
	ColConst cc;
	cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
	cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
	cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
	const MetCol mcol = met_col(p, col);
	const int kw = (k0 > 0) ? k0 - 1 : 0;      // first block row evaluated (k0 - 1: for the carry only)
	// The window: block row k reads U, V, rho*theta, rho of the levels k - 1, k, k + 1 and W of the interfaces k - 1, k, k + 1; with the
	// carry, the metric terms of level k and of the interfaces k, k + 1 only.  Levels k + 2 and k + 3 are in flight as raw values
	// (their metric terms are arithmetic on per-column factors and the LDS table: evaluated when a level becomes the window's top).
	struct Raw { double un, vn, pn, rn, we; };
	auto raw = [&](int l) {      // (out-of-range levels carry the values of the nearest one, as load_node / load_edge)
		Raw r;
		const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l), le = l < 0 ? 0 : (l > L ? L : l);
		r.un = xin[TMX_SLAB_U(L, lc) * NS + col]; r.vn = xin[TMX_SLAB_V(L, lc) * NS + col];
		r.pn = xin[TMX_SLAB_T(L, lc) * NS + col]; r.rn = xin[TMX_SLAB_R(L, lc) * NS + col];
		r.we = xin[TMX_SLAB_W(L, le) * NS + col];
		return r;
	};
	auto node_of = [&](const Raw & r, int l) {
		NodeLev n; n.un = r.un; n.vn = r.vn; n.pn = r.pn; n.rn = r.rn;
		const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l);
		metric_node3<MM>(p, mcol, lc, col, n.ca2, n.cb2, n.cx2, etal);
		return n;
	};
	auto edge_of = [&](const Raw & r, int l) {
		EdgeLev e; e.we = r.we;
		const int le = l < 0 ? 0 : (l > L ? L : l);
		metric_edge<MM>(p, mcol, le, col, e.ce0, e.ce1, e.ce2, etal);
		return e;
	};
	Raw rA = raw(kw - 1), rN = raw(kw + 2);
	NodeLev nB = load_node<MM>(p, mcol, xin, kw, col, etal), nC = load_node<MM>(p, mcol, xin, kw + 1, col, etal);
	EdgeLev eB = load_edge<MM>(p, mcol, xin, kw, col, etal), eC = load_edge<MM>(p, mcol, xin, kw + 1, col, etal);
	double pm2 = 0.0, wm2 = 0.0;      // rho*theta of level k - 2, W of interface k - 2 (what has left the window)
	double tr[5] = { 0, 0, 0, 0, 0 }, wr[5] = { 0, 0, 0, 0, 0 };      // reference rho*theta on the levels, W on the interfaces k - 2 .. k + 2
	auto ref_T = [&](int l) { return (UD && l >= 0 && l < L) ? xref[(size_t)TMX_SLAB_T(L, l) * NS + col] : 0.0; };
	auto ref_W = [&](int l) { return (UD && l >= 0 && l <= L) ? xref[(size_t)TMX_SLAB_W(L, l) * NS + col] : 0.0; };
	if (UD) {
		if (kw - 2 >= 0) { pm2 = xin[(size_t)TMX_SLAB_T(L, kw - 2) * NS + col]; wm2 = xin[(size_t)TMX_SLAB_W(L, kw - 2) * NS + col]; }
#pragma unroll
		for (int t = 0; t < 5; t++) { tr[t] = ref_T(kw - 2 + t); wr[t] = ref_W(kw - 2 + t); }
	}
	double upT = 0.0, upR = 0.0, upW = 0.0;
	if (kw == k0) {
		upW = xup[TMX_SLAB_W(L, kw) * NS + col];
		if (kw < L) { upT = xup[TMX_SLAB_T(L, kw) * NS + col]; upR = xup[TMX_SLAB_R(L, kw) * NS + col]; }
	}
	BlkCarry cy = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll 1
	for (int k = kw; k < k1; k++) {
		// two levels ahead: the level that enters the window after the next block row; the reference's entering values and the
		// values the next block row updates
		const Raw rNN = raw(k + 3);
		const double trn = ref_T(k + 3), wrn = ref_W(k + 3);
		double upTn = 0.0, upRn = 0.0, upWn = 0.0;
		if (k + 1 < k1) {
			upWn = xup[TMX_SLAB_W(L, k + 1) * NS + col];
			if (k + 1 < L) { upTn = xup[TMX_SLAB_T(L, k + 1) * NS + col]; upRn = xup[TMX_SLAB_R(L, k + 1) * NS + col]; }
		}
		double udP = 0.0, udW = 0.0;
		if (UD) {
			const double tw[5] = { pm2, rA.pn, nB.pn, nC.pn, rN.pn }, ww[5] = { wm2, rA.we, eB.we, eC.we, rN.we };
			if (k < L) {
				double dd = 0.0, ddr = 0.0;
#pragma unroll
				for (int off = -2; off <= 2; off++) {
					const int l = k + off;
					if (l < 0 || l >= L) continue;
					const double c = OPCL(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, off);
					dd += c * tw[off + 2];
					ddr += c * tr[off + 2];
				}
				udP = cs * (dd - ddr);
			}
			if (k > 0 && k < L) {
				double dd = 0.0, ddr = 0.0;
#pragma unroll
				for (int off = -2; off <= 2; off++) {
					const int l = k + off;
					if (l < 0 || l > L) continue;
					const double c = OPCL(TMX_OP_DIFFDIFF_REDGE_TO_REDGE, k, off);
					dd += c * ww[off + 2];
					ddr += c * wr[off + 2];
				}
				udW = cw * (dd - ddr);
			}
		}
		double rP[TMX_BW], rW[TMX_BW], rR[TMX_BW], fP, fW, fR;
		{
			NodeLev nA; nA.un = rA.un; nA.vn = rA.vn; nA.pn = rA.pn; nA.rn = rA.rn; nA.ca2 = 0.0; nA.cb2 = 0.0; nA.cx2 = 0.0;      // (with the carry a block row reads no metric term of level k - 1)
			EdgeLev eA; eA.we = rA.we; eA.ce0 = 0.0; eA.ce1 = 0.0; eA.ce2 = 0.0;
			compute_block<true, UD>(p, opsl, cc, k, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, udP, udW, rmtab);
		}
		const double resT = upT - dt * fP, resR = upR - dt * fR, resW = upW - dt * fW;
		pm2 = rA.pn; wm2 = rA.we;
		rA.un = nB.un; rA.vn = nB.vn; rA.pn = nB.pn; rA.rn = nB.rn; rA.we = eB.we;
		nB = nC; eB = eC;
		nC = node_of(rN, k + 2); eC = edge_of(rN, k + 2);
		rN = rNN;
		tr[0] = tr[1]; tr[1] = tr[2]; tr[2] = tr[3]; tr[3] = tr[4]; tr[4] = trn;
		wr[0] = wr[1]; wr[1] = wr[2]; wr[2] = wr[3]; wr[3] = wr[4]; wr[4] = wrn;
		upT = upTn; upR = upRn; upW = upWn;
		// The block row's stores LAST, behind the window's move (which evaluates the metric terms of the level that enters, i.e. reads values
		// loaded an iteration ago): the stores sit in a branch, the compiler cannot count them, and a loaded value first used behind them waits
		// for everything in flight -- the stores' own completion and the prefetch of two levels ahead -- once per block row.  The values the
		// next block row updates are made to land before the stores too (they were loaded with this block row's operands).
		asm volatile("" : "+v"(upT), "+v"(upR), "+v"(upW));
		if (k >= k0) {
			if (k < L) {
				xup[TMX_SLAB_T(L, k) * NS + col] = resT;
				xup[TMX_SLAB_R(L, k) * NS + col] = resR;
			}
			xup[TMX_SLAB_W(L, k) * NS + col] = resW;
		}
	}
}
#undef OPCL

void tmxk_vi_terms_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_uv) {
	const int nt_ = NTILES(e, p), xm = e->xcd_vertical;
	const size_t lds_slide = ((size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW + 2 * p.L + 1 + TMX_RMTAB_DOUBLES) * sizeof(double);
	if (e->opt_vite_walk < 0 && !with_uv && lds_slide <= 64 * 1024) {      // a thread walks (a segment of) its column; -n = n segments, -1000 = chosen from the grid size
		int nseg = -e->opt_vite_walk;
		if (e->opt_vite_walk == -1000) for (nseg = 2; nt_ * nseg < 2048 && (p.L + 1) / (nseg + 2) >= 5; nseg += 2) { }
		nseg = std::max(1, std::min(nseg, p.L + 1));
		dim3 blk(64, 2), grd(xcd_column_grid(xm, nt_, (nseg + 1) / 2));
		const bool ud = e->udiff && e->fully_explicit;
		const double z2 = e->cfg.ztop * e->cfg.ztop;
		const double cs = ud ? e->cfg.uniform_diffusion_scalar / z2 : 0.0, cw = ud ? e->cfg.uniform_diffusion_vector / z2 : 0.0;
#define LAUNCH_VS(UD_, CL_) hipLaunchKernelGGL((k_vi_terms_explicit_slide<UD_, CL_>), grd, blk, lds_slide, e->stream, p, xin, xup, dt, (const double *)(ud ? e->d_ref : nullptr), cs, cw, nt_, xm, nseg)
		if (ud) { if (p.closed) LAUNCH_VS(true, true); else LAUNCH_VS(true, false); }
		else { if (p.closed) LAUNCH_VS(false, true); else LAUNCH_VS(false, false); }
#undef LAUNCH_VS
		return;
	}
	dim3 blk(64, KT_VC), grd(xcd_column_grid(xm, nt_, (p.L + 1 + KT_VC - 1) / KT_VC));
	if (e->udiff && e->fully_explicit) {
		const double z2 = e->cfg.ztop * e->cfg.ztop;
#if TMX_EXP      // (option "vx_fused", measured slower: experiments flavour only)
		if (with_uv) hipLaunchKernelGGL((k_vi_terms_explicit<true, true>), grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)e->d_ref,
			e->cfg.uniform_diffusion_scalar / z2, e->cfg.uniform_diffusion_vector / z2, nt_, xm);
		else
#endif
		hipLaunchKernelGGL((k_vi_terms_explicit<true, false>), grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)e->d_ref,
			e->cfg.uniform_diffusion_scalar / z2, e->cfg.uniform_diffusion_vector / z2, nt_, xm);
	} else
		hipLaunchKernelGGL((k_vi_terms_explicit<false, false>), grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)nullptr, 0.0, 0.0, nt_, xm);
}

}      // namespace lu_fma / lu_nofma
