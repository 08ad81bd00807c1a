// tmx_k_vertical.hip -- hand-written CDNA4 (gfx950) kernels of the spectral-element hot path (one translation unit per operator family;
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

template <bool UDV>
__global__ __launch_bounds__(64 * KT_VE) void k_v_explicit(KParams p, const double * __restrict__ xin,
	double * __restrict__ xup, double dt, const double * __restrict__ xref, double cf, int ntile, int xmode)
{
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (p.L + KT_VE - 1) / KT_VE, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + threadIdx.x;
	const int k = by * KT_VE + WAVE_UNIFORM(threadIdx.y);
	if (k >= p.L || col >= p.ncol) return;
	v_explicit_point<UDV>(p, xin, xup, dt, xref, cf, col, k);
}

// The same update by a thread that walks a column (or one of `nseg` segments of it) level by level: U, V of the levels k - 2 .. k + 2
// (and, UDV, the reference's) in a sliding register window, xi_dot of every interface evaluated once (the level-parallel form evaluates
// it for the level below and again for the level above), the entering level loaded an iteration ahead, operator coefficients and
// the 1 - eta table in LDS (see k_v_tracers_explicit_slide).  Same statements on the same operands as v_explicit_point: bit-identical (tested).
#define OPCL(op, k, off) opsl[(((op) * (L + 1)) + (k)) * TMX_OPW + ((off) + 2)]
template <bool UDV, bool CLOSED>
__global__ __launch_bounds__(128) void k_v_explicit_slide(KParams p, const double * __restrict__ xin, double * __restrict__ xup, double dt,
	const double * __restrict__ xref, double cf, int ntile, int xmode, int nseg)
{
	extern __shared__ double opsl[];      // [TMX_OP_COUNT][L + 1][TMX_OPW], then 1 - eta [2 L + 1]
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	double * etal = opsl + TMX_OP_COUNT * (L + 1) * TMX_OPW;
	{
		const int tid = threadIdx.y * 64 + threadIdx.x;
		for (int t = tid; t < TMX_OP_COUNT * (L + 1) * TMX_OPW; t += 128) opsl[t] = p.ops[t];
		if (CLOSED) for (int t = tid; t < 2 * L + 1; t += 128) etal[t] = p.eta[t];
		__syncthreads();
	}
	constexpr int MM = CLOSED ? 1 : 2;
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (nseg + 1) / 2, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + threadIdx.x;
	if (col >= p.ncol) return;
	const int sg = WAVE_UNIFORM(by * 2 + (int)threadIdx.y);
	const int seg = (L + nseg - 1) / nseg, k0 = sg * seg, k1 = min(L, k0 + seg);
	if (sg >= nseg || k0 >= k1) return;
	const MetCol mc = met_col(p, col);
	auto in = [&](int l) { return l >= 0 && l < L; };
	// window: levels k - 2 .. k + 2, at the (virtual) level k = k0 - 1 the walk starts from
	double uw[5], vw[5], ur[5], vr[5];
#pragma unroll
	for (int j = 0; j < 5; j++) {
		const int l = k0 - 3 + j;
		const bool ok = in(l);
		uw[j] = ok ? xin[(size_t)TMX_SLAB_U(L, l) * NS + col] : 0.0;
		vw[j] = ok ? xin[(size_t)TMX_SLAB_V(L, l) * NS + col] : 0.0;
		ur[j] = (UDV && ok) ? xref[(size_t)TMX_SLAB_U(L, l) * NS + col] : 0.0;
		vr[j] = (UDV && ok) ? xref[(size_t)TMX_SLAB_V(L, l) * NS + col] : 0.0;
	}
	double xd_lo = 0.0, upU = 0.0, upV = 0.0;
	double wm = (k0 >= 1 && k0 <= L - 1) ? xin[(size_t)TMX_SLAB_W(L, k0) * NS + col] : 0.0;      // W on the interface of the current iteration
#pragma unroll 1
	for (int k = k0 - 1; k < k1; k++) {
		const int m = k + 1;
		// the level that enters the window after this iteration, W of the next interface, the values the next level updates
		const int ln = k + 3;
		const bool okn = in(ln) && ln <= k1 + 1;
		const double eU = okn ? xin[(size_t)TMX_SLAB_U(L, ln) * NS + col] : 0.0, eV = okn ? xin[(size_t)TMX_SLAB_V(L, ln) * NS + col] : 0.0;
		const double eUR = (UDV && okn) ? xref[(size_t)TMX_SLAB_U(L, ln) * NS + col] : 0.0, eVR = (UDV && okn) ? xref[(size_t)TMX_SLAB_V(L, ln) * NS + col] : 0.0;
		const double wn = (m + 1 >= 1 && m + 1 <= L - 1 && m + 1 <= k1) ? xin[(size_t)TMX_SLAB_W(L, m + 1) * NS + col] : 0.0;
		const double upUn = (k + 1 < k1) ? xup[(size_t)TMX_SLAB_U(L, k + 1) * NS + col] : 0.0, upVn = (k + 1 < k1) ? xup[(size_t)TMX_SLAB_V(L, k + 1) * NS + col] : 0.0;
		// xi_dot on interface m (xidot_edge): window entries 1 .. 4 are the levels m - 2 .. m + 1
		double xd_hi = 0.0;
		if (m >= 1 && m <= L - 1) {
			double ue = 0.0, ve = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				const double c = OPCL(TMX_OP_INTERP_NODE_TO_REDGE, m, off);
				ue += c * uw[off + 3];
				ve += c * vw[off + 3];
			}
			double e0, e1, e2;
			metric_edge<MM>(p, mc, m, col, e0, e1, e2, etal);
			xd_hi = e0 * ue + e1 * ve + e2 * wm;
		}
		double res[2] = { 0.0, 0.0 };
		if (k >= k0) {      // level k: window entry 2
			const double w_hi = dt * fabs(xd_hi), w_lo = dt * fabs(xd_lo);
#pragma unroll
			for (int v = 0; v < 2; v++) {
				const double * x = v ? vw : uw, * xr = v ? vr : ur;
				const double x0 = x[2], xm = x[1], xp = x[3];
				double out = v ? upV : upU;
				if (k < L - 1) {
					double b = 0.0;
					b += OPCL(TMX_OP_PENALTY_LEFT, k, 0) * x0;
					b += OPCL(TMX_OP_PENALTY_LEFT, k, 1) * xp;
					out += b * w_hi;
				}
				if (k > 0) {
					double b = 0.0;
					b += OPCL(TMX_OP_PENALTY_RIGHT, k, -1) * xm;
					b += OPCL(TMX_OP_PENALTY_RIGHT, k, 0) * x0;
					out += b * w_lo;
				}
				if (UDV) {
					double dd = 0.0, ddr = 0.0;
#pragma unroll
					for (int off = -2; off <= 2; off++) {
						const int l = k + off;
						if (l < 0 || l >= L) continue;
						const double c = OPCL(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, off);
						dd += c * x[off + 2];
						ddr += c * xr[off + 2];
					}
					out += dt * cf * (dd - ddr);
				}
				res[v] = out;
			}
		}
		xd_lo = xd_hi; wm = wn; upU = upUn; upV = upVn;
#pragma unroll
		for (int j = 0; j < 4; j++) { uw[j] = uw[j + 1]; vw[j] = vw[j + 1]; ur[j] = ur[j + 1]; vr[j] = vr[j + 1]; }
		uw[4] = eU; vw[4] = eV; ur[4] = eUR; vr[4] = eVR;
		// the stores last, the entering level landed before them (see k_v_tracers_explicit_slide)
		asm volatile("" : "+v"(uw[4]), "+v"(vw[4]), "+v"(wm), "+v"(upU), "+v"(upV));
		if (UDV) asm volatile("" : "+v"(ur[4]), "+v"(vr[4]));
		if (k >= k0) { xup[(size_t)k * NS + col] = res[0]; xup[(size_t)(L + k) * NS + col] = res[1]; }
	}
}
#undef OPCL

void tmxk_v_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_udiff_uv) {
	const int nt_ = NTILES(e, p), xm = e->xcd_vertical;
	const size_t lds_slide = ((size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW + 2 * p.L + 1) * sizeof(double);
	if (e->opt_vx_walk < 0 && lds_slide <= 64 * 1024) {      // a thread walks (a segment of) its column; -n = n segments, -1000 = chosen from the grid size
		int nseg = -e->opt_vx_walk;
		// (a light kernel, four resident wavefronts per SIMD: ne30 L40 on one GPU 1 / 2 / 4 segments 3.84 / 3.76 / 3.75 ms per step of BASELINE config 4's
		// shape, the level-parallel kernel 3.98)
		if (e->opt_vx_walk == -1000) for (nseg = 2; nt_ * nseg < 4096 && p.L / (nseg + 2) >= 5; nseg += 2) { }
		nseg = std::max(1, std::min(nseg, p.L));
		dim3 blk(64, 2), grd(xcd_column_grid(xm, nt_, (nseg + 1) / 2));
		const double cf = with_udiff_uv ? e->cfg.uniform_diffusion_vector / (e->cfg.ztop * e->cfg.ztop) : 0.0;
#define LAUNCH_VXS(UD_, CL_) hipLaunchKernelGGL((k_v_explicit_slide<UD_, CL_>), grd, blk, lds_slide, e->stream, p, xin, xup, dt, (const double *)(with_udiff_uv ? e->d_ref : nullptr), cf, nt_, xm, nseg)
		if (with_udiff_uv) { if (p.closed) LAUNCH_VXS(true, true); else LAUNCH_VXS(true, false); }
		else { if (p.closed) LAUNCH_VXS(false, true); else LAUNCH_VXS(false, false); }
#undef LAUNCH_VXS
		return;
	}
	dim3 blk(64, KT_VE), grd(xcd_column_grid(xm, nt_, (p.L + KT_VE - 1) / KT_VE));
	if (with_udiff_uv)
		hipLaunchKernelGGL(k_v_explicit<true>, grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)e->d_ref,
			e->cfg.uniform_diffusion_vector / (e->cfg.ztop * e->cfg.ztop), nt_, xm);
	else
		hipLaunchKernelGGL(k_v_explicit<false>, grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)nullptr, 0.0, nt_, xm);
}

template <int LW>
__global__ __launch_bounds__(64) void k_vi_tracers(KParams p, int nt, const double * __restrict__ xin, const double * __restrict__ w0,
	const double * xbase, double * xup, double dt, int nunique, const int * __restrict__ ucol, const int * __restrict__ udep, int * __restrict__ flag,
	int expl, double ks, const double * __restrict__ xref)
{
	extern __shared__ double smt[];
	const int L = p.L, lane = threadIdx.x;
	const size_t NS = (size_t)p.NS;
	double * A = smt;                        // [L][4][LW] band matrix, rebuilt and factorised per tracer
	double * F = A + (size_t)L * 4 * LW;     // [L][LW] right-hand side / solution
	double * xd0 = F + (size_t)L * LW;       // [L+1][LW] xi_dot of the initial column on interfaces
	double * xd1 = xd0 + (size_t)(L + 1) * LW;   // [L+1][LW] xi_dot with the updated W
	double * qn = xd1 + (size_t)(L + 1) * LW;    // [L][LW] tracer density of the column
	const int u = blockIdx.x * LW + lane;
	if (lane >= LW || u >= nunique) return;
	const int col = ucol ? ucol[u] : u;
	const MetCol mc = met_col(p, col);
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col];
	double * mixr = A;                       // explicit mode: [L][LW] q / rho - (q / rho)_ref   (the matrix is not stored)
	double * rhoe = A + (size_t)L * LW;      // explicit mode: [L+1][LW] rho on interfaces
#define ABW(i_, j_) A[((size_t)(j_) * 4 + 2 + (i_) - (j_)) * LW + lane]
	// U,V on interfaces (InterpolateNodeToREdge of the initial column), xi_dot initial / updated (:3943-3957, :4059-4086)
	for (int k = 0; k <= L; k++) {
		double x0v = 0.0, x1v = 0.0;
		if (k >= 1 && k <= L - 1) {
			double ue = 0.0, ve = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = k + off;
				if (l < 0 || l >= L) continue;
				const double cc = OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, off);
				ue += cc * xin[(size_t)TMX_SLAB_U(L, l) * NS + col];
				ve += cc * xin[(size_t)TMX_SLAB_V(L, l) * NS + col];
			}
			double e0, e1, e2;
			metric_edge(p, mc, k, col, e0, e1, e2);
			x0v = e0 * ue + e1 * ve + e2 * w0[(size_t)k * NS + col];
			x1v = expl ? x0v : e0 * ue + e1 * ve + e2 * xup[(size_t)TMX_SLAB_W(L, k) * NS + col];
		}
		xd0[(size_t)k * LW + lane] = x0v; xd1[(size_t)k * LW + lane] = x1v;
	}
	// The matrix (the same for every tracer; the reference factorises it once) is cheap to rebuild from xi_dot, which
	// keeps the LDS footprint at (8L + 2) x 512 bytes.
	auto assemble = [&]() {
		for (int t = 0; t < L * 4; t++) A[(size_t)t * LW + lane] = 0.0;
		// d F_k / d q_n: DiffREdgeToNode x J_e / J_n x InterpNodeToREdge x xi_dot (:3959-3976); J is level-independent
		for (int k = 0; k < L; k++)
			for (int mo = 0; mo <= 1; mo++) {
				const int m = k + mo;                          // DiffREdgeToNode row k touches interfaces k, k+1
				const double dc = OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, mo);
				for (int no = -2; no <= 1; no++) {
					const int n = m + no;                      // InterpNodeToREdge row m: nodes m-1, m (two-sided at the ends)
					if (n < 0 || n >= L) continue;
					if (n - k < -1 || n - k > 1) continue;     // outside the band the coefficient is structurally zero
					ABW(k, n) += dc * je / jn * OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, no) * xd0[(size_t)m * LW + lane];
				}
			}
		// upwinding in the Jacobian (:3978-4016): interface a, weight |xi_dot_a|
		for (int a = 1; a < L; a++) {
			const double wgt = fabs(xd0[(size_t)a * LW + lane]);
			ABW(a - 1, a - 1) -= wgt * OPC(TMX_OP_PENALTY_LEFT, a - 1, 0);
			ABW(a - 1, a) -= wgt * OPC(TMX_OP_PENALTY_LEFT, a - 1, 1);
			ABW(a, a - 1) -= wgt * OPC(TMX_OP_PENALTY_RIGHT, a, -1);
			ABW(a, a) -= wgt * OPC(TMX_OP_PENALTY_RIGHT, a, 0);
		}
		for (int k = 0; k < L; k++) ABW(k, k) += 1.0 / dt;
	};

	int dep[3];
#pragma unroll
	for (int t = 0; t < 3; t++) dep[t] = udep ? udep[u * 3 + t] : -1;
	bool singular = false, factored = false;
	unsigned long long jpmask = 0, jpmask_hi = 0, zeromask = 0, zeromask_hi = 0;      // interchanges / zero pivots of the factorisation, by column
	if (expl && ks != 0.0) {
		// rho on interfaces: InterpolateNodeToREdge of the initial column (PrepareColumn :1905-1916)
		for (int m = 0; m <= L; m++) {
			double re = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				re += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * xin[(size_t)TMX_SLAB_R(L, l) * NS + col];
			}
			rhoe[(size_t)m * LW + lane] = re;
		}
	}
	for (int c = 0; c < nt; c++) {
		for (int k = 0; k < L; k++) qn[(size_t)k * LW + lane] = xin[(size_t)TMX_SLAB_Q(L, c, k) * NS + col];
		if (expl && ks != 0.0)
			for (int k = 0; k < L; k++) {
				double a = qn[(size_t)k * LW + lane] / xin[(size_t)TMX_SLAB_R(L, k) * NS + col];
				a -= xref[(size_t)TMX_SLAB_Q(L, c, k) * NS + col] / xref[(size_t)TMX_SLAB_R(L, k) * NS + col];
				mixr[(size_t)k * LW + lane] = a;
			}
		// mass flux with the updated xi_dot, its divergence (:4092-4140)
		for (int k = 0; k < L; k++) {
			double mf[2];
			for (int mo = 0; mo <= 1; mo++) {
				const int m = k + mo;
				double qe = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					qe += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * qn[(size_t)l * LW + lane];
				}
				mf[mo] = (m == 0 || m == L) ? 0.0 : je * qe * xd1[(size_t)m * LW + lane];
				if (expl && ks != 0.0 && m > 0 && m < L) {
					double dq = 0.0;       // DifferentiateNodeToREdge of the mixing-ratio difference
#pragma unroll
					for (int off = -2; off <= 1; off++) {
						const int l = m + off;
						if (l < 0 || l >= L) continue;
						dq += OPC(TMX_OP_DIFF_NODE_TO_REDGE, m, off) * mixr[(size_t)l * LW + lane];
					}
					mf[mo] -= ks * rhoe[(size_t)m * LW + lane] * dq;
				}
			}
			double dmf = 0.0;
			dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 0) * mf[0];
			dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 1) * mf[1];
			F[(size_t)k * LW + lane] = dmf / jn;
		}
		// upwinding with the initial xi_dot (LinearColumnDiscPenaltyFEM::Apply; :4153-4181)
		for (int k = 0; k < L; k++) {
			double aux = 0.0;
			if (k < L - 1) {
				double b = 0.0;
				b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * qn[(size_t)k * LW + lane];
				b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * qn[(size_t)(k + 1) * LW + lane];
				aux += b * fabs(xd0[(size_t)(k + 1) * LW + lane]);
			}
			if (k > 0) {
				double b = 0.0;
				b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * qn[(size_t)(k - 1) * LW + lane];
				b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * qn[(size_t)k * LW + lane];
				aux += b * fabs(xd0[(size_t)k * LW + lane]);
			}
			F[(size_t)k * LW + lane] -= aux;
		}
		if (expl) {
			// diagonal matrix 1/dt: dgbtrs leaves b_j / (1/dt)
			const double idt = 1.0 / dt;
			for (int k = 0; k < L; k++) {
				const size_t so = (size_t)TMX_SLAB_Q(L, c, k) * NS;
				double bj = F[(size_t)k * LW + lane];
				if (bj != 0.0) bj /= idt;
				xup[so + col] = xbase[so + col] - bj;
			}
			continue;
		}
		// implicit velocity correction (:4183-4233)
		for (int a = 1; a < L; a++) {
			const double xa = xd0[(size_t)a * LW + lane];
			double e0, e1, e2;
			metric_edge(p, mc, a, col, e0, e1, e2);
			const double sw = (xa > 0.0) ? 1.0 * e2 : ((xa < 0.0) ? -1.0 * e2 : 0.0);
			const double jump = sw * (xup[(size_t)TMX_SLAB_W(L, a) * NS + col] - w0[(size_t)a * NS + col]);
			double f = F[(size_t)(a - 1) * LW + lane];
			f -= OPC(TMX_OP_PENALTY_LEFT, a - 1, 0) * qn[(size_t)(a - 1) * LW + lane] * jump;
			f -= OPC(TMX_OP_PENALTY_LEFT, a - 1, 1) * qn[(size_t)a * LW + lane] * jump;
			F[(size_t)(a - 1) * LW + lane] = f;
			f = F[(size_t)a * LW + lane];
			f -= OPC(TMX_OP_PENALTY_RIGHT, a, -1) * qn[(size_t)(a - 1) * LW + lane] * jump;
			f -= OPC(TMX_OP_PENALTY_RIGHT, a, 0) * qn[(size_t)a * LW + lane] * jump;
			F[(size_t)a * LW + lane] = f;
		}
		// dgbtf2 (kl = ku = 1, kv = 2) once per column -- the matrix is the same for every tracer, the reference factorises it
		// once (:4028-4040) --, then per tracer dgbtrs: forward substitution with the stored multipliers and interchanges, dtbsv
		if (!factored) {
			factored = true;
			assemble();
			int ju = 0;
			for (int jj = 0; jj < L; jj++) {
				if (jj + 2 < L) ABW(jj + 2 - 2, jj + 2) = 0.0;         // fill-in super-diagonal of column jj + kv
				const int km = (jj < L - 1) ? 1 : 0;
				int jp = 0;
				if (km && fabs(ABW(jj + 1, jj)) > fabs(ABW(jj, jj))) jp = 1;
				const double piv = ABW(jj + jp, jj);
				if (piv != 0.0) {
					int tt = jj + 1 + jp; if (tt > L - 1) tt = L - 1;
					if (tt > ju) ju = tt;
					if (jp) {
						jpmask |= 1ull << (jj & 63); if (jj >= 64) jpmask_hi |= 1ull << (jj - 64);
						for (int cc = jj; cc <= ju; cc++) { const double tmp = ABW(jj + 1, cc); ABW(jj + 1, cc) = ABW(jj, cc); ABW(jj, cc) = tmp; }
					}
					if (km) {
						const double r = 1.0 / ABW(jj, jj);
						ABW(jj + 1, jj) *= r;
						for (int cc = jj + 1; cc <= ju; cc++) {
							const double uu = ABW(jj, cc);
							if (uu != 0.0) ABW(jj + 1, cc) = LU_UPD(ABW(jj + 1, cc), ABW(jj + 1, jj), uu);
						}
					}
				} else { singular = true; zeromask |= 1ull << (jj & 63); if (jj >= 64) zeromask_hi |= 1ull << (jj - 64); }
			}
		}
		// forward substitution (dgbtrs): interchange, then eliminate; a zero pivot leaves its column untouched, as above
		for (int jj = 0; jj < L - 1; jj++) {
			const unsigned long long bit = 1ull << (jj & 63);
			const bool isz = (jj < 64) ? (zeromask & bit) != 0 : (zeromask_hi & (1ull << (jj - 64))) != 0;
			if (isz) continue;
			const bool jp = (jj < 64) ? (jpmask & bit) != 0 : (jpmask_hi & (1ull << (jj - 64))) != 0;
			if (jp) { const double tmp = F[(size_t)(jj + 1) * LW + lane]; F[(size_t)(jj + 1) * LW + lane] = F[(size_t)jj * LW + lane]; F[(size_t)jj * LW + lane] = tmp; }
			F[(size_t)(jj + 1) * LW + lane] = LU_UPD(F[(size_t)(jj + 1) * LW + lane], F[(size_t)jj * LW + lane], ABW(jj + 1, jj));
		}
		for (int jj = L - 1; jj >= 0; jj--) {
			double bj = F[(size_t)jj * LW + lane];
			if (bj != 0.0) {
				bj /= ABW(jj, jj);
				F[(size_t)jj * LW + lane] = bj;
				const int lo = (jj - 2 < 0) ? 0 : jj - 2;
				for (int ii = jj - 1; ii >= lo; ii--) F[(size_t)ii * LW + lane] = LU_UPD(F[(size_t)ii * LW + lane], bj, ABW(ii, jj));
			}
		}
		for (int k = 0; k < L; k++) {
			const size_t so = (size_t)TMX_SLAB_Q(L, c, k) * NS;
			const double val = xbase[so + col] - F[(size_t)k * LW + lane];
			xup[so + col] = val;
#pragma unroll
			for (int t = 0; t < 3; t++) if (dep[t] >= 0) xup[so + dep[t]] = val;
		}
	}
	if (singular) atomicOr(flag, TMX_FLAG_SINGULAR);
#undef ABW
}

// The same column update (implicit mode) with the level-parallel parts spread over NR row lanes per column: a workgroup =
// 16 columns x NR row lanes.  Row lane t of a column evaluates xi_dot, the matrix rows, the tracer loads, the
// right-hand side and the result stores of the levels k = t, t + NR, ...; every matrix entry and every right-hand-side entry is
// formed by the lane that owns its row, in the order the sequential loops of k_vi_tracers accumulate it (flux terms, upwinding
// of interface k then k + 1, 1/dt; velocity correction of interface k then k + 1), so the results are bit-identical; only the
// factorisation and the two substitutions stay on one lane per column (pivot row / running entries in registers, the
// next row prefetched from LDS).  LDS: (9L + 3) x 16 doubles per workgroup, so that at L = 30 five workgroups share a CU.
template <int NR, int LWB> __global__ __launch_bounds__(NR << LWB) void k_vi_tracers_rows(KParams p, int nt, const double * __restrict__ xin, const double * __restrict__ w0,
	const double * xbase, double * xup, double dt, int nunique, const int * __restrict__ ucol, const int * __restrict__ udep, int * __restrict__ flag)
{
	constexpr int LW = 1 << LWB;        // columns per workgroup (16 or 8)
	extern __shared__ double smt[];
	const int L = p.L, lane = threadIdx.x & (LW - 1), t = threadIdx.x >> LWB;
	const size_t NS = (size_t)p.NS;
	double * A = smt;                                // [L][4][LW]
	double * F = A + (size_t)L * 4 * LW;             // [L][LW]
	double * xd0 = F + (size_t)L * LW;               // [L+1][LW]
	double * xd1 = xd0 + (size_t)(L + 1) * LW;       // [L+1][LW]
	double * qn = xd1 + (size_t)(L + 1) * LW;        // [L][LW]
	double * jmp = qn + (size_t)L * LW;              // [L+1][LW] upwind sign x metric x (W updated - W initial) per interface
	const int uraw = blockIdx.x * LW + lane;
	const bool valid = uraw < nunique;
	const int u = valid ? uraw : nunique - 1;        // lanes of a ragged last workgroup redo the last column (no stores): barriers below
	const int col = ucol ? ucol[u] : u;
	const MetCol mc = met_col(p, col);
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col];
	// row k of the band matrix in slots 0..2 (sub-, main, super-diagonal); after the factorisation: slot 0 of row j + 1 = the
	// multiplier of step j, slots 1..3 of row j = row j of U (diagonal and the two entries right of it)
#define AR(k_, s_) A[((size_t)(k_) * 4 + (s_)) * LW + lane]
	// xi_dot on interfaces, initial and with the updated W (:3943-3957, :4059-4086)
	for (int k = t; k <= L; k += NR) {
		double x0v = 0.0, x1v = 0.0, jv = 0.0;
		if (k >= 1 && k <= L - 1) {
			double ue = 0.0, ve = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = k + off;
				if (l < 0 || l >= L) continue;
				const double cc = OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, off);
				ue += cc * xin[(size_t)TMX_SLAB_U(L, l) * NS + col];
				ve += cc * xin[(size_t)TMX_SLAB_V(L, l) * NS + col];
			}
			double e0, e1, e2;
			metric_edge(p, mc, k, col, e0, e1, e2);
			const double wi = w0[(size_t)k * NS + col], wu = xup[(size_t)TMX_SLAB_W(L, k) * NS + col];
			x0v = e0 * ue + e1 * ve + e2 * wi;
			x1v = e0 * ue + e1 * ve + e2 * wu;
			const double sw = (x0v > 0.0) ? 1.0 * e2 : ((x0v < 0.0) ? -1.0 * e2 : 0.0);
			jv = sw * (wu - wi);
		}
		xd0[(size_t)k * LW + lane] = x0v; xd1[(size_t)k * LW + lane] = x1v; jmp[(size_t)k * LW + lane] = jv;
	}
	__syncthreads();
	// matrix rows (:3959-4016), each entry by the owner of its row
	for (int k = t; k < L; k += NR) {
#pragma unroll
		for (int d = -1; d <= 1; d++) {
			const int n = k + d;
			if (n < 0 || n >= L) { AR(k, d + 1) = 0.0; continue; }
			double a = 0.0;
			// d F_k / d q_n: DiffREdgeToNode x J_e / J_n x InterpNodeToREdge x xi_dot; interface m = k + mo, node n = m + no
#pragma unroll
			for (int mo = 0; mo <= 1; mo++) {
				const int m = k + mo, no = n - m;
				if (no < -2 || no > 1) continue;
				a += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, mo) * je / jn * OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, no) * xd0[(size_t)m * LW + lane];
			}
			// upwinding: interface a = k (this row as "a"), then a = k + 1 (this row as "a - 1")
			if (k >= 1 && k <= L - 1) {
				const double wgt = fabs(xd0[(size_t)k * LW + lane]);
				if (d == -1) a -= wgt * OPC(TMX_OP_PENALTY_RIGHT, k, -1);
				if (d == 0) a -= wgt * OPC(TMX_OP_PENALTY_RIGHT, k, 0);
			}
			if (k + 1 <= L - 1) {
				const double wgt = fabs(xd0[(size_t)(k + 1) * LW + lane]);
				if (d == 0) a -= wgt * OPC(TMX_OP_PENALTY_LEFT, k, 0);
				if (d == 1) a -= wgt * OPC(TMX_OP_PENALTY_LEFT, k, 1);
			}
			if (d == 0) a += 1.0 / dt;
			AR(k, d + 1) = a;
		}
	}
	__syncthreads();
	// dgbtf2 (kl = ku = 1, kv = 2), one lane per column, with the pivot candidate row carried in registers: at step j the
	// candidate has entries (c0, c1) in columns j, j + 1 (the fill-in column j + 2 holds the zero dgbtf2 stores there); row j + 1
	// is (l, d, u) in columns j .. j + 2.  Interchange when |l| > |c0| (idamax keeps the first of equals), multiplier = the other
	// row's leading entry x (1 / pivot) (dscal), rank-1 update fused as in LU_UPD, skipped for a zero in the pivot row (dger).
	bool singular = false;
	unsigned long long jpmask = 0, jpmask_hi = 0, zeromask = 0, zeromask_hi = 0;
	if (t == 0) {
		double c0 = AR(0, 1), c1 = AR(0, 2);
		double nl = (L > 1) ? AR(1, 0) : 0.0, nd = (L > 1) ? AR(1, 1) : 0.0, nu = (L > 1) ? AR(1, 2) : 0.0;
		for (int jj = 0; jj < L - 1; jj++) {
			const double l = nl, d = nd, un = nu;
			if (jj + 2 < L) { nl = AR(jj + 2, 0); nd = AR(jj + 2, 1); nu = AR(jj + 2, 2); }
			const bool jp = fabs(l) > fabs(c0);
			const double piv = jp ? l : c0;
			if (piv != 0.0) {
				const double p1 = jp ? d : c1, p2 = jp ? un : 0.0;          // pivot row right of the diagonal
				const double o0 = jp ? c0 : l, o1 = jp ? c1 : d, o2 = jp ? 0.0 : un;
				if (jp) { jpmask |= 1ull << (jj & 63); if (jj >= 64) jpmask_hi |= 1ull << (jj - 64); }
				const double r = 1.0 / piv;
				const double m = o0 * r;
				AR(jj, 1) = piv; AR(jj, 2) = p1; AR(jj, 3) = p2; AR(jj + 1, 0) = m;
				c0 = (p1 != 0.0) ? LU_UPD(o1, m, p1) : o1;
				c1 = (jp && p2 != 0.0) ? LU_UPD(o2, m, p2) : o2;
			} else {
				singular = true; zeromask |= 1ull << (jj & 63); if (jj >= 64) zeromask_hi |= 1ull << (jj - 64);
				AR(jj, 1) = c0; AR(jj, 2) = c1; AR(jj, 3) = 0.0;       // l == c0 == 0: nothing eliminated, row j + 1 becomes the candidate
				c0 = d; c1 = un;
			}
		}
		AR(L - 1, 1) = c0; AR(L - 1, 2) = 0.0; AR(L - 1, 3) = 0.0;
		if (c0 == 0.0) { singular = true; zeromask |= 1ull << ((L - 1) & 63); if (L - 1 >= 64) zeromask_hi |= 1ull << (L - 1 - 64); }
	}
	int dep[3];
#pragma unroll
	for (int q = 0; q < 3; q++) dep[q] = udep ? udep[u * 3 + q] : -1;
	for (int c = 0; c < nt; c++) {
		__syncthreads();        // the previous tracer's stores have read F
		for (int k = t; k < L; k += NR) qn[(size_t)k * LW + lane] = xin[(size_t)TMX_SLAB_Q(L, c, k) * NS + col];
		__syncthreads();
		for (int k = t; k < L; k += NR) {
			// mass flux with the updated xi_dot, its divergence (:4092-4140)
			double mf[2];
#pragma unroll
			for (int mo = 0; mo <= 1; mo++) {
				const int m = k + mo;
				double qe = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					qe += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * qn[(size_t)l * LW + lane];
				}
				mf[mo] = (m == 0 || m == L) ? 0.0 : je * qe * xd1[(size_t)m * LW + lane];
			}
			double dmf = 0.0;
			dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 0) * mf[0];
			dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 1) * mf[1];
			double f = dmf / jn;
			// upwinding with the initial xi_dot (:4153-4181)
			double aux = 0.0;
			if (k < L - 1) {
				double b = 0.0;
				b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * qn[(size_t)k * LW + lane];
				b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * qn[(size_t)(k + 1) * LW + lane];
				aux += b * fabs(xd0[(size_t)(k + 1) * LW + lane]);
			}
			if (k > 0) {
				double b = 0.0;
				b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * qn[(size_t)(k - 1) * LW + lane];
				b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * qn[(size_t)k * LW + lane];
				aux += b * fabs(xd0[(size_t)k * LW + lane]);
			}
			f -= aux;
			// implicit velocity correction (:4183-4233): interface a = k (this row as "a"), then a = k + 1 (as "a - 1")
#pragma unroll
			for (int s_ = 0; s_ <= 1; s_++) {
				const int a = k + s_;
				if (a < 1 || a > L - 1) continue;
				const double jump = jmp[(size_t)a * LW + lane];
				if (s_ == 0) {
					f -= OPC(TMX_OP_PENALTY_RIGHT, a, -1) * qn[(size_t)(a - 1) * LW + lane] * jump;
					f -= OPC(TMX_OP_PENALTY_RIGHT, a, 0) * qn[(size_t)a * LW + lane] * jump;
				} else {
					f -= OPC(TMX_OP_PENALTY_LEFT, a - 1, 0) * qn[(size_t)(a - 1) * LW + lane] * jump;
					f -= OPC(TMX_OP_PENALTY_LEFT, a - 1, 1) * qn[(size_t)a * LW + lane] * jump;
				}
			}
			F[(size_t)k * LW + lane] = f;
		}
		__syncthreads();
		if (t == 0) {
			// dgbtrs: forward substitution with the stored multipliers and interchanges, the running entry in a register ...
			double fc = F[lane], fn = (L > 1) ? F[(size_t)LW + lane] : 0.0, mm = (L > 1) ? AR(1, 0) : 0.0;
			for (int jj = 0; jj < L - 1; jj++) {
				const double f1 = fn, m = mm;
				if (jj + 2 < L) { fn = F[(size_t)(jj + 2) * LW + lane]; mm = AR(jj + 2, 0); }
				const unsigned long long bit = 1ull << (jj & 63);
				const bool isz = (jj < 64) ? (zeromask & bit) != 0 : (zeromask_hi & (1ull << (jj - 64))) != 0;
				const bool jp = (jj < 64) ? (jpmask & bit) != 0 : (jpmask_hi & (1ull << (jj - 64))) != 0;
				const double top = (jp && !isz) ? f1 : fc, bot = (jp && !isz) ? fc : f1;
				F[(size_t)jj * LW + lane] = top;
				fc = isz ? bot : LU_UPD(bot, top, m);
			}
			// ... then dtbsv (upper, no transpose, non-unit), column oriented: entry j is divided by the diagonal and then leaves
			// entries j - 1 and j - 2; entry i therefore receives column i + 2 before column i + 1.  b1 = entry j - 1 so far.
			double cur = fc, b1 = (L > 1) ? F[(size_t)(L - 2) * LW + lane] : 0.0;
			double dg = AR(L - 1, 1), u1 = (L > 1) ? AR(L - 2, 2) : 0.0, u2 = (L > 2) ? AR(L - 3, 3) : 0.0, fr = (L > 2) ? F[(size_t)(L - 3) * LW + lane] : 0.0;
			for (int jj = L - 1; jj >= 0; jj--) {
				const double dgj = dg, u1j = u1, u2j = u2, b2 = fr;
				if (jj >= 1) dg = AR(jj - 1, 1);
				if (jj >= 2) u1 = AR(jj - 2, 2);
				if (jj >= 3) { u2 = AR(jj - 3, 3); fr = F[(size_t)(jj - 3) * LW + lane]; }
				const bool nz = cur != 0.0;
				const double x = nz ? cur / dgj : cur;
				F[(size_t)jj * LW + lane] = x;
				cur = nz ? LU_UPD(b1, x, u1j) : b1;
				b1 = nz ? LU_UPD(b2, x, u2j) : b2;
			}
		}
		__syncthreads();
		if (valid)
			for (int k = t; k < L; k += NR) {
				const size_t so = (size_t)TMX_SLAB_Q(L, c, k) * NS;
				const double val = xbase[so + col] - F[(size_t)k * LW + lane];
				xup[so + col] = val;
#pragma unroll
				for (int q = 0; q < 3; q++) if (dep[q] >= 0) xup[so + dep[q]] = val;
			}
	}
	if (singular && valid) atomicOr(flag, TMX_FLAG_SINGULAR);
#undef AR
}

static size_t vt_rows_lds(int L, int lw = 16) { return ((size_t)L * 4 + (size_t)L * 2 + (size_t)(L + 1) * 3) * lw * sizeof(double); }
// row lanes per column of k_vi_tracers_rows: 16 (four wavefronts per 16 columns) up to 48 levels, 32 above; TMX_VT_NR = 4 | 8 | 16 | 32;
// 8 columns per workgroup up to 48 levels (half the LDS per workgroup, twice the workgroups per CU), 16 above; TMX_VT_LW8 = 0 | 1
static void launch_vt_rows(tmx_engine * e, const KParams & p, int ncols, const double * xin, const double * w0, const double * xbase, double * xup,
	double dt, const int * ucol, const int * udep, size_t lds)
{
	const int nr_env = e->opt_vt_nr, lw8_env = e->opt_vt_lw8;      // options "vt_row_lanes", "vt_lw8"
	const bool lw8 = lw8_env >= 0 ? lw8_env == 1 : p.L <= 48;    // ne30, 2 tracers: L30 1.308 (8 columns) / 1.358 (16) ms per step, L60 3.35 / 3.21
	const int nr = nr_env ? nr_env : (p.L > 48 ? 32 : 16);       // measured at ne30: L30 1.36 (16) / 1.49 (32) ms, L60 3.50 (16) / 3.25 (32) ms per step
#define LAUNCH_VTR(NR_, LWB_) do { \
		const size_t lds_ = vt_rows_lds(p.L, 1 << LWB_); \
		hipFuncSetAttribute((const void *)k_vi_tracers_rows<NR_, LWB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
		hipLaunchKernelGGL((k_vi_tracers_rows<NR_, LWB_>), dim3((ncols + (1 << LWB_) - 1) >> LWB_), dim3(NR_ << LWB_), lds_, e->stream, p, e->nt, xin, w0, xbase, xup, dt, \
			ncols, ucol, udep, e->d_flag); } while (0)
	if (lw8) { if (nr == 8) LAUNCH_VTR(8, 3); else if (nr == 32) LAUNCH_VTR(32, 3); else LAUNCH_VTR(16, 3); }
	else if (nr == 4) LAUNCH_VTR(4, 4); else if (nr == 8) LAUNCH_VTR(8, 4); else if (nr == 32) LAUNCH_VTR(32, 4); else LAUNCH_VTR(16, 4);
#undef LAUNCH_VTR
}

// lanes per workgroup such that the LDS working set (8L + 2) x LW x 8 bytes fits one CU
static int vi_tracers_lanes(int L, size_t & lds) {
	for (int lw = 64; lw >= 32; lw >>= 1) {
		lds = ((size_t)L * 4 + (size_t)L * 2 + (size_t)(L + 1) * 2) * lw * sizeof(double);
		if (lds <= 160 * 1024) return lw;
	}
	return 0;
}

int tmxk_vi_tracers(tmx_engine * e, const KParams & p, const double * xin, const double * w0, const double * xbase, double * xup, double dt) {
	if (e->nunique == 0 || e->nt == 0) return 0;
	if (e->opt_vt_rows) {      // default: 16 columns x 4 row lanes per workgroup
		const size_t l2 = vt_rows_lds(p.L);
		if (l2 <= 160 * 1024) {
			launch_vt_rows(e, p, e->nunique, xin, w0, xbase, xup, dt, (const int *)e->d_ucol, (const int *)e->d_udep, l2);
			return 0;
		}
	}
	size_t lds; int lw = vi_tracers_lanes(p.L, lds);
	if (!lw) return -1;
	// Fewer columns per workgroup = more workgroups per CU: the kernel lives in LDS (dependent read-modify-write chains of
	// ~100 cycles each) and one 64-column workgroup fills a CU's LDS, i.e. ONE wavefront per CU.  With 16 columns per
	// workgroup five wavefronts share a CU and hide each other's LDS latency (TMX_VT_LANES: 64, 32, 16, 8 for A/B).
	int want = 16;
	want = e->opt_vt_lanes;
	if (want < lw && (want == 32 || want == 16 || want == 8)) { lw = want; lds = ((size_t)p.L * 4 + (size_t)p.L * 2 + (size_t)(p.L + 1) * 2) * lw * sizeof(double); }
#define LAUNCH_VT(LW_) do { hipFuncSetAttribute((const void *)k_vi_tracers<LW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
	hipLaunchKernelGGL(k_vi_tracers<LW_>, dim3((e->nunique + LW_ - 1) / LW_), dim3(64), lds, e->stream, p, e->nt, xin, w0, xbase, xup, dt, \
		e->nunique, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_flag, 0, 0.0, (const double *)nullptr); } while (0)
	if (lw == 64) LAUNCH_VT(64); else if (lw == 32) LAUNCH_VT(32); else if (lw == 16) LAUNCH_VT(16); else LAUNCH_VT(8);
#undef LAUNCH_VT
	return 0;
}

#if TMX_EXP      // the form without LDS staging (option "vt_explicit_v1"), superseded by k_v_tracers_explicit_tile: experiments flavour only
// UpdateColumnTracers in the fully explicit vertical mode, level-parallel.  There the matrix is the diagonal 1/dt, so the
// update of (column, level, tracer) only needs the column within two levels: one thread per (column, level) evaluates
// exactly the statements of k_vi_tracers' explicit branch (same operands, same order: bit-identical, tested) instead of one
// lane walking the whole column out of LDS -- that form took 3.4 ms per launch at ne30 L40 (32 columns per workgroup, 82 KB
// of LDS each), 75 % of a supercell step.
__global__ __launch_bounds__(256) void k_v_tracers_explicit(KParams p, int nt, const double * __restrict__ xin, double * xup, double dt,
	double ks, const double * __restrict__ xref)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = TILE_X(p) * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (k >= L || col >= p.ncol) return;
	const MetCol mc = met_col(p, col);
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col];
	const double * w0 = xin + (size_t)TMX_SLAB_W(L, 0) * NS;
	// xi_dot of the initial column on the interfaces k and k+1 (zero at the boundaries)
	double xd[2];
#pragma unroll
	for (int mo = 0; mo <= 1; mo++) {
		const int m = k + mo;
		double x0v = 0.0;
		if (m >= 1 && m <= L - 1) {
			double ue = 0.0, ve = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				const double cc = OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off);
				ue += cc * xin[(size_t)TMX_SLAB_U(L, l) * NS + col];
				ve += cc * xin[(size_t)TMX_SLAB_V(L, l) * NS + col];
			}
			double e0, e1, e2;
			metric_edge(p, mc, m, col, e0, e1, e2);
			x0v = e0 * ue + e1 * ve + e2 * w0[(size_t)m * NS + col];
		}
		xd[mo] = x0v;
	}
	// rho on the two interfaces (uniform diffusion only)
	double rhoe[2] = { 0.0, 0.0 };
	if (ks != 0.0) {
#pragma unroll
		for (int mo = 0; mo <= 1; mo++) {
			const int m = k + mo;
			double re = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				re += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * xin[(size_t)TMX_SLAB_R(L, l) * NS + col];
			}
			rhoe[mo] = re;
		}
	}
	// rho of the five levels k-2 .. k+2 the stencils reach (each value is loaded once and divided once per tracer)
	double rh5[5] = { 1.0, 1.0, 1.0, 1.0, 1.0 }, rr5[5] = { 1.0, 1.0, 1.0, 1.0, 1.0 };
	if (ks != 0.0) {
#pragma unroll
		for (int t = 0; t < 5; t++) {
			const int l = k - 2 + t;
			if (l < 0 || l >= L) continue;
			rh5[t] = xin[(size_t)TMX_SLAB_R(L, l) * NS + col];
			rr5[t] = xref[(size_t)TMX_SLAB_R(L, l) * NS + col];
		}
	}
	for (int c = 0; c < nt; c++) {
		double q5[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 }, mr5[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
		for (int t = 0; t < 5; t++) {
			const int l = k - 2 + t;
			if (l < 0 || l >= L) continue;
			q5[t] = xin[(size_t)TMX_SLAB_Q(L, c, l) * NS + col];
			if (ks != 0.0) {
				double a_ = q5[t] / rh5[t];
				a_ -= xref[(size_t)TMX_SLAB_Q(L, c, l) * NS + col] / rr5[t];
				mr5[t] = a_;
			}
		}
		auto qn = [&](int l) -> double { return q5[l - k + 2]; };          // l in [k-2, k+2], unrolled: a register
		auto mixr = [&](int l) -> double { return mr5[l - k + 2]; };
		double mf[2];
#pragma unroll
		for (int mo = 0; mo <= 1; mo++) {
			const int m = k + mo;
			double qe = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				qe += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * qn(l);
			}
			mf[mo] = (m == 0 || m == L) ? 0.0 : je * qe * xd[mo];
			if (ks != 0.0 && m > 0 && m < L) {
				double dq = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					dq += OPC(TMX_OP_DIFF_NODE_TO_REDGE, m, off) * mixr(l);
				}
				mf[mo] -= ks * rhoe[mo] * dq;
			}
		}
		double dmf = 0.0;
		dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 0) * mf[0];
		dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 1) * mf[1];
		double F = dmf / jn;
		double aux = 0.0;
		if (k < L - 1) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * qn(k);
			b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * qn(k + 1);
			aux += b * fabs(xd[1]);
		}
		if (k > 0) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * qn(k - 1);
			b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * qn(k);
			aux += b * fabs(xd[0]);
		}
		F -= aux;
		const double idt = 1.0 / dt;
		if (F != 0.0) F /= idt;
		const size_t so = (size_t)TMX_SLAB_Q(L, c, k) * NS;
		xup[so + col] = xup[so + col] - F;
	}
}
#endif      // TMX_EXP

// The same update (k_v_tracers_explicit, experiments flavour: one thread per (column, level) with every operand from memory) with the shared operands of a tile staged once in LDS: a workgroup = 64 columns x 8 levels; xi_dot (and, with
// uniform diffusion, rho) on the tile's 9 interfaces, rho on its 12 levels and per tracer the 12 column values and mixing-ratio
// deviations (two fp64 divisions each) are evaluated by one thread each instead of by every thread whose stencil reaches them
// (5 x for the divisions: 372 -> 264 us per launch at ne30 L40 with three tracers).  Every value is produced by the same expression as above, so the results are bit-identical.
__global__ __launch_bounds__(512) void k_v_tracers_explicit_tile(KParams p, int nt, const double * __restrict__ xin, double * xup, double dt,
	double ks, const double * __restrict__ xref, int ntile, int xmode)
{
	constexpr int KT = 8, NL = KT + 4, NI = KT + 1;
	__shared__ double s_xd[NI][64], s_re[NI][64], s_rh[NL][64], s_rr[NL][64], s_q[NL][64], s_mr[NL][64];
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int tx = threadIdx.x, y = WAVE_UNIFORM(threadIdx.y);
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (L + KT - 1) / KT, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + tx;
	const int k0 = by * KT, k = k0 + y;
	const double * w0 = xin + (size_t)TMX_SLAB_W(L, 0) * NS;
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col];
	// the column values a thread stages (levels y and y + KT of the tile's 12) and the value it updates are loaded one tracer
	// ahead of their use, so that their latency overlaps the previous tracer's arithmetic instead of following a barrier
	double pq[2] = { 0.0, 0.0 }, pr[2] = { 0.0, 0.0 }, pup = 0.0;
	const bool mine = k < L && col < p.ncol;
	auto prefetch = [&](int c) {
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const int li = y + h * KT, l = k0 - 2 + li;
			if (li < NL && l >= 0 && l < L) {
				pq[h] = xin[(size_t)TMX_SLAB_Q(L, c, l) * NS + col];
				if (ks != 0.0) pr[h] = xref[(size_t)TMX_SLAB_Q(L, c, l) * NS + col];
			}
		}
		if (mine) pup = xup[(size_t)TMX_SLAB_Q(L, c, k) * NS + col];
	};
	if (nt > 0) prefetch(0);       // in flight while the shared operands below are staged
	{
		const MetCol mc = met_col(p, col);
		for (int mi = y; mi < NI; mi += KT) {
			const int m = k0 + mi;
			double x0v = 0.0, re = 0.0;
			if (m >= 1 && m <= L - 1) {
				double ue = 0.0, ve = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					const double cc = OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off);
					ue += cc * xin[(size_t)TMX_SLAB_U(L, l) * NS + col];
					ve += cc * xin[(size_t)TMX_SLAB_V(L, l) * NS + col];
				}
				double e0, e1, e2;
				metric_edge(p, mc, m, col, e0, e1, e2);
				x0v = e0 * ue + e1 * ve + e2 * w0[(size_t)m * NS + col];
			}
			if (ks != 0.0 && m <= L) {
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					re += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * xin[(size_t)TMX_SLAB_R(L, l) * NS + col];
				}
			}
			s_xd[mi][tx] = x0v; s_re[mi][tx] = re;
		}
		for (int li = y; li < NL; li += KT) {
			const int l = k0 - 2 + li;
			double a = 1.0, b = 1.0;
			if (ks != 0.0 && l >= 0 && l < L) { a = xin[(size_t)TMX_SLAB_R(L, l) * NS + col]; b = xref[(size_t)TMX_SLAB_R(L, l) * NS + col]; }
			s_rh[li][tx] = a; s_rr[li][tx] = b;
		}
	}
	for (int c = 0; c < nt; c++) {
		__syncthreads();           // first pass: the staging above; later passes: the previous tracer's reads of s_q / s_mr
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const int li = y + h * KT, l = k0 - 2 + li;
			if (li >= NL) continue;
			double qv = 0.0, mr = 0.0;
			if (l >= 0 && l < L) {
				qv = pq[h];
				if (ks != 0.0) {
					double a_ = qv / s_rh[li][tx];
					a_ -= pr[h] / s_rr[li][tx];
					mr = a_;
				}
			}
			s_q[li][tx] = qv; s_mr[li][tx] = mr;
		}
		const double up0 = pup;
		if (c + 1 < nt) prefetch(c + 1);
		__syncthreads();
		if (k >= L || col >= p.ncol) continue;
		auto qn = [&](int l) -> double { return s_q[l - k0 + 2][tx]; };
		auto mixr = [&](int l) -> double { return s_mr[l - k0 + 2][tx]; };
		double mf[2];
#pragma unroll
		for (int mo = 0; mo <= 1; mo++) {
			const int m = k + mo;
			double qe = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				qe += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * qn(l);
			}
			mf[mo] = (m == 0 || m == L) ? 0.0 : je * qe * s_xd[y + mo][tx];
			if (ks != 0.0 && m > 0 && m < L) {
				double dq = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					dq += OPC(TMX_OP_DIFF_NODE_TO_REDGE, m, off) * mixr(l);
				}
				mf[mo] -= ks * s_re[y + mo][tx] * dq;
			}
		}
		double dmf = 0.0;
		dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 0) * mf[0];
		dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 1) * mf[1];
		double F = dmf / jn;
		double aux = 0.0;
		if (k < L - 1) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * qn(k);
			b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * qn(k + 1);
			aux += b * fabs(s_xd[y + 1][tx]);
		}
		if (k > 0) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * qn(k - 1);
			b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * qn(k);
			aux += b * fabs(s_xd[y][tx]);
		}
		F -= aux;
		const double idt = 1.0 / dt;
		if (F != 0.0) F /= idt;
		const size_t so = (size_t)TMX_SLAB_Q(L, c, k) * NS;
		xup[so + col] = up0 - F;
	}
}

#if TMX_EXP      // superseded by the sliding-window form below (measured: -2 % against the tiled kernel at 4 levels per thread, slower at 8 and 10): experiments flavour only
// The same update by column segments: a thread = (column, KC consecutive levels), no LDS and no barrier.  The operands a segment's
// stencils reach -- U, V, rho on KC + 3 / KC + 4 levels, per tracer the KC + 4 column values and mixing-ratio deviations -- are loaded
// once into registers and every interface flux is formed once (the level-parallel forms evaluate the flux of interface k + 1 for level
// k and again for level k + 1: the same expression on the same operands, so once is the same value).  Statement for statement the
// arithmetic of k_v_tracers_explicit_tile: bit-identical (tested).
template <int KC>
__global__ __launch_bounds__(64) void k_v_tracers_explicit_walk(KParams p, int nt, const double * __restrict__ xin, double * xup, double dt,
	double ks, const double * __restrict__ xref, int ntile, int xmode)
{
	constexpr int NL = KC + 4, NI = KC + 1;
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (L + KC - 1) / KC, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + threadIdx.x;
	if (col >= p.ncol) return;
	const int k0 = WAVE_UNIFORM(by * KC);
	const int nk = min(KC, L - k0);
	const double * w0 = xin + (size_t)TMX_SLAB_W(L, 0) * NS;
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col];
	double xd[NI], re[NI], rh[NL], rr[NL];
	{
		// xi_dot of the initial column on the segment's interfaces (zero at the boundaries), rho there (uniform diffusion only)
		const MetCol mc = met_col(p, col);
		double uw[KC + 3], vw[KC + 3];
#pragma unroll
		for (int t = 0; t < KC + 3; t++) {
			const int l = k0 - 2 + t;
			uw[t] = 0.0; vw[t] = 0.0;
			if (l >= 0 && l < L) { uw[t] = xin[(size_t)TMX_SLAB_U(L, l) * NS + col]; vw[t] = xin[(size_t)TMX_SLAB_V(L, l) * NS + col]; }
		}
#pragma unroll
		for (int li = 0; li < NL; li++) {
			const int l = k0 - 2 + li;
			rh[li] = 1.0; rr[li] = 1.0;
			if (ks != 0.0 && l >= 0 && l < L) { rh[li] = xin[(size_t)TMX_SLAB_R(L, l) * NS + col]; rr[li] = xref[(size_t)TMX_SLAB_R(L, l) * NS + col]; }
		}
#pragma unroll
		for (int mi = 0; mi < NI; mi++) {
			const int m = k0 + mi;
			double x0v = 0.0, r0 = 0.0;
			if (mi <= nk && m >= 1 && m <= L - 1) {
				double ue = 0.0, ve = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					const double cc = OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off);
					ue += cc * uw[mi + off + 2];
					ve += cc * vw[mi + off + 2];
				}
				double e0, e1, e2;
				metric_edge(p, mc, m, col, e0, e1, e2);
				x0v = e0 * ue + e1 * ve + e2 * w0[(size_t)m * NS + col];
			}
			if (ks != 0.0 && mi <= nk && m <= L) {
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					r0 += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * rh[mi + off + 2];
				}
			}
			xd[mi] = x0v; re[mi] = r0;
		}
	}
	const double idt = 1.0 / dt;
	for (int c = 0; c < nt; c++) {
		double q[NL], mr[NL], up0[KC];
#pragma unroll
		for (int li = 0; li < NL; li++) {
			const int l = k0 - 2 + li;
			double qv = 0.0, a_ = 0.0;
			if (l >= 0 && l < L) {
				qv = xin[(size_t)TMX_SLAB_Q(L, c, l) * NS + col];
				if (ks != 0.0) {
					a_ = qv / rh[li];
					a_ -= xref[(size_t)TMX_SLAB_Q(L, c, l) * NS + col] / rr[li];
				}
			}
			q[li] = qv; mr[li] = a_;
		}
#pragma unroll
		for (int i = 0; i < KC; i++) up0[i] = (i < nk) ? xup[(size_t)TMX_SLAB_Q(L, c, k0 + i) * NS + col] : 0.0;
		double mf[NI];
#pragma unroll
		for (int mi = 0; mi < NI; mi++) {
			const int m = k0 + mi;
			double v = 0.0;
			if (mi <= nk) {
				double qe = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					qe += OPC(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * q[mi + off + 2];
				}
				v = (m == 0 || m == L) ? 0.0 : je * qe * xd[mi];
				if (ks != 0.0 && m > 0 && m < L) {
					double dq = 0.0;
#pragma unroll
					for (int off = -2; off <= 1; off++) {
						const int l = m + off;
						if (l < 0 || l >= L) continue;
						dq += OPC(TMX_OP_DIFF_NODE_TO_REDGE, m, off) * mr[mi + off + 2];
					}
					v -= ks * re[mi] * dq;
				}
			}
			mf[mi] = v;
		}
#pragma unroll
		for (int i = 0; i < KC; i++) {
			if (i >= nk) continue;
			const int k = k0 + i;
			double dmf = 0.0;
			dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 0) * mf[i];
			dmf += OPC(TMX_OP_DIFF_REDGE_TO_NODE, k, 1) * mf[i + 1];
			double F = dmf / jn;
			double aux = 0.0;
			if (k < L - 1) {
				double b = 0.0;
				b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * q[i + 2];
				b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * q[i + 3];
				aux += b * fabs(xd[i + 1]);
			}
			if (k > 0) {
				double b = 0.0;
				b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * q[i + 1];
				b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * q[i + 2];
				aux += b * fabs(xd[i]);
			}
			F -= aux;
			if (F != 0.0) F /= idt;
			xup[(size_t)TMX_SLAB_Q(L, c, k) * NS + col] = up0[i] - F;
		}
	}
}

#endif      // TMX_EXP

// The same update by a thread that walks a column (or one of `nseg` segments of it) level by level with the stencils' operands in a
// sliding register window -- the four levels k - 1 .. k + 2 of U, V, rho and, for NTR tracers at once, of the column values and
// mixing-ratio deviations -- so that every operand is loaded once (a segment re-reads the four levels around its ends), the two
// divisions of a deviation and every interface flux are evaluated once, and the level entering the window is loaded an iteration
// ahead of its first use.  No LDS, no barrier.  Statement for statement the arithmetic of k_v_tracers_explicit_tile: bit-identical (tested).
#define OPCL(op, k, off) opsl[(((op) * (L + 1)) + (k)) * TMX_OPW + ((off) + 2)]
// The operator coefficients and the 1 - eta table come from LDS: the kernel stores to the update instance inside the loop, after which
// the compiler may not read them through the scalar cache any more (the stores might alias them) -- as vector loads each one waits
// for every load in flight (s_waitcnt vmcnt(0)), the prefetched level included.
template <int NTR, bool CLOSED>
__global__ __launch_bounds__(128) void k_v_tracers_explicit_slide(KParams p, int c0, const double * __restrict__ xin, double * xup, double dt,
	double ks, const double * __restrict__ xref, int ntile, int xmode, int nseg)
{
	extern __shared__ double opsl[];      // [TMX_OP_COUNT][L + 1][TMX_OPW], then 1 - eta [2 L + 1]
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	double * etal = opsl + TMX_OP_COUNT * (L + 1) * TMX_OPW;
	{
		const int tid = threadIdx.y * 64 + threadIdx.x;
		for (int t = tid; t < TMX_OP_COUNT * (L + 1) * TMX_OPW; t += 128) opsl[t] = p.ops[t];
		if (CLOSED) for (int t = tid; t < 2 * L + 1; t += 128) etal[t] = p.eta[t];
		__syncthreads();
	}
	constexpr int MM = CLOSED ? 1 : 2;
	int bx, by;
	if (!xcd_column_tile(xmode, ntile, (nseg + 1) / 2, bx, by)) return;
	const int col = (p.quads ? p.quads[bx] : bx) * 64 + threadIdx.x;
	if (col >= p.ncol) return;
	const int sg = WAVE_UNIFORM(by * 2 + (int)threadIdx.y);      // the two wavefronts of a workgroup: two segments of the same columns
	const int seg = (L + nseg - 1) / nseg, k0 = sg * seg, k1 = min(L, k0 + seg);
	if (sg >= nseg || k0 >= k1) return;
	const double * w0 = xin + (size_t)TMX_SLAB_W(L, 0) * NS;
	const double jn = p.g2d[G2_JN * NS + col], je = p.g2d[G2_JE * NS + col];
	const MetCol mc = met_col(p, col);
	const bool ud = ks != 0.0;
	const double idt = 1.0 / dt;
	auto in = [&](int l) { return l >= 0 && l < L; };
	// deviation of the mixing ratio from the reference's (uniform diffusion only), as the staging of the tiled kernel forms it
	auto deviation = [&](bool ok, double qv, double rhv, double qrv, double rrv) -> double {
		double a_ = 0.0;
		if (ok && ud) { a_ = qv / rhv; a_ -= qrv / rrv; }
		return a_;
	};
	// window: levels k - 1 .. k + 2, at the (virtual) level k = k0 - 1 the walk starts from
	double uw[4], vw[4], rh[4], q[NTR][4], mr[NTR][4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int l = k0 - 2 + j;
		const bool ok = in(l);
		uw[j] = ok ? xin[(size_t)TMX_SLAB_U(L, l) * NS + col] : 0.0;
		vw[j] = ok ? xin[(size_t)TMX_SLAB_V(L, l) * NS + col] : 0.0;
		const double rhv = (ok && ud) ? xin[(size_t)TMX_SLAB_R(L, l) * NS + col] : 1.0, rrv = (ok && ud) ? xref[(size_t)TMX_SLAB_R(L, l) * NS + col] : 1.0;
		rh[j] = rhv;
#pragma unroll
		for (int t = 0; t < NTR; t++) {
			const double qv = ok ? xin[(size_t)TMX_SLAB_Q(L, c0 + t, l) * NS + col] : 0.0;
			const double qrv = (ok && ud) ? xref[(size_t)TMX_SLAB_Q(L, c0 + t, l) * NS + col] : 0.0;
			q[t][j] = qv; mr[t][j] = deviation(ok, qv, rhv, qrv, rrv);
		}
	}
	double xd_lo = 0.0, mf_lo[NTR], up[NTR];
#pragma unroll
	for (int t = 0; t < NTR; t++) { mf_lo[t] = 0.0; up[t] = 0.0; }
	double wm = (k0 >= 1 && k0 <= L - 1) ? w0[(size_t)k0 * NS + col] : 0.0;      // W on the interface of the current iteration
	for (int k = k0 - 1; k < k1; k++) {
		const int m = k + 1;
		// the level that enters the window after this iteration, W of the next interface, the values the next level updates
		const int ln = k + 3;
		const bool okn = in(ln) && ln <= k1 + 1;
		const double eU = okn ? xin[(size_t)TMX_SLAB_U(L, ln) * NS + col] : 0.0, eV = okn ? xin[(size_t)TMX_SLAB_V(L, ln) * NS + col] : 0.0;
		const double eR = (okn && ud) ? xin[(size_t)TMX_SLAB_R(L, ln) * NS + col] : 1.0, eRR = (okn && ud) ? xref[(size_t)TMX_SLAB_R(L, ln) * NS + col] : 1.0;
		const double wn = (m + 1 >= 1 && m + 1 <= L - 1 && m + 1 <= k1) ? w0[(size_t)(m + 1) * NS + col] : 0.0;
		double eQ[NTR], eQR[NTR], upn[NTR];
#pragma unroll
		for (int t = 0; t < NTR; t++) {
			eQ[t] = okn ? xin[(size_t)TMX_SLAB_Q(L, c0 + t, ln) * NS + col] : 0.0;
			eQR[t] = (okn && ud) ? xref[(size_t)TMX_SLAB_Q(L, c0 + t, ln) * NS + col] : 0.0;
			upn[t] = (k + 1 < k1) ? xup[(size_t)TMX_SLAB_Q(L, c0 + t, k + 1) * NS + col] : 0.0;
		}
		// xi_dot (zero at the boundaries) and, with uniform diffusion, rho on interface m
		double xd_hi = 0.0, re_hi = 0.0;
		if (m >= 1 && m <= L - 1) {
			double ue = 0.0, ve = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				const double cc = OPCL(TMX_OP_INTERP_NODE_TO_REDGE, m, off);
				ue += cc * uw[off + 2];
				ve += cc * vw[off + 2];
			}
			double e0, e1, e2;
			metric_edge<MM>(p, mc, m, col, e0, e1, e2, etal);
			xd_hi = e0 * ue + e1 * ve + e2 * wm;
		}
		if (ud && m <= L) {
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				re_hi += OPCL(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * rh[off + 2];
			}
		}
		double res[NTR];
#pragma unroll
		for (int t = 0; t < NTR; t++) {
			double qe = 0.0;
#pragma unroll
			for (int off = -2; off <= 1; off++) {
				const int l = m + off;
				if (l < 0 || l >= L) continue;
				qe += OPCL(TMX_OP_INTERP_NODE_TO_REDGE, m, off) * q[t][off + 2];
			}
			double mf_hi = (m == 0 || m == L) ? 0.0 : je * qe * xd_hi;
			if (ud && m > 0 && m < L) {
				double dq = 0.0;
#pragma unroll
				for (int off = -2; off <= 1; off++) {
					const int l = m + off;
					if (l < 0 || l >= L) continue;
					dq += OPCL(TMX_OP_DIFF_NODE_TO_REDGE, m, off) * mr[t][off + 2];
				}
				mf_hi -= ks * re_hi * dq;
			}
			if (k >= k0) {      // level k: window entries 0, 1, 2 are the levels k - 1, k, k + 1
				double dmf = 0.0;
				dmf += OPCL(TMX_OP_DIFF_REDGE_TO_NODE, k, 0) * mf_lo[t];
				dmf += OPCL(TMX_OP_DIFF_REDGE_TO_NODE, k, 1) * mf_hi;
				double F = dmf / jn;
				double aux = 0.0;
				if (k < L - 1) {
					double b = 0.0;
					b += OPCL(TMX_OP_PENALTY_LEFT, k, 0) * q[t][1];
					b += OPCL(TMX_OP_PENALTY_LEFT, k, 1) * q[t][2];
					aux += b * fabs(xd_hi);
				}
				if (k > 0) {
					double b = 0.0;
					b += OPCL(TMX_OP_PENALTY_RIGHT, k, -1) * q[t][0];
					b += OPCL(TMX_OP_PENALTY_RIGHT, k, 0) * q[t][1];
					aux += b * fabs(xd_lo);
				}
				F -= aux;
				if (F != 0.0) F /= idt;
				res[t] = up[t] - F;
			} else res[t] = 0.0;
			mf_lo[t] = mf_hi;
			// the window moves up one level
			q[t][0] = q[t][1]; q[t][1] = q[t][2]; q[t][2] = q[t][3]; q[t][3] = eQ[t];
			mr[t][0] = mr[t][1]; mr[t][1] = mr[t][2]; mr[t][2] = mr[t][3]; mr[t][3] = deviation(okn, eQ[t], eR, eQR[t], eRR);
			up[t] = upn[t];
		}
		xd_lo = xd_hi; wm = wn;
		uw[0] = uw[1]; uw[1] = uw[2]; uw[2] = uw[3]; uw[3] = eU;
		vw[0] = vw[1]; vw[1] = vw[2]; vw[2] = vw[3]; vw[3] = eV;
		rh[0] = rh[1]; rh[1] = rh[2]; rh[2] = rh[3]; rh[3] = eR;
		// The level's stores LAST, behind everything that reads a value loaded in this iteration: the stores sit in a branch, the compiler
		// cannot count them, and a loaded value first used behind them waits for everything in flight (s_waitcnt vmcnt(0)) -- i.e. for the
		// stores' own completion, once per tracer and level (the entering level's values are made to land first: the empty asm statements).
		asm volatile("" : "+v"(uw[3]), "+v"(vw[3]), "+v"(wm));
#pragma unroll
		for (int t = 0; t < NTR; t++) asm volatile("" : "+v"(q[t][3]), "+v"(mr[t][3]), "+v"(up[t]));
		if (k >= k0) {
#pragma unroll
			for (int t = 0; t < NTR; t++) xup[(size_t)TMX_SLAB_Q(L, c0 + t, k) * NS + col] = res[t];
		}
	}
}

#undef OPCL

// UpdateColumnTracers in the fully explicit vertical mode: every stored column, in place on the update instance
int tmxk_vi_tracers_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) {
	if (e->nt == 0 || p.ncol == 0) return 0;
	if (!e->opt_vt_column) {      // level-parallel / column-walking forms; option "vt_column": the one-lane-per-column LDS kernel, for A/B and tests
		const size_t lds_slide = ((size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW + 2 * p.L + 1) * sizeof(double);
#if TMX_EXP
		if (e->opt_vt_explicit_v1) {      // the form without LDS staging, for A/B and tests
			dim3 blk(64, 4), grd(NTILES(e, p), (p.L + 3) / 4);
			hipLaunchKernelGGL(k_v_tracers_explicit, grd, blk, 0, e->stream, p, e->nt, xin, xup, dt,
				e->udiff ? e->cfg.uniform_diffusion_scalar : 0.0, (const double *)e->d_ref);
		} else
#endif
		if (e->opt_vt_walk < 0 && lds_slide <= 64 * 1024) {      // a thread walks (a segment of) its column: -n = n segments per column
			const int nt_ = NTILES(e, p), xm = e->xcd_vertical;
			// default (-1000): segments in pairs (the two wavefronts of a workgroup) until the chip has two wavefronts per SIMD, at least five
			// levels each -- ne30 L40 on one GPU: 1 350 tiles x 2 (measured: 1 / 2 / 3 / 4 segments 4.27 / 4.08 / 4.28 / 4.20 ms per step of BASELINE
			// config 4's shape, the LDS-tiled kernel 4.67)
			int nseg = -e->opt_vt_walk;
			if (e->opt_vt_walk == -1000) for (nseg = 2; nt_ * nseg < 2048 && p.L / (nseg + 2) >= 5; nseg += 2) { }
			nseg = std::max(1, std::min(nseg, p.L));
			dim3 blk(64, 2), grd(xcd_column_grid(xm, nt_, (nseg + 1) / 2));
			const double ks_ = e->udiff ? e->cfg.uniform_diffusion_scalar : 0.0;
			const size_t lds = lds_slide;
			for (int c0 = 0; c0 < e->nt; c0 += 3) {
				const int ntr = std::min(3, e->nt - c0);
#define LAUNCH_SLIDE(N_) do { if (p.closed) hipLaunchKernelGGL((k_v_tracers_explicit_slide<N_, true>), grd, blk, lds, e->stream, p, c0, xin, xup, dt, ks_, (const double *)e->d_ref, nt_, xm, nseg); \
				else hipLaunchKernelGGL((k_v_tracers_explicit_slide<N_, false>), grd, blk, lds, e->stream, p, c0, xin, xup, dt, ks_, (const double *)e->d_ref, nt_, xm, nseg); } while (0)
				if (ntr == 3) LAUNCH_SLIDE(3); else if (ntr == 2) LAUNCH_SLIDE(2); else LAUNCH_SLIDE(1);
#undef LAUNCH_SLIDE
			}
		}
#if TMX_EXP
		else if (e->opt_vt_walk == 4 || e->opt_vt_walk == 5 || e->opt_vt_walk == 6 || e->opt_vt_walk == 8 || e->opt_vt_walk == 10) {      // by column segments of that many levels
			const int nt_ = NTILES(e, p), xm = e->xcd_vertical, kc = e->opt_vt_walk;
			dim3 blk(64), grd(xcd_column_grid(xm, nt_, (p.L + kc - 1) / kc));
#define LAUNCH_WALK(KC_) hipLaunchKernelGGL(k_v_tracers_explicit_walk<KC_>, grd, blk, 0, e->stream, p, e->nt, xin, xup, dt, \
				e->udiff ? e->cfg.uniform_diffusion_scalar : 0.0, (const double *)e->d_ref, nt_, xm)
			if (kc == 4) LAUNCH_WALK(4); else if (kc == 5) LAUNCH_WALK(5); else if (kc == 6) LAUNCH_WALK(6); else if (kc == 8) LAUNCH_WALK(8); else LAUNCH_WALK(10);
#undef LAUNCH_WALK
		}
#endif
		else {
			const int nt_ = NTILES(e, p), xm = e->xcd_vertical;
			dim3 blk(64, 8), grd(xcd_column_grid(xm, nt_, (p.L + 7) / 8));
			hipLaunchKernelGGL(k_v_tracers_explicit_tile, grd, blk, 0, e->stream, p, e->nt, xin, xup, dt,
				e->udiff ? e->cfg.uniform_diffusion_scalar : 0.0, (const double *)e->d_ref, nt_, xm);
		}
		return 0;
	}
	size_t lds; const int lw = vi_tracers_lanes(p.L, lds);
	if (!lw) return -1;
#define LAUNCH_VT(LW_) do { hipFuncSetAttribute((const void *)k_vi_tracers<LW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
	hipLaunchKernelGGL(k_vi_tracers<LW_>, dim3((p.ncol + LW_ - 1) / LW_), dim3(64), lds, e->stream, p, e->nt, xin, xin + (size_t)TMX_SLAB_W(p.L, 0) * p.NS, \
		(const double *)xup, xup, dt, p.ncol, (const int *)nullptr, (const int *)nullptr, e->d_flag, 1, \
		e->udiff ? e->cfg.uniform_diffusion_scalar : 0.0, (const double *)e->d_ref); } while (0)
	if (lw == 64) LAUNCH_VT(64); else LAUNCH_VT(32);
#undef LAUNCH_VT
	return 0;
}

// UpdateColumnTracers at the end of StepImplicitTermsExplicitly (VerticalDynamicsFEM.cpp:600-608): the implicit column update
// of the tracers, on every stored column (the reference loops over all nodes there, :541-542), in place on the update instance
int tmxk_vi_tracers_all(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) {
	if (e->nt == 0 || p.ncol == 0) return 0;
	if (e->opt_vt_rows) {
		const size_t l2 = vt_rows_lds(p.L);
		if (l2 <= 160 * 1024) {
			launch_vt_rows(e, p, p.ncol, xin, xin + (size_t)TMX_SLAB_W(p.L, 0) * p.NS, (const double *)xup, xup, dt, nullptr, nullptr, l2);
			return 0;
		}
	}
	size_t lds; const int lw = vi_tracers_lanes(p.L, lds);
	if (!lw) return -1;
#define LAUNCH_VT(LW_) do { hipFuncSetAttribute((const void *)k_vi_tracers<LW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
	hipLaunchKernelGGL(k_vi_tracers<LW_>, dim3((p.ncol + LW_ - 1) / LW_), dim3(64), lds, e->stream, p, e->nt, xin, xin + (size_t)TMX_SLAB_W(p.L, 0) * p.NS, \
		(const double *)xup, xup, dt, p.ncol, (const int *)nullptr, (const int *)nullptr, e->d_flag, 0, 0.0, (const double *)nullptr); } while (0)
	if (lw == 64) LAUNCH_VT(64); else LAUNCH_VT(32);
#undef LAUNCH_VT
	return 0;
}

__global__ __launch_bounds__(256) void k_v_filter_tracers(KParams p, int nt, const double * __restrict__ area, double * x) {
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = blockIdx.x * 256 + threadIdx.x;
	const int c = blockIdx.y;
	if (col >= p.ncol || c >= nt) return;
	double total = 0.0, nonneg = 0.0;
	for (int k = 0; k < L; k++) {
		const double q = x[(size_t)TMX_SLAB_Q(L, c, k) * NS + col];
		const double pm = q * area[(size_t)k * NS + col];
		total += pm;
		if (q >= 0.0) nonneg += pm;
	}
	const double r = total / nonneg;
	for (int k = 0; k < L; k++) {
		const size_t o = (size_t)TMX_SLAB_Q(L, c, k) * NS + col;
		const double q = x[o];
		x[o] = (q > 0.0) ? q * r : 0.0;
	}
}

void tmxk_v_filter_tracers(tmx_engine * e, const KParams & p, double * x) {
	if (e->nt == 0) return;
	hipLaunchKernelGGL(k_v_filter_tracers, dim3((p.ncol + 255) / 256, e->nt), dim3(256), 0, e->stream, p, e->nt, (const double *)e->d_area, x);
}

// Column physics: HeldSuarezPhysics::Perform (src/atm/HeldSuarezPhysics.cpp:60-301), Lorenz staggering.  Pointwise
// in (column, level): backward-Euler boundary-layer friction on U,V (sigma from PressureFromRhoTheta(rho * rhotheta),
// :107-126, as the reference writes it) and relaxation of rho*theta towards the equilibrium temperature with the
// "Ullrich update" (:186-247).  The surface pressure is PressureFromRhoTheta(rho_e0 * rhotheta_e0) of the tracked surface
// slots (`surf` = [2][NS]: what the reference finds in the interface entries of rho and rho*theta, :113-116), or, when the
// caller pinned it, the input of tmx_set_patch_physics_inputs.

}      // namespace lu_fma / lu_nofma
