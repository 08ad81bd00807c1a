// tmx_k_horizontal.hip -- hand-written CDNA4 (gfx950) kernels of the spectral-element hot path (one translation unit per operator family;
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

// ---------------------------------------------------------------------------------------------
// H explicit, part 1: HorizontalDynamicsFEM::StepNonhydrostaticPrimitive, updates on levels
// (src/atm/HorizontalDynamicsFEM.cpp:848-1421).  The prologue interpolations (:817-831) are
// recomputed on the fly instead of being stored.  (u x zeta)_xi is handed to part 2 via `ucz`.

// FUSE_V: also apply VerticalDynamicsFEM::StepExplicit (k_v_explicit) to the freshly updated U,V in
// registers -- same statements in the same order, so the result is bit-identical to the two calls.
// The H-only values of levels 0,1 are kept in `uv01` for the bottom boundary condition of W, which the
// reference evaluates before V.StepExplicit runs (HorizontalDynamicsFEM.cpp:1615-1638).
// MF (opt-in, TMX_MFMA=1): the twelve 4x4 contractions of a wavefront's four elements as v_mfma_f64_4x4x4_4b_f64 instead
// of LDS reads + VALU.  Lane layout of that instruction, probed on gfx950 with exact integer data (tools/probe/
// mfma_f64_probe.hip): A[m][k] of block b at lane m + 4b + 16k, B[k][n] at lane n + 4b + 16k, D[m][n] at lane n + 4b + 16m.
// With the lane -> node map  lane = j + 4 e + 16 i  (j beta node, e element, i alpha node) a field IS the B operand
// of the alpha contraction (k = alpha node), the result lands on the lane of its own node, and no LDS is touched; the
// beta contraction takes the field as A with alpha and beta exchanged, one in-register lane transpose (two
// ds_bpermute_b32) per field.  Global accesses stay one contiguous 512-byte row per wavefront (the map permutes lanes
// inside it).  The matrix unit accumulates with fused multiply-adds, the reference with separate multiplies and adds:
// results agree to rounding (1e-15), not bit for bit -- which is why this path is not the default.
// Diagnostic builds (-DTMX_H_TIMING, tools/h_timing.py): shader cycles per wavefront of k_h_explicit by phase, accumulated per
// instantiation (index = number of combination terms): [0] prologue up to the last load issued, [1] until the loaded values are
// there and the pointwise arithmetic in front of the Exner function is done, [2] Exner function + LDS writes, [3] contractions,
// [4] pointwise updates up to the W hand-over barrier, [5] barrier + W, [6] stores / seam exchange, [7] wavefronts counted.
#ifdef TMX_H_TIMING
// (one record per wavefront, plain stores: atomics on a handful of addresses serialise chip-wide and stall the very phases measured)
#define HT_WAVES 65536
__device__ unsigned long long g_h_rec[8][HT_WAVES][8];
#define HT_DECL long long ht_ = __builtin_readcyclecounter(); long long hts_[7] = { 0, 0, 0, 0, 0, 0, 0 }
#define HT_MARK(i) do { const long long t1_ = __builtin_readcyclecounter(); hts_[i] += t1_ - ht_; ht_ = t1_; } while (0)
#define HT_END(NT_) do { if (threadIdx.x == 0) { const unsigned w_ = ((blockIdx.x * gridDim.y + blockIdx.y) * KT_H + threadIdx.y) % HT_WAVES; \
	for (int q_ = 0; q_ < 7; q_++) g_h_rec[(NT_) & 7][w_][q_] = (unsigned long long)hts_[q_]; g_h_rec[(NT_) & 7][w_][7] = 1ull; } } while (0)
void tmxk_h_timing_read(unsigned long long * out) {
	static std::vector<unsigned long long> h((size_t)8 * HT_WAVES * 8);
	hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_h_rec), h.size() * sizeof(unsigned long long));
	for (int i = 0; i < 128; i++) out[i] = 0;
	for (int nt = 0; nt < 8; nt++) for (int w = 0; w < HT_WAVES; w++) if (h[((size_t)nt * HT_WAVES + w) * 8 + 7])
		for (int q = 0; q < 8; q++) out[nt * 8 + q] += h[((size_t)nt * HT_WAVES + w) * 8 + q];
	std::fill(h.begin(), h.end(), 0ull);
	hipMemcpyToSymbol(HIP_SYMBOL(g_h_rec), h.data(), h.size() * sizeof(unsigned long long));
}
#else
#define HT_DECL
#define HT_MARK(i)
#define HT_END(NT_)
void tmxk_h_timing_read(unsigned long long * out) { for (int i = 0; i < 128; i++) out[i] = 0; }
#endif
#ifndef TMX_H_BURST
#define TMX_H_BURST 0      // 1: raw terms of the stage combination loaded in bursts (lin_load) instead of slab by slab (base_value_u): one memory round trip
                           // instead of five per wavefront, but 185-190 registers (two resident wavefronts per SIMD instead of three): measured equal (element-major) or slower (node-unique)
#endif
#ifndef TMX_H_EXNER_LDS
#define TMX_H_EXNER_LDS 1  // exp / log tables of the Exner function in LDS (one more barrier) instead of two dependent gathers from global memory
#endif
#ifndef TMX_H_EM_GROUPS
#define TMX_H_EM_GROUPS 2
#endif
#ifndef TMX_H_LATE_BASE_D
#define TMX_H_LATE_BASE_D 1      // the same on the element-major layout
#endif
#ifndef TMX_H_LATE_BASE
#define TMX_H_LATE_BASE 1  // node-unique layout, stage combinations of three and more terms: the combination's raw terms are loaded and reduced at the end of
                           // the kernel, next to the stores, where almost nothing else is live -- in the prologue's burst they take the kernel over 128 registers
                           // (three resident wavefronts per SIMD instead of four)
#endif
// UQ: node-unique state layout (tmx_device.h, seam_store)
// PM: the update instance enters its own stage combination (LinTerms::premul), a compile-time parameter so that the loads of the
// combination's raw terms carry no control flow (lin_load)
// EM: also form and store the prefix of a later stage's combination from the loaded terms (EmitTerms, tmx_device.h)
// DM (with UQ): some sources are element-major instances whose copies of a node may differ (LinTerms::dmask) -- the stage reads them per
// copy, as the element-major kernel does, and still averages and stores node-unique results
template <bool FUSE_V, int NT, bool MF, bool UQ, bool PM, bool EM, bool DM = false>
#ifndef TMX_H_MINWG
#define TMX_H_MINWG 1      // ablation: minimum workgroups per CU the register allocation must allow (4 caps at 128 registers: slower, spills)
#endif
__global__ __launch_bounds__(64 * KT_H, TMX_H_MINWG) void k_h_explicit(KParams p, const double * __restrict__ xin,
	LinTerms xbase, double * xup, double * __restrict__ ucz, double * __restrict__ uv01, double dt, UvTerms uvt, EmitTerms em)
{
	__shared__ double s[KT_H][MF ? 1 : 9][TMX_UQ_ROWLEN(UQ)];
	__shared__ double sDw[MF ? 1 : KT_H][16], sSw[MF ? 1 : KT_H][16];      // every wavefront its own copy of the operator tables (no workgroup barrier before the contractions)
	__shared__ double rmtab[TMX_H_EXNER_LDS ? TMX_RMTAB_DOUBLES : 1];      // exp / log tables of the Exner function: from global memory they are two dependent gathers per wavefront
	const int lane = tile_lane<UQ>(), ty = WAVE_UNIFORM(threadIdx.y);
	const int tid = ty * 64 + lane;
	const int lp = lds_pos<UQ>(lane);      // where this lane's values sit in an LDS row
	if (!MF && lane < 16) { sDw[ty][lane] = p.dx[lane]; sSw[ty][lane] = p.stiff[lane]; }
	const double * sD = sDw[MF ? 0 : ty], * sS = sSw[MF ? 0 : ty];
	if (TMX_H_EXNER_LDS) tmx_rm_tables_to_lds(rmtab, tid, 64 * KT_H);
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	// column of this lane: element-major (e, i, j) in memory; MF: lane = j + 4 e + 16 i
	int tix, yb;
	if (!wg_tile<UQ>(p, (L + KT_H - 1) / KT_H, tix, yb)) return;
	const int tile = p.quads ? p.quads[tix] : tix;       // 64-column tile of this workgroup
	const int col = MF ? tile * 64 + ((lane >> 2) & 3) * 16 + (lane >> 4) * 4 + (lane & 3) : tile * 64 + lane;
	const int k = yb * KT_H + ty;
	// UQ: geometry, LDS and the hand-over arrays keep the thread's own column; the state is gathered from the unique slabs
	const int uc = UQ ? tile_ucol(p, tile, lane) : col;
	const size_t SS = UQ ? (size_t)p.NSS : NS;
	const bool act = (k < L) && (UQ ? (uc >= 0) : (col < p.ncol));

	double ua = 0, ub = 0, rt = 1, rh = 1, ux = 0;
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, jn = 1, j2d = 0, fcor = 0, ca2 = 0, cb2 = 0, cx2 = 0, dra = 0, drb = 0;
	double conUa = 0, conUb = 0, conUx = 0, dxua = 0, dxub = 0;
	double v_uam = 0, v_uap = 0, v_ubm = 0, v_ubp = 0, w_hi = 0, w_lo = 0;
	// base of the update (a plain instance or the fused stage combination): loaded in the same burst as the state
	// and reduced to four values at once -- loading it at the point of use made the compiler keep all 48 LDS reads
	// and the 4 x NT combination sources live together (180-200 VGPRs, occupancy 2)
	HT_DECL;
	constexpr bool LB = TMX_H_LATE_BASE && (UQ || TMX_H_LATE_BASE_D) && FUSE_V && !MF && NT >= 3 && !TMX_H_BURST;
	double bU = 0, bV = 0, bT = 0, bR = 0, bW = 0;
	constexpr int NX = NT > 0 ? NT : 1;
	double xU[NX], xV[NX], xT[NX], xR[NX], xW[NX];
#pragma unroll
	for (int m = 0; m < NX; m++) { xU[m] = 0; xV[m] = 0; xT[m] = 0; xR[m] = 0; xW[m] = 0; }
	if (act) {
		const int km = (k > 0) ? k - 1 : 0, kp = (k < L - 1) ? k + 1 : L - 1;
		const unsigned cb = (unsigned)uc * 8u, gb = (unsigned)col * 8u;      // byte offsets inside a state slab / a geometry slab
		const double * __restrict__ xuv = uvt.xin;      // U,V of the initial instance (its own slot or the one it shares)
		// (DM: the initial instance may be element-major -- the thread's own copy, slab stride NS)
		const bool xd = DM && (xbase.dmask >> 31) != 0u;
		const size_t SX = xd ? NS : SS;
		const unsigned cx = xd ? gb : cb;
		ua = ldu(xuv + TMX_SLAB_U(L, k) * SX, cx);
		ub = ldu(xuv + TMX_SLAB_V(L, k) * SX, cx);
		rt = ldu(xin + TMX_SLAB_T(L, k) * SX, cx);
		rh = ldu(xin + TMX_SLAB_R(L, k) * SX, cx);
		const double we0 = ldu(xin + TMX_SLAB_W(L, k) * SX, cx);
		const double we1 = ldu(xin + TMX_SLAB_W(L, k + 1) * SX, cx);
		const double uam = ldu(xuv + TMX_SLAB_U(L, km) * SX, cx), uap = ldu(xuv + TMX_SLAB_U(L, kp) * SX, cx);
		const double ubm = ldu(xuv + TMX_SLAB_V(L, km) * SX, cx), ubp = ldu(xuv + TMX_SLAB_V(L, kp) * SX, cx);
		// raw terms of the base of U,V (issued with the state, before anything is waited for; reduced below); those of
		// rho*theta, rho and W follow behind the barrier, when the registers of the prologue are free again, and arrive
		// while the contractions run -- all of them in one burst take the kernel from three resident wavefronts per SIMD to two
		if (TMX_H_BURST) { lin_load<NT, PM>(uvt.src, TMX_SLAB_U(L, k) * SS, cb, xU); lin_load<NT, PM>(uvt.src, TMX_SLAB_V(L, k) * SS, cb, xV); }
		else if (LB) { }
		else if (EM) {
			// every copy of a node forms the same prefix (the sources are consistent): each stores it, to the same entry
			double pf = 0.0;
			bU = base_value_u2<NT, true>(xbase, uvt.src, em, TMX_SLAB_U(L, k) * SS, cb, pf); stu(em.xp + TMX_SLAB_U(L, k) * SS, cb, pf);
			bV = base_value_u2<NT, true>(xbase, uvt.src, em, TMX_SLAB_V(L, k) * SS, cb, pf); stu(em.xp + TMX_SLAB_V(L, k) * SS, cb, pf);
			bT = base_value_u2<NT, true>(xbase, xbase.src, em, TMX_SLAB_T(L, k) * SS, cb, pf); stu(em.xp + TMX_SLAB_T(L, k) * SS, cb, pf);
			bR = base_value_u2<NT, true>(xbase, xbase.src, em, TMX_SLAB_R(L, k) * SS, cb, pf); stu(em.xp + TMX_SLAB_R(L, k) * SS, cb, pf);
			if (TMX_H_FUSE_W && ty > 0) { bW = base_value_u2<NT, true>(xbase, xbase.src, em, TMX_SLAB_W(L, k) * SS, cb, pf); stu(em.xp + TMX_SLAB_W(L, k) * SS, cb, pf); }
		} else if (DM) {
			bU = base_value_dm<NT>(xbase, uvt.src, TMX_SLAB_U(L, k), SS, NS, cb, gb); bV = base_value_dm<NT>(xbase, uvt.src, TMX_SLAB_V(L, k), SS, NS, cb, gb);
			bT = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_T(L, k), SS, NS, cb, gb); bR = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_R(L, k), SS, NS, cb, gb);
			if (TMX_H_FUSE_W && ty > 0) bW = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_W(L, k), SS, NS, cb, gb);
		} else {
			bU = base_value_uv<NT>(xbase, uvt, TMX_SLAB_U(L, k) * SS, cb); bV = base_value_uv<NT>(xbase, uvt, TMX_SLAB_V(L, k) * SS, cb);
			bT = base_value_u<NT>(xbase, TMX_SLAB_T(L, k) * SS, cb); bR = base_value_u<NT>(xbase, TMX_SLAB_R(L, k) * SS, cb);
			if (TMX_H_FUSE_W && ty > 0) bW = base_value_u<NT>(xbase, TMX_SLAB_W(L, k) * SS, cb);
		}
		c2a0 = ldu(p.g2d + G2_C2A0 * NS, gb); c2a1 = ldu(p.g2d + G2_C2A1 * NS, gb); c2b1 = ldu(p.g2d + G2_C2B1 * NS, gb);
		jn = ldu(p.g2d + G2_JN * NS, gb); j2d = ldu(p.g2d + G2_J2D * NS, gb); fcor = ldu(p.g2d + G2_F * NS, gb);
		const MetCol mc = met_col(p, col);
		metric_node(p, mc, k, col, ca2, cb2, cx2, dra, drb);
		// W on levels (InterpolateREdgeToNode, GridPatchGLL.cpp:111-143)
		ux = 0.0; ux += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * we0; ux += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * we1;
		// contravariant velocities, kinetic energy (:884-930)
		conUa = c2a0 * ua + c2a1 * ub + ca2 * ux;
		conUb = c2a1 * ua + c2b1 * ub + cb2 * ux;
		conUx = ca2 * ua + cb2 * ub + cx2 * ux;
		// vertical derivatives of the covariant velocities (DifferentiateNodeToNode, :975-982)
		const double dm = OPC(TMX_OP_DIFF_NODE_TO_NODE, k, -1), d0 = OPC(TMX_OP_DIFF_NODE_TO_NODE, k, 0), dp = OPC(TMX_OP_DIFF_NODE_TO_NODE, k, 1);
		dxua = 0.0; dxua += dm * uam; dxua += d0 * ua; dxua += dp * uap;
		dxub = 0.0; dxub += dm * ubm; dxub += d0 * ub; dxub += dp * ubp;
		if (FUSE_V) {
			// xi_dot on the interfaces k and k+1 (VerticalDynamicsFEM.cpp:816-828), zero at the boundaries
			v_uam = uam; v_uap = uap; v_ubm = ubm; v_ubp = ubp;
			double xlo = 0.0, xhi = 0.0;
			if (k >= 1) {
				double ue = 0.0, ve = 0.0;
				ue += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * uam; ue += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * ua;
				ve += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * ubm; ve += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * ub;
				double e0, e1, e2;
				metric_edge(p, mc, k, col, e0, e1, e2);
				xlo = e0 * ue + e1 * ve + e2 * we0;
			}
			if (k + 1 <= L - 1) {
				double ue = 0.0, ve = 0.0;
				ue += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, -1) * ua; ue += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, 0) * uap;
				ve += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, -1) * ub; ve += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, 0) * ubp;
				double e0, e1, e2;
				metric_edge(p, mc, k + 1, col, e0, e1, e2);
				xhi = e0 * ue + e1 * ve + e2 * we1;
			}
			w_hi = dt * fabs(xhi); w_lo = dt * fabs(xlo);
		}
	}
	HT_MARK(0);
	if (TMX_H_BURST) { bU = lin_reduce<NT, PM>(xbase, xU); bV = lin_reduce<NT, PM>(xbase, xV); }
	const double ke = 0.5 * (conUa * ua + conUb * ub + conUx * ux);
#ifdef TMX_H_TIMING
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	HT_MARK(1);
	if (TMX_H_EXNER_LDS) __syncthreads();      // the tables are in LDS (every thread's loads are in flight by now)
	const double exn = act ? (TMX_H_EXNER_LDS ? exner_from_rhotheta_lds(p, rt, rmtab) : exner_from_rhotheta(p, rt)) : 0.0;
	const double fa = jn * conUa, fb = jn * conUb;
	double daub = 0, daux = 0, dbua = 0, dbux = 0, dap = 0, dbp = 0, dake = 0, dbke = 0;
	double darf = 0, dapf = 0, dbrf = 0, dbpf = 0;
	int sdst = 0, sred = 0;
	if (MF) {
		// operator entries of this lane: D[k][m] and S[m][k] with m = lane & 3, k = lane >> 4 (A of the alpha contraction
		// d/d_alpha f (i,j) = sum_t D[t][i] f(t,j), B of the beta contraction d/d_beta f (i,j) = sum_t f(i,t) D[t][j])
		const int m_ = lane & 3, k_ = lane >> 4;
		const double cD = p.dx[k_ * 4 + m_], cS = p.stiff[m_ * 4 + k_];
		// alpha <-> beta exchange inside the element: value of lane (lane >> 4) + 4 e + 16 (lane & 3)
		const int src = (k_ + (lane & 12) + 16 * m_) * 4;
		auto tr = [&](double v) -> double {
			const long long b = __double_as_longlong(v);
			const int lo = __builtin_amdgcn_ds_bpermute(src, (int)(b & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(src, (int)(b >> 32));
			return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
		};
#define MFMA4(A_, B_) __builtin_amdgcn_mfma_f64_4x4x4f64((A_), (B_), 0.0, 0, 0, 0)
		daub = MFMA4(cD, ub); daux = MFMA4(cD, ux); dap = MFMA4(cD, exn); dake = MFMA4(cD, ke);
		darf = -MFMA4(cS, fa * rh); dapf = -MFMA4(cS, fa * rt);
		dbua = MFMA4(tr(ua), cD); dbux = MFMA4(tr(ux), cD); dbp = MFMA4(tr(exn), cD); dbke = MFMA4(tr(ke), cD);
		dbrf = -MFMA4(tr(fb * rh), cS); dbpf = -MFMA4(tr(fb * rt), cS);
#undef MFMA4
		if (!act) return;
		if (TMX_H_BURST) {
			const unsigned cb2 = (unsigned)uc * 8u;
			lin_load<NT, PM>(xbase.src, TMX_SLAB_T(L, k) * SS, cb2, xT); lin_load<NT, PM>(xbase.src, TMX_SLAB_R(L, k) * SS, cb2, xR);
		}
	} else {
	s[ty][0][lp] = ua;  s[ty][1][lp] = ub;  s[ty][2][lp] = ux;
	s[ty][3][lp] = exn; s[ty][4][lp] = ke;
	s[ty][5][lp] = fa * rh; s[ty][6][lp] = fa * rt;
	s[ty][7][lp] = fb * rh; s[ty][8][lp] = fb * rt;
	wave_sync();      // the rows are this wavefront's own
	HT_MARK(2);
	if (!act) return;
#ifdef TMX_H_SCHED_BARRIER
	__builtin_amdgcn_sched_barrier(0);
#endif
	// what the thread does with its results (seam_store): loaded behind the barrier, when the registers of the prologue are free
	// again (in front of it they cost the first stage's kernel its fourth wavefront per SIMD), used at the end
	if (UQ) { sdst = p.t_sdst[col]; sred = p.t_sred[col]; }
	if (TMX_H_BURST) {
		const unsigned cb2 = (unsigned)uc * 8u;
		lin_load<NT, PM>(xbase.src, TMX_SLAB_T(L, k) * SS, cb2, xT); lin_load<NT, PM>(xbase.src, TMX_SLAB_R(L, k) * SS, cb2, xR);
		if (TMX_H_FUSE_W && ty > 0) lin_load<NT, PM>(xbase.src, TMX_SLAB_W(L, k) * SS, cb2, xW);
	}

	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lds_pos<UQ>(lane & 48);
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j;        // node (t, j)
		const double D = sD[t * 4 + i], S = sS[i * 4 + t];
		daub += s[ty][1][la] * D;
		daux += s[ty][2][la] * D;
		dap  += s[ty][3][la] * D;
		dake += s[ty][4][la] * D;
		darf -= s[ty][5][la] * S;
		dapf -= s[ty][6][la] * S;
	}
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int lb = eb + 4 * i + t;        // node (i, t)
		const double D = sD[t * 4 + j], S = sS[j * 4 + t];
		dbua += s[ty][0][lb] * D;
		dbux += s[ty][2][lb] * D;
		dbp  += s[ty][3][lb] * D;
		dbke += s[ty][4][lb] * D;
		dbrf -= s[ty][7][lb] * S;
		dbpf -= s[ty][8][lb] * S;
	}
	}
	HT_MARK(3);
	const double ida = p.g2d[G2_IDA * NS + col], idb = p.g2d[G2_IDB * NS + col];      // 1 / element spacing of the column's patch
	daub *= ida; daux *= ida; dbua *= idb; dbux *= idb;
	darf *= ida; dbrf *= idb; dapf *= ida; dbpf *= idb;
	dap *= ida; dbp *= idb; dake *= ida; dbke *= idb;

	// u x zeta (:1011-1037)
	const double jza = dbux - dxub, jzb = dxua - daux, jzx = daub - dbua;
	const double ucza = conUb * jzx - conUx * jzb;
	const double uczb = conUx * jza - conUa * jzx;
	const double uczx = -conUa * daux - conUb * dbux;

	// pointwise updates (:1321-1421)
	const double invJ = 1.0 / jn;
	double upa = 0.0, upb = 0.0;
	upa += ucza; upb += uczb;
	upa += fcor * j2d * conUb;
	upb -= fcor * j2d * conUa;
	const double pgfa = dap * rt / rh, pgfb = dbp * rt / rh;
	const double daphi = p.grav * dra, dbphi = p.grav * drb;
	upa -= (pgfa + dake + daphi);
	upb -= (pgfb + dbke + dbphi);
	const size_t sU = TMX_SLAB_U(L, k) * NS, sV = TMX_SLAB_V(L, k) * NS, sT = TMX_SLAB_T(L, k) * NS, sR = TMX_SLAB_R(L, k) * NS;
	const unsigned colb = (unsigned)col * 8u;
	// out = base + dt * rhs; base == out for the reference's in-place accumulation, base == initial
	// instance when the preceding CopyData(initial -> update) is fused away
	// (LB: the increments are kept apart and added to the base, in this order, at the end of the kernel)
	const double tU = dt * upa, tV = dt * upb;
	double outU = bU + tU;
	double outV = bV + tV;
	double tUhi = 0, tVhi = 0, tUlo = 0, tVlo = 0;
	if (FUSE_V) {
		if (!LB && k < 2) { uv01[(size_t)k * NS + col] = outU; uv01[(size_t)(2 + k) * NS + col] = outV; }
		// first-order upwind penalty on U,V (VerticalDynamicsFEM.cpp:984-1023): left then right operator
		if (k < L - 1) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * ua; b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * v_uap;
			tUhi = b * w_hi; outU += tUhi;
			b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * ub; b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * v_ubp;
			tVhi = b * w_hi; outV += tVhi;
		}
		if (k > 0) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * v_uam; b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * ua;
			tUlo = b * w_lo; outU += tUlo;
			b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * v_ubm; b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * ub;
			tVlo = b * w_lo; outV += tVlo;
		}
	}
	if (TMX_H_BURST) {
		bT = lin_reduce<NT, PM>(xbase, xT); bR = lin_reduce<NT, PM>(xbase, xR);
		if (TMX_H_FUSE_W && ty > 0) bW = lin_reduce<NT, PM>(xbase, xW);
	}
	const double tR = dt * invJ * (darf + dbrf), tT = dt * invJ * (dapf + dbpf);
	double outR = bR - tR, outT = bT - tT;
	if (!UQ && !LB) {
		stu(xup + sU, colb, outU);
		stu(xup + sV, colb, outV);
		stu(xup + sR, colb, outR);
		stu(xup + sT, colb, outT);
	}
	double outW = 0.0, tW = 0.0;
	if (TMX_H_FUSE_W) {
		// (MF: the matrix-unit variant keeps one 512-byte row per wavefront for exactly this hand-over; round 3 measured it without,
		// i.e. with every interface left to k_h_w_update)
		// W on the interfaces inside the tile (part 2 below, :1612-1659): interface k takes (u x zeta)_xi of the levels k - 1 and k,
		// which wavefront ty - 1 hands over through its own (now free) LDS row; only the tile's first interface is left to
		// k_h_w_update, which therefore needs the hand-off array for the first and the last level of a tile only
		s[ty][0][lp] = uczx;
		if (ty == 0 || ty == KT_H - 1) stu(ucz + (size_t)k * NS, colb, uczx);
		HT_MARK(4);
		__syncthreads();       // wavefronts above the top level have left; the barrier counts the ones still running
		HT_MARK(5);
		if (ty > 0) {
			double v = 0.0;
			v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * s[ty - 1][0][lp];
			v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * uczx;
			tW = dt * v;
			outW = bW + tW;
			if (!UQ && !LB) stu(xup + TMX_SLAB_W(L, k) * NS, colb, outW);
		}
	} else
		stu(ucz + (size_t)k * NS, colb, uczx);
	if (LB) {
		// the stage combination, now: NT loads per slab in flight beside five increments (and, EM, the prefix of a later stage's
		// combination formed from the same terms: every copy of a node forms the same one and stores it, to the same entry)
#if TMX_H_LATE_BASE == 2
		__builtin_amdgcn_sched_barrier(0);
#endif
		unsigned cb = (unsigned)uc * 8u;
#if TMX_H_LATE_BASE == 1
		asm volatile("" : "+v"(cb) : "v"(tR), "v"(tT), "v"(outU), "v"(outV));      // (the addresses exist only now: the loads cannot be moved up into the contractions)
#endif
		const bool withW = TMX_H_FUSE_W && ty > 0;
		double pU = 0, pV = 0, pT = 0, pR = 0, pW = 0;
		if (DM) {
			unsigned gbl = (unsigned)col * 8u;
			asm volatile("" : "+v"(gbl) : "v"(cb));
			bU = base_value_dm<NT>(xbase, uvt.src, TMX_SLAB_U(L, k), SS, NS, cb, gbl); bV = base_value_dm<NT>(xbase, uvt.src, TMX_SLAB_V(L, k), SS, NS, cb, gbl);
			bT = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_T(L, k), SS, NS, cb, gbl); bR = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_R(L, k), SS, NS, cb, gbl);
			if (withW) bW = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_W(L, k), SS, NS, cb, gbl);
		} else {
		bU = base_value_u2<NT, EM>(xbase, uvt.src, em, TMX_SLAB_U(L, k) * SS, cb, pU);
		bV = base_value_u2<NT, EM>(xbase, uvt.src, em, TMX_SLAB_V(L, k) * SS, cb, pV);
#if TMX_H_EM_GROUPS >= 2
		// EM: both sums need a slab's terms, and all five slabs' terms in flight at once cost the fourth resident wavefront again: two (three) rounds
		if (EM) asm volatile("" : "+v"(cb) : "v"(bU), "v"(bV), "v"(pU), "v"(pV));
#endif
		bT = base_value_u2<NT, EM>(xbase, xbase.src, em, TMX_SLAB_T(L, k) * SS, cb, pT);
#if TMX_H_EM_GROUPS >= 3
		if (EM) asm volatile("" : "+v"(cb) : "v"(bT), "v"(pT));
#endif
		bR = base_value_u2<NT, EM>(xbase, xbase.src, em, TMX_SLAB_R(L, k) * SS, cb, pR);
		if (withW) bW = base_value_u2<NT, EM>(xbase, xbase.src, em, TMX_SLAB_W(L, k) * SS, cb, pW);
		}
		if (EM) {
			stu(em.xp + TMX_SLAB_U(L, k) * SS, cb, pU); stu(em.xp + TMX_SLAB_V(L, k) * SS, cb, pV);
			stu(em.xp + TMX_SLAB_T(L, k) * SS, cb, pT); stu(em.xp + TMX_SLAB_R(L, k) * SS, cb, pR);
			if (withW) stu(em.xp + TMX_SLAB_W(L, k) * SS, cb, pW);
		}
		outU = bU + tU; outV = bV + tV;
		if (k < 2) { uv01[(size_t)k * NS + col] = outU; uv01[(size_t)(2 + k) * NS + col] = outV; }
		if (k < L - 1) { outU += tUhi; outV += tVhi; }
		if (k > 0) { outU += tUlo; outV += tVlo; }
		outR = bR - tR; outT = bT - tT;
		if (withW) outW = bW + tW;
		if (!UQ) {
			stu(xup + sU, colb, outU); stu(xup + sV, colb, outV); stu(xup + sR, colb, outR); stu(xup + sT, colb, outT);
			if (withW) stu(xup + TMX_SLAB_W(L, k) * NS, colb, outW);
		}
	}
	if (UQ && !MF) {
		// rows 1..5 of the wavefront's LDS block (row 0 is being read by the wavefront above); W only where this kernel updates it
		const double v[5] = { outU, outV, outR, outT, outW };
		const int slab[5] = { TMX_SLAB_U(L, k), TMX_SLAB_V(L, k), TMX_SLAB_R(L, k), TMX_SLAB_T(L, k), TMX_SLAB_W(L, k) };
		seam_store<5, TMX_UQ_ROWLEN(UQ)>(p, &s[ty][1], lane, sdst, sred, v, xup, slab, (TMX_H_FUSE_W && ty > 0) ? 5 : 4);
	}
#ifdef TMX_H_TIMING
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	HT_MARK(6);
	HT_END(NT);
}

// H explicit, part 2: W on interfaces (src/atm/HorizontalDynamicsFEM.cpp:1612-1659): bottom
// boundary from the UPDATED U,V extrapolated to the surface, interior += dt * interp((u x zeta)_xi).
// stride > 1: only the interfaces k = 0, stride, 2 stride, ... (the first of every level tile of k_h_explicit, which has done
// the others) and the top one
template <int NT, bool UQ, bool EM, bool DM = false>
__global__ __launch_bounds__(256) void k_h_w_update(KParams p, LinTerms xbase, double * xup,
	const double * __restrict__ ucz, const double * __restrict__ uv01, double dt, int stride, EmitTerms em)
{
	__shared__ double sw[UQ ? 4 : 1][1][64];
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int ty = WAVE_UNIFORM(threadIdx.y);
	int tix, yb;
	const int nfirst = (L + stride - 1) / stride;          // interfaces idx * stride < L, then the top one
	if (!wg_tile<UQ>(p, (nfirst + 1 + 3) / 4, tix, yb)) return;
	const int lane = tile_lane<UQ>();
	const int tile = p.quads ? p.quads[tix] : tix;
	const int col = tile * 64 + lane;
	const int idx = yb * 4 + ty;
	const int k = (idx < nfirst) ? idx * stride : ((idx == nfirst) ? L : L + 1);
	const int uc = UQ ? tile_ucol(p, tile, lane) : col;    // the node's entry in the state slabs (UQ: unique slabs, stride NSS)
	const size_t SS = UQ ? (size_t)p.NSS : NS;
	if (k > L || (UQ ? uc < 0 : col >= p.ncol)) return;
	double wnew;
	if (k == L && !UQ) {      // the top interface is not touched by the H step; carry it over when base != out
		if (NT != 0 || xbase.src[0] != xup) xup[TMX_SLAB_W(L, L) * SS + uc] = base_value<NT>(xbase, TMX_SLAB_W(L, L) * SS + uc);
		return;
	}
	if (k == L) {
		// UQ: the base's value goes through seam_store like every other result of the stage, so that the partial slots of the nodes on tile
		// and patch edges hold it when the DSS kernel averages them (round 5: it used to be stored to the node directly, and the DSS then
		// averaged whatever the slots of this slab held -- zeros, which is what W on the rigid lid always is; a state with W != 0 there
		// would have been changed).  DM: the copies may carry different base values (element-major sources): the node gets their average.
		if (DM) wnew = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_W(L, L), SS, NS, (unsigned)uc * 8u, (unsigned)col * 8u);
		else if (EM) {
			double pf = 0.0;
			wnew = base_value_u2<NT, true>(xbase, xbase.src, em, TMX_SLAB_W(L, L) * SS, (unsigned)uc * 8u, pf);
			em.xp[TMX_SLAB_W(L, L) * SS + uc] = pf;
		} else wnew = base_value<NT>(xbase, TMX_SLAB_W(L, L) * SS + uc);
	} else if (k == 0) {
		double u0 = 0.0, v0 = 0.0;
		// updated U,V of levels 0,1 as the H step left them (uv01 when V.StepExplicit was fused in)
		const double * pu0 = uv01 ? uv01 + col : xup + TMX_SLAB_U(L, 0) * NS + col;
		const double * pu1 = uv01 ? uv01 + NS + col : xup + TMX_SLAB_U(L, 1) * NS + col;
		const double * pv0 = uv01 ? uv01 + 2 * NS + col : xup + TMX_SLAB_V(L, 0) * NS + col;
		const double * pv1 = uv01 ? uv01 + 3 * NS + col : xup + TMX_SLAB_V(L, 1) * NS + col;
		u0 += OPC(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * (*pu0);
		u0 += OPC(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * (*pu1);
		v0 += OPC(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * (*pv0);
		v0 += OPC(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * (*pv1);
		double cx0, cx1, cx2;
		metric_edge(p, met_col(p, col), 0, col, cx0, cx1, cx2);
		wnew = -(cx0 * u0 + cx1 * v0) / cx2;
	} else {
		double v = 0.0;
		v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * ucz[(size_t)(k - 1) * NS + col];
		v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * ucz[(size_t)k * NS + col];
		if (EM) {
			double pf = 0.0;
			wnew = base_value_u2<NT, true>(xbase, xbase.src, em, TMX_SLAB_W(L, k) * SS, (unsigned)uc * 8u, pf) + dt * v;
			em.xp[TMX_SLAB_W(L, k) * SS + uc] = pf;
		} else if (DM)
			wnew = base_value_dm<NT>(xbase, xbase.src, TMX_SLAB_W(L, k), SS, NS, (unsigned)uc * 8u, (unsigned)col * 8u) + dt * v;
		else
		wnew = base_value<NT>(xbase, TMX_SLAB_W(L, k) * SS + uc) + dt * v;
	}
	if (UQ) {
		const double v1[1] = { wnew };
		const int slab[1] = { TMX_SLAB_W(L, k) };
		seam_store<1, 64>(p, &sw[ty][0], lane, p.t_sdst[col], p.t_sred[col], v1, xup, slab, 1);
	} else
		xup[TMX_SLAB_W(L, k) * NS + col] = wnew;
}

// xin_uv / xbase_uv / src_uv: where the U,V slabs of the initial instance, the base instance and the combination sources
// live when that is not the instance's own slot (NULL: the same pointers)
void tmxk_h_explicit(tmx_engine * e, const KParams & p, const double * xin, const double * xbase_ptr, double * xup, double dt, int fuse_v,
	int nterms, const double * const * src, const double * coef, int premul, const double * xin_uv, const double * xbase_uv, const double * const * src_uv,
	const EmitTerms * emit, unsigned dmask) {
	// the column-segment walk (tmx_k_hwalk.hip) does the whole stage, W of every interface included (node-unique layout: with the vertical part fused in)
	if (tmxk_h_walk_ok(e, p, fuse_v, nterms, emit && emit->xp)) {
		tmxk_h_walk(e, p, xin, xbase_ptr, xup, dt, fuse_v, nterms, src, coef, premul, xin_uv, xbase_uv, src_uv, dmask, emit);
		return;
	}
	// base of the update: a plain instance (nterms == 0) or a linear combination evaluated in the kernels
	LinTerms xbase;
	UvTerms uvt;
	xbase.n = nterms; xbase.premul = premul; xbase.dmask = dmask;
	xbase.src[0] = xbase_ptr; xbase.coef[0] = 0.0;
	uvt.xin = xin_uv ? xin_uv : xin; uvt.src[0] = xbase_uv ? xbase_uv : xbase_ptr;
	for (int m = 0; m < nterms; m++) { xbase.src[m] = src[m]; xbase.coef[m] = coef[m]; uvt.src[m] = src_uv ? src_uv[m] : src[m]; }
	for (int m = (nterms > 0 ? nterms : 1); m < 12; m++) { xbase.src[m] = xbase.src[0]; xbase.coef[m] = 0.0; uvt.src[m] = uvt.src[0]; }
	const int ntile = p.quads ? e->launch_tiles : p.NS / 64;
	const bool uq = p.t_ucol != nullptr;      // node-unique layout (tmxu_params): 1-D launches, XCD-aware tile order
	if (uq) e->u.slots_by_blocks = false;      // (wavefront-level seam exchange: the DSS kernel runs its full list)
	// copy-by-copy reads of element-major sources exist in the node-unique kernels with the vertical part fused in only
	if (dmask && !(uq && fuse_v)) { tmx_set_error("internal: explicit stage with element-major sources outside the fused node-unique kernels"); hipMemsetAsync(e->d_flag, 0xff, sizeof(int), e->stream); return; }
	KParams q = p; q.u_ntiles = ntile;
	const int wstride = TMX_H_FUSE_W ? KT_H : 1;
	const int nyb = (p.L + KT_H - 1) / KT_H, nyb2 = ((p.L + wstride - 1) / wstride + 1 + 3) / 4;
	dim3 blk(64, KT_H), grd = wg_grid(uq, q, ntile, nyb);
	double * uv01 = e->d_scratch + (size_t)p.L * p.NS;     // 4 slabs behind the (u x zeta)_xi hand-off
	dim3 blk2(64, 4), grd2 = wg_grid(uq, q, ntile, nyb2);
	const double * uvp = fuse_v ? (const double *)uv01 : (const double *)nullptr;
	// PM_: does the update instance enter its own combination?  (no combination: irrelevant, one instantiation)
	EmitTerms em0; em0.xp = nullptr; em0.first = -1; for (int m = 0; m < 12; m++) em0.coef[m] = 0.0;
	// (the prefix of a later stage's combination: node-unique layout with the vertical part fused in, no own coefficient, at least two terms)
	const bool emit_ok = TMX_EXP && emit && emit->xp && uq && fuse_v && !premul && nterms >= 3 && nterms <= 8;
	const EmitTerms em = emit_ok ? *emit : em0;
#define LAUNCH_HV2(NT_, PM_) do { \
		if (TMX_EXP && uq && fuse_v && emit_ok && NT_ >= 3 && NT_ <= 8 && !PM_) hipLaunchKernelGGL((k_h_explicit<true, (NT_ >= 3 && NT_ <= 8) ? NT_ : 3, false, true, false, TMX_EXP != 0>), grd, blk, 0, e->stream, q, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em); \
		else if (uq && fuse_v && dmask) hipLaunchKernelGGL((k_h_explicit<true, NT_, false, true, PM_, false, true>), grd, blk, 0, e->stream, q, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em0); \
		else if (uq && fuse_v) hipLaunchKernelGGL((k_h_explicit<true, NT_, false, true, PM_, false>), grd, blk, 0, e->stream, q, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em0); \
		else if (uq) hipLaunchKernelGGL((k_h_explicit<false, NT_, false, true, PM_, false>), grd, blk, 0, e->stream, q, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em0); \
		else if (fuse_v && e->use_mfma) hipLaunchKernelGGL((k_h_explicit<true, NT_, true, false, PM_, false>), grd, blk, 0, e->stream, p, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em0); \
		else if (fuse_v) hipLaunchKernelGGL((k_h_explicit<true, NT_, false, false, PM_, false>), grd, blk, 0, e->stream, p, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em0); \
		else hipLaunchKernelGGL((k_h_explicit<false, NT_, false, false, PM_, false>), grd, blk, 0, e->stream, p, xin, xbase, xup, e->d_scratch, uv01, dt, uvt, em0); } while (0)
#define LAUNCH_HV(NT_) do { \
		if (NT_ == 0 || premul) LAUNCH_HV2(NT_, true); else LAUNCH_HV2(NT_, false); \
		if (TMX_EXP && uq && emit_ok && NT_ >= 3 && NT_ <= 8) hipLaunchKernelGGL((k_h_w_update<(NT_ >= 3 && NT_ <= 8) ? NT_ : 3, true, TMX_EXP != 0>), grd2, blk2, 0, e->stream, q, xbase, xup, (const double *)e->d_scratch, uvp, dt, wstride, em); \
		else if (uq && dmask) hipLaunchKernelGGL((k_h_w_update<NT_, true, false, true>), grd2, blk2, 0, e->stream, q, xbase, xup, (const double *)e->d_scratch, uvp, dt, wstride, em0); \
		else if (uq) hipLaunchKernelGGL((k_h_w_update<NT_, true, false>), grd2, blk2, 0, e->stream, q, xbase, xup, (const double *)e->d_scratch, uvp, dt, wstride, em0); \
		else hipLaunchKernelGGL((k_h_w_update<NT_, false, false>), grd2, blk2, 0, e->stream, p, xbase, xup, (const double *)e->d_scratch, uvp, dt, wstride, em0); } while (0)
	switch (nterms) {
		case 0: LAUNCH_HV(0); break;
		case 2: LAUNCH_HV(2); break;
		case 3: LAUNCH_HV(3); break;
		case 4: LAUNCH_HV(4); break;
		case 5: LAUNCH_HV(5); break;
		case 6: LAUNCH_HV(6); break;
		case 7: LAUNCH_HV(7); break;
		case 8: LAUNCH_HV(8); break;
		default: LAUNCH_HV(12); break;      // entries beyond n carry coefficient 0 and a valid pointer (see below)
	}
#undef LAUNCH_HV
#undef LAUNCH_HV2
}

// HorizontalDynamicsFEM::ApplyRayleighFriction (src/atm/HorizontalDynamicsFEM.cpp:2420-2570): where the strength
// is non-zero, ten sub-cycled backward-Euler relaxations x <- f x + (1 - f) x_ref, f = 1 / (1 + dt nu / 10), of
// U, V, rho*theta (strength on levels) and W (strength on interfaces); rho is left alone (:2455-2461).
// FROM_U: the instance comes in node-unique form (xs, slab stride NSS, node of a stored column = ucd[col]) and leaves element-major (x): the
// strength and the reference state are per stored copy, so the copies of a seam node may end different (the reference applies the relaxation
// per copy); slabs and points the relaxation leaves alone are copied across -- the kernel's load is the conversion (cf. k_held_suarez<FROM_U>)
template <bool FROM_U>
__global__ __launch_bounds__(256) void k_rayleigh(int L, int NS_, int ncol, double * __restrict__ x,
	const double * __restrict__ nu, const double * __restrict__ ref, double dt, const double * __restrict__ xs, size_t NSS, const int * __restrict__ ucd)
{
	const size_t NS = (size_t)NS_;
	const int col = blockIdx.x * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (k > L || col >= ncol) return;
	const double * __restrict__ xr = FROM_U ? xs : x;
	const size_t SR = FROM_U ? NSS : NS;
	const size_t sc = FROM_U ? (size_t)ucd[col] : (size_t)col;
	const double factor = 1.0 / 10;
	if (k < L) {
		const double dnu = nu[(size_t)k * NS + col];
		if (FROM_U) x[(size_t)TMX_SLAB_R(L, k) * NS + col] = xr[(size_t)TMX_SLAB_R(L, k) * SR + sc];      // rho is left alone (:2455-2461)
		if (dnu != 0.0 || FROM_U) {
#pragma unroll
			for (int v = 0; v < 3; v++) {
				const size_t o = (size_t)(v * L + k) * NS + col;        // U, V, rho*theta slabs are [0, 3L)
				double xv = xr[(size_t)(v * L + k) * SR + sc];
				if (dnu != 0.0) {
					const double r = ref[o];
					for (int si = 0; si < 10; si++) {
						const double f = 1.0 / (1.0 + factor * dt * dnu);
						xv = f * xv + (1.0 - f) * r;
					}
				}
				x[o] = xv;
			}
		}
	}
	const double dnu = nu[(size_t)(L + k) * NS + col];
	if (dnu != 0.0 || FROM_U) {
		const size_t o = (size_t)TMX_SLAB_W(L, k) * NS + col;
		double xv = xr[(size_t)TMX_SLAB_W(L, k) * SR + sc];
		if (dnu != 0.0) {
			const double r = ref[(size_t)(3 * L + k) * NS + col];
			for (int si = 0; si < 10; si++) {
				const double f = 1.0 / (1.0 + factor * dt * dnu);
				xv = f * xv + (1.0 - f) * r;
			}
		}
		x[o] = xv;
	}
}

// xs != null: read node-unique (stride NSS, ucd = node of every stored column), write element-major to x
void tmxk_rayleigh(tmx_engine * e, const KParams & p, double * x, double dt, const double * xs, size_t NSS, const int * ucd) {
	dim3 blk(64, 4), grd(p.NS / 64, (p.L + 1 + 3) / 4);
	if (xs) hipLaunchKernelGGL(k_rayleigh<true>, grd, blk, 0, e->stream, p.L, p.NS, p.ncol, x, (const double *)e->d_ray_nu, (const double *)e->d_ray_ref, dt, xs, NSS, ucd);
	else hipLaunchKernelGGL(k_rayleigh<false>, grd, blk, 0, e->stream, p.L, p.NS, p.ncol, x, (const double *)e->d_ray_nu, (const double *)e->d_ray_ref, dt, (const double *)nullptr, (size_t)0, (const int *)nullptr);
}

// ---------------------------------------------------------------------------------------------
// Tracer transport.  Tracer densities are node scalars stored as extra slabs
// TMX_SLAB_Q(L, c, k) behind the state slabs, so copies, linear combinations, packing and the exchange treat
// them like any other slab.  The kernels below are the tracer parts of the reference's state operators.

// H explicit, tracer part (HorizontalDynamicsFEM.cpp:1090-1100, :1531-1553) + FilterNegativeTracers (:213-317):
// rho q <- base - dt / J * (weak divergence of J u^a rho q, J u^b rho q) with the mass-flux base of the state
// update (same contravariant velocities, recomputed here), then per element and level: clip negatives, rescale the
// positive values so that the element mass is unchanged.
// UD: uniform diffusion of the tracer mixing ratio q / rho in the fluxes (HorizontalDynamicsFEM.cpp:1105-1148), ks = K_scalar.
// NT > 0: the base is the stage's linear combination of the tracer slabs of NT instances, evaluated here as in the state
// kernels (base_value) instead of by a k_lincomb pass over the tracer slabs in front of this kernel.
template <bool UD, int NT>
__global__ __launch_bounds__(256) void k_h_tracers(KParams p, int nt, const double * __restrict__ area,
	const double * __restrict__ xin, LinTerms xbase, double * xup, double dt, double ks, const double * __restrict__ xuv)      // xuv: slot holding xin's U,V slabs
{
	__shared__ double s[4][5][64];
	__shared__ double sS[16], sD[16];
	const int lane = threadIdx.x, ty = WAVE_UNIFORM(threadIdx.y);
	if (ty * 64 + lane < 16) { sS[ty * 64 + lane] = p.stiff[ty * 64 + lane]; sD[ty * 64 + lane] = p.dx[ty * 64 + lane]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = TILE_X(p) * 64 + lane;
	const double ida = p.g2d[G2_IDA * (size_t)p.NS + col], idb = p.g2d[G2_IDB * (size_t)p.NS + col];      // 1 / element spacing of the column's patch
	const int k = blockIdx.y * 4 + ty;
	const bool act = (k < L) && (col < p.ncol);
	const int kc = (k < L) ? k : L - 1;
	const unsigned cb = (unsigned)col * 8u;
	double fa = 0.0, fb = 0.0, invJ = 0.0, ar = 0.0;
	double dc2a0 = 0.0, dc2a1 = 0.0, dc2b1 = 0.0, djn = 0.0, rho = 1.0;      // uniform diffusion only
	if (act) {
		const double ua = ldu(xuv + TMX_SLAB_U(L, kc) * NS, cb), ub = ldu(xuv + TMX_SLAB_V(L, kc) * NS, cb);
		const double we0 = ldu(xin + TMX_SLAB_W(L, kc) * NS, cb), we1 = ldu(xin + TMX_SLAB_W(L, kc + 1) * NS, cb);
		const double c2a0 = ldu(p.g2d + G2_C2A0 * NS, cb), c2a1 = ldu(p.g2d + G2_C2A1 * NS, cb), c2b1 = ldu(p.g2d + G2_C2B1 * NS, cb);
		const double jn = ldu(p.g2d + G2_JN * NS, cb);
		double ca2, cb2, cx2;
		metric_node3(p, met_col(p, col), kc, col, ca2, cb2, cx2);
		double ux = 0.0; ux += OPC(TMX_OP_INTERP_REDGE_TO_NODE, kc, 0) * we0; ux += OPC(TMX_OP_INTERP_REDGE_TO_NODE, kc, 1) * we1;
		const double conUa = c2a0 * ua + c2a1 * ub + ca2 * ux;
		const double conUb = c2a1 * ua + c2b1 * ub + cb2 * ux;
		fa = jn * conUa; fb = jn * conUb; invJ = 1.0 / jn;
		ar = ldu(area + (size_t)kc * NS, cb);
		if (UD) { dc2a0 = c2a0; dc2a1 = c2a1; dc2b1 = c2b1; djn = jn; rho = ldu(xin + TMX_SLAB_R(L, kc) * NS, cb); }
	}
	s[ty][3][lane] = ar;
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;
	for (int c = 0; c < nt; c++) {
		const size_t so = (size_t)TMX_SLAB_Q(L, c, kc) * NS;
		const double qv = act ? ldu(xin + so, cb) : 0.0;
		const double bv = act ? base_value_u<NT>(xbase, so, cb) : 0.0;
		double taf = fa * qv, tbf = fb * qv;
		if (UD) {
			__syncthreads();
			s[ty][4][lane] = qv / rho;
			__syncthreads();
			double daq = 0.0, dbq = 0.0;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				daq += s[ty][4][eb + 4 * t + j] * sD[t * 4 + i];
				dbq += s[ty][4][eb + 4 * i + t] * sD[t * 4 + j];
			}
			daq *= ida; dbq *= idb;
			const double cdaq = dc2a0 * daq + dc2a1 * dbq;
			const double cdbq = dc2a1 * daq + dc2b1 * dbq;
			taf -= ks * djn * rho * cdaq;
			tbf -= ks * djn * rho * cdbq;
		}
		__syncthreads();
		s[ty][0][lane] = taf; s[ty][1][lane] = tbf;
		__syncthreads();
		double da = 0.0, db = 0.0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			da -= s[ty][0][eb + 4 * t + j] * sS[i * 4 + t];
			db -= s[ty][1][eb + 4 * i + t] * sS[j * 4 + t];
		}
		da *= ida; db *= idb;
		double out = bv - dt * invJ * (da + db);
		s[ty][2][lane] = out;
		__syncthreads();
		double total = 0.0, nonneg = 0.0;
#pragma unroll
		for (int n = 0; n < 16; n++) {
			const double qq = s[ty][2][eb + n];
			const double pm = qq * s[ty][3][eb + n];
			total += pm;
			if (qq >= 0.0) nonneg += pm;
		}
		const double r = total / nonneg;
		out = (out > 0.0) ? out * r : 0.0;
		if (act) stu(xup + so, cb, out);
	}
}

void tmxk_h_tracers(tmx_engine * e, const KParams & p, const double * xin, const double * xbase_ptr, double * xup, double dt,
	int nterms, const double * const * src, const double * coef, int premul, const double * xin_uv) {
	if (!xin_uv) xin_uv = xin;
	LinTerms xbase;
	xbase.n = nterms; xbase.premul = premul;
	xbase.src[0] = xbase_ptr; xbase.coef[0] = 0.0;
	for (int m = 0; m < nterms; m++) { xbase.src[m] = src[m]; xbase.coef[m] = coef[m]; }
	for (int m = (nterms > 0 ? nterms : 1); m < 12; m++) { xbase.src[m] = xbase.src[0]; xbase.coef[m] = 0.0; }
	dim3 blk(64, 4), grd(NTILES(e, p), (p.L + 3) / 4);
#define LAUNCH_HT(NT_) do { \
		if (e->udiff) hipLaunchKernelGGL((k_h_tracers<true, NT_>), grd, blk, 0, e->stream, p, e->nt, (const double *)e->d_area, xin, xbase, xup, dt, e->cfg.uniform_diffusion_scalar, xin_uv); \
		else hipLaunchKernelGGL((k_h_tracers<false, NT_>), grd, blk, 0, e->stream, p, e->nt, (const double *)e->d_area, xin, xbase, xup, dt, 0.0, xin_uv); } while (0)
	switch (nterms) {
		case 0: LAUNCH_HT(0); break;
		case 2: LAUNCH_HT(2); break;
		case 3: LAUNCH_HT(3); break;
		case 4: LAUNCH_HT(4); break;
		case 5: LAUNCH_HT(5); break;
		case 6: LAUNCH_HT(6); break;
		case 7: LAUNCH_HT(7); break;
		case 8: LAUNCH_HT(8); break;
		default: LAUNCH_HT(12); break;      // entries beyond n carry coefficient 0 and a valid pointer, as in tmxk_h_explicit
	}
#undef LAUNCH_HT
}

// ApplyScalarHyperdiffusion, tracer branch (HorizontalDynamicsFEM.cpp:1999-2035, :2076-2165) and, after the second
// pass, FilterNegativeTracers (:2707).  xbase == nullptr: base 0 (the zeroed working instance of the first pass).
__global__ __launch_bounds__(256) void k_hypervis_tracers(KParams p, int nt, const double * __restrict__ area,
	const double * __restrict__ xsrc, const double * xbase, double * xout, double dt, double nu_in, int filter, int scale)
{
	__shared__ double s[4][4][64];
	__shared__ double sD[16], sS[16];
	const int lane = threadIdx.x, ty = WAVE_UNIFORM(threadIdx.y);
	if (ty * 64 + lane < 16) { sD[ty * 64 + lane] = p.dx[ty * 64 + lane]; sS[ty * 64 + lane] = p.stiff[ty * 64 + lane]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = TILE_X(p) * 64 + lane;
	const double ida = p.g2d[G2_IDA * (size_t)p.NS + col], idb = p.g2d[G2_IDB * (size_t)p.NS + col];      // 1 / element spacing of the column's patch
	const double nu = nu_in * (scale ? p.g2d[G2_NUS * (size_t)p.NS + col] : 1.0);
	const int k = blockIdx.y * 4 + ty;
	const bool act = (k < L) && (col < p.ncol);
	const int kc = (k < L) ? k : L - 1;
	const unsigned cb = (unsigned)col * 8u;
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, jn = 1, ar = 0;
	if (act) {
		c2a0 = ldu(p.g2d + G2_C2A0 * NS, cb); c2a1 = ldu(p.g2d + G2_C2A1 * NS, cb); c2b1 = ldu(p.g2d + G2_C2B1 * NS, cb);
		jn = ldu(p.g2d + G2_JN * NS, cb);
		if (filter) ar = ldu(area + (size_t)kc * NS, cb);
	}
	s[ty][3][lane] = ar;
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;
	for (int c = 0; c < nt; c++) {
		const size_t so = (size_t)TMX_SLAB_Q(L, c, kc) * NS;
		const double psi = act ? ldu(xsrc + so, cb) : 0.0;
		const double bv = (act && xbase) ? ldu(xbase + so, cb) : 0.0;
		__syncthreads();
		s[ty][0][lane] = psi;
		__syncthreads();
		double da = 0.0, db = 0.0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			da += s[ty][0][eb + 4 * t + j] * sD[t * 4 + i];
			db += s[ty][0][eb + 4 * i + t] * sD[t * 4 + j];
		}
		da *= ida; db *= idb;
		s[ty][1][lane] = jn * (c2a0 * da + c2a1 * db);
		s[ty][2][lane] = jn * (c2a1 * da + c2b1 * db);
		__syncthreads();
		double ua = 0.0, ub = 0.0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			ua += s[ty][1][eb + 4 * t + j] * sS[i * 4 + t];
			ub += s[ty][2][eb + 4 * i + t] * sS[j * 4 + t];
		}
		ua *= ida; ub *= idb;
		double out = bv - dt * (1.0 / jn) * nu * (ua + ub);
		if (filter) {
			__syncthreads();
			s[ty][0][lane] = out;
			__syncthreads();
			double total = 0.0, nonneg = 0.0;
#pragma unroll
			for (int n = 0; n < 16; n++) {
				const double qq = s[ty][0][eb + n];
				const double pm = qq * s[ty][3][eb + n];
				total += pm;
				if (qq >= 0.0) nonneg += pm;
			}
			const double r = total / nonneg;
			out = (out > 0.0) ? out * r : 0.0;
		}
		if (act) stu(xout + so, cb, out);
	}
}

void tmxk_hypervis_tracers(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout, double dt, double nu, int filter, int scale) {
	dim3 blk(64, 4), grd(NTILES(e, p), (p.L + 3) / 4);
	hipLaunchKernelGGL(k_hypervis_tracers, grd, blk, 0, e->stream, p, e->nt, (const double *)e->d_area, xsrc, xbase, xout, dt, nu, filter, scale);
}

// VerticalDynamicsFEM::UpdateColumnTracers (VerticalDynamicsFEM.cpp:3783-4282), implicit mode, vertical order 1:
// one lane per unique column.  The tridiagonal matrix (kl = ku = 1) is kept in LAPACK band storage
// AB(i, j) = sm[(j * 4 + 2 + i - j)] and factorised / solved with the dgbtf2 / dgbtrs loops (first-maximum
// pivoting, column-oriented back substitution) -- the restatement oracle/tmx_oracle.c:orc_dgbsv pins against
// LAPACK -- once per tracer on a fresh copy, which is what dgbtrf + dgbtrs per tracer compute.
// xin: initial instance (U, V, tracer densities), w0: W of the initial column [L+1][NS] (a saved copy when the step
// runs in place and the state kernel has already overwritten it), xup: update instance (updated W; receives the tracers),
// xbase: instance holding the tracer values the update is subtracted from (the update instance itself in the
// reference; the initial instance when the preceding CopyData was fused away).
// expl (fully explicit vertical mode, :3910-3912, :4047-4063, :4117-4141, :4166-4170, :4193): every stored column is
// advanced on its own (ucol == udep == nullptr; the update instance has not been through the DSS yet), the matrix is
// the diagonal 1/dt (dgbtrs then is one division per level), xi_dot comes from the initial W for both uses, there
// is no velocity correction, and with ks != 0 the mass flux carries the uniform diffusion of q / rho - (q / rho)_ref.
// LW = columns per workgroup (the lane stride of the LDS arrays): 64, or 32 when (8L + 2) x 512 bytes exceed the 160 KB of a
// CU (L >= 40, e.g. the 40 levels of the DCMIP2016 supercell) -- half-filled wavefronts, twice the workgroups.
// ---------------------------------------------------------------------------------------------
// Hyperviscosity pass: scalar Laplacian of rho*theta, W, rho and vector Laplacian of (U,V)
// (ApplyScalarHyperdiffusion / ApplyVectorHyperdiffusion / ComputeCurlAndDiv,
//  src/atm/HorizontalDynamicsFEM.cpp:1867-2203, 2207-2414; src/atm/GridPatchCSGLL.cpp:1132-1305).
// out = base - dt * (...)(src); base = 0 when xbase == nullptr (first pass into the zeroed work
// instance), otherwise the instance that StepAfterSubCycle copied into the update instance.

// scale != 0: the coefficients are multiplied by the patch's (delta_alpha / reference length)^3.2 (fScaleNuLocally, :1968-1975)
// PULL: the source instance holds values that have NOT been averaged over co-located nodes yet (the first pass's Laplacians):
// every thread forms its own node's average while loading -- its group's copies, combined exactly as k_dss would have
// combined them for this member (dss_combine in the member's own frame, (U,V) of copies on other panels rotated into it,
// copies of other ranks from the ghost buffer) -- so the DSS pass between the two hyperviscosity passes (a read and a write
// of the whole instance) is not run at all.  The partner rows are the neighbouring elements' rows, which other threads
// load anyway: L2 hits, no extra HBM traffic.  Bit-identical to pass 1 + k_dss + pass 2.
struct DssPull {
	const int4 * colpart;     // [NS] {partner 1, partner 2, partner 3, info}: the OTHER copies of a stored column in the order of its group
	                          // (column index, >= NS: ghost NS + index, -1: none); info = n | me << 3 | type << 6 | (matrix index + 1) << 8; info 0: no copies
	const double * xmat, * ghost;
	const int * gbase, * gstride;
};
template <int NF, bool uv>
__device__ __forceinline__ void dss_pull_fields(const KParams & p, const DssPull & d, const double * __restrict__ x, int col, const int (&slab)[NF], double (&out)[NF]) {
	// out[f] = this node's DSS average of field slab[f]; with uv the last two fields are (U, V) (rotated between panels)
	const size_t NS = (size_t)p.NS;
	const int4 cp = d.colpart[col];
	double own[NF];
#pragma unroll
	for (int f = 0; f < NF; f++) own[f] = x[(size_t)slab[f] * NS + col];
	if (cp.w == 0) {
#pragma unroll
		for (int f = 0; f < NF; f++) out[f] = own[f];
		return;
	}
	const int n = cp.w & 7, me = (cp.w >> 3) & 3, ty = (cp.w >> 6) & 3, xi = (cp.w >> 8) - 1;
	const int oc[3] = { cp.x, cp.y, cp.z };
	// v[f][m]: copies in group order; this column sits at position me, the others keep their order around it
	double v[NF][4];
#pragma unroll
	for (int m = 0; m < 4; m++) {
#pragma unroll
		for (int f = 0; f < NF; f++) v[f][m] = 0.0;
	}
#pragma unroll
	for (int q = 0; q < 3; q++) {
		const int m = q + (q >= me ? 1 : 0);      // group position of the q-th other copy
		const bool on = m < n;
		const int c = oc[q];
		const bool loc = on && c < (int)NS;
		const double * ptr = !on ? x : (loc ? x + c : d.ghost + d.gbase[c - (int)NS]);
		const size_t str = !on ? 0 : (loc ? NS : (size_t)d.gstride[c - (int)NS]);
#pragma unroll
		for (int f = 0; f < NF; f++) {
			const double w = on ? ptr[(size_t)slab[f] * str] : 0.0;
#pragma unroll
			for (int mm = 0; mm < 4; mm++) if (mm == m) v[f][mm] = w;
		}
	}
#pragma unroll
	for (int f = 0; f < NF; f++) {
#pragma unroll
		for (int mm = 0; mm < 4; mm++) if (mm == me) v[f][mm] = own[f];
	}
	auto comb = [&](double w0, double w1, double w2, double w3) -> double {
		switch (me) {
			case 0: return dss_combine<0>(w0, w1, w2, w3, n, ty);
			case 1: return dss_combine<1>(w0, w1, w2, w3, n, ty);
			case 2: return dss_combine<2>(w0, w1, w2, w3, n, ty);
			default: return dss_combine<3>(w0, w1, w2, w3, n, ty);
		}
	};
	const int nscal = uv ? NF - 2 : NF;
#pragma unroll
	for (int f = 0; f < NF; f++) if (f < nscal) out[f] = comb(v[f][0], v[f][1], v[f][2], v[f][3]);
	if constexpr (uv) {
		double a[4], b[4];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			if (xi < 0 || q == me || q >= n) { a[q] = v[NF - 2][q]; b[q] = v[NF - 1][q]; }
			else {
				const double * T = d.xmat + (size_t)xi * 64 + (me * 4 + q) * 4;
				a[q] = T[0] * v[NF - 2][q] + T[1] * v[NF - 1][q]; b[q] = T[2] * v[NF - 2][q] + T[3] * v[NF - 1][q];
			}
		}
		out[NF - 2] = comb(a[0], a[1], a[2], a[3]); out[NF - 1] = comb(b[0], b[1], b[2], b[3]);
	}
	(void)xi;
}

// UQ: node-unique state layout (tmx_device.h, seam_store): operands gathered from the unique slabs, results averaged over the
// seams inside the wavefront or handed to the partial slots
// Registers: 131-133 left to itself (three workgroups per CU).  Held to 128 = four per CU the node-unique form gains 6 % (139 against
// 147 us per step, 20 bytes of scratch); the element-major form loses 12 % that way (148 against 132 us) and keeps its three.
#ifndef TMX_HV_MINWG
#define TMX_HV_MINWG 1
#endif
// HB: the launch has a base instance (second pass, or the one-pass order-2 form); without one (first pass: plain Laplacians) the five base
// values are compile-time zeros instead of registers -- that instantiation then fits its 128 registers without the 20 bytes of scratch
// (the second pass keeps its 20 bytes: loading two or four of the base values at their point of use instead leaves it at 20 / 44 bytes)
// BLK (with UQ): the workgroup's four wavefronts are the wavefronts of ONE BLOCK of the thread order at ONE level (instead of one wavefront's
// tile at four levels); the seams between them are averaged through LDS (block_seam_store, tmx_device.h), one barrier per workgroup
template <bool PULL, bool UQ, bool HB = true, bool BLK = false>
__global__ __launch_bounds__(256, (UQ && TMX_HV_MINWG < 4) ? 4 : TMX_HV_MINWG) void k_hypervis(KParams p, const double * __restrict__ xsrc,
	const double * __restrict__ xbase, double * __restrict__ xout, double dt, double nu_s_in, double nu_d_in, double nu_v_in, int scale, DssPull pull)
{
	// every LDS row here belongs to one wavefront (its level's fields, its own copy of the operator tables): the kernel has no
	// workgroup barrier, the four wavefronts of a workgroup run decoupled (wave_sync) -- up to the one exchange of the block form
	__shared__ double s[4][8][TMX_UQ_ROWLEN(UQ)];
	__shared__ double sDw[4][16], sSw[4][16];
	__shared__ double xs[BLK ? 5 : 1][BLK ? 256 : 1];
	const int lane = tile_lane<UQ>(), ty = WAVE_UNIFORM(threadIdx.y);
	if (lane < 16) { sDw[ty][lane] = p.dx[lane]; sSw[ty][lane] = p.stiff[lane]; }
	const double * sD = sDw[ty], * sS = sSw[ty];
	const int lp = lds_pos<UQ>(lane);
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	int tix, yb;
	if (!wg_tile<UQ>(p, BLK ? L + 1 : (L + 1 + 3) / 4, tix, yb)) return;
	int tile;
	if (BLK) {
		const int info = p.blk_info[p.bquads ? p.bquads[tix] : tix];
		if (ty >= (info & 7)) return;      // (a wavefront that has ended no longer counts at the workgroup's barrier)
		tile = (info >> 3) + ty;
	} else tile = p.quads ? p.quads[tix] : tix;
	const int col = tile * 64 + lane;                                 // the thread's own column: geometry, LDS
	const int kk = BLK ? yb : yb * 4 + ty;
	const int uc = UQ ? tile_ucol(p, tile, lane) : col;               // where its node's state lives
	const size_t SS = UQ ? (size_t)p.NSS : NS;                        // state slab stride
	const bool here = UQ ? (uc >= 0) : (col < p.ncol);
	const bool actw = (kk <= L) && here;     // interface level (W)
	const bool actn = (kk < L) && here;      // model level (U,V,rho*theta,rho)
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lds_pos<UQ>(lane & 48);

	double psiT = 0, psiR = 0, psiW = 0, ua = 0, ub = 0;
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, j2d = 1, jn = 1, je = 1;
	if (actw) {
		c2a0 = p.g2d[G2_C2A0 * NS + col]; c2a1 = p.g2d[G2_C2A1 * NS + col]; c2b1 = p.g2d[G2_C2B1 * NS + col];
		j2d = p.g2d[G2_J2D * NS + col]; jn = p.g2d[G2_JN * NS + col]; je = p.g2d[G2_JE * NS + col];
		if (!PULL) psiW = xsrc[TMX_SLAB_W(L, kk) * SS + uc];
	}
	// the base of the update (second pass: the instance the viscosity is applied to), loaded with the operands: at the point
	// of use its latency would follow three barriers and both contractions
	double bW = 0.0, bT = 0.0, bR = 0.0, bU = 0.0, bV = 0.0;
	if (HB && xbase && actw) {
		bW = xbase[TMX_SLAB_W(L, kk) * SS + uc];
		if (actn) {
			bT = xbase[TMX_SLAB_T(L, kk) * SS + uc]; bR = xbase[TMX_SLAB_R(L, kk) * SS + uc];
			bU = xbase[TMX_SLAB_U(L, kk) * SS + uc]; bV = xbase[TMX_SLAB_V(L, kk) * SS + uc];
		}
	}
	if (!PULL && actn) {
		psiT = xsrc[TMX_SLAB_T(L, kk) * SS + uc];
		psiR = xsrc[TMX_SLAB_R(L, kk) * SS + uc];
		ua = xsrc[TMX_SLAB_U(L, kk) * SS + uc];
		ub = xsrc[TMX_SLAB_V(L, kk) * SS + uc];
	}
	if (PULL && actw) {
		if (actn) {
			const int sl[5] = { TMX_SLAB_W(L, kk), TMX_SLAB_T(L, kk), TMX_SLAB_R(L, kk), TMX_SLAB_U(L, kk), TMX_SLAB_V(L, kk) };
			double o[5];
			dss_pull_fields<5, true>(p, pull, xsrc, col, sl, o);
			psiW = o[0]; psiT = o[1]; psiR = o[2]; ua = o[3]; ub = o[4];
		} else {
			const int sl[1] = { TMX_SLAB_W(L, kk) };
			double o[1];
			dss_pull_fields<1, false>(p, pull, xsrc, col, sl, o);
			psiW = o[0];
		}
	}
	const double cua = c2a0 * ua + c2a1 * ub;
	const double cub = c2a1 * ua + c2b1 * ub;
	s[ty][0][lp] = psiT; s[ty][1][lp] = psiR; s[ty][2][lp] = psiW;
	s[ty][3][lp] = ua;   s[ty][4][lp] = ub;
	s[ty][5][lp] = j2d * cua; s[ty][6][lp] = j2d * cub;
	wave_sync();

	double daT = 0, dbT = 0, daR = 0, dbR = 0, daW = 0, dbW = 0, daub = 0, dbua = 0, dajua = 0, dbjub = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
		const double Da = sD[t * 4 + i], Db = sD[t * 4 + j];
		daT += s[ty][0][la] * Da; dbT += s[ty][0][lb] * Db;
		daR += s[ty][1][la] * Da; dbR += s[ty][1][lb] * Db;
		daW += s[ty][2][la] * Da; dbW += s[ty][2][lb] * Db;
		daub += s[ty][4][la] * Da; dbua += s[ty][3][lb] * Db;
		dajua += s[ty][5][la] * Da; dbjub += s[ty][6][lb] * Db;
	}
	const double ida = p.g2d[G2_IDA * NS + col], idb = p.g2d[G2_IDB * NS + col];
	const double nsc = scale ? p.g2d[G2_NUS * NS + col] : 1.0;
	const double nu_s = nu_s_in * nsc, nu_d = nu_d_in * nsc, nu_v = nu_v_in * nsc;
	daT *= ida; dbT *= idb; daR *= ida; dbR *= idb; daW *= ida; dbW *= idb;
	daub *= ida; dbua *= idb; dajua *= ida; dbjub *= idb;
	const double invJ2 = 1.0 / j2d;
	const double dv = (dajua + dbjub) * invJ2, cl = (daub - dbua) * invJ2;
	// what the thread does with its results (seam_store): loaded here, behind the first contraction, used at the end
	const int sdst = UQ ? (BLK ? p.b_sdst[col] : p.t_sdst[col]) : 0, sred = UQ ? (BLK ? p.b_sred[col] : p.t_sred[col]) : 0;
	wave_sync();
	s[ty][0][lp] = jn * (c2a0 * daT + c2a1 * dbT); s[ty][1][lp] = jn * (c2a1 * daT + c2b1 * dbT);
	s[ty][2][lp] = jn * (c2a0 * daR + c2a1 * dbR); s[ty][3][lp] = jn * (c2a1 * daR + c2b1 * dbR);
	s[ty][4][lp] = je * (c2a0 * daW + c2a1 * dbW); s[ty][5][lp] = je * (c2a1 * daW + c2b1 * dbW);
	s[ty][6][lp] = dv; s[ty][7][lp] = cl;
	wave_sync();
	if (!BLK && !actw) return;      // (BLK: padding lanes stay for the exchange below -- a whole padding wavefront must reach the barrier its block's others wait at)

	double uaT = 0, ubT = 0, uaR = 0, ubR = 0, uaW = 0, ubW = 0, dad = 0, dbd = 0, dac = 0, dbc = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
		const double Sa = sS[i * 4 + t], Sb = sS[j * 4 + t];
		uaT += s[ty][0][la] * Sa; ubT += s[ty][1][lb] * Sb;
		uaR += s[ty][2][la] * Sa; ubR += s[ty][3][lb] * Sb;
		uaW += s[ty][4][la] * Sa; ubW += s[ty][5][lb] * Sb;
		dad -= Sa * s[ty][6][la]; dbd -= Sb * s[ty][6][lb];
		dac -= Sa * s[ty][7][la]; dbc -= Sb * s[ty][7][lb];
	}
	uaT *= ida; ubT *= idb; uaR *= ida; ubR *= idb; uaW *= ida; ubW *= idb;
	dad *= ida; dbd *= idb; dac *= ida; dbc *= idb;

	const double oW = bW - dt * (1.0 / je) * nu_s * (uaW + ubW);
	double oT = 0, oR = 0, oU = 0, oV = 0;
	if (actn) {
		const double invJ = 1.0 / jn;
		oT = bT - dt * invJ * nu_s * (uaT + ubT);
		oR = bR - dt * invJ * nu_s * (uaR + ubR);
		const double upa = nu_d * dad - nu_v * j2d * (c2a1 * dac + c2b1 * dbc);
		const double upb = nu_d * dbd + nu_v * j2d * (c2a0 * dac + c2a1 * dbc);
		oU = bU - dt * upa;
		oV = bV - dt * upb;
	}
	if (UQ) {
		// (level activity is wave-uniform, so all lanes of a wavefront agree on nv; the LDS rows were last read by this wavefront)
		const double v[5] = { oW, oT, oR, oU, oV };
		const int slab[5] = { TMX_SLAB_W(L, kk), TMX_SLAB_T(L, kk), TMX_SLAB_R(L, kk), TMX_SLAB_U(L, kk), TMX_SLAB_V(L, kk) };
		if constexpr (BLK) block_seam_store<5>(p, xs, ty * 64 + lane, sdst, sred, v, xout, slab, (kk < L) ? 31u : 1u);
		else seam_store<5, TMX_UQ_ROWLEN(UQ)>(p, &s[ty][0], lane, sdst, sred, v, xout, slab, (kk < L) ? 5 : 1);
		return;
	}
	xout[TMX_SLAB_W(L, kk) * NS + col] = oW;
	if (actn) {
		xout[TMX_SLAB_T(L, kk) * NS + col] = oT;
		xout[TMX_SLAB_R(L, kk) * NS + col] = oR;
		xout[TMX_SLAB_U(L, kk) * NS + col] = oU;
		xout[TMX_SLAB_V(L, kk) * NS + col] = oV;
	}
}

void tmxk_hypervis(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout,
	double dt, double nu_s, double nu_d, double nu_v, int scale, bool pull_dss) {
	const int ntiles = p.quads ? e->launch_tiles : p.NS / 64, nyb = (p.L + 1 + 3) / 4;
	dim3 blk(64, 4);
	DssPull d = { (const int4 *)e->d_colref, (const double *)e->d_xmat, (const double *)e->d_ghost, (const int *)e->d_ghost_base, (const int *)e->d_ghost_stride };
	if (tmxk_hv_walk_ok(e, p)) { tmxk_hv_walk(e, p, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale); return; }      // node-unique layout: the pass as a walk (tmx_k_hwalk.hip)
	if (p.t_ucol) {      // node-unique layout (tmxu_params)
		KParams q = p; q.u_ntiles = ntiles;
		const bool blocks = tmxu_blocks_on(e, 2);      // a workgroup = a block of the thread order at one level: its seams are averaged in the kernel
		e->u.slots_by_blocks = blocks;
		if (blocks) {
			const int nblk = p.bquads ? e->launch_blocks : e->u.nblocks;
			q.u_ntiles = nblk;
			if (xbase) hipLaunchKernelGGL((k_hypervis<false, true, true, TMX_EXP != 0>), wg_grid(true, q, nblk, p.L + 1), blk, 0, e->stream, q, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
			else hipLaunchKernelGGL((k_hypervis<false, true, false, TMX_EXP != 0>), wg_grid(true, q, nblk, p.L + 1), blk, 0, e->stream, q, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
		}
		else if (xbase) hipLaunchKernelGGL((k_hypervis<false, true, true>), wg_grid(true, q, ntiles, nyb), blk, 0, e->stream, q, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
		else hipLaunchKernelGGL((k_hypervis<false, true, false>), wg_grid(true, q, ntiles, nyb), blk, 0, e->stream, q, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
	}
#if TMX_EXP
	else if (pull_dss) hipLaunchKernelGGL((k_hypervis<true, false>), dim3(ntiles, nyb), blk, 0, e->stream, p, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
#endif
	else hipLaunchKernelGGL((k_hypervis<false, false>), dim3(ntiles, nyb), blk, 0, e->stream, p, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
}

#if TMX_EXP      // archived experiment (option "hvis_block"): in the experiments flavour of the library only
// ---------------------------------------------------------------------------------------------
// Hyperviscosity pass FUSED with the DSS of the seams inside a patch (round 3 prototype of "remove DSS passes": DESIGN.md
// section 7).  A workgroup owns a block of up to EB x EB elements of one patch plus the ring of elements around it (where the
// ring lies inside the patch), evaluates k_hypervis's arithmetic -- statement for statement -- for all of them with the
// per-element contractions in LDS, leaves the five results of every node in LDS, and then averages every seam node of its
// inner elements that is NOT on the patch's boundary from the copies in LDS, in the reference's order (alpha seams before
// beta seams: 0.5 (x + x'), 0.5 (0.5 (x + x_alpha) + 0.5 (x_beta + x_diag)); inside a patch every copy receives the same
// bits).  Only the inner elements are written, averaged; nodes on the patch boundary are written raw, and k_dss runs on the
// groups that span patches only (tmx_engine::ngroups_inpatch: 91 % of the groups at ne30 on 24 patches never reach it).
// The ring elements are evaluated redundantly by the neighbouring blocks (same operands, same statements: same bits).
#define TMX_HB_E 5                              // inner elements per block side
#define TMX_HB_W (TMX_HB_E + 2)                 // with the ring
#define TMX_HB_N (TMX_HB_W * TMX_HB_W * 16)     // nodes a workgroup holds
#define TMX_HB_T ((TMX_HB_N + 63) / 64 * 64)    // threads
#define TMX_HB_KT 4                             // levels per workgroup (geometry stays in registers)
struct HvBlock { int elem_base, nea, neb, a0, b0; };
__global__ __launch_bounds__(TMX_HB_T) void k_hypervis_block(KParams p, const HvBlock * __restrict__ blocks, const double * __restrict__ xsrc,
	const double * __restrict__ xbase, double * __restrict__ xout, double dt, double nu_s_in, double nu_d_in, double nu_v_in, int scale)
{
	__shared__ double s[8][TMX_HB_N];
	__shared__ double sD[16], sS[16];
	const int t = threadIdx.x;
	if (t < 16) { sD[t] = p.dx[t]; sS[t] = p.stiff[t]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const HvBlock B = blocks[blockIdx.x];
	const int er = t >> 4, q = t & 15, i = q >> 2, j = q & 3;
	const int ea = er / TMX_HB_W, eb_ = er - ea * TMX_HB_W;                   // position in the ringed block
	const int a = B.a0 - 1 + ea, b = B.b0 - 1 + eb_;                          // element of the patch
	const int na_in = min(TMX_HB_E, B.nea - B.a0), nb_in = min(TMX_HB_E, B.neb - B.b0);
	const bool held = t < TMX_HB_N && ea <= na_in + 1 && eb_ <= nb_in + 1 && a >= 0 && a < B.nea && b >= 0 && b < B.neb;
	const bool inner = held && ea >= 1 && ea <= na_in && eb_ >= 1 && eb_ <= nb_in;
	const int col = held ? (B.elem_base + a * B.neb + b) * 16 + q : 0;
	const int eb16 = er * 16;

	// seam partners of an inner node, as indices into the workgroup's node arrays (-1: none; -2: the node lies on the
	// patch boundary, its copies belong to another patch: left to k_dss)
	int ta = -1, tb = -1, td = -1;
	if (inner) {
		const int ia = (i == 0) ? -1 : ((i == 3) ? 1 : 0), jb = (j == 0) ? -1 : ((j == 3) ? 1 : 0);
		const bool oa = ia != 0 && (a + ia < 0 || a + ia >= B.nea), ob = jb != 0 && (b + jb < 0 || b + jb >= B.neb);
		if (oa || ob) { ta = -2; }
		else {
			if (ia != 0) ta = ((ea + ia) * TMX_HB_W + eb_) * 16 + (3 - i) * 4 + j;
			if (jb != 0) tb = (ea * TMX_HB_W + eb_ + jb) * 16 + i * 4 + (3 - j);
			if (ia != 0 && jb != 0) td = ((ea + ia) * TMX_HB_W + eb_ + jb) * 16 + (3 - i) * 4 + (3 - j);
		}
	}
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, j2d = 1, jn = 1, je = 1, ida = 0, idb = 0, nsc = 1.0;
	if (held) {
		c2a0 = p.g2d[G2_C2A0 * NS + col]; c2a1 = p.g2d[G2_C2A1 * NS + col]; c2b1 = p.g2d[G2_C2B1 * NS + col];
		j2d = p.g2d[G2_J2D * NS + col]; jn = p.g2d[G2_JN * NS + col]; je = p.g2d[G2_JE * NS + col];
		ida = p.g2d[G2_IDA * NS + col]; idb = p.g2d[G2_IDB * NS + col];
		nsc = scale ? p.g2d[G2_NUS * NS + col] : 1.0;
	}
	const double nu_s = nu_s_in * nsc, nu_d = nu_d_in * nsc, nu_v = nu_v_in * nsc;
	__syncthreads();
	for (int kq = 0; kq < TMX_HB_KT; kq++) {
		const int kk = blockIdx.y * TMX_HB_KT + kq;
		if (kk > L) break;                              // uniform
		const bool actw = held, actn = held && kk < L;
		double psiT = 0, psiR = 0, psiW = 0, ua = 0, ub = 0;
		if (actw) psiW = xsrc[TMX_SLAB_W(L, kk) * NS + col];
		if (actn) {
			psiT = xsrc[TMX_SLAB_T(L, kk) * NS + col];
			psiR = xsrc[TMX_SLAB_R(L, kk) * NS + col];
			ua = xsrc[TMX_SLAB_U(L, kk) * NS + col];
			ub = xsrc[TMX_SLAB_V(L, kk) * NS + col];
		}
		const double cua = c2a0 * ua + c2a1 * ub;
		const double cub = c2a1 * ua + c2b1 * ub;
		if (t < TMX_HB_N) {
			s[0][t] = psiT; s[1][t] = psiR; s[2][t] = psiW;
			s[3][t] = ua;   s[4][t] = ub;
			s[5][t] = j2d * cua; s[6][t] = j2d * cub;
		}
		__syncthreads();
		double daT = 0, dbT = 0, daR = 0, dbR = 0, daW = 0, dbW = 0, daub = 0, dbua = 0, dajua = 0, dbjub = 0;
		if (held) {
#pragma unroll
			for (int tt = 0; tt < 4; tt++) {
				const int la = eb16 + 4 * tt + j, lb = eb16 + 4 * i + tt;
				const double Da = sD[tt * 4 + i], Db = sD[tt * 4 + j];
				daT += s[0][la] * Da; dbT += s[0][lb] * Db;
				daR += s[1][la] * Da; dbR += s[1][lb] * Db;
				daW += s[2][la] * Da; dbW += s[2][lb] * Db;
				daub += s[4][la] * Da; dbua += s[3][lb] * Db;
				dajua += s[5][la] * Da; dbjub += s[6][lb] * Db;
			}
		}
		daT *= ida; dbT *= idb; daR *= ida; dbR *= idb; daW *= ida; dbW *= idb;
		daub *= ida; dbua *= idb; dajua *= ida; dbjub *= idb;
		const double invJ2 = 1.0 / j2d;
		const double dv = (dajua + dbjub) * invJ2, cl = (daub - dbua) * invJ2;
		__syncthreads();
		if (t < TMX_HB_N) {
			s[0][t] = jn * (c2a0 * daT + c2a1 * dbT); s[1][t] = jn * (c2a1 * daT + c2b1 * dbT);
			s[2][t] = jn * (c2a0 * daR + c2a1 * dbR); s[3][t] = jn * (c2a1 * daR + c2b1 * dbR);
			s[4][t] = je * (c2a0 * daW + c2a1 * dbW); s[5][t] = je * (c2a1 * daW + c2b1 * dbW);
			s[6][t] = dv; s[7][t] = cl;
		}
		__syncthreads();
		double uaT = 0, ubT = 0, uaR = 0, ubR = 0, uaW = 0, ubW = 0, dad = 0, dbd = 0, dac = 0, dbc = 0;
		if (held) {
#pragma unroll
			for (int tt = 0; tt < 4; tt++) {
				const int la = eb16 + 4 * tt + j, lb = eb16 + 4 * i + tt;
				const double Sa = sS[i * 4 + tt], Sb = sS[j * 4 + tt];
				uaT += s[0][la] * Sa; ubT += s[1][lb] * Sb;
				uaR += s[2][la] * Sa; ubR += s[3][lb] * Sb;
				uaW += s[4][la] * Sa; ubW += s[5][lb] * Sb;
				dad -= Sa * s[6][la]; dbd -= Sb * s[6][lb];
				dac -= Sa * s[7][la]; dbc -= Sb * s[7][lb];
			}
		}
		uaT *= ida; ubT *= idb; uaR *= ida; ubR *= idb; uaW *= ida; ubW *= idb;
		dad *= ida; dbd *= idb; dac *= ida; dbc *= idb;
		// the five results of this node, exactly k_hypervis's statements; the base value is only known for written nodes,
		// so the copies in LDS hold the results with THEIR OWN base: every copy of a seam node has one (duplicated storage)
		double rW = 0, rT = 0, rR = 0, rU = 0, rV = 0;
		if (actw) {
			const size_t o = TMX_SLAB_W(L, kk) * NS + col;
			const double bs = xbase ? xbase[o] : 0.0;
			rW = bs - dt * (1.0 / je) * nu_s * (uaW + ubW);
		}
		if (actn) {
			const double invJ = 1.0 / jn;
			size_t o = TMX_SLAB_T(L, kk) * NS + col;
			double bs = xbase ? xbase[o] : 0.0;
			rT = bs - dt * invJ * nu_s * (uaT + ubT);
			o = TMX_SLAB_R(L, kk) * NS + col;
			bs = xbase ? xbase[o] : 0.0;
			rR = bs - dt * invJ * nu_s * (uaR + ubR);
			const double upa = nu_d * dad - nu_v * j2d * (c2a1 * dac + c2b1 * dbc);
			const double upb = nu_d * dbd + nu_v * j2d * (c2a0 * dac + c2a1 * dbc);
			o = TMX_SLAB_U(L, kk) * NS + col;
			bs = xbase ? xbase[o] : 0.0;
			rU = bs - dt * upa;
			o = TMX_SLAB_V(L, kk) * NS + col;
			bs = xbase ? xbase[o] : 0.0;
			rV = bs - dt * upb;
		}
		__syncthreads();
		if (t < TMX_HB_N) { s[0][t] = rW; s[1][t] = rT; s[2][t] = rR; s[3][t] = rU; s[4][t] = rV; }
		__syncthreads();
		if (inner) {
			double o5[5] = { rW, rT, rR, rU, rV };
			if (ta >= 0 || tb >= 0) {
#pragma unroll
				for (int f = 0; f < 5; f++) {
					const double x = o5[f];
					if (td >= 0) o5[f] = 0.5 * (0.5 * (x + s[f][ta]) + 0.5 * (s[f][tb] + s[f][td]));
					else o5[f] = 0.5 * (x + s[f][ta >= 0 ? ta : tb]);
				}
			}
			xout[TMX_SLAB_W(L, kk) * NS + col] = o5[0];
			if (kk < L) {
				xout[TMX_SLAB_T(L, kk) * NS + col] = o5[1];
				xout[TMX_SLAB_R(L, kk) * NS + col] = o5[2];
				xout[TMX_SLAB_U(L, kk) * NS + col] = o5[3];
				xout[TMX_SLAB_V(L, kk) * NS + col] = o5[4];
			}
		}
		__syncthreads();
	}
}

void tmxk_hypervis_block(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout,
	double dt, double nu_s, double nu_d, double nu_v, int scale) {
	dim3 blk(TMX_HB_T), grd(e->n_hvblocks, (p.L + 1 + TMX_HB_KT - 1) / TMX_HB_KT);
	hipLaunchKernelGGL(k_hypervis_block, grd, blk, 0, e->stream, p, (const HvBlock *)e->d_hvblocks, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale);
}
#endif      // TMX_EXP

// ---------------------------------------------------------------------------------------------
// Uniform diffusion at the end of HorizontalDynamicsFEM::StepExplicit (src/atm/HorizontalDynamicsFEM.cpp:1817-1859):
//   ApplyVectorHyperdiffusion(initial, update, dt, -K_v, -K_v)          U,V -= dt * (-K_v) * vector Laplacian(U,V)
//   ApplyVectorHyperdiffusion(REFERENCE, update, dt, +K_v, +K_v)        U,V -= dt * (+K_v) * vector Laplacian(U,V)_ref
//   ApplyScalarHyperdiffusion(initial, update, dt, K_s, comp 2, remove ref)   rho*theta, psi = initial - reference
//   ApplyScalarHyperdiffusion(initial, update, dt, K_v, comp 3, remove ref)   W on interfaces
// in that order per variable, one launch; the update instance is modified in place.  Same element kernels as
// k_hypervis (ApplyScalarHyperdiffusion :1867-2203, ApplyVectorHyperdiffusion :2207-2414).
__global__ __launch_bounds__(256) void k_uniform_diffusion(KParams p, const double * __restrict__ xsrc,
	const double * __restrict__ xref, double * xout, double dt, double ks, double kv)
{
	__shared__ double s[4][10][64];
	__shared__ double sD[16], sS[16];
	const int lane = threadIdx.x, ty = WAVE_UNIFORM(threadIdx.y);
	const int tid = ty * 64 + lane;
	if (tid < 16) { sD[tid] = p.dx[tid]; sS[tid] = p.stiff[tid]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = TILE_X(p) * 64 + lane;
	const int kk = blockIdx.y * 4 + ty;
	const bool actw = (kk <= L) && (col < p.ncol);
	const bool actn = (kk < L) && (col < p.ncol);
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;

	double psiT = 0, psiW = 0, ua = 0, ub = 0, ra = 0, rb = 0;
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, j2d = 1, jn = 1, je = 1;
	if (actw) {
		c2a0 = p.g2d[G2_C2A0 * NS + col]; c2a1 = p.g2d[G2_C2A1 * NS + col]; c2b1 = p.g2d[G2_C2B1 * NS + col];
		j2d = p.g2d[G2_J2D * NS + col]; jn = p.g2d[G2_JN * NS + col]; je = p.g2d[G2_JE * NS + col];
		psiW = xsrc[TMX_SLAB_W(L, kk) * NS + col];
		psiW -= xref[TMX_SLAB_W(L, kk) * NS + col];
	}
	// the values to update, loaded with the operands (at the point of use their latency would follow the barriers)
	double oW = 0.0, oT = 0.0, oU = 0.0, oV = 0.0;
	if (actw) oW = xout[TMX_SLAB_W(L, kk) * NS + col];
	if (actn) { oT = xout[TMX_SLAB_T(L, kk) * NS + col]; oU = xout[TMX_SLAB_U(L, kk) * NS + col]; oV = xout[TMX_SLAB_V(L, kk) * NS + col]; }
	if (actn) {
		psiT = xsrc[TMX_SLAB_T(L, kk) * NS + col];
		psiT -= xref[TMX_SLAB_T(L, kk) * NS + col];
		ua = xsrc[TMX_SLAB_U(L, kk) * NS + col]; ub = xsrc[TMX_SLAB_V(L, kk) * NS + col];
		ra = xref[TMX_SLAB_U(L, kk) * NS + col]; rb = xref[TMX_SLAB_V(L, kk) * NS + col];
	}
	s[ty][0][lane] = psiT; s[ty][1][lane] = psiW;
	s[ty][2][lane] = ua; s[ty][3][lane] = ub; s[ty][4][lane] = j2d * (c2a0 * ua + c2a1 * ub); s[ty][5][lane] = j2d * (c2a1 * ua + c2b1 * ub);
	s[ty][6][lane] = ra; s[ty][7][lane] = rb; s[ty][8][lane] = j2d * (c2a0 * ra + c2a1 * rb); s[ty][9][lane] = j2d * (c2a1 * ra + c2b1 * rb);
	__syncthreads();

	double daT = 0, dbT = 0, daW = 0, dbW = 0;
	double daub = 0, dbua = 0, dajua = 0, dbjub = 0, daubr = 0, dbuar = 0, dajuar = 0, dbjubr = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
		const double Da = sD[t * 4 + i], Db = sD[t * 4 + j];
		daT += s[ty][0][la] * Da; dbT += s[ty][0][lb] * Db;
		daW += s[ty][1][la] * Da; dbW += s[ty][1][lb] * Db;
		daub += s[ty][3][la] * Da; dbua += s[ty][2][lb] * Db;
		dajua += s[ty][4][la] * Da; dbjub += s[ty][5][lb] * Db;
		daubr += s[ty][7][la] * Da; dbuar += s[ty][6][lb] * Db;
		dajuar += s[ty][8][la] * Da; dbjubr += s[ty][9][lb] * Db;
	}
	const double ida = p.g2d[G2_IDA * NS + col], idb = p.g2d[G2_IDB * NS + col];
	daT *= ida; dbT *= idb; daW *= ida; dbW *= idb;
	daub *= ida; dbua *= idb; dajua *= ida; dbjub *= idb;
	daubr *= ida; dbuar *= idb; dajuar *= ida; dbjubr *= idb;
	const double invJ2 = 1.0 / j2d;
	const double dv = (dajua + dbjub) * invJ2, cl = (daub - dbua) * invJ2;
	const double dvr = (dajuar + dbjubr) * invJ2, clr = (daubr - dbuar) * invJ2;
	__syncthreads();
	s[ty][0][lane] = jn * (c2a0 * daT + c2a1 * dbT); s[ty][1][lane] = jn * (c2a1 * daT + c2b1 * dbT);
	s[ty][2][lane] = je * (c2a0 * daW + c2a1 * dbW); s[ty][3][lane] = je * (c2a1 * daW + c2b1 * dbW);
	s[ty][4][lane] = dv; s[ty][5][lane] = cl; s[ty][6][lane] = dvr; s[ty][7][lane] = clr;
	__syncthreads();
	if (!actw) return;

	double uaT = 0, ubT = 0, uaW = 0, ubW = 0, dad = 0, dbd = 0, dac = 0, dbc = 0, dadr = 0, dbdr = 0, dacr = 0, dbcr = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
		const double Sa = sS[i * 4 + t], Sb = sS[j * 4 + t];
		uaT += s[ty][0][la] * Sa; ubT += s[ty][1][lb] * Sb;
		uaW += s[ty][2][la] * Sa; ubW += s[ty][3][lb] * Sb;
		dad -= Sa * s[ty][4][la]; dbd -= Sb * s[ty][4][lb];
		dac -= Sa * s[ty][5][la]; dbc -= Sb * s[ty][5][lb];
		dadr -= Sa * s[ty][6][la]; dbdr -= Sb * s[ty][6][lb];
		dacr -= Sa * s[ty][7][la]; dbcr -= Sb * s[ty][7][lb];
	}
	uaT *= ida; ubT *= idb; uaW *= ida; ubW *= idb;
	dad *= ida; dbd *= idb; dac *= ida; dbc *= idb; dadr *= ida; dbdr *= idb; dacr *= ida; dbcr *= idb;
	{
		const size_t o = TMX_SLAB_W(L, kk) * NS + col;
		xout[o] = oW - dt * (1.0 / je) * kv * (uaW + ubW);
	}
	if (actn) {
		size_t o = TMX_SLAB_T(L, kk) * NS + col;
		xout[o] = oT - dt * (1.0 / jn) * ks * (uaT + ubT);
		const double nkv = -kv;
		const double upa = nkv * dad - nkv * j2d * (c2a1 * dac + c2b1 * dbc);
		const double upb = nkv * dbd + nkv * j2d * (c2a0 * dac + c2a1 * dbc);
		const double upar = kv * dadr - kv * j2d * (c2a1 * dacr + c2b1 * dbcr);
		const double upbr = kv * dbdr + kv * j2d * (c2a0 * dacr + c2a1 * dbcr);
		o = TMX_SLAB_U(L, kk) * NS + col;
		double v = oU; v -= dt * upa; v -= dt * upar; xout[o] = v;
		o = TMX_SLAB_V(L, kk) * NS + col;
		v = oV; v -= dt * upb; v -= dt * upbr; xout[o] = v;
	}
}

void tmxk_uniform_diffusion(tmx_engine * e, const KParams & p, const double * xsrc, const double * xref, double * xout,
	double dt, double ks, double kv) {
	dim3 blk(64, 4), grd(NTILES(e, p), (p.L + 1 + 3) / 4);
	hipLaunchKernelGGL(k_uniform_diffusion, grd, blk, 0, e->stream, p, xsrc, xref, xout, dt, ks, kv);
}

// Vertical uniform diffusion of U, V in VerticalDynamicsFEM::StepExplicit (src/atm/VerticalDynamicsFEM.cpp:1059-1105):
// update += dt * K_v / ztop^2 * (DiffDiffNodeToNode(U) - DiffDiffNodeToNode(U_ref)), same for V.
__global__ __launch_bounds__(256) void k_v_uniform_diffusion_uv(KParams p, const double * __restrict__ xin,
	const double * __restrict__ xref, double * __restrict__ xup, double dt, double cf)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = TILE_X(p) * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (k >= L || col >= p.ncol) return;
#pragma unroll
	for (int v = 0; v < 2; v++) {
		const int sb = v * L;
		double dd = 0.0, ddr = 0.0;
#pragma unroll
		for (int off = -2; off <= 2; off++) {
			const int l = k + off;
			if (l < 0 || l >= L) continue;
			const double c = OPC(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, off);
			dd += c * xin[(size_t)(sb + l) * NS + col];
			ddr += c * xref[(size_t)(sb + l) * NS + col];
		}
		xup[(size_t)(sb + k) * NS + col] += dt * cf * (dd - ddr);
	}
}

void tmxk_v_uniform_diffusion_uv(tmx_engine * e, const KParams & p, const double * xin, const double * xref, double * xup, double dt, double cf) {
	dim3 blk(64, 4), grd(NTILES(e, p), (p.L + 3) / 4);
	hipLaunchKernelGGL(k_v_uniform_diffusion_uv, grd, blk, 0, e->stream, p, xin, xref, xup, dt, cf);
}

// ---------------------------------------------------------------------------------------------
// Shallow water: HorizontalDynamicsFEM::StepShallowWater (src/atm/HorizontalDynamicsFEM.cpp:321-647).
// One level; H is kept in the rho*theta slab.  out = base + dt * rhs(in).

__global__ __launch_bounds__(64) void k_sw_explicit(KParams p, const double * __restrict__ xin,
	const double * xbase, double * xup, double dt)
{
	__shared__ double s[5][64];
	__shared__ double sD[16], sS[16];
	const int lane = threadIdx.x;
	if (lane < 16) { sD[lane] = p.dx[lane]; sS[lane] = p.stiff[lane]; }
	const size_t NS = (size_t)p.NS;
	const int col = TILE_X(p) * 64 + lane;
	const bool act = col < p.ncol;
	double ua = 0, ub = 0, h = 0, c2a0 = 0, c2a1 = 0, c2b1 = 0, j2d = 1, fcor = 0, zs = 0;
	if (act) {
		ua = xin[0 * NS + col]; ub = xin[1 * NS + col]; h = xin[2 * NS + col];
		c2a0 = p.g2d[G2_C2A0 * NS + col]; c2a1 = p.g2d[G2_C2A1 * NS + col]; c2b1 = p.g2d[G2_C2B1 * NS + col];
		j2d = p.g2d[G2_J2D * NS + col]; fcor = p.g2d[G2_F * NS + col]; zs = p.g2d[G2_ZS * NS + col];
	}
	const double cua = c2a0 * ua + c2a1 * ub;
	const double cub = c2a1 * ua + c2b1 * ub;
	double ke = 0.5 * (cua * ua + cub * ub);
	ke += p.grav * h;
	const double ba = j2d * cua, bb = j2d * cub;
	s[0][lane] = ua; s[1][lane] = ub; s[2][lane] = ke;
	s[3][lane] = ba * (h - zs); s[4][lane] = bb * (h - zs);
	__syncthreads();
	if (!act) return;
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;
	double daub = 0, dbua = 0, dake = 0, dbke = 0, dafa = 0, dbfb = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j;
		dafa -= s[3][la] * sS[i * 4 + t];
		daub += s[1][la] * sD[t * 4 + i];
		dake += s[2][la] * sD[t * 4 + i];
	}
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int lb = eb + 4 * i + t;
		dbfb -= s[4][lb] * sS[j * 4 + t];
		dbua += s[0][lb] * sD[t * 4 + j];
		dbke += s[2][lb] * sD[t * 4 + j];
	}
	const double ida = p.g2d[G2_IDA * NS + col], idb = p.g2d[G2_IDB * NS + col];
	dafa *= ida; daub *= ida; dake *= ida; dbfb *= idb; dbua *= idb; dbke *= idb;
	double upa = 0.0, upb = 0.0;
	const double zeta = (daub - dbua);
	const double ucza = cub * zeta, uczb = -cua * zeta;
	upa += fcor * j2d * cub;
	upb -= fcor * j2d * cua;
	upa += -dake + ucza;
	upb += -dbke + uczb;
	const double invJ2 = 1.0 / j2d;
	xup[0 * NS + col] = xbase[0 * NS + col] + dt * upa;
	xup[1 * NS + col] = xbase[1 * NS + col] + dt * upb;
	xup[2 * NS + col] = xbase[2 * NS + col] - dt * invJ2 * (dafa + dbfb);
}

void tmxk_sw_explicit(tmx_engine * e, const KParams & p, const double * xin, const double * xbase, double * xup, double dt) {
	hipLaunchKernelGGL(k_sw_explicit, dim3(NTILES(e, p)), dim3(64), 0, e->stream, p, xin, xbase, xup, dt);
}

