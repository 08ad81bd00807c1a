// tmx_k_hwalk.hip -- the explicit stage of the node-unique layout as a COLUMN-SEGMENT WALK (round 6).
//
// Same arithmetic, statement for statement, as k_h_explicit<FUSE_V, NT, .., UQ> + k_h_w_update (tmx_k_horizontal.hip):
// HorizontalDynamicsFEM::StepNonhydrostaticPrimitive (src/atm/HorizontalDynamicsFEM.cpp:848-1421, W on interfaces :1612-1659),
// VerticalDynamicsFEM::StepExplicit in the implicit mode (src/atm/VerticalDynamicsFEM.cpp:816-828, 984-1023) and the stage's
// LinearCombineData (src/atm/GridPatch.cpp:1436-1508) -- but a wavefront no longer owns ONE level of its four elements: it walks a
// segment of levels k0 .. k1-1 upwards.
//   * the 17 per-column geometry values, the seam tables and the operator rows are loaded once per segment instead of once per level
//     (16 of the level-parallel kernel's 26 + 5 NT loads per thread and level);
//   * U, V of the levels k-1, k, k+1, W of the interfaces k, k+1 live in a sliding register window: 5 state loads per level instead of 10;
//   * the loads of level k+1 and the stage combination's raw terms of level k are issued BEFORE level k's arithmetic (Exner function,
//     twelve contractions, pointwise updates), so a wavefront waits for memory once per level, behind ~3 000 cycles of its own work;
//   * xi_dot of interface k+1 is evaluated once (the level-parallel kernel forms it in the wavefronts of both adjacent levels);
//   * (u x zeta)_xi of the level below stays in a register, so W of EVERY interface is updated here: the segment's first interface from a
//     recomputed (u x zeta)_xi of level k0-1 (one contraction row: U, V, W of that level are in the window anyway), the bottom boundary
//     condition from the H-only U, V of levels 0, 1 kept in registers, the rigid lid by the last segment.  k_h_w_update and its hand-over
//     arrays (ucz, uv01) are not used on this path.
// One workgroup = one wavefront (no workgroup barrier anywhere); LDS: nine contraction rows (reused by the seam exchange), the
// operator tables, the exp / log tables of the Exner function.
#include "tmx_device.h"

#ifdef TMX_H_TIMING
// shader cycles per wavefront by phase, accumulated over the levels of a segment; one record per wavefront (tools/h_timing.py)
// [0] prologue (geometry, window, halo level), [1] loads of the next level + combination terms issued, [2] pointwise + Exner + LDS writes,
// [3] contractions, [4] pointwise updates, [5] wait for the combination's terms + reduce, [6] seam exchange + stores, [7] wavefronts
#define HW_WAVES 32768
__device__ unsigned long long g_hw_rec[8][HW_WAVES][8];
#define HW_DECL long long hw_ = __builtin_readcyclecounter(); long long hws_[7] = { 0, 0, 0, 0, 0, 0, 0 }
#define HW_MARK(i) do { const long long t1_ = __builtin_readcyclecounter(); hws_[i] += t1_ - hw_; hw_ = t1_; } while (0)
#define HW_END(NT_) do { if (threadIdx.x == 0) { const unsigned w_ = blockIdx.x % HW_WAVES; \
	for (int q_ = 0; q_ < 7; q_++) g_hw_rec[(NT_) & 7][w_][q_] = (unsigned long long)hws_[q_]; g_hw_rec[(NT_) & 7][w_][7] = 1ull; } } while (0)
void tmxk_h_walk_timing_read(unsigned long long * out) {
	static std::vector<unsigned long long> h((size_t)8 * HW_WAVES * 8);
	hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_hw_rec), h.size() * sizeof(unsigned long long));
	for (int i = 0; i < 128; i++) out[i] = 0;
	for (int nt = 0; nt < 8; nt++) for (int w = 0; w < HW_WAVES; w++) if (h[((size_t)nt * HW_WAVES + w) * 8 + 7])
		for (int q = 0; q < 8; q++) out[nt * 8 + q] += h[((size_t)nt * HW_WAVES + w) * 8 + q];
	std::fill(h.begin(), h.end(), 0ull);
	hipMemcpyToSymbol(HIP_SYMBOL(g_hw_rec), h.data(), h.size() * sizeof(unsigned long long));
}
#else
#define HW_DECL
#define HW_MARK(i)
#define HW_END(NT_)
void tmxk_h_walk_timing_read(unsigned long long * out) { for (int i = 0; i < 128; i++) out[i] = 0; }
#endif

// where term m of the stage combination reads slab `slab`: node-unique (stride SS, the node's entry) or, DM and bit m of the mask set,
// the thread's own stored copy of an element-major instance (stride NS, the thread's column)
// (NS / gb here: stride and byte offset of the thread's stored column in the ELEMENT-MAJOR layout -- KParams::NSD, t_dcol)
template <bool DM>
__device__ __forceinline__ double term_load(const double * src, unsigned dmask, int m, int slab, size_t SS, size_t NS, unsigned cb, unsigned gb) {
	const bool d = DM && ((dmask >> m) & 1u);
	return ldu_term(src + (size_t)slab * (d ? NS : SS), d ? gb : cb);
}

template <int NT, bool PM, bool DM>
__device__ __forceinline__ void terms_load(const double * const (&src)[12], unsigned dmask, int slab, size_t SS, size_t NS, unsigned cb, unsigned gb, double (&x)[NT > 0 ? NT : 1]) {
	if (NT == 0) { x[0] = term_load<DM>(src[0], dmask, 0, slab, SS, NS, cb, gb); return; }
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || PM) ? term_load<DM>(src[m], dmask, m, slab, SS, NS, cb, gb) : 0.0;
}

// Operator rows and the 1 - eta table through the CONSTANT address space: the walk stores results before it loads the next level's
// coefficients, and a load behind a store that may alias is not a scalar load any more (the compiler fetched every coefficient with a
// vector load of one address: ~15 vector-memory instructions and 30 registers per level).  The tables are written by tmx_finalize only.
typedef const double __attribute__((address_space(4))) * tmx_cdp;
#define OPK(op, k, off) opsc[(((op) * (L + 1)) + (k)) * TMX_OPW + ((off) + 2)]

#ifndef TMX_HW_WAVES_PER_EU
#define TMX_HW_WAVES_PER_EU 2      // register budget: 256 per lane (the raw terms of a seven-term combination alone are 70 registers in flight)
#endif

// BLK: a workgroup = the (up to four) wavefronts of a block of the thread order, walking the same segment in step; the seams between them
// are averaged through LDS (block_seam_store, tables KParams::b_sdst / b_sred), W of the surface and of the lid travel with levels 1 and L-1
// EM: the stage also forms and stores the leading partial sum of a LATER stage's combination from the terms it has loaded anyway (EmitTerms,
// emit_value in tmx_device.h; prefix_plan in tmx_step.hip): the later stage then reads that one instance instead of the members
// UQ = false: the ELEMENT-MAJOR layout (every configuration the node-unique one does not cover: tracers, uniform diffusion, fully explicit
// vertical dynamics, a step the engine keeps element-major): a thread reads and writes its own stored column, no seam exchange, the DSS
// kernel averages afterwards as the reference does.  FV = false: H.StepExplicit alone (the caller runs V.StepExplicit as a kernel of its own).
// UD (element-major): the horizontal uniform diffusion at the end of HorizontalDynamicsFEM::StepExplicit (:1817-1859; k_uniform_diffusion,
// tmx_k_horizontal.hip, statement for statement) applied to the level's results in registers before they are stored -- that kernel's pass over the
// initial, the reference and the update instance (20 slab reads / writes per point) shrinks to four slabs: the reference's rho*theta and W, and the
// diffusion increments of the REFERENCE's U, V, which do not change from step to step and are formed once (k_ref_diffusion_terms below: the
// same statements on the same operands, stored instead of used at once).
// UD with FV: V.StepExplicit's part on U, V follows in the same registers (VerticalDynamicsFEM.cpp:984-1023 upwind penalty, :1059-1105 vertical
// uniform diffusion -- k_v_explicit_slide, tmx_k_vertical.hip, statement for statement): the window holds U, V of the levels k - 2 .. k + 2, the
// reference's DiffDiff sums come from the stored terms too.  k_v_explicit_slide's pass over U, V, W, the reference and the update is not run.
struct UdiffArgs { const double * xref; double ks, kv; const double * xrefd; double cfv; int udv; };
// slabs of the stored reference terms [4 L][NS]: horizontal increments of U, V (upar, upbr of k_uniform_diffusion), DiffDiffNodeToNode of U_ref, V_ref
#define TMX_REFD_HU(L, k) (k)
#define TMX_REFD_HV(L, k) ((L) + (k))
#define TMX_REFD_DU(L, k) (2 * (L) + (k))
#define TMX_REFD_DV(L, k) (3 * (L) + (k))

// The reference state's share of the uniform diffusion of U, V, once per reference state: one wavefront = four elements at one level.
__global__ __launch_bounds__(64) void k_ref_diffusion_terms(KParams p, const double * __restrict__ xref, double * __restrict__ xrefd, double kv)
{
	__shared__ double s[4][64];
	__shared__ double sD[16], sS[16];
	const int lane = threadIdx.x;
	if (lane < 16) { sD[lane] = p.dx[lane]; sS[lane] = p.stiff[lane]; }
	const int L = p.L, k = blockIdx.y;
	const size_t NS = (size_t)p.NS;
	const int colr = blockIdx.x * 64 + lane;
	const bool here = colr < p.ncol;
	const int col = here ? colr : p.ncol - 1;
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;
	const double c2a0 = p.g2d[G2_C2A0 * NS + col], c2a1 = p.g2d[G2_C2A1 * NS + col], c2b1 = p.g2d[G2_C2B1 * NS + col];
	const double j2d = p.g2d[G2_J2D * NS + col], ida = p.g2d[G2_IDA * NS + col], idb = p.g2d[G2_IDB * NS + col];
	const double ra = xref[TMX_SLAB_U(L, k) * NS + col], rb = xref[TMX_SLAB_V(L, k) * NS + col];
	s[0][lane] = ra; s[1][lane] = rb; s[2][lane] = j2d * (c2a0 * ra + c2a1 * rb); s[3][lane] = j2d * (c2a1 * ra + c2b1 * rb);
	__syncthreads();
	double daubr = 0, dbuar = 0, dajuar = 0, dbjubr = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
		const double Da = sD[t * 4 + i], Db = sD[t * 4 + j];
		daubr += s[1][la] * Da; dbuar += s[0][lb] * Db;
		dajuar += s[2][la] * Da; dbjubr += s[3][lb] * Db;
	}
	daubr *= ida; dbuar *= idb; dajuar *= ida; dbjubr *= idb;
	const double invJ2 = 1.0 / j2d;
	const double dvr = (dajuar + dbjubr) * invJ2, clr = (daubr - dbuar) * invJ2;
	__syncthreads();
	s[0][lane] = dvr; s[1][lane] = clr;
	__syncthreads();
	double dadr = 0, dbdr = 0, dacr = 0, dbcr = 0;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
		const double Sa = sS[i * 4 + t], Sb = sS[j * 4 + t];
		dadr -= Sa * s[0][la]; dbdr -= Sb * s[0][lb];
		dacr -= Sa * s[1][la]; dbcr -= Sb * s[1][lb];
	}
	dadr *= ida; dbdr *= idb; dacr *= ida; dbcr *= idb;
	const double upar = kv * dadr - kv * j2d * (c2a1 * dacr + c2b1 * dbcr);
	const double upbr = kv * dbdr + kv * j2d * (c2a0 * dacr + c2a1 * dbcr);
	// DiffDiffNodeToNode of the reference's U, V (the accumulation of k_v_explicit_slide / v_explicit_point: in-range levels, offsets ascending)
	double ddr[2] = { 0.0, 0.0 };
#pragma unroll
	for (int off = -2; off <= 2; off++) {
		const int l = k + off;
		if (l < 0 || l >= L) continue;
		const double c = p.ops[((TMX_OP_DIFFDIFF_NODE_TO_NODE * (L + 1)) + k) * TMX_OPW + (off + 2)];
		ddr[0] += c * xref[TMX_SLAB_U(L, l) * NS + col];
		ddr[1] += c * xref[TMX_SLAB_V(L, l) * NS + col];
	}
	if (!here) return;
	xrefd[TMX_REFD_HU(L, k) * NS + col] = upar; xrefd[TMX_REFD_HV(L, k) * NS + col] = upbr;
	xrefd[TMX_REFD_DU(L, k) * NS + col] = ddr[0]; xrefd[TMX_REFD_DV(L, k) * NS + col] = ddr[1];
}
template <int NT, bool PM, bool DM, bool BLK = false, bool EM = false, bool UQ = true, bool FV = true, bool UD = false>
__global__ __launch_bounds__(BLK ? 256 : 64, TMX_HW_WAVES_PER_EU) void k_h_walk(KParams p, const double * __restrict__ xin, LinTerms xbase, double * xup, double dt, UvTerms uvt, int nseg, EmitTerms em,
	UdiffArgs ud)
{
	constexpr int RL = TMX_UQ_ROWLEN(true);
	constexpr int NW = BLK ? 4 : 1;
	__shared__ double sw[NW][9][RL];
	__shared__ double xs[BLK ? 2 : 1][BLK ? 6 : 1][BLK ? 256 : 1];
	__shared__ double sD[16], sS[16];
	__shared__ double rmtab[TMX_RMTAB_DOUBLES];
	const int wv = BLK ? WAVE_UNIFORM(threadIdx.y) : 0;
	double (*s)[RL] = sw[wv];
	const int lane = tile_lane<true>();      // position e * 16 + i * 4 + j of this thread inside its tile (tmx_device.h: the identity unless TMX_UQ_ROWLANES)
	const int lp = lds_pos<true>(lane);      // where its values sit in an LDS row
	HW_DECL;
	// the exp / log tables (256 + 256 words) on their way to LDS: a lane's loads in flight with the geometry and the first window
	constexpr int NTL = BLK ? 1 : 4;
	double tlog[NTL]; unsigned long long texp[NTL];
	const int tidw = wv * 64 + (int)threadIdx.x;
#pragma unroll
	for (int t = 0; t < NTL; t++) { tlog[t] = tmx_rm_log_tab[t * 64 + tidw]; texp[t] = tmx_rm_exp_tab[t * 64 + tidw]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS, SS = UQ ? (size_t)p.NSS : NS, NSD = (size_t)p.NSD;
	int tix, seg;
	if (!wg_tile<UQ>(p, nseg, tix, seg)) return;
	int tile;
	bool idle = false;      // BLK: a wavefront beyond the block's last one only keeps the barriers company
	if (BLK) {
		const int info = p.blk_info[p.bquads ? p.bquads[tix] : tix];
		tile = (info >> 3) + wv; idle = wv >= (info & 7);
		if (idle) tile = info >> 3;
	} else tile = p.quads ? p.quads[tix] : tix;
	const int k0 = WAVE_UNIFORM((seg * L) / nseg), k1 = WAVE_UNIFORM(((seg + 1) * L) / nseg);
	const int col = tile * 64 + lane;
	// padding lanes (a whole element of the last tile): node 0, nothing stored (UQ: t_sdst = -1; element-major: col >= ncol)
	const int uc = UQ ? max(tile_ucol(p, tile, lane), 0) : min(col, p.ncol - 1);
	const bool here = UQ || col < p.ncol;
	unsigned cb = (unsigned)uc * 8u, gb = (unsigned)col * 8u;
	// where the thread's own stored copy sits in an element-major instance (mixed reads, DM): its stored column, that layout's stride
	unsigned gd = DM ? (unsigned)max(p.t_dcol[col], 0) * 8u : gb;
	// the initial instance may be element-major (DM, bit 31): the thread's own copy
	const bool xd = DM && (xbase.dmask >> 31) != 0u;
	const size_t SX = xd ? NSD : SS;
	unsigned cx = xd ? gd : cb;
	const double * __restrict__ xuv = uvt.xin;
	const unsigned dmask = xbase.dmask;
	const tmx_cdp opsc = (tmx_cdp)(unsigned long long)p.ops, etac = (tmx_cdp)(unsigned long long)p.eta;

	// ---- per-column constants (once per segment)
	const double c2a0 = ldu(p.g2d + G2_C2A0 * NS, gb), c2a1 = ldu(p.g2d + G2_C2A1 * NS, gb), c2b1 = ldu(p.g2d + G2_C2B1 * NS, gb);
	const double jn = ldu(p.g2d + G2_JN * NS, gb), j2d = ldu(p.g2d + G2_J2D * NS, gb), fcor = ldu(p.g2d + G2_F * NS, gb);
	const MetCol mc = met_col(p, col);
	const double ida = ldu(p.g2d + G2_IDA * NS, gb), idb = ldu(p.g2d + G2_IDB * NS, gb);
	const int sdst = UQ ? (BLK ? p.b_sdst[col] : p.t_sdst[col]) : 0, sred = UQ ? (BLK ? p.b_sred[col] : p.t_sred[col]) : 0;
	const int pos = wv * 64 + lane;      // BLK: position in the block
	const double je = UD ? ldu(p.g2d + G2_JE * NS, gb) : 1.0;      // UD: Jacobian on interfaces
	double xW0 = 0.0;      // UD: the diffusion increment of W on the surface, formed at level 0, applied when level 1 knows W there
	// ---- window of the first level
	double uM, u0, uP, vM, v0, vP, w0, w1, t0, r0;
	double uMM = 0.0, vMM = 0.0;      // FV with UD: level k - 2 (the DiffDiff stencil)
	if (FV && UD) {
		const int kmm = (k0 > 1) ? k0 - 2 : 0;
		uMM = ldu(xuv + TMX_SLAB_U(L, kmm) * SX, cx); vMM = ldu(xuv + TMX_SLAB_V(L, kmm) * SX, cx);
	}
	{
		const int km = (k0 > 0) ? k0 - 1 : 0, kp = (k0 < L - 1) ? k0 + 1 : L - 1;
		uM = ldu(xuv + TMX_SLAB_U(L, km) * SX, cx); u0 = ldu(xuv + TMX_SLAB_U(L, k0) * SX, cx); uP = ldu(xuv + TMX_SLAB_U(L, kp) * SX, cx);
		vM = ldu(xuv + TMX_SLAB_V(L, km) * SX, cx); v0 = ldu(xuv + TMX_SLAB_V(L, k0) * SX, cx); vP = ldu(xuv + TMX_SLAB_V(L, kp) * SX, cx);
		w0 = ldu(xin + TMX_SLAB_W(L, k0) * SX, cx); w1 = ldu(xin + TMX_SLAB_W(L, k0 + 1) * SX, cx);
		t0 = ldu(xin + TMX_SLAB_T(L, k0) * SX, cx); r0 = ldu(xin + TMX_SLAB_R(L, k0) * SX, cx);
	}
	const double wH = (k0 > 0) ? ldu(xin + TMX_SLAB_W(L, k0 - 1) * SX, cx) : 0.0;
	const double invJ = 1.0 / jn;
	const double fj = fcor * j2d;
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lds_pos<true>(lane & 48);
	if (tidw < 16) { sD[tidw] = p.dx[tidw]; sS[tidw] = p.stiff[tidw]; }
#pragma unroll
	for (int t = 0; t < NTL; t++) { rmtab[t * 64 + tidw] = tlog[t]; ((unsigned long long *)(rmtab + 256))[t * 64 + tidw] = texp[t]; }
	if (BLK) {
		lds_barrier();      // the tables are the workgroup's
		if (idle) return;      // (a wavefront that has ended no longer counts at the workgroup's barriers)
	} else wave_sync();      // operator and exp / log tables are this wavefront's own
	// ---- (u x zeta)_xi of the level below the segment (:1011-1037, the xi component alone) and xi_dot of the segment's first interface
	double ucz_prev = 0.0, xlo = 0.0;
	if (k0 > 0) {
		const int kh = k0 - 1;
		double ca2, cb2, cx2, dra, drb;
		met_eval(mc, etac[kh], ca2, cb2, cx2, dra, drb);
		double ux = 0.0; ux += OPK(TMX_OP_INTERP_REDGE_TO_NODE, kh, 0) * wH; ux += OPK(TMX_OP_INTERP_REDGE_TO_NODE, kh, 1) * w0;
		const double conUa = c2a0 * uM + c2a1 * vM + ca2 * ux;
		const double conUb = c2a1 * uM + c2b1 * vM + cb2 * ux;
		s[2][lp] = ux;
		wave_sync();
		double daux = 0, dbux = 0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			daux += s[2][eb + 4 * t + j] * sD[t * 4 + i];
			dbux += s[2][eb + 4 * i + t] * sD[t * 4 + j];
		}
		daux *= ida; dbux *= idb;
		ucz_prev = -conUa * daux - conUb * dbux;
		double ue = 0.0, ve = 0.0;
		ue += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k0, -1) * uM; ue += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k0, 0) * u0;
		ve += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k0, -1) * vM; ve += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k0, 0) * v0;
		double e0, e1, e2, ea, eb_;
		met_eval(mc, etac[L + k0], e0, e1, e2, ea, eb_);
		xlo = e0 * ue + e1 * ve + e2 * w0;
		wave_sync();
	}
	double uh0 = 0.0, vh0 = 0.0;      // H-only U, V of level 0 (bottom boundary condition of W, :1615-1638)
	HW_MARK(0);

	constexpr int NX = NT > 0 ? NT : 1;
	for (int kl = k0; kl < k1; kl++) {
		// (the level index through readfirstlane: opaque to the loop optimiser, which would otherwise turn every  base + k * stride + lane  into a
		// per-lane 64-bit induction pointer -- two registers per stream, 10 + 10 NT of them -- instead of scalar base + 32-bit lane offset)
		const int k = WAVE_UNIFORM(kl);
		// (and the lane offsets redefined per level: hoisted out of the loop, base + lane becomes one 64-bit register pair per source instance)
		asm volatile("" : "+v"(cb), "+v"(gd), "+v"(cx));
		// ---- loads of the next level's window and of this level's combination terms: in flight during this level's arithmetic
		double uN = uP, vN = vP, wN = 0.0, tN = 0.0, rN = 0.0;
		if (k + 1 < k1 || (FV && UD && k + 2 < L)) {      // (FV with UD: level k + 2 enters the DiffDiff stencil of a segment's last level too)
			const int kpp = (k + 2 < L) ? k + 2 : L - 1;
			uN = ldu(xuv + TMX_SLAB_U(L, kpp) * SX, cx); vN = ldu(xuv + TMX_SLAB_V(L, kpp) * SX, cx);
		}
		if (k + 1 < k1) {
			wN = ldu(xin + TMX_SLAB_W(L, k + 2) * SX, cx);
			tN = ldu(xin + TMX_SLAB_T(L, k + 1) * SX, cx); rN = ldu(xin + TMX_SLAB_R(L, k + 1) * SX, cx);
		}
		double xU[NX], xV[NX], xT[NX], xR[NX], xW[NX];
		terms_load<NT, PM, DM>(uvt.src, dmask, TMX_SLAB_U(L, k), SS, NSD, cb, gd, xU);
		terms_load<NT, PM, DM>(uvt.src, dmask, TMX_SLAB_V(L, k), SS, NSD, cb, gd, xV);
		terms_load<NT, PM, DM>(xbase.src, dmask, TMX_SLAB_T(L, k), SS, NSD, cb, gd, xT);
		terms_load<NT, PM, DM>(xbase.src, dmask, TMX_SLAB_R(L, k), SS, NSD, cb, gd, xR);
		terms_load<NT, PM, DM>(xbase.src, dmask, TMX_SLAB_W(L, (k > 0) ? k : 1), SS, NSD, cb, gd, xW);      // (interface 0 has no base: boundary condition)
		double rfT = 0.0, rfW = 0.0, hdU = 0.0, hdV = 0.0, ddU = 0.0, ddV = 0.0;      // UD: the reference's rho*theta, W of this level / interface, its stored U, V terms
		if (UD) {
			rfT = ldu(ud.xref + TMX_SLAB_T(L, k) * NS, cb); rfW = ldu(ud.xref + TMX_SLAB_W(L, k) * NS, cb);
			hdU = ldu(ud.xrefd + TMX_REFD_HU(L, k) * NS, cb); hdV = ldu(ud.xrefd + TMX_REFD_HV(L, k) * NS, cb);
			if (FV && ud.udv) { ddU = ldu(ud.xrefd + TMX_REFD_DU(L, k) * NS, cb); ddV = ldu(ud.xrefd + TMX_REFD_DV(L, k) * NS, cb); }
		}
		__builtin_amdgcn_sched_barrier(0);
		HW_MARK(1);

		// ---- level k, pointwise (:884-982)
		double ca2, cb2, cx2, dra, drb;
		met_eval(mc, etac[k], ca2, cb2, cx2, dra, drb);
		double ux = 0.0; ux += OPK(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * w0; ux += OPK(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * w1;
		const double conUa = c2a0 * u0 + c2a1 * v0 + ca2 * ux;
		const double conUb = c2a1 * u0 + c2b1 * v0 + cb2 * ux;
		const double conUx = ca2 * u0 + cb2 * v0 + cx2 * ux;
		const double dm = OPK(TMX_OP_DIFF_NODE_TO_NODE, k, -1), d0 = OPK(TMX_OP_DIFF_NODE_TO_NODE, k, 0), dp = OPK(TMX_OP_DIFF_NODE_TO_NODE, k, 1);
		double dxua = 0.0; dxua += dm * uM; dxua += d0 * u0; dxua += dp * uP;
		double dxub = 0.0; dxub += dm * vM; dxub += d0 * v0; dxub += dp * vP;
		// xi_dot on interface k+1 (VerticalDynamicsFEM.cpp:816-828), zero at the lid; interface k's is carried from the level below
		double xhi = 0.0;
		if (FV && k + 1 <= L - 1) {
			double ue = 0.0, ve = 0.0;
			ue += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, -1) * u0; ue += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, 0) * uP;
			ve += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, -1) * v0; ve += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k + 1, 0) * vP;
			double e0, e1, e2, ea, eb_;
			met_eval(mc, etac[L + k + 1], e0, e1, e2, ea, eb_);
			xhi = e0 * ue + e1 * ve + e2 * w1;
		}
		const double w_hi = dt * fabs(xhi), w_lo = dt * fabs(xlo);
		const double ke = 0.5 * (conUa * u0 + conUb * v0 + conUx * ux);
		const double exn = exner_from_rhotheta_lds(p, t0, rmtab);
		const double fa = jn * conUa, fb = jn * conUb;
		s[0][lp] = u0;  s[1][lp] = v0;  s[2][lp] = ux;
		s[3][lp] = exn; s[4][lp] = ke;
		s[5][lp] = fa * r0; s[6][lp] = fa * t0;
		s[7][lp] = fb * r0; s[8][lp] = fb * t0;
		wave_sync();
		HW_MARK(2);
		// the twelve 4 x 4 contractions in four groups of three (each sum in the level-parallel kernel's order, t = 0 .. 3); the empty asm
		// statements keep the compiler from issuing all 80 LDS reads at once (160 registers, measured: the loop's pressure peak)
		double daub = 0, daux = 0, dbua = 0, dbux = 0, dap = 0, dbp = 0, dake = 0, dbke = 0;
		double darf = 0, dapf = 0, dbrf = 0, dbpf = 0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int la = eb + 4 * t + j;        // node (t, j)
			const double D = sD[t * 4 + i];
			daub += s[1][la] * D;
			daux += s[2][la] * D;
			dap  += s[3][la] * D;
		}
		asm volatile("" : "+v"(daub), "+v"(daux), "+v"(dap) : : "memory");      // this group's sums exist before the next group's LDS reads are issued
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int la = eb + 4 * t + j;
			const double D = sD[t * 4 + i], S = sS[i * 4 + t];
			dake += s[4][la] * D;
			darf -= s[5][la] * S;
			dapf -= s[6][la] * S;
		}
		asm volatile("" : "+v"(dake), "+v"(darf), "+v"(dapf) : : "memory");      // this group's sums exist before the next group's LDS reads are issued
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int lb = eb + 4 * i + t;        // node (i, t)
			const double D = sD[t * 4 + j];
			dbua += s[0][lb] * D;
			dbux += s[2][lb] * D;
			dbp  += s[3][lb] * D;
		}
		asm volatile("" : "+v"(dbua), "+v"(dbux), "+v"(dbp) : : "memory");      // this group's sums exist before the next group's LDS reads are issued
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int lb = eb + 4 * i + t;
			const double D = sD[t * 4 + j], S = sS[j * 4 + t];
			dbke += s[4][lb] * D;
			dbrf -= s[7][lb] * S;
			dbpf -= s[8][lb] * S;
		}
		asm volatile("" : "+v"(dbke), "+v"(dbrf), "+v"(dbpf) : : "memory");      // this group's sums exist before the next group's LDS reads are issued
		HW_MARK(3);
		daub *= ida; daux *= ida; dbua *= idb; dbux *= idb;
		darf *= ida; dbrf *= idb; dapf *= ida; dbpf *= idb;
		dap *= ida; dbp *= idb; dake *= ida; dbke *= idb;
		// u x zeta (:1011-1037)
		const double jza = dbux - dxub, jzb = dxua - daux, jzx = daub - dbua;
		const double ucza = conUb * jzx - conUx * jzb;
		const double uczb = conUx * jza - conUa * jzx;
		const double uczx = -conUa * daux - conUb * dbux;
		// pointwise updates (:1321-1421)
		double upa = 0.0, upb = 0.0;
		upa += ucza; upb += uczb;
		upa += fj * conUb;
		upb -= fj * conUa;
		const double pgfa = dap * t0 / r0, pgfb = dbp * t0 / r0;
		const double daphi = p.grav * dra, dbphi = p.grav * drb;
		upa -= (pgfa + dake + daphi);
		upb -= (pgfb + dbke + dbphi);
		const double tU = dt * upa, tV = dt * upb;
		// first-order upwind penalty on U,V (VerticalDynamicsFEM.cpp:984-1023): left then right operator
		double tUhi = 0, tVhi = 0, tUlo = 0, tVlo = 0;
		if (FV && k < L - 1) {
			double b = 0.0;
			b += OPK(TMX_OP_PENALTY_LEFT, k, 0) * u0; b += OPK(TMX_OP_PENALTY_LEFT, k, 1) * uP;
			tUhi = b * w_hi;
			b = 0.0;
			b += OPK(TMX_OP_PENALTY_LEFT, k, 0) * v0; b += OPK(TMX_OP_PENALTY_LEFT, k, 1) * vP;
			tVhi = b * w_hi;
		}
		if (FV && k > 0) {
			double b = 0.0;
			b += OPK(TMX_OP_PENALTY_RIGHT, k, -1) * uM; b += OPK(TMX_OP_PENALTY_RIGHT, k, 0) * u0;
			tUlo = b * w_lo;
			b = 0.0;
			b += OPK(TMX_OP_PENALTY_RIGHT, k, -1) * vM; b += OPK(TMX_OP_PENALTY_RIGHT, k, 0) * v0;
			tVlo = b * w_lo;
		}
		const double tR = dt * invJ * (darf + dbrf), tT = dt * invJ * (dapf + dbpf);
		// W on interface k (:1612-1659): (u x zeta)_xi of the levels k - 1 and k
		double tW = 0.0;
		if (k > 0) {
			double v = 0.0;
			v += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * ucz_prev;
			v += OPK(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * uczx;
			tW = dt * v;
		}
		HW_MARK(4);
		// ---- the stage combination (reference order), then base + increments in the level-parallel kernel's order
		const double bU = lin_reduce<NT, PM>(xbase, xU), bV = lin_reduce<NT, PM>(xbase, xV);
		const double bT = lin_reduce<NT, PM>(xbase, xT), bR = lin_reduce<NT, PM>(xbase, xR), bW = lin_reduce<NT, PM>(xbase, xW);
		if (EM) {
			// every copy of a node forms the same prefix (the sources are node-unique): each stores it, to the same entry (one store per node
			// -- the representative copy alone -- and the later destination's position resolved by a uniform switch instead of selects both
			// measured the same: what the prefix costs is its 61 MB of stores, 0.02 ms per step of the 0.05 the fourth stage saves)
			stu(em.xp + TMX_SLAB_U(L, k) * SS, cb, emit_value<NT>(em, xU)); stu(em.xp + TMX_SLAB_V(L, k) * SS, cb, emit_value<NT>(em, xV));
			stu(em.xp + TMX_SLAB_T(L, k) * SS, cb, emit_value<NT>(em, xT)); stu(em.xp + TMX_SLAB_R(L, k) * SS, cb, emit_value<NT>(em, xR));
			if (k > 0) stu(em.xp + TMX_SLAB_W(L, k) * SS, cb, emit_value<NT>(em, xW));
		}
		HW_MARK(5);
		double outU = bU + tU, outV = bV + tV;
		const double uh = outU, vh = outV;      // H-only values (levels 0, 1: the bottom boundary condition below)
		if (FV && !UD && k < L - 1) { outU += tUhi; outV += tVhi; }
		if (FV && !UD && k > 0) { outU += tUlo; outV += tVlo; }
		const double outR = bR - tR;
		double outT = bT - tT, outW = bW + tW;
		if constexpr (UD) {
			// horizontal uniform diffusion of rho*theta, W (relative to the reference state) and of U, V (divergence / curl form, state and reference
			// apart), k_uniform_diffusion's statements on this level's values; the stage combination's terms are reduced by now (their registers are free)
			wave_sync();
			const double psiT = t0 - rfT, psiW = w0 - rfW;
			s[0][lp] = psiT; s[1][lp] = psiW;
			s[2][lp] = u0; s[3][lp] = v0; s[4][lp] = j2d * (c2a0 * u0 + c2a1 * v0); s[5][lp] = j2d * (c2a1 * u0 + c2b1 * v0);
			wave_sync();
			double daT = 0, dbT = 0, daW = 0, dbW = 0;
			double dUb = 0, dbUa = 0, dajua = 0, dbjub = 0;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
				const double Da = sD[t * 4 + i], Db = sD[t * 4 + j];
				daT += s[0][la] * Da; dbT += s[0][lb] * Db;
				daW += s[1][la] * Da; dbW += s[1][lb] * Db;
				dUb += s[3][la] * Da; dbUa += s[2][lb] * Db;
				dajua += s[4][la] * Da; dbjub += s[5][lb] * Db;
			}
			daT *= ida; dbT *= idb; daW *= ida; dbW *= idb;
			dUb *= ida; dbUa *= idb; dajua *= ida; dbjub *= idb;
			const double invJ2 = 1.0 / j2d;
			const double dv = (dajua + dbjub) * invJ2, cl = (dUb - dbUa) * invJ2;
			wave_sync();
			s[0][lp] = jn * (c2a0 * daT + c2a1 * dbT); s[1][lp] = jn * (c2a1 * daT + c2b1 * dbT);
			s[2][lp] = je * (c2a0 * daW + c2a1 * dbW); s[3][lp] = je * (c2a1 * daW + c2b1 * dbW);
			s[4][lp] = dv; s[5][lp] = cl;
			wave_sync();
			double uaT = 0, ubT = 0, uaW = 0, ubW = 0, dad = 0, dbd = 0, dac = 0, dbc = 0;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
				const double Sa = sS[i * 4 + t], Sb = sS[j * 4 + t];
				uaT += s[0][la] * Sa; ubT += s[1][lb] * Sb;
				uaW += s[2][la] * Sa; ubW += s[3][lb] * Sb;
				dad -= Sa * s[4][la]; dbd -= Sb * s[4][lb];
			}
			asm volatile("" : "+v"(uaT), "+v"(ubT), "+v"(uaW), "+v"(ubW) : : "memory");
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
				const double Sa = sS[i * 4 + t], Sb = sS[j * 4 + t];
				dac -= Sa * s[5][la]; dbc -= Sb * s[5][lb];
			}
			uaT *= ida; ubT *= idb; uaW *= ida; ubW *= idb;
			dad *= ida; dbd *= idb; dac *= ida; dbc *= idb;
			const double xW = dt * (1.0 / je) * ud.kv * (uaW + ubW);
			if (k == 0) xW0 = xW; else outW = outW - xW;
			outT = outT - dt * (1.0 / jn) * ud.ks * (uaT + ubT);
			const double nkv = -ud.kv;
			const double upa = nkv * dad - nkv * j2d * (c2a1 * dac + c2b1 * dbc);
			const double upb = nkv * dbd + nkv * j2d * (c2a0 * dac + c2a1 * dbc);
			outU -= dt * upa; outU -= dt * hdU;      // (hdU, hdV: the reference's increments upar, upbr, k_ref_diffusion_terms)
			outV -= dt * upb; outV -= dt * hdV;
			wave_sync();
			if constexpr (FV) {
				// V.StepExplicit on U, V after the horizontal diffusion, as the reference orders them: upwind penalty, then the vertical uniform diffusion
				if (k < L - 1) { outU += tUhi; outV += tVhi; }
				if (k > 0) { outU += tUlo; outV += tVlo; }
				if (ud.udv) {
					double ddu = 0.0, ddv = 0.0;
					if (k >= 2) { const double c = OPK(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, -2); ddu += c * uMM; ddv += c * vMM; }
					if (k >= 1) { const double c = OPK(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, -1); ddu += c * uM; ddv += c * vM; }
					{ const double c = OPK(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, 0); ddu += c * u0; ddv += c * v0; }
					if (k + 1 < L) { const double c = OPK(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, 1); ddu += c * uP; ddv += c * vP; }
					if (k + 2 < L) { const double c = OPK(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, 2); ddu += c * uN; ddv += c * vN; }
					outU += dt * ud.cfv * (ddu - ddU);
					outV += dt * ud.cfv * (ddv - ddV);
				}
			}
		}
		if constexpr (!UQ) {
			// element-major: the thread's own column
			if (here) {
				stu(xup + TMX_SLAB_U(L, k) * NS, cb, outU); stu(xup + TMX_SLAB_V(L, k) * NS, cb, outV);
				stu(xup + TMX_SLAB_R(L, k) * NS, cb, outR); stu(xup + TMX_SLAB_T(L, k) * NS, cb, outT);
				if (k > 0) stu(xup + TMX_SLAB_W(L, k) * NS, cb, outW);
			}
			if (k == 0) { uh0 = uh; vh0 = vh; }
			if (k == 1) {
				// bottom boundary: W of interface 0 from the H-only U, V of levels 0 and 1 extrapolated to the surface (:1615-1638)
				double ub0 = 0.0, vb0 = 0.0;
				ub0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * uh0; ub0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * uh;
				vb0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * vh0; vb0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * vh;
				double e0, e1, e2, ea, eb_;
				met_eval(mc, etac[L], e0, e1, e2, ea, eb_);
				double wb = -(e0 * ub0 + e1 * vb0) / e2;
				if (UD) wb = wb - xW0;      // (the surface's diffusion increment, formed at level 0 from the initial W there)
				if (here) stu(xup + TMX_SLAB_W(L, 0) * NS, cb, wb);
			}
		} else if constexpr (BLK) {
			// one exchange per level for the whole block; W of the surface (known once level 1's H-only U, V are) and of the lid ride along
			double extra = 0.0; int xslab = TMX_SLAB_W(L, 0); unsigned vmask = (k > 0) ? 31u : 15u;
			if (k == 0) { uh0 = uh; vh0 = vh; }
			if (k == 1) {
				double ub0 = 0.0, vb0 = 0.0;
				ub0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * uh0; ub0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * uh;
				vb0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * vh0; vb0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * vh;
				double e0, e1, e2, ea, eb_;
				met_eval(mc, etac[L], e0, e1, e2, ea, eb_);
				extra = -(e0 * ub0 + e1 * vb0) / e2; vmask |= 32u;
			}
			if (k == L - 1) {
				double xWl[NX];
				terms_load<NT, PM, DM>(xbase.src, dmask, TMX_SLAB_W(L, L), SS, NSD, cb, gd, xWl);
				extra = lin_reduce<NT, PM>(xbase, xWl); xslab = TMX_SLAB_W(L, L); vmask |= 32u;
				if (EM) stu(em.xp + TMX_SLAB_W(L, L) * SS, cb, emit_value<NT>(em, xWl));
			}
			const double v[6] = { outU, outV, outR, outT, outW, extra };
			const int slab[6] = { TMX_SLAB_U(L, k), TMX_SLAB_V(L, k), TMX_SLAB_R(L, k), TMX_SLAB_T(L, k), TMX_SLAB_W(L, k), xslab };
			block_seam_store<6>(p, xs[k & 1], pos, sdst, sred, v, xup, slab, vmask);
		} else {
			{
				const double v[5] = { outU, outV, outR, outT, outW };
				const int slab[5] = { TMX_SLAB_U(L, k), TMX_SLAB_V(L, k), TMX_SLAB_R(L, k), TMX_SLAB_T(L, k), TMX_SLAB_W(L, k) };
				seam_store<5, RL>(p, &s[0], lane, sdst, sred, v, xup, slab, (k > 0) ? 5 : 4);
			}
			if (k == 0) { uh0 = uh; vh0 = vh; }
			if (k == 1) {
				// bottom boundary: W of interface 0 from the H-only U, V of levels 0 and 1 extrapolated to the surface (:1615-1638)
				wave_sync();
				double ub0 = 0.0, vb0 = 0.0;
				ub0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * uh0; ub0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * uh;
				vb0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 0) * vh0; vb0 += OPK(TMX_OP_INTERP_NODE_TO_REDGE, 0, 1) * vh;
				double e0, e1, e2, ea, eb_;
				met_eval(mc, etac[L], e0, e1, e2, ea, eb_);
				const double v1[1] = { -(e0 * ub0 + e1 * vb0) / e2 };
				const int slab1[1] = { TMX_SLAB_W(L, 0) };
				seam_store<1, RL>(p, &s[0], lane, sdst, sred, v1, xup, slab1, 1);
			}
		}
		wave_sync();
		HW_MARK(6);
		// ---- slide the window
		if (FV && UD) { uMM = uM; vMM = vM; }
		uM = u0; u0 = uP; uP = uN; vM = v0; v0 = vP; vP = vN;
		w0 = w1; w1 = wN; t0 = tN; r0 = rN;
		ucz_prev = uczx; xlo = xhi;
	}
	if (!BLK && k1 == L) {
		// the rigid lid: the H step leaves W there alone; the base's value goes through the seam exchange like every other result
		double xW[NX];
		terms_load<NT, PM, DM>(xbase.src, dmask, TMX_SLAB_W(L, L), SS, NSD, cb, gd, xW);
		const double v1[1] = { lin_reduce<NT, PM>(xbase, xW) };
		if constexpr (!UQ) {
			double wl = v1[0];
			if constexpr (UD) {
				// the lid's W diffuses too (k_uniform_diffusion's interface L: W alone); w0 holds the initial W of the lid after the last slide
				const double psiW = w0 - ldu(ud.xref + TMX_SLAB_W(L, L) * NS, cb);
				wave_sync();
				s[1][lp] = psiW;
				wave_sync();
				double daW = 0, dbW = 0;
#pragma unroll
				for (int t = 0; t < 4; t++) { daW += s[1][eb + 4 * t + j] * sD[t * 4 + i]; dbW += s[1][eb + 4 * i + t] * sD[t * 4 + j]; }
				daW *= ida; dbW *= idb;
				wave_sync();
				s[2][lp] = je * (c2a0 * daW + c2a1 * dbW); s[3][lp] = je * (c2a1 * daW + c2b1 * dbW);
				wave_sync();
				double uaW = 0, ubW = 0;
#pragma unroll
				for (int t = 0; t < 4; t++) { uaW += s[2][eb + 4 * t + j] * sS[i * 4 + t]; ubW += s[3][eb + 4 * i + t] * sS[j * 4 + t]; }
				uaW *= ida; ubW *= idb;
				wl = wl - dt * (1.0 / je) * ud.kv * (uaW + ubW);
			}
			if (here) stu(xup + TMX_SLAB_W(L, L) * NS, cb, wl);
			HW_MARK(6); HW_END(NT); return;
		}
		if (EM) stu(em.xp + TMX_SLAB_W(L, L) * SS, cb, emit_value<NT>(em, xW));
		const int slab1[1] = { TMX_SLAB_W(L, L) };
		seam_store<1, RL>(p, &s[0], lane, sdst, sred, v1, xup, slab1, 1);
	}
	HW_MARK(6);
	HW_END(NT);
}

// Does this launch have a walk form?  (node-unique layout with the vertical part fused in, closed-form metric, no stored prefix, up to
// eight terms; segments of at least two levels so that levels 0 and 1 -- the bottom boundary condition -- belong to one wavefront)
bool tmxk_h_walk_ok(const tmx_engine * e, const KParams & p, int fuse_v, int nterms, bool emit) {
	const bool uq = p.t_ucol != nullptr;
	(void)emit;      // (a stored prefix: the walk's own instantiation)
	// node-unique layout: with the vertical part fused in (what the steppers run); element-major layout: with or without it
	return (uq ? fuse_v != 0 : true) && p.closed && nterms != 1 && nterms <= 8 && e->opt_h_walk != 0 && p.L >= 2 && !e->use_mfma;
}

// Segments per column.  One wavefront per segment, two resident per SIMD (256 registers), three for the plain stage and the two-term
// combination (168).  A segment costs its wavefront a prologue (geometry, tables, first window, halo level) worth about 1.3 levels
// (profiles/r06_walk_phase_timing.txt: 9 600 cycles against 7 500 per level), and the launch takes as many rounds as its wavefronts need of
// the chip's slots: the count that minimises  rounds x (1.3 + levels per segment).  ne30: 3 segments (4 050 wavefronts in 4 096 slots, two
// rounds) for the combinations, 2 for the plain stage; the 8-rank share of ne30 (169 tiles): 12; measured beside it: ne30 3 / 4 / 6 segments
// 0.404 / 0.430 / 0.415 ms per step, the 8-rank share 6 / 10 / 15 segments 0.178 / 0.148 / 0.129 against 0.151 of the level-parallel kernel.
int tmxk_h_walk_segments(const tmx_engine * e, const KParams & p, int ntiles, int nterms, int waves_per_simd) {
	int nseg = 1;
	if (waves_per_simd != 4 && e->opt_h_walk < 0 && e->opt_h_walk > -1000) nseg = -e->opt_h_walk;
	else if (waves_per_simd == 4 && e->opt_hv_walk < 0) nseg = -e->opt_hv_walk;      // (the hyperviscosity walk's call)
	else {
		static int cus = 0;      // (one device kind per process)
		if (!cus) { int dev = 0, n = 0; cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
		const double slots = (double)cus * 4 * (waves_per_simd ? waves_per_simd : (nterms <= 2 ? 3 : 2));
		double best = 1e300;
		for (int n = 1; n <= p.L / 2 || n == 1; n++) {
			const double rounds = (double)ntiles * n / slots;
			const double t = ((rounds <= 1.02) ? 1.0 : (double)(long long)(rounds + 0.98)) * (1.3 + (double)p.L / n);
			if (t < best) { best = t; nseg = n; }
		}
	}
	if (nseg > p.L / 2) nseg = p.L / 2;
	if (nseg < 1) nseg = 1;
	return nseg;
}

// Does the walk apply the horizontal uniform diffusion itself?  (element-major layout: the uniform-diffusion configurations; with V.StepExplicit's
// U, V part behind it -- fuse_v -- from option h_walk_udiff = 2 on)
bool tmxk_h_walk_fuses_udiff(const tmx_engine * e, const KParams & p, int fuse_v, int nterms) {
	return e->udiff && !p.t_ucol && (fuse_v ? e->opt_h_walk_udiff >= 2 : e->opt_h_walk_udiff != 0) && tmxk_h_walk_ok(e, p, fuse_v, nterms, false);
}

// the stored reference terms (UdiffArgs::xrefd), formed when the reference state has changed since they were
static const double * ref_diffusion_terms(tmx_engine * e, const KParams & p) {
	if (!e->d_refd) {
		const size_t rb = (size_t)4 * p.L * p.NS * sizeof(double);
		if (hipMalloc((void **)&e->d_refd, rb) != hipSuccess) { e->d_refd = nullptr; return nullptr; }
		hipMemsetAsync(e->d_refd, 0, rb, e->stream);
		e->hbm_bytes += rb; e->refd_valid = false;
	}
	if (!e->refd_valid) {
		KParams q = p; q.quads = nullptr;      // every tile, whatever list the stage runs on
		hipLaunchKernelGGL(k_ref_diffusion_terms, dim3(p.NS / 64, p.L), dim3(64), 0, e->stream, q, (const double *)e->d_ref, e->d_refd, e->cfg.uniform_diffusion_vector);
		e->refd_valid = true;
	}
	return e->d_refd;
}

// Before a step is captured into a graph: what the walk would allocate or compute on first use (neither is allowed while the stream captures)
void tmxk_h_walk_prepare(tmx_engine * e, const KParams & p) {
	if (e->udiff && !e->refd_valid && e->d_ref && tmxk_h_walk_fuses_udiff(e, p, 0, 0)) (void)ref_diffusion_terms(e, p);
}

void tmxk_h_walk(tmx_engine * e, const KParams & p, const double * xin, const double * xbase_ptr, double * xup, double dt, int fuse_v,
	int nterms, const double * const * src, const double * coef, int premul, const double * xin_uv, const double * xbase_uv, const double * const * src_uv, unsigned dmask,
	const EmitTerms * emit) {
	LinTerms xbase;
	UvTerms uvt;
	xbase.n = nterms; xbase.premul = premul; xbase.dmask = dmask;
	xbase.src[0] = xbase_ptr; xbase.coef[0] = 0.0;
	uvt.xin = xin_uv ? xin_uv : xin; uvt.src[0] = xbase_uv ? xbase_uv : xbase_ptr;
	for (int m = 0; m < nterms; m++) { xbase.src[m] = src[m]; xbase.coef[m] = coef[m]; uvt.src[m] = src_uv ? src_uv[m] : src[m]; }
	for (int m = (nterms > 0 ? nterms : 1); m < 12; m++) { xbase.src[m] = xbase.src[0]; xbase.coef[m] = 0.0; uvt.src[m] = uvt.src[0]; }
	const int ntile = p.quads ? e->launch_tiles : p.NS / 64;
	// (the kernels that apply the uniform diffusion hold 190 - 256 registers whatever the number of terms: two wavefronts per SIMD)
	// (three wavefronts per SIMD -- at most 168 registers -- for the plain stage and the two-term combination without a coefficient of its own)
	const int nseg = tmxk_h_walk_segments(e, p, ntile, nterms, ((!p.t_ucol && tmxk_h_walk_fuses_udiff(e, p, fuse_v, nterms)) || (nterms == 2 && premul)) ? 2 : 0);
	KParams q = p; q.u_ntiles = ntile;
	if (!p.t_ucol) {
		// element-major layout: 2-D launch (tile list or all tiles) x segments
		EmitTerms em1; em1.xp = nullptr; em1.first = -1; for (int m = 0; m < 12; m++) em1.coef[m] = 0.0;
		const dim3 grd(ntile, nseg), blk(64);
		const bool udf = tmxk_h_walk_fuses_udiff(e, p, fuse_v, nterms);
		// (fuse_v with uniform diffusion: the vertical diffusion of U, V rides along where k_v_explicit_slide would apply it -- the fully explicit mode)
		const UdiffArgs ud0 = { nullptr, 0.0, 0.0, nullptr, 0.0, 0 }, ud1 = { (const double *)e->d_ref, e->cfg.uniform_diffusion_scalar, e->cfg.uniform_diffusion_vector,
			udf ? ref_diffusion_terms(e, p) : nullptr, e->cfg.uniform_diffusion_vector / (e->cfg.ztop * e->cfg.ztop), (e->fully_explicit && !e->opt_udv_separate) ? 1 : 0 };
#define LAUNCH_D2(NT_, PM_) do { \
			if (fuse_v && udf) hipLaunchKernelGGL((k_h_walk<NT_, PM_, false, false, false, false, true, true>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em1, ud1); \
			else if (fuse_v) hipLaunchKernelGGL((k_h_walk<NT_, PM_, false, false, false, false, true>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em1, ud0); \
			else if (udf) hipLaunchKernelGGL((k_h_walk<NT_, PM_, false, false, false, false, false, true>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em1, ud1); \
			else hipLaunchKernelGGL((k_h_walk<NT_, PM_, false, false, false, false, false>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em1, ud0); } while (0)
#define LAUNCH_D(NT_) do { if (NT_ == 0 || premul) LAUNCH_D2(NT_, true); else LAUNCH_D2(NT_, false); } while (0)
		switch (nterms) {
			case 0: LAUNCH_D(0); break;
			case 2: LAUNCH_D(2); break;
			case 3: LAUNCH_D(3); break;
			case 4: LAUNCH_D(4); break;
			case 5: LAUNCH_D(5); break;
			case 6: LAUNCH_D(6); break;
			case 7: LAUNCH_D(7); break;
			default: LAUNCH_D(8); break;
		}
#undef LAUNCH_D
#undef LAUNCH_D2
		return;
	}
	// block kernels: the launch is over the blocks of the thread order (four wavefronts each), the seams between a block's wavefronts never reach the partial slots
	const bool blocks = tmxu_blocks_on(e, 1) && p.L >= 3;
	const int nblk = p.bquads ? e->launch_blocks : e->u.nblocks;
	if (blocks) q.u_ntiles = nblk;
	e->u.slots_by_blocks = blocks;
	const dim3 blk(64, blocks ? 4 : 1), grd = wg_grid(true, q, blocks ? nblk : ntile, nseg);
	EmitTerms em0; em0.xp = nullptr; em0.first = -1; for (int m = 0; m < 12; m++) em0.coef[m] = 0.0;
	const UdiffArgs udn = { nullptr, 0.0, 0.0, nullptr, 0.0, 0 };
	// (the prefix of a later stage's combination: no own coefficient, three to eight terms, node-unique sources, wavefront-level kernel)
	const bool emit_ok = emit && emit->xp && !premul && !dmask && !blocks && nterms >= 3 && nterms <= 8;
	const EmitTerms em = emit_ok ? *emit : em0;
#define LAUNCH_W3(NT_, PM_, DM_) do { \
		if (emit_ok && NT_ >= 3 && !PM_ && !DM_) hipLaunchKernelGGL((k_h_walk<(NT_ >= 3 ? NT_ : 3), false, false, false, true>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em, udn); \
		else if (TMX_EXP && blocks) hipLaunchKernelGGL((k_h_walk<NT_, PM_, DM_, TMX_EXP != 0>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em0, udn); \
		else hipLaunchKernelGGL((k_h_walk<NT_, PM_, DM_, false>), grd, blk, 0, e->stream, q, xin, xbase, xup, dt, uvt, nseg, em0, udn); } while (0)
#define LAUNCH_W2(NT_, PM_) do { if (dmask) LAUNCH_W3(NT_, PM_, true); else LAUNCH_W3(NT_, PM_, false); } while (0)
#define LAUNCH_W(NT_) do { if (NT_ == 0 || premul) LAUNCH_W2(NT_, true); else LAUNCH_W2(NT_, false); } while (0)
	switch (nterms) {
		case 0: LAUNCH_W(0); break;
		case 2: LAUNCH_W(2); break;
		case 3: LAUNCH_W(3); break;
		case 4: LAUNCH_W(4); break;
		case 5: LAUNCH_W(5); break;
		case 6: LAUNCH_W(6); break;
		case 7: LAUNCH_W(7); break;
		default: LAUNCH_W(8); break;
	}
#undef LAUNCH_W3
#undef LAUNCH_W
#undef LAUNCH_W2
}

// ---------------------------------------------------------------------------------------------
// The hyperviscosity pass of the node-unique layout as a walk: k_hypervis<false, UQ, HB> (tmx_k_horizontal.hip; ApplyScalarHyperdiffusion /
// ApplyVectorHyperdiffusion, HorizontalDynamicsFEM.cpp:1867-2414), statement for statement, with a wavefront walking a segment of the L + 1
// levels (the last one holds W of the lid alone): the nine per-column geometry values, the seam tables and the viscosity coefficients once
// per segment instead of once per level (11 of the level-parallel kernel's 21 loads per thread and level), the next level's five fields and
// five base values in flight during this level's two rounds of contractions.  No coupling between levels: any split into segments will do.
template <bool HB>
__global__ __launch_bounds__(64, 4) void k_hv_walk(KParams p, const double * __restrict__ xsrc, const double * __restrict__ xbase, double * __restrict__ xout,
	double dt, double nu_s_in, double nu_d_in, double nu_v_in, int scale, int nseg)
{
	constexpr int RL = TMX_UQ_ROWLEN(true);
	__shared__ double s[8][RL];
	__shared__ double sD[16], sS[16];
	const int lane = tile_lane<true>(), lp = lds_pos<true>(lane);
	if (threadIdx.x < 16) { sD[threadIdx.x] = p.dx[threadIdx.x]; sS[threadIdx.x] = p.stiff[threadIdx.x]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS, SS = (size_t)p.NSS;
	int tix, seg;
	if (!wg_tile<true>(p, nseg, tix, seg)) return;
	const int tile = p.quads ? p.quads[tix] : tix;
	const int k0 = WAVE_UNIFORM((seg * (L + 1)) / nseg), k1 = WAVE_UNIFORM(((seg + 1) * (L + 1)) / nseg);
	const int col = tile * 64 + lane;
	const int uc0 = tile_ucol(p, tile, lane);
	const bool here = uc0 >= 0;
	unsigned cb = (unsigned)max(uc0, 0) * 8u;
	const unsigned gb = (unsigned)col * 8u;
	// per-column constants (padding lanes: the level-parallel kernel's neutral values)
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, j2d = 1, jn = 1, je = 1;
	if (here) {
		c2a0 = ldu(p.g2d + G2_C2A0 * NS, gb); c2a1 = ldu(p.g2d + G2_C2A1 * NS, gb); c2b1 = ldu(p.g2d + G2_C2B1 * NS, gb);
		j2d = ldu(p.g2d + G2_J2D * NS, gb); jn = ldu(p.g2d + G2_JN * NS, gb); je = ldu(p.g2d + G2_JE * NS, gb);
	}
	const double ida = ldu(p.g2d + G2_IDA * NS, gb), idb = ldu(p.g2d + G2_IDB * NS, gb);
	const double nsc = scale ? ldu(p.g2d + G2_NUS * NS, gb) : 1.0;
	const double nu_s = nu_s_in * nsc, nu_d = nu_d_in * nsc, nu_v = nu_v_in * nsc;
	const int sdst = p.t_sdst[col], sred = p.t_sred[col];
	const double invJ2 = 1.0 / j2d, invJ = 1.0 / jn, rje = 1.0 / je;
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lds_pos<true>(lane & 48);
	// a level's fields: W of interface k and, below the lid, rho*theta, rho, U, V of level k (a level >= L has W alone: the others re-read level L - 1,
	// unused); padding lanes read node 0 and store nothing.  Every load unconditional, so that the loads in flight can be counted (s_waitcnt vmcnt(n)).
	struct Lev { double W, T, R, U, V, bW, bT, bR, bU, bV; };
	auto load_level = [&](Lev & v, int k) {
		const int kw = min(k, L), kn = min(k, L - 1);
		v.W = ldu(xsrc + (size_t)TMX_SLAB_W(L, kw) * SS, cb);
		v.T = ldu(xsrc + (size_t)TMX_SLAB_T(L, kn) * SS, cb); v.R = ldu(xsrc + (size_t)TMX_SLAB_R(L, kn) * SS, cb);
		v.U = ldu(xsrc + (size_t)TMX_SLAB_U(L, kn) * SS, cb); v.V = ldu(xsrc + (size_t)TMX_SLAB_V(L, kn) * SS, cb);
		if (HB) {
			v.bW = ldu(xbase + (size_t)TMX_SLAB_W(L, kw) * SS, cb);
			v.bT = ldu(xbase + (size_t)TMX_SLAB_T(L, kn) * SS, cb); v.bR = ldu(xbase + (size_t)TMX_SLAB_R(L, kn) * SS, cb);
			v.bU = ldu(xbase + (size_t)TMX_SLAB_U(L, kn) * SS, cb); v.bV = ldu(xbase + (size_t)TMX_SLAB_V(L, kn) * SS, cb);
		} else { v.bW = v.bT = v.bR = v.bU = v.bV = 0.0; }
	};
	// Level k with its fields in `c`; `nx` holds the level after it and becomes `c`; the loads of level kload go into `ld` (= nx) at the top.  The fields of the next level are made to ARRIVE before this level's stores are issued: the stores sit in
	// divergent branches (seam_store), the compiler cannot count them and waits for everything in flight wherever a load issued before them is
	// first used behind them -- that is for the stores' own completion, once per level (measured: the wait at the loop's end).
	auto level = [&](int kk, Lev & c, Lev & nx, Lev & ld, int kload) {
		asm volatile("" : "+v"(cb));      // (lane offset redefined per level: base + lane stays scalar base + 32-bit offset, cf. k_h_walk)
		const bool actn = kk < L;
		load_level(ld, kload);
		__builtin_amdgcn_sched_barrier(0);
		const double psiT = actn ? c.T : 0.0, psiR = actn ? c.R : 0.0, ua = actn ? c.U : 0.0, ub = actn ? c.V : 0.0, psiW = c.W;
		const double cua = c2a0 * ua + c2a1 * ub;
		const double cub = c2a1 * ua + c2b1 * ub;
		s[0][lp] = psiT; s[1][lp] = psiR; s[2][lp] = psiW;
		s[3][lp] = ua;   s[4][lp] = ub;
		s[5][lp] = j2d * cua; s[6][lp] = j2d * cub;
		wave_sync();
		double daT = 0, dbT = 0, daR = 0, dbR = 0, daW = 0, dbW = 0, daub = 0, dbua = 0, dajua = 0, dbjub = 0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
			const double Da = sD[t * 4 + i], Db = sD[t * 4 + j];
			daT += s[0][la] * Da; dbT += s[0][lb] * Db;
			daR += s[1][la] * Da; dbR += s[1][lb] * Db;
			daW += s[2][la] * Da; dbW += s[2][lb] * Db;
			daub += s[4][la] * Da; dbua += s[3][lb] * Db;
			dajua += s[5][la] * Da; dbjub += s[6][lb] * Db;
		}
		daT *= ida; dbT *= idb; daR *= ida; dbR *= idb; daW *= ida; dbW *= idb;
		daub *= ida; dbua *= idb; dajua *= ida; dbjub *= idb;
		const double dv = (dajua + dbjub) * invJ2, cl = (daub - dbua) * invJ2;
		wave_sync();
		s[0][lp] = jn * (c2a0 * daT + c2a1 * dbT); s[1][lp] = jn * (c2a1 * daT + c2b1 * dbT);
		s[2][lp] = jn * (c2a0 * daR + c2a1 * dbR); s[3][lp] = jn * (c2a1 * daR + c2b1 * dbR);
		s[4][lp] = je * (c2a0 * daW + c2a1 * dbW); s[5][lp] = je * (c2a1 * daW + c2b1 * dbW);
		s[6][lp] = dv; s[7][lp] = cl;
		wave_sync();
		double uaT = 0, ubT = 0, uaR = 0, ubR = 0, uaW = 0, ubW = 0, dad = 0, dbd = 0, dac = 0, dbc = 0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int la = eb + 4 * t + j, lb = eb + 4 * i + t;
			const double Sa = sS[i * 4 + t], Sb = sS[j * 4 + t];
			uaT += s[0][la] * Sa; ubT += s[1][lb] * Sb;
			uaR += s[2][la] * Sa; ubR += s[3][lb] * Sb;
			uaW += s[4][la] * Sa; ubW += s[5][lb] * Sb;
			dad -= Sa * s[6][la]; dbd -= Sb * s[6][lb];
			dac -= Sa * s[7][la]; dbc -= Sb * s[7][lb];
		}
		uaT *= ida; ubT *= idb; uaR *= ida; ubR *= idb; uaW *= ida; ubW *= idb;
		dad *= ida; dbd *= idb; dac *= ida; dbc *= idb;
		const double oW = c.bW - dt * rje * nu_s * (uaW + ubW);
		double oT = 0, oR = 0, oU = 0, oV = 0;
		if (actn) {
			oT = c.bT - dt * invJ * nu_s * (uaT + ubT);
			oR = c.bR - dt * invJ * nu_s * (uaR + ubR);
			const double upa = nu_d * dad - nu_v * j2d * (c2a1 * dac + c2b1 * dbc);
			const double upb = nu_d * dbd + nu_v * j2d * (c2a0 * dac + c2a1 * dbc);
			oU = c.bU - dt * upa;
			oV = c.bV - dt * upb;
		}
		wave_sync();
		// the next level's fields are here before this level's stores leave
		c = nx;
		asm volatile("" : "+v"(c.W), "+v"(c.T), "+v"(c.R), "+v"(c.U), "+v"(c.V));
		if (HB) asm volatile("" : "+v"(c.bW), "+v"(c.bT), "+v"(c.bR), "+v"(c.bU), "+v"(c.bV));
		{
			const double v[5] = { oW, oT, oR, oU, oV };
			const int slab[5] = { TMX_SLAB_W(L, kk), TMX_SLAB_T(L, kk), TMX_SLAB_R(L, kk), TMX_SLAB_U(L, kk), TMX_SLAB_V(L, kk) };
			seam_store<5, RL>(p, &s[0], lane, here ? sdst : -1, sred, v, xout, slab, actn ? 5 : 1);
		}
		wave_sync();
	};
	// (measured beside it, ne30: the level after next loaded too -- two staging sets -- 0.110 against 0.108 ms per step for the two passes, and 0.144 where
	// the third set spills; profiles/r06_hv_walk_segments_ab.txt)
	Lev C, N;
	load_level(C, k0);
	wave_sync();
	for (int kl = k0; kl < k1; kl++) {
		const int kk = WAVE_UNIFORM(kl);
		level(kk, C, N, N, min(kk + 1, k1 - 1));
	}
}

bool tmxk_hv_walk_ok(const tmx_engine * e, const KParams & p) { return p.t_ucol != nullptr && e->opt_hv_walk != 0 && !tmxu_blocks_on(e, 2); }

void tmxk_hv_walk(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout, double dt, double nu_s, double nu_d, double nu_v, int scale) {
	const int ntile = p.quads ? e->launch_tiles : p.NS / 64;
	KParams q = p; q.u_ntiles = ntile;
	KParams pl = p; pl.L = p.L + 1;      // (segments over the L + 1 levels)
	const int nseg = tmxk_h_walk_segments(e, pl, ntile, 0, 4);      // (128 registers: four wavefronts per SIMD; ne30: 3 segments = 4 050 wavefronts in 4 096 slots)
	e->u.slots_by_blocks = false;
	const dim3 grd = wg_grid(true, q, ntile, nseg);
	if (xbase) hipLaunchKernelGGL(k_hv_walk<true>, grd, dim3(64), 0, e->stream, q, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, nseg);
	else hipLaunchKernelGGL(k_hv_walk<false>, grd, dim3(64), 0, e->stream, q, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, nseg);
}
