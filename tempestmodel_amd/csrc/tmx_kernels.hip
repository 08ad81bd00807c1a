// tmx_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the spectral-element hot path.
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
#include "tmx_internal.h"
#include "tmx_refmath.h"

#ifndef TMX_H_FUSE_W
#define TMX_H_FUSE_W 1      // W of the interfaces inside a level tile updated by k_h_explicit itself (0: all of them by k_h_w_update)
#endif
#ifndef KT_H
#define KT_H 4      // levels per workgroup of the horizontal kernels (256 threads); 2 / 3 measure the same, 5 / 6 / 8: explicit stage 0.77 / 0.75 / 0.69 against 0.56 ms per step
                    // (three resident wavefronts per SIMD hold whole 4-wavefront workgroups only; taller tiles leave slots empty)
#endif

// Update statement a - l*u of the band LU (dgbtf2's rank-1 update, dgbtrs' forward sweep, dtbsv's column sweep).
// LAPACK builds them from dger / dtbsv, and the optimised BLAS libraries the reference links (MKL, OpenBLAS) evaluate
// those with fused multiply-adds: the reference linked against either agrees BIT FOR BIT, and the C oracle reproduces
// its implicit step bit for bit only with the update fused (tests/golden/lapack_spread.json,
// test_percall_chain).  So the fused form IS the reference's arithmetic; everything else in this file stays unfused
// (-ffp-contract=off), as the reference's own C++ is compiled without FMA contraction.
// The band LU's updates a - l * u: fused multiply-adds, as OpenBLAS's and MKL's kernels evaluate them on x86-64-v3 hosts
// (DESIGN.md section 2).  A reference linked to a BLAS WITHOUT fused multiply-adds (a plain netlib build, MKL on some AMD
// hosts) is reproduced by the library's second build flavour, `make nofma` -> libtempest_mi355x_nofma.so (-DTMX_LU_NOFMA,
// selected by TMX_LU_FMA=0 in engine.py / by linking it): multiply and subtract, rounded separately (-ffp-contract=off).
#ifdef TMX_LU_NOFMA
#define LU_UPD(a, l, u) ((a) - (l) * (u))
#else
#define LU_UPD(a, l, u) __builtin_fma(-(l), (u), (a))
#endif
#ifndef TMX_STORE_PRED
#define TMX_STORE_PRED true
#endif
#ifndef TMX_TAIL
#define TMX_TAIL 2
#endif
#ifndef TMX_SEARCH_TREE
#define TMX_SEARCH_TREE 0      // the tree form of the pivot search costs the pair kernel 8-10 spilled registers
#endif

#define OPC(op, k, off) p.ops[(((op) * (p.L + 1)) + (k)) * TMX_OPW + ((off) + 2)]

// Linear combination of state instances in the reference's accumulation order
// (GridPatch::LinearCombineData, src/atm/GridPatch.cpp:1436-1508): v = c_d * X_d (or 0), then += c_m * X_m.
struct LinTerms { const double * src[12]; double coef[12]; int n; int premul; };
// Where the U,V slabs of the instances named by a LinTerms (and of the stage's initial instance) live: the instance's own slot,
// or the slot of the instance it was copied from when the stepper left the CopyData of U,V out (tmx_host.hip: the implicit
// column solve rewrites rho*theta, W, rho of the copy but leaves U,V identical to the source's)
struct UvTerms { const double * src[12]; const double * xin; };

// base value of the update instance at flat offset o: either a plain instance (n == 0: src[0] is the base
// pointer) or the linear combination evaluated on the fly, so that LinearCombineData + StepExplicit is one
// pass instead of write + re-read of the combined state.  Same operations in the same order.
// NT = number of entries (compile time, so that every load of the combination is issued at once; a
// run-time loop over the kernel-argument arrays serialises one memory round trip per term).
// Every kernel here is launched with blockDim.x == 64: one wavefront per threadIdx.y.  Telling the compiler
// that the level index is wave-uniform moves the operator coefficients, the 1 - eta table and all slab
// offsets to scalar loads / SALU and frees the vector registers they occupied.
#define WAVE_UNIFORM(v) __builtin_amdgcn_readfirstlane((int)(v))
// 64-column tile of a workgroup: all tiles of the rank's patches, or the entries of a tile list (KParams.quads: the early /
// late tiles of the boundary-first stages); NTILES is the matching grid size on the host
#define TILE_X(p) ((p).quads ? (p).quads[blockIdx.x] : (int)blockIdx.x)
#define NTILES(e, p) ((p).quads ? (e)->launch_tiles : (p).NS / 64)

// Addressing: every global access here is  base pointer + (wave-uniform slab offset) + (lane's column).  Written
// as (uniform 64-bit pointer) + (32-bit per-lane BYTE offset) it maps onto the saddr form of global_load /
// global_store -- one 32-bit VGPR offset shared by all accesses of a thread instead of a 64-bit VGPR address
// (2 registers + a v_lshl_add_u64) per access.  An instance is 104 MB at ne30 L30; a slab row (NS columns) must
// stay below 4 GiB of bytes, checked at finalize.
__device__ __forceinline__ double ldu(const double * uniform_base, unsigned lane_byte_off) {
	return *(const double *)((const char *)uniform_base + lane_byte_off);
}
__device__ __forceinline__ void stu(double * uniform_base, unsigned lane_byte_off, double v) {
	*(double *)((char *)uniform_base + lane_byte_off) = v;
}

// base_value with the slab offset (uniform, in doubles) and the lane's byte offset kept apart
template <int NT>
__device__ __forceinline__ double base_value_u(const LinTerms & t, size_t slab_off, unsigned colb) {
	if (NT == 0) return ldu(t.src[0] + slab_off, colb);
	double x[NT > 0 ? NT : 1];
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? ldu(t.src[m] + slab_off, colb) : 0.0;
	double v = t.premul ? x[0] * t.coef[0] : 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) v += t.coef[m] * x[m];
	return v;
}

// the same for a U or V slab: sources from the UvTerms
template <int NT>
__device__ __forceinline__ double base_value_uv(const LinTerms & t, const UvTerms & uv, size_t slab_off, unsigned colb) {
	if (NT == 0) return ldu(uv.src[0] + slab_off, colb);
	double x[NT > 0 ? NT : 1];
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? ldu(uv.src[m] + slab_off, colb) : 0.0;
	double v = t.premul ? x[0] * t.coef[0] : 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) v += t.coef[m] * x[m];
	return v;
}

template <int NT>
__device__ __forceinline__ double base_value(const LinTerms & t, size_t o) {
	if (NT == 0) return t.src[0][o];
	double x[NT > 0 ? NT : 1];
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? t.src[m][o] : 0.0;
	double v = t.premul ? x[0] * t.coef[0] : 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) v += t.coef[m] * x[m];
	return v;
}

// XCD-aware tile order for the gather-heavy DSS kernel.  Workgroups are dealt round-robin over the 8 XCDs
// (b and b+8 share one, each XCD has its own L2), so a 1-D launch is unfolded such that every XCD sweeps a
// contiguous range of group tiles: the 128-byte element rows a tile touches are shared with its
// neighbouring tiles and now stay in one L2 (measured: k_dss 96 -> 79 us at ne30 L30; the streaming H and
// hyperviscosity kernels did not gain and keep the plain 2-D launch).  Placement only affects speed.
__device__ __forceinline__ bool xcd_tile(int nx, int ny, int & x, int & y) {
	const int b = blockIdx.x;
	const int xcd = b & 7, r = b >> 3;
	const int per = (nx + 7) >> 3;            // column tiles per XCD
	y = r / per;                              // level-major inside the XCD
	x = xcd * per + r % per;
	return y < ny && x < nx;
}

// The same for the vertical stencil kernels of the fully explicit mode (explicitly treated terms, U,V penalty, tracer columns):
// a thread of level k reads the levels k - 2 .. k + 2 of its column, so the level blocks of one 64-column tile share most of
// their operands.  In the plain 2-D launch they are a whole row of tiles apart in dispatch order and land on different XCDs
// (the halo levels were fetched 2.7 times over, PMC); here the 1-D launch is unfolded so that the level blocks of a column
// tile are consecutive workgroups of ONE XCD and find the halo in its L2.  mode 0: the plain order (TMX_XCD_VERTICAL=0, A/B).
__device__ __forceinline__ bool xcd_column_tile(int mode, int nx, int ny, int & x, int & y) {
	const int b = blockIdx.x;
	if (!mode) { x = b % nx; y = b / nx; return y < ny; }
	const int xcd = b & 7, r = b >> 3;
	y = r % ny;
	x = (r / ny) * 8 + xcd;
	return x < nx;
}
static inline int xcd_column_grid(int mode, int nx, int ny) { return mode ? ((nx + 7) / 8) * 8 * ny : nx * ny; }

// 3-D terrain-following metric (GridPatchCSGLL.cpp:441-566).  Stored mode streams the reference's arrays;
// closed mode evaluates the same expressions, operation for operation, from eight per-column factors and the
// 1 - eta table (accepted by tmx_set_patch_metric_factors only after a bit-for-bit comparison), which removes
// 8 doubles per point from the explicit kernels' HBM traffic and 6 from the column solve's.
struct MetCol { double mp, ma, mb, mc, daz, dbz, idx, idx2; };
__device__ __forceinline__ MetCol met_col(const KParams & p, int col) {
	MetCol m = { 0, 0, 0, 0, 0, 0, 0, 0 };
	if (p.closed) {
		const size_t NS = (size_t)p.NS;
		m.mp = p.g2d[G2_MP * NS + col]; m.ma = p.g2d[G2_MA * NS + col]; m.mb = p.g2d[G2_MB * NS + col]; m.mc = p.g2d[G2_MC * NS + col];
		m.daz = p.g2d[G2_DAZ * NS + col]; m.dbz = p.g2d[G2_DBZ * NS + col]; m.idx = p.g2d[G2_IDX * NS + col]; m.idx2 = p.g2d[G2_IDX2 * NS + col];
	}
	return m;
}
__device__ __forceinline__ void met_eval(const MetCol & m, double e, double & c0, double & c1, double & c2, double & dra, double & drb) {
	dra = e * m.daz; drb = e * m.dbz;
	c0 = m.mp * (m.ma * dra + m.mb * drb);
	c1 = m.mp * (m.mb * dra + m.mc * drb);
	c2 = m.idx2 - m.idx * (c0 * dra + c1 * drb);
}
// node level k: g^{alpha xi}, g^{beta xi}, g^{xi xi}, d_alpha R, d_beta R
__device__ __forceinline__ void metric_node(const KParams & p, const MetCol & m, int k, int col,
	double & ca2, double & cb2, double & cx2, double & dra, double & drb) {
	if (p.closed) { met_eval(m, p.eta[k], ca2, cb2, cx2, dra, drb); return; }
	const size_t s3 = (size_t)p.L * p.NS, o = (size_t)k * p.NS + col;
	ca2 = p.g3n[G3N_CA2 * s3 + o]; cb2 = p.g3n[G3N_CB2 * s3 + o]; cx2 = p.g3n[G3N_CX2 * s3 + o];
	dra = p.g3n[G3N_DRA * s3 + o]; drb = p.g3n[G3N_DRB * s3 + o];
}
// MM: 0 = metric mode read from p.closed at run time, 1 - eta table in global memory (the explicit kernels);
//     1 = closed form, table in LDS (`etal`); 2 = stored arrays.  1 and 2 are k_vi_fused's compile-time variants:
//     a fixed number of loads per block row lets the compiler count vmcnt exactly, and a pointer that may be LDS
//     or global would become a flat load, which also counts on vmcnt.
template <int MM = 0>
__device__ __forceinline__ void metric_node3(const KParams & p, const MetCol & m, int k, int col, double & ca2, double & cb2, double & cx2,
	const double * etal = nullptr) {
	if (MM == 1 || (MM == 0 && p.closed)) { double a, b; met_eval(m, MM == 1 ? etal[k] : p.eta[k], ca2, cb2, cx2, a, b); return; }
	const size_t s3 = (size_t)p.L * p.NS, o = (size_t)k * p.NS + col;
	ca2 = p.g3n[G3N_CA2 * s3 + o]; cb2 = p.g3n[G3N_CB2 * s3 + o]; cx2 = p.g3n[G3N_CX2 * s3 + o];
}
// interface k: g^{xi alpha}, g^{xi beta}, g^{xi xi}
template <int MM = 0>
__device__ __forceinline__ void metric_edge(const KParams & p, const MetCol & m, int k, int col, double & cx0, double & cx1, double & cx2,
	const double * etal = nullptr) {
	if (MM == 1 || (MM == 0 && p.closed)) { double a, b; met_eval(m, MM == 1 ? etal[p.L + k] : p.eta[p.L + k], cx0, cx1, cx2, a, b); return; }
	const size_t s3 = (size_t)(p.L + 1) * p.NS, o = (size_t)k * p.NS + col;
	cx0 = p.g3e[G3E_CX0 * s3 + o]; cx1 = p.g3e[G3E_CX1 * s3 + o]; cx2 = p.g3e[G3E_CX2 * s3 + o];
}

__device__ __forceinline__ double exner_from_rhotheta(const KParams & p, double rt) {
	// PhysicalConstants::ExnerPressureFromRhoTheta (src/atm/PhysicalConstants.h:397-399)
	return p.cp * tmx_ref_exp(p.Rd / (p.cp - p.Rd) * tmx_ref_log(p.Rd / p.p0 * rt));      // glibc's exp / log, bit for bit (tmx_refmath.h)
}
// The same with the two lookup tables of exp / log in LDS (`rmtab`: 256 doubles {1/c, log c} of log, then the 256 words of
// exp; tmx_rm_tables_to_lds fills it).  From global memory the two dependent table gathers of every evaluation are
// vector-memory loads: in the column kernels each one cost an s_waitcnt vmcnt(0), i.e. a wait for the level prefetches
// (and U-row stores) in flight as well.
#define TMX_RMTAB_DOUBLES 512
__device__ __forceinline__ void tmx_rm_tables_to_lds(double * rmtab, int tid, int nthreads) {
	for (int t = tid; t < 256; t += nthreads) {
		rmtab[t] = tmx_rm_log_tab[t];
		((unsigned long long *)(rmtab + 256))[t] = tmx_rm_exp_tab[t];
	}
}
// Branch-free form for arguments in the main range of log (positive, normal, finite; the caller checks all lanes with
// exner_arg_in_main_range first): both branches of log are evaluated and selected, exp needs none for these arguments
// (|x| < 2^-54 gives 1.0 on its main path as well).  Same values as exner_from_rhotheta_lds there.
__device__ __forceinline__ bool exner_arg_in_main_range(const KParams & p, double rt) {
	const unsigned top = (unsigned)(tmx_rm_bits(p.Rd / p.p0 * rt) >> 48);
	return top - 0x0010u < 0x7ff0u - 0x0010u;
}
__device__ __forceinline__ double exner_from_rhotheta_bf(const KParams & p, double rt, const double * rmtab) {
	return p.cp * tmx_ref_exp_main(p.Rd / (p.cp - p.Rd) * tmx_ref_log_bf(p.Rd / p.p0 * rt, rmtab), (const unsigned long long *)(rmtab + 256));
}
__device__ __forceinline__ double exner_from_rhotheta_lds(const KParams & p, double rt, const double * rmtab) {
	return p.cp * tmx_ref_exp_t(p.Rd / (p.cp - p.Rd) * tmx_ref_log_t(p.Rd / p.p0 * rt, rmtab), (const unsigned long long *)(rmtab + 256));
}

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
template <bool FUSE_V, int NT, bool MF>
#ifndef TMX_H_MINWG
#define TMX_H_MINWG 1      // ablation: minimum workgroups per CU the register allocation must allow (4 caps at 128 registers: slower, spills)
#endif
__global__ __launch_bounds__(64 * KT_H, TMX_H_MINWG) void k_h_explicit(KParams p, const double * __restrict__ xin,
	LinTerms xbase, double * xup, double * __restrict__ ucz, double * __restrict__ uv01, double dt, UvTerms uvt)
{
	__shared__ double s[MF ? 1 : KT_H][MF ? 1 : 9][64];
	__shared__ double sD[16], sS[16];
	const int lane = threadIdx.x, ty = WAVE_UNIFORM(threadIdx.y);
	const int tid = ty * 64 + lane;
	if (!MF && tid < 16) { sD[tid] = p.dx[tid]; sS[tid] = p.stiff[tid]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	// column of this lane: element-major (e, i, j) in memory; MF: lane = j + 4 e + 16 i
	const int tile = p.quads ? p.quads[blockIdx.x] : blockIdx.x;       // 64-column tile of this workgroup
	const int col = MF ? tile * 64 + ((lane >> 2) & 3) * 16 + (lane >> 4) * 4 + (lane & 3) : tile * 64 + lane;
	const int k = blockIdx.y * KT_H + ty;
	const bool act = (k < L) && (col < p.ncol);

	double ua = 0, ub = 0, rt = 1, rh = 1, ux = 0;
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, jn = 1, j2d = 0, fcor = 0, ca2 = 0, cb2 = 0, cx2 = 0, dra = 0, drb = 0;
	double conUa = 0, conUb = 0, conUx = 0, dxua = 0, dxub = 0;
	double v_uam = 0, v_uap = 0, v_ubm = 0, v_ubp = 0, w_hi = 0, w_lo = 0;
	// base of the update (a plain instance or the fused stage combination): loaded in the same burst as the state
	// and reduced to four values at once -- loading it at the point of use made the compiler keep all 48 LDS reads
	// and the 4 x NT combination sources live together (180-200 VGPRs, occupancy 2)
	double bU = 0, bV = 0, bT = 0, bR = 0, bW = 0;
	if (act) {
		const int km = (k > 0) ? k - 1 : 0, kp = (k < L - 1) ? k + 1 : L - 1;
		const unsigned cb = (unsigned)col * 8u;
		const double * __restrict__ xuv = uvt.xin;      // U,V of the initial instance (its own slot or the one it shares)
		ua = ldu(xuv + TMX_SLAB_U(L, k) * NS, cb);
		ub = ldu(xuv + TMX_SLAB_V(L, k) * NS, cb);
		rt = ldu(xin + TMX_SLAB_T(L, k) * NS, cb);
		rh = ldu(xin + TMX_SLAB_R(L, k) * NS, cb);
		const double we0 = ldu(xin + TMX_SLAB_W(L, k) * NS, cb);
		const double we1 = ldu(xin + TMX_SLAB_W(L, k + 1) * NS, cb);
		const double uam = ldu(xuv + TMX_SLAB_U(L, km) * NS, cb), uap = ldu(xuv + TMX_SLAB_U(L, kp) * NS, cb);
		const double ubm = ldu(xuv + TMX_SLAB_V(L, km) * NS, cb), ubp = ldu(xuv + TMX_SLAB_V(L, kp) * NS, cb);
		bU = base_value_uv<NT>(xbase, uvt, TMX_SLAB_U(L, k) * NS, cb); bV = base_value_uv<NT>(xbase, uvt, TMX_SLAB_V(L, k) * NS, cb);
		bT = base_value_u<NT>(xbase, TMX_SLAB_T(L, k) * NS, cb); bR = base_value_u<NT>(xbase, TMX_SLAB_R(L, k) * NS, cb);
		if (TMX_H_FUSE_W && !MF && ty > 0) bW = base_value_u<NT>(xbase, TMX_SLAB_W(L, k) * NS, cb);
		c2a0 = ldu(p.g2d + G2_C2A0 * NS, cb); c2a1 = ldu(p.g2d + G2_C2A1 * NS, cb); c2b1 = ldu(p.g2d + G2_C2B1 * NS, cb);
		jn = ldu(p.g2d + G2_JN * NS, cb); j2d = ldu(p.g2d + G2_J2D * NS, cb); fcor = ldu(p.g2d + G2_F * NS, cb);
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
	const double ke = 0.5 * (conUa * ua + conUb * ub + conUx * ux);
	const double exn = act ? exner_from_rhotheta(p, rt) : 0.0;
	const double fa = jn * conUa, fb = jn * conUb;
	double daub = 0, daux = 0, dbua = 0, dbux = 0, dap = 0, dbp = 0, dake = 0, dbke = 0;
	double darf = 0, dapf = 0, dbrf = 0, dbpf = 0;
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
	} else {
	s[ty][0][lane] = ua;  s[ty][1][lane] = ub;  s[ty][2][lane] = ux;
	s[ty][3][lane] = exn; s[ty][4][lane] = ke;
	s[ty][5][lane] = fa * rh; s[ty][6][lane] = fa * rt;
	s[ty][7][lane] = fb * rh; s[ty][8][lane] = fb * rt;
	__syncthreads();
	if (!act) return;
#ifdef TMX_H_SCHED_BARRIER
	__builtin_amdgcn_sched_barrier(0);
#endif

	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;
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
	double outU = bU + dt * upa;
	double outV = bV + dt * upb;
	if (FUSE_V) {
		if (k < 2) { uv01[(size_t)k * NS + col] = outU; uv01[(size_t)(2 + k) * NS + col] = outV; }
		// first-order upwind penalty on U,V (VerticalDynamicsFEM.cpp:984-1023): left then right operator
		if (k < L - 1) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * ua; b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * v_uap;
			outU += b * w_hi;
			b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * ub; b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * v_ubp;
			outV += b * w_hi;
		}
		if (k > 0) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * v_uam; b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * ua;
			outU += b * w_lo;
			b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * v_ubm; b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * ub;
			outV += b * w_lo;
		}
	}
	stu(xup + sU, colb, outU);
	stu(xup + sV, colb, outV);
	stu(xup + sR, colb, bR - dt * invJ * (darf + dbrf));
	stu(xup + sT, colb, bT - dt * invJ * (dapf + dbpf));
	if (TMX_H_FUSE_W && !MF) {
		// W on the interfaces inside the tile (part 2 below, :1612-1659): interface k takes (u x zeta)_xi of the levels k - 1 and k,
		// which wavefront ty - 1 hands over through its own (now free) LDS row; only the tile's first interface is left to
		// k_h_w_update, which therefore needs the hand-off array for the first and the last level of a tile only
		s[ty][0][lane] = uczx;
		if (ty == 0 || ty == KT_H - 1) stu(ucz + (size_t)k * NS, colb, uczx);
		__syncthreads();       // wavefronts above the top level have left; the barrier counts the ones still running
		if (ty > 0) {
			double v = 0.0;
			v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * s[ty - 1][0][lane];
			v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * uczx;
			stu(xup + TMX_SLAB_W(L, k) * NS, colb, bW + dt * v);
		}
	} else
		stu(ucz + (size_t)k * NS, colb, uczx);
}

// H explicit, part 2: W on interfaces (src/atm/HorizontalDynamicsFEM.cpp:1612-1659): bottom
// boundary from the UPDATED U,V extrapolated to the surface, interior += dt * interp((u x zeta)_xi).
// stride > 1: only the interfaces k = 0, stride, 2 stride, ... (the first of every level tile of k_h_explicit, which has done
// the others) and the top one
template <int NT>
__global__ __launch_bounds__(256) void k_h_w_update(KParams p, LinTerms xbase, double * xup,
	const double * __restrict__ ucz, const double * __restrict__ uv01, double dt, int stride)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = (p.quads ? p.quads[blockIdx.x] : blockIdx.x) * 64 + threadIdx.x;
	const int idx = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	const int nfirst = (L + stride - 1) / stride;          // interfaces idx * stride < L, then the top one
	const int k = (idx < nfirst) ? idx * stride : ((idx == nfirst) ? L : L + 1);
	if (k > L || col >= p.ncol) return;
	if (k == L) {      // the top interface is not touched by the H step; carry it over when base != out
		if (NT != 0 || xbase.src[0] != xup) xup[TMX_SLAB_W(L, L) * NS + col] = base_value<NT>(xbase, TMX_SLAB_W(L, L) * NS + col);
		return;
	}
	if (k == 0) {
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
		xup[TMX_SLAB_W(L, 0) * NS + col] = -(cx0 * u0 + cx1 * v0) / cx2;
	} else {
		double v = 0.0;
		v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, -1) * ucz[(size_t)(k - 1) * NS + col];
		v += OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, 0) * ucz[(size_t)k * NS + col];
		xup[TMX_SLAB_W(L, k) * NS + col] = base_value<NT>(xbase, TMX_SLAB_W(L, k) * NS + col) + dt * v;
	}
}

// xin_uv / xbase_uv / src_uv: where the U,V slabs of the initial instance, the base instance and the combination sources
// live when that is not the instance's own slot (NULL: the same pointers)
void tmxk_h_explicit(tmx_engine * e, const KParams & p, const double * xin, const double * xbase_ptr, double * xup, double dt, int fuse_v,
	int nterms, const double * const * src, const double * coef, int premul, const double * xin_uv, const double * xbase_uv, const double * const * src_uv) {
	// base of the update: a plain instance (nterms == 0) or a linear combination evaluated in the kernels
	LinTerms xbase;
	UvTerms uvt;
	xbase.n = nterms; xbase.premul = premul;
	xbase.src[0] = xbase_ptr; xbase.coef[0] = 0.0;
	uvt.xin = xin_uv ? xin_uv : xin; uvt.src[0] = xbase_uv ? xbase_uv : xbase_ptr;
	for (int m = 0; m < nterms; m++) { xbase.src[m] = src[m]; xbase.coef[m] = coef[m]; uvt.src[m] = src_uv ? src_uv[m] : src[m]; }
	for (int m = (nterms > 0 ? nterms : 1); m < 12; m++) { xbase.src[m] = xbase.src[0]; xbase.coef[m] = 0.0; uvt.src[m] = uvt.src[0]; }
	const int ntile = p.quads ? e->launch_tiles : p.NS / 64;
	dim3 blk(64, KT_H), grd(ntile, (p.L + KT_H - 1) / KT_H);
	double * uv01 = e->d_scratch + (size_t)p.L * p.NS;     // 4 slabs behind the (u x zeta)_xi hand-off
	const int wstride = (TMX_H_FUSE_W && !(fuse_v && e->use_mfma)) ? KT_H : 1;
	dim3 blk2(64, 4), grd2(ntile, ((p.L + wstride - 1) / wstride + 1 + 3) / 4);
	const double * uvp = fuse_v ? (const double *)uv01 : (const double *)nullptr;
#define LAUNCH_HV(NT_) do { \
		if (fuse_v && e->use_mfma) hipLaunchKernelGGL((k_h_explicit<true, NT_, true>), grd, blk, 0, e->stream, p, xin, xbase, xup, e->d_scratch, uv01, dt, uvt); \
		else if (fuse_v) hipLaunchKernelGGL((k_h_explicit<true, NT_, false>), grd, blk, 0, e->stream, p, xin, xbase, xup, e->d_scratch, uv01, dt, uvt); \
		else hipLaunchKernelGGL((k_h_explicit<false, NT_, false>), grd, blk, 0, e->stream, p, xin, xbase, xup, e->d_scratch, uv01, dt, uvt); \
		hipLaunchKernelGGL((k_h_w_update<NT_>), grd2, blk2, 0, e->stream, p, xbase, xup, (const double *)e->d_scratch, uvp, dt, wstride); } while (0)
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
}

// ---------------------------------------------------------------------------------------------
// V explicit (implicit mode): xi_dot on interfaces and first-order upwind penalty on U,V
// (src/atm/VerticalDynamicsFEM.cpp:816-828, 984-1023; LinearColumnOperatorFEM.cpp:1863-1888).

__device__ __forceinline__ double xidot_edge(const KParams & p, const double * __restrict__ xin, int kk, int col) {
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	if (kk <= 0 || kk >= L) return 0.0;
	// U,V on interfaces = InterpolateNodeToREdge of the node values (H prologue, :817-831)
	double ue = 0.0, ve = 0.0;
#pragma unroll
	for (int off = -2; off <= 1; off++) {
		const int l = kk + off;
		if (l < 0 || l >= L) continue;
		const double c = OPC(TMX_OP_INTERP_NODE_TO_REDGE, kk, off);
		ue += c * xin[TMX_SLAB_U(L, l) * NS + col];
		ve += c * xin[TMX_SLAB_V(L, l) * NS + col];
	}
	const double we = xin[TMX_SLAB_W(L, kk) * NS + col];
	double e0, e1, e2;
	metric_edge(p, met_col(p, col), kk, col, e0, e1, e2);
	return e0 * ue + e1 * ve + e2 * we;
}

// UDV: the vertical uniform diffusion of U,V of the fully explicit mode (k_v_uniform_diffusion_uv below, :1059-1105) is added
// to the same value before it is stored -- the sum the two kernels form through memory, in the same order
#ifndef KT_VE
#define KT_VE 4       // levels per workgroup (vertical stencils: see KT_VC)
#endif
// the update of U,V at (col, level k), k < L
template <bool UDV>
__device__ __forceinline__ void v_explicit_point(const KParams & p, const double * __restrict__ xin, double * __restrict__ xup, double dt,
	const double * __restrict__ xref, double cf, int col, int k)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const double up0[2] = { xup[(size_t)k * NS + col], xup[(size_t)(L + k) * NS + col] };      // U, V to update: loaded with the operands
	const double w_hi = dt * fabs(xidot_edge(p, xin, k + 1, col));   // weight of interface k+1
	const double w_lo = dt * fabs(xidot_edge(p, xin, k, col));       // weight of interface k
	const int km = (k > 0) ? k - 1 : 0, kp = (k < L - 1) ? k + 1 : L - 1;
#pragma unroll
	for (int v = 0; v < 2; v++) {
		const int sb = v * L;
		const double x0 = xin[(size_t)(sb + k) * NS + col];
		const double xm = xin[(size_t)(sb + km) * NS + col], xp = xin[(size_t)(sb + kp) * NS + col];
		double out = up0[v];
		if (k < L - 1) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 0) * x0;
			b += OPC(TMX_OP_PENALTY_LEFT, k, 1) * xp;
			out += b * w_hi;
		}
		if (k > 0) {
			double b = 0.0;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, -1) * xm;
			b += OPC(TMX_OP_PENALTY_RIGHT, k, 0) * x0;
			out += b * w_lo;
		}
		if (UDV) {
			double dd = 0.0, ddr = 0.0;
#pragma unroll
			for (int off = -2; off <= 2; off++) {
				const int l = k + off;
				if (l < 0 || l >= L) continue;
				const double c = OPC(TMX_OP_DIFFDIFF_NODE_TO_NODE, k, off);
				dd += c * xin[(size_t)(sb + l) * NS + col];
				ddr += c * xref[(size_t)(sb + l) * NS + col];
			}
			out += dt * cf * (dd - ddr);
		}
		xup[(size_t)(sb + k) * NS + col] = out;
	}
}

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

void tmxk_v_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_udiff_uv) {
	const int nt_ = NTILES(e, p), xm = e->xcd_vertical;
	dim3 blk(64, KT_VE), grd(xcd_column_grid(xm, nt_, (p.L + KT_VE - 1) / KT_VE));
	if (with_udiff_uv)
		hipLaunchKernelGGL(k_v_explicit<true>, grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)e->d_ref,
			e->cfg.uniform_diffusion_vector / (e->cfg.ztop * e->cfg.ztop), nt_, xm);
	else
		hipLaunchKernelGGL(k_v_explicit<false>, grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)nullptr, 0.0, nt_, xm);
}

// ---------------------------------------------------------------------------------------------
// DSS: average co-located copies (GridCSGLL::ApplyDSS, src/atm/GridCSGLL.cpp:435-781) expressed per group of
// co-located nodes instead of per patch edge, in the reference's own order of operations so that the result is the
// reference's bit for bit: two copies 0.5 * (x + y); four copies 0.5 * (0.5 * (x + x_alpha) + 0.5 * (x_beta + x_diag))
// with the partners named in the frame of the member's own patch (alpha seams are averaged before beta seams); cube
// corners (1/3) * ((x + x_alpha) + x_beta).  The host stores the members as [m0, alpha, beta, diagonal] of m0 and a
// 2-bit pairing type per member (tmx_host.hip, "device group tables"); within one panel every member has type 0 and
// all copies receive the same bits, on panel edges with rotated axes the copies differ in the last bit exactly as
// they do in the reference.  (U,V) of copies on other panels are rotated into the member's frame with the covector
// panel transform of the ring entry through which its patch sees them (GridPatchCSGLL.cpp:1783-1924).  Copies owned
// by other ranks are read from the ghost buffer filled by the exchange.

// Every load of the group (5 slabs x up to 4 copies) is issued before the first use, so a wave keeps up to
// 20 gathers in flight (measured 90 -> 74 us per call at ne30 L30 against loading slab by slab), and the tiles
// of one XCD run level-major, so that the partially used 128-byte element rows of a 4-level tile are completed
// by the neighbouring group tiles while they still sit in that XCD's L2 (74 -> 68 us).
// member M's combination of the copies w0..w3 (values already in M's frame); ty = M's 2-bit pairing type
template <int M>
__device__ __forceinline__ double dss_combine(double w0, double w1, double w2, double w3, int n, int ty) {
	if (n == 2) return 0.5 * (w0 + w1);
	if (n == 4) {
		const double p0 = (ty == 0) ? w0 + w1 : ((ty == 1) ? w0 + w2 : w0 + w3);
		const double p1 = (ty == 0) ? w2 + w3 : ((ty == 1) ? w1 + w3 : w1 + w2);
		return 0.5 * (0.5 * p0 + 0.5 * p1);
	}
	// cube corner: (1/3) * ((x + first) + second), first = next member (ty 0) or previous member (ty 1)
	const double me = (M == 0) ? w0 : ((M == 1) ? w1 : w2);
	const double nx = (M == 0) ? w1 : ((M == 1) ? w2 : w0);
	const double pv = (M == 0) ? w2 : ((M == 1) ? w0 : w1);
	return (1.0 / 3.0) * ((me + (ty ? pv : nx)) + (ty ? nx : pv));
}

#ifndef TMX_DSS_LPT
#define TMX_DSS_LPT 1      // slab-levels per thread of k_dss
#endif
// five workgroups (20 wavefronts) per CU: the register allocation is held to 96 (94 used; it settled at 98 = four per SIMD on its
// own), which the gathers use (-2 % per launch; six workgroups, 80 registers, serialise the loads: +26 %)
#ifndef TMX_DSS_MINWG
#define TMX_DSS_MINWG 5
#endif
__global__ __launch_bounds__(256, TMX_DSS_MINWG) void k_dss(KParams p, double * __restrict__ x, const double * __restrict__ ghost,
	const int * __restrict__ gbase, const int * __restrict__ gstride, int ngroups, const int * __restrict__ gcols,
	const int * __restrict__ gn, const int * __restrict__ gx, const int * __restrict__ gtype, const double * __restrict__ xmat, int nt, int g0)
{
	// groups [g0, ngroups) of the engine's list
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	int bx, by;
	if (!xcd_tile((ngroups - g0 + 63) / 64, (L + 1 + 4 * TMX_DSS_LPT - 1) / (4 * TMX_DSS_LPT), bx, by)) return;
	const int g = g0 + bx * 64 + threadIdx.x;
	const int kk0 = (by * 4 + WAVE_UNIFORM(threadIdx.y)) * TMX_DSS_LPT;
	if (g >= ngroups || kk0 > L) return;
	const int n = gn[g];
	const int4 c4 = *(const int4 *)(gcols + g * 4);
	const int c[4] = { c4.x, c4.y, c4.z, c4.w };
	const int xi = gx[g];
	const int gt = gtype[g];
	const double * ptr[4]; size_t str[4]; bool loc[4];
#pragma unroll
	for (int m = 0; m < 4; m++) {
		loc[m] = (c[m] < (int)NS);
		const bool on = (m < n);
		if (!on) { ptr[m] = x; str[m] = 0; loc[m] = false; }
		else if (loc[m]) { ptr[m] = x + c[m]; str[m] = NS; }
		else { ptr[m] = ghost + gbase[c[m] - (int)NS]; str[m] = (size_t)gstride[c[m] - (int)NS]; }
	}
#pragma unroll
	for (int rep = 0; rep < TMX_DSS_LPT; rep++) {
	const int kk = kk0 + rep;
	if (kk > L) break;
	const bool lev = (kk < L);
	int slabs[5];
	slabs[0] = TMX_SLAB_W(L, kk);
	slabs[1] = lev ? TMX_SLAB_T(L, kk) : slabs[0]; slabs[2] = lev ? TMX_SLAB_R(L, kk) : slabs[0];
	slabs[3] = lev ? TMX_SLAB_U(L, kk) : slabs[0]; slabs[4] = lev ? TMX_SLAB_V(L, kk) : slabs[0];
	double v[5][4];
#pragma unroll
	for (int t = 0; t < 5; t++)
#pragma unroll
		for (int m = 0; m < 4; m++) v[t][m] = (m < n && (t == 0 || lev)) ? ptr[m][(size_t)slabs[t] * str[m]] : 0.0;
	// gt == 0 with two or four copies: every member has the same pairing, one value for all of them
	const bool same = (gt == 0) && (n != 3);
	const int t0 = gt & 3, t1 = (gt >> 2) & 3, t2 = (gt >> 4) & 3, t3 = (gt >> 6) & 3;
#define DSS_SCALAR(W0, W1, W2, W3, O0, O1, O2, O3) do { \
		if (same) { const double a_ = dss_combine<0>(W0, W1, W2, W3, n, 0); O0 = a_; O1 = a_; O2 = a_; O3 = a_; } \
		else { O0 = dss_combine<0>(W0, W1, W2, W3, n, t0); O1 = dss_combine<1>(W0, W1, W2, W3, n, t1); \
		       O2 = dss_combine<2>(W0, W1, W2, W3, n, t2); O3 = dss_combine<3>(W0, W1, W2, W3, n, t3); } } while (0)
	double out[5][4];
#pragma unroll
	for (int t = 0; t < 3; t++) DSS_SCALAR(v[t][0], v[t][1], v[t][2], v[t][3], out[t][0], out[t][1], out[t][2], out[t][3]);
	if (xi < 0) {
		DSS_SCALAR(v[3][0], v[3][1], v[3][2], v[3][3], out[3][0], out[3][1], out[3][2], out[3][3]);
		DSS_SCALAR(v[4][0], v[4][1], v[4][2], v[4][3], out[4][0], out[4][1], out[4][2], out[4][3]);
	} else {
		// copies on other panels: every member sees the others rotated into its own frame
		const double * Mx = xmat + (size_t)xi * 64;
#define DSS_ROT(M_, Q_, A_, B_) do { if (Q_ == M_ || Q_ >= n) { A_ = v[3][Q_]; B_ = v[4][Q_]; } else { const double * T_ = Mx + (M_ * 4 + Q_) * 4; \
			A_ = T_[0] * v[3][Q_] + T_[1] * v[4][Q_]; B_ = T_[2] * v[3][Q_] + T_[3] * v[4][Q_]; } } while (0)
#define DSS_VEC(M_, TY_) do { double a0, a1, a2, a3, b0, b1, b2, b3; \
			DSS_ROT(M_, 0, a0, b0); DSS_ROT(M_, 1, a1, b1); DSS_ROT(M_, 2, a2, b2); DSS_ROT(M_, 3, a3, b3); \
			out[3][M_] = dss_combine<M_>(a0, a1, a2, a3, n, TY_); out[4][M_] = dss_combine<M_>(b0, b1, b2, b3, n, TY_); } while (0)
		DSS_VEC(0, t0); DSS_VEC(1, t1); DSS_VEC(2, t2); DSS_VEC(3, t3);
#undef DSS_VEC
#undef DSS_ROT
	}
#pragma unroll
	for (int t = 0; t < 5; t++)
#pragma unroll
		for (int m = 0; m < 4; m++)
			if (loc[m] && (t == 0 || lev)) x[(size_t)slabs[t] * NS + c[m]] = out[t][m];
	// tracer densities: scalars on levels (ApplyDSS(DataType_Tracers))
	if (lev) {
		for (int q = 0; q < nt; q++) {
			const size_t sl = (size_t)TMX_SLAB_Q(L, q, kk);
			const double w0 = (0 < n) ? ptr[0][sl * str[0]] : 0.0, w1 = (1 < n) ? ptr[1][sl * str[1]] : 0.0;
			const double w2 = (2 < n) ? ptr[2][sl * str[2]] : 0.0, w3 = (3 < n) ? ptr[3][sl * str[3]] : 0.0;
			double o0, o1, o2, o3;
			DSS_SCALAR(w0, w1, w2, w3, o0, o1, o2, o3);
			if (loc[0]) x[sl * NS + c[0]] = o0;
			if (loc[1]) x[sl * NS + c[1]] = o1;
			if (loc[2]) x[sl * NS + c[2]] = o2;
			if (loc[3]) x[sl * NS + c[3]] = o3;
		}
	}
#undef DSS_SCALAR
	}
}

void tmxk_dss(tmx_engine * e, const KParams & p, double * x, int g0, int g1) {
	if (g1 <= g0) return;
	const int nx = (g1 - g0 + 63) / 64, ny = (p.L + 1 + 4 * TMX_DSS_LPT - 1) / (4 * TMX_DSS_LPT);
	dim3 blk(64, 4), grd(8 * ((nx + 7) / 8) * ny);
	hipLaunchKernelGGL(k_dss, grd, blk, 0, e->stream, p, x, (const double *)e->d_ghost, (const int *)e->d_ghost_base,
		(const int *)e->d_ghost_stride, g1,
		(const int *)e->d_grp_cols, (const int *)e->d_grp_n, (const int *)e->d_grp_x, (const int *)e->d_grp_type, (const double *)e->d_xmat, e->nt, g0);
}

// HorizontalDynamicsFEM::ApplyRayleighFriction (src/atm/HorizontalDynamicsFEM.cpp:2420-2570): where the strength
// is non-zero, ten sub-cycled backward-Euler relaxations x <- f x + (1 - f) x_ref, f = 1 / (1 + dt nu / 10), of
// U, V, rho*theta (strength on levels) and W (strength on interfaces); rho is left alone (:2455-2461).
__global__ __launch_bounds__(256) void k_rayleigh(int L, int NS_, int ncol, double * __restrict__ x,
	const double * __restrict__ nu, const double * __restrict__ ref, double dt)
{
	const size_t NS = (size_t)NS_;
	const int col = blockIdx.x * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (k > L || col >= ncol) return;
	const double factor = 1.0 / 10;
	if (k < L) {
		const double dnu = nu[(size_t)k * NS + col];
		if (dnu != 0.0) {
#pragma unroll
			for (int v = 0; v < 3; v++) {
				const size_t o = (size_t)(v * L + k) * NS + col;        // U, V, rho*theta slabs are [0, 3L)
				double xv = x[o];
				const double r = ref[o];
				for (int si = 0; si < 10; si++) {
					const double f = 1.0 / (1.0 + factor * dt * dnu);
					xv = f * xv + (1.0 - f) * r;
				}
				x[o] = xv;
			}
		}
	}
	const double dnu = nu[(size_t)(L + k) * NS + col];
	if (dnu != 0.0) {
		const size_t o = (size_t)TMX_SLAB_W(L, k) * NS + col;
		double xv = x[o];
		const double r = ref[(size_t)(3 * L + k) * NS + col];
		for (int si = 0; si < 10; si++) {
			const double f = 1.0 / (1.0 + factor * dt * dnu);
			xv = f * xv + (1.0 - f) * r;
		}
		x[o] = xv;
	}
}

void tmxk_rayleigh(tmx_engine * e, const KParams & p, double * x, double dt) {
	dim3 blk(64, 4), grd(p.NS / 64, (p.L + 1 + 3) / 4);
	hipLaunchKernelGGL(k_rayleigh, grd, blk, 0, e->stream, p.L, p.NS, p.ncol, x, (const double *)e->d_ray_nu, (const double *)e->d_ray_ref, dt);
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
	static const int nr_env = getenv("TMX_VT_NR") ? atoi(getenv("TMX_VT_NR")) : 0;
	static const int lw8_env = getenv("TMX_VT_LW8") ? atoi(getenv("TMX_VT_LW8")) : -1;
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
	if (!(getenv("TMX_VT_ROWS") && atoi(getenv("TMX_VT_ROWS")) == 0)) {      // default: 16 columns x 4 row lanes per workgroup
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
	if (const char * m = getenv("TMX_VT_LANES")) want = atoi(m);
	if (want < lw && (want == 32 || want == 16 || want == 8)) { lw = want; lds = ((size_t)p.L * 4 + (size_t)p.L * 2 + (size_t)(p.L + 1) * 2) * lw * sizeof(double); }
#define LAUNCH_VT(LW_) do { hipFuncSetAttribute((const void *)k_vi_tracers<LW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
	hipLaunchKernelGGL(k_vi_tracers<LW_>, dim3((e->nunique + LW_ - 1) / LW_), dim3(64), lds, e->stream, p, e->nt, xin, w0, xbase, xup, dt, \
		e->nunique, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_flag, 0, 0.0, (const double *)nullptr); } while (0)
	if (lw == 64) LAUNCH_VT(64); else if (lw == 32) LAUNCH_VT(32); else if (lw == 16) LAUNCH_VT(16); else LAUNCH_VT(8);
#undef LAUNCH_VT
	return 0;
}

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

// The same update with the shared operands of a tile staged once in LDS: a workgroup = 64 columns x 8 levels; xi_dot (and, with
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

// UpdateColumnTracers in the fully explicit vertical mode: every stored column, in place on the update instance
int tmxk_vi_tracers_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) {
	if (e->nt == 0 || p.ncol == 0) return 0;
	if (!getenv("TMX_VT_COLUMN")) {      // level-parallel form (default); TMX_VT_COLUMN=1: the one-lane-per-column kernel, for A/B and tests
		const bool v1 = getenv("TMX_VT_EXPLICIT_V1") != nullptr;       // the form without LDS staging, for A/B and tests
		if (v1) {
			dim3 blk(64, 4), grd(NTILES(e, p), (p.L + 3) / 4);
			hipLaunchKernelGGL(k_v_tracers_explicit, grd, blk, 0, e->stream, p, e->nt, xin, xup, dt,
				e->udiff ? e->cfg.uniform_diffusion_scalar : 0.0, (const double *)e->d_ref);
		} else {
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
	if (!(getenv("TMX_VT_ROWS") && atoi(getenv("TMX_VT_ROWS")) == 0)) {
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

// ---------------------------------------------------------------------------------------------
// Restart image: GridPatch::GetDataContainerActiveState() of one patch (GridPatch.cpp:359-361,387), i.e.
// what OutputManagerComposite::Output writes per patch (OutputManagerComposite.cpp:331-349): after the 8-byte patch index,
//   node [5][na][nb][L], redge [5][na][nb][L+1], tracers [nt][na][nb][L]      (k fastest, one-node halo ring)
// One workgroup = 64 device columns (4 elements), one wavefront = one element.  Per variable the slabs of the tile are
// staged through LDS (coalesced 512 B rows of the slab layout on one side, contiguous L x 8 B columns of the image on the
// other).  PACK also forms the three derived arrays the reference keeps beside the prognostic ones (W on levels, U and V
// on interfaces: HorizontalDynamicsFEM.cpp:817-831), with the same statements as tmx_download_state.
struct ImageArgs {
	int c0, ncp, na, nb, neb, nt;     // first device column of the patch, its column count, patch box, elements along beta, tracers
	double * x;                       // instance base (slab layout)
	double * surf;                    // its two surface slots or nullptr
	double * img;                     // image, in doubles, header excluded
};

template <bool PACK>
__global__ __launch_bounds__(256) void k_active_state_image(KParams p, ImageArgs a) {
	extern __shared__ double tile[];          // [L+1][65]
	const int L = p.L, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t NS = (size_t)p.NS, nn = (size_t)a.na * a.nb;
	const int cb = blockIdx.x * 64;                 // first patch column of the tile
	const int cl = cb + threadIdx.x % 64;           // this lane's column when moving slab rows
	const bool col_ok = cl < a.ncp;
	const int el = (cb >> 4) + wv;                   // this wavefront's element
	const bool el_ok = (el << 4) < a.ncp;
	const int ea = el / a.neb, eb = el % a.neb;
	// host node offset of column q (0..15) of the wavefront's element
	auto hoff = [&](int q) { return (size_t)(1 + ea * 4 + (q >> 2)) * a.nb + (1 + eb * 4 + (q & 3)); };
	double * node = a.img, * redge = a.img + 5 * nn * L, * trc = redge + 5 * nn * (L + 1);
#define TL(k_, c_) tile[(k_) * 65 + (c_)]
	// one variable: slabs [slab0, slab0 + nlev) <-> image array `arr` component `comp` with nlev entries per node
	auto move = [&](int slab0, int nlev, double * arr, int comp) {
		if (PACK) {
			for (int k = wv; k < nlev; k += 4) if (col_ok) TL(k, lane) = a.x[(size_t)(slab0 + k) * NS + a.c0 + cl];
			__syncthreads();
			if (el_ok)
				for (int q = 0; q < 16; q++) {
					double * dst = arr + ((size_t)comp * nn + hoff(q)) * nlev;
					for (int k = lane; k < nlev; k += 64) dst[k] = TL(k, wv * 16 + q);
				}
		} else {
			if (el_ok)
				for (int q = 0; q < 16; q++) {
					const double * src = arr + ((size_t)comp * nn + hoff(q)) * nlev;
					for (int k = lane; k < nlev; k += 64) TL(k, wv * 16 + q) = src[k];
				}
			__syncthreads();
			for (int k = wv; k < nlev; k += 4) if (col_ok) a.x[(size_t)(slab0 + k) * NS + a.c0 + cl] = TL(k, lane);
		}
	};
	const int nodevar[4] = { 0, 1, 2, 4 };
	for (int v = 0; v < 4; v++) {
		move(v * L, L, node, nodevar[v]);
		if (PACK && v < 2 && el_ok) {
			// U (v = 0) or V (v = 1) on interfaces from the column in the tile: InterpolateNodeToREdge
			for (int q = 0; q < 16; q++) {
				double * dst = redge + ((size_t)v * nn + hoff(q)) * (L + 1);
				for (int k = lane; k <= L; k += 64) {
					double ue = 0.0;
					for (int off = -2; off <= 1; off++) {
						const int l = k + off;
						if (l < 0 || l >= L) continue;
						const double cc = OPC(TMX_OP_INTERP_NODE_TO_REDGE, k, off);
						if (cc == 0.0) continue;
						ue += cc * TL(l, wv * 16 + q);
					}
					dst[k] = ue;
				}
			}
		}
		__syncthreads();
	}
	move(4 * L, L + 1, redge, 3);
	if (PACK && el_ok) {
		for (int q = 0; q < 16; q++) {      // W on levels: InterpolateREdgeToNode
			double * dst = node + ((size_t)3 * nn + hoff(q)) * L;
			for (int k = lane; k < L; k += 64) {
				double w = 0.0;
				w += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * TL(k, wv * 16 + q);
				w += OPC(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * TL(k + 1, wv * 16 + q);
				dst[k] = w;
			}
		}
	}
	__syncthreads();
	for (int c = 0; c < a.nt; c++) { move(TMX_SLAB_Q(L, c, 0), L, trc, c); __syncthreads(); }
	// surface slots: interface-level-0 entries of rho (component 4) and rho*theta (component 2)
	if (a.surf && el_ok && lane < 16) {
		const size_t col = (size_t)a.c0 + (el << 4) + lane;
		double * r4 = redge + ((size_t)4 * nn + hoff(lane)) * (L + 1), * r2 = redge + ((size_t)2 * nn + hoff(lane)) * (L + 1);
		if (PACK) { *r4 = a.surf[col]; *r2 = a.surf[NS + col]; }
		else { a.surf[col] = *r4; a.surf[NS + col] = *r2; }
	}
#undef TL
}

void tmxk_active_state_image(tmx_engine * e, const KParams & p, bool pack, int c0, int ncp, int na, int nb, int neb, double * x, double * surf, double * img) {
	ImageArgs a; a.c0 = c0; a.ncp = ncp; a.na = na; a.nb = nb; a.neb = neb; a.nt = e->nt; a.x = x; a.surf = surf; a.img = img;
	const size_t lds = (size_t)(p.L + 1) * 65 * sizeof(double);
	if (lds > 48 * 1024) {
		hipFuncSetAttribute((const void *)k_active_state_image<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipFuncSetAttribute((const void *)k_active_state_image<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	}
	if (pack) hipLaunchKernelGGL(k_active_state_image<true>, dim3((ncp + 63) / 64), dim3(256), lds, e->stream, p, a);
	else hipLaunchKernelGGL(k_active_state_image<false>, dim3((ncp + 63) / 64), dim3(256), lds, e->stream, p, a);
}

// VerticalDynamicsFEM::FilterNegativeTracers (VerticalDynamicsFEM.cpp:4286-4347): per stored column and tracer
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
__global__ __launch_bounds__(256) void k_held_suarez(KParams p, double * __restrict__ x, const double * __restrict__ surf, double dt, double gamma, double kappa, double pscal)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = blockIdx.x * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (k >= L || col >= p.ncol) return;
	const unsigned cb = (unsigned)col * 8u;
	const double bsig = 0.7, kf = 1.0 / 86400.0, ka = (1.0 / 40.0) / 86400.0, ks = (1.0 / 4.0) / 86400.0;
	const double dty = 60.0, dthz = 10.0, tmin = 200.0, tmax = 315.0;
	const double ps = surf ? pscal * tmx_ref_exp(tmx_ref_log(ldu(surf, cb) * ldu(surf + NS, cb)) * gamma) : ldu(p.g2d + G2_PS * NS, cb);
	const double rho = ldu(x + TMX_SLAB_R(L, k) * NS, cb), rt = ldu(x + TMX_SLAB_T(L, k) * NS, cb);
	{
		const double pr = pscal * tmx_ref_exp(tmx_ref_log(rho * rt) * gamma);
		const double sigma = pr / ps;
		double bs = (sigma - bsig) / (1.0 - bsig);
		if (bs < 0.0) bs = 0.0;
		const double u = ldu(x + TMX_SLAB_U(L, k) * NS, cb), v = ldu(x + TMX_SLAB_V(L, k) * NS, cb);
		stu(x + TMX_SLAB_U(L, k) * NS, cb, u / (1.0 + kf * bs * dt));
		stu(x + TMX_SLAB_V(L, k) * NS, cb, v / (1.0 + kf * bs * dt));
	}
	const double pr = pscal * tmx_ref_exp(tmx_ref_log(rt) * gamma);
	const double sigma = pr / ps;
	double bs = (sigma - bsig) / (1.0 - bsig);
	if (bs < 0.0) bs = 0.0;
	const double T = pr / (rho * p.Rd);
	const double sl = ldu(p.g2d + G2_SINLAT * NS, cb), cl = ldu(p.g2d + G2_COSLAT * NS, cb);      // host libm values
	const double c4 = cl * cl * cl * cl;
	const double kt = ka + (ks - ka) * bs * c4;
	double teq = tmax - dty * sl * sl - dthz * tmx_ref_log(pr / p.p0) * cl * cl;
	teq *= tmx_ref_pow(pr / p.p0, kappa);      // glibc's pow, bit for bit (tmx_refmath.h)
	if (teq < tmin) teq = tmin;
	const double ddh = -kt / gamma * (1.0 + (gamma - 1.0) * teq / T);
	const double dh = -kt / gamma * (1.0 - teq / T);
	stu(x + TMX_SLAB_T(L, k) * NS, cb, rt * (1.0 + dt / (1.0 - dt * ddh) * dh));
}

void tmxk_held_suarez(tmx_engine * e, const KParams & p, double * x, const double * surf, double dt) {
	// PhysicalConstants::GetGamma / GetKappa / pressure scaling (src/atm/PhysicalConstants.h:355-376), host libm as in the reference
	const double gamma = p.cp / (p.cp - p.Rd), kappa = p.Rd / p.cp;
	const double pscal = p.p0 * pow(p.Rd / p.p0, gamma);
	dim3 blk(64, 4), grd(p.NS / 64, (p.L + 3) / 4);
	hipLaunchKernelGGL(k_held_suarez, grd, blk, 0, e->stream, p, x, surf, dt, gamma, kappa, pscal);
}

// Column physics: KesslerPhysics::Perform (test/dcmip2016/KesslerPhysics.cpp:83-285) around SUBROUTINE KESSLER
// (test/dcmip2016/interface/kessler.f90:64-185), Lorenz staggering / FORMULATION_RHOTHETA_PI; tracers 0,1,2 = RhoQv,
// RhoQc, RhoQr.  One lane per stored column (the reference visits every interior node of a patch, duplicated element-edge
// nodes included), three sweeps over the levels; the column's work arrays live in an HBM scratch [8][L][NS] (coalesced
// across lanes) instead of the Fortran automatic arrays.  The arithmetic is the subroutine's as the reference's build
// compiles it (mk/defs.make:35: no default-real-8 flag): r, rhalf, velqr, sed, pc are SINGLE precision, real literals
// without d0 are single precision values promoted to double, AMAX1 / AMIN1 convert their arguments to single precision
// (oracle/tmx_oracle.c:orc_kessler_column is the same restatement, bit for bit against the compiled Fortran).
// Level k only reads levels k and k+1 of the previous state of the sub-cycle (upstream sedimentation), so the
// sedimentation term is formed on the fly while sweeping upwards.
__device__ __forceinline__ double kes_amax1(double a, double b) { const float x = (float)a, y = (float)b; return (double)((x > y) ? x : y); }
__device__ __forceinline__ double kes_amin1(double a, double b, double c) {
	const float x = (float)a, y = (float)b, z = (float)c;
	const float m = (x < y) ? x : y;
	return (double)((m < z) ? m : z);
}
__device__ __forceinline__ float kes_velqr(double qr, float r, float rhalf) {
	return (float)(36.34 * tmx_ref_pow(qr * (double)r, (double)0.1364f) * (double)rhalf);      // kessler.f90:113, :170
}

enum { KES_TH = 0, KES_QV, KES_QC, KES_QR, KES_PK, KES_RHOD, KES_PC, KES_VEL, KES_NF };

__global__ __launch_bounds__(256) void k_kessler(KParams p, double * __restrict__ x, const double * __restrict__ zlev,
	double * __restrict__ w, double * __restrict__ prect, double dt, double gamma, double pscal)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= p.ncol) return;
#define WK(f, k) w[((size_t)(f) * L + (k)) * NS + col]
	const double xk = .2875, psl = 1000.0, rhoqr = 1000.0, f2x = 17.27;
	const double f5 = 237.3 * f2x * 2500000.0 / 1003.0;
	const double inv_xk = (double)1.0f / xk;
	// ---- KesslerPhysics.cpp:147-220: column inputs; kessler.f90:107-123: r, rhalf, pc, velqr, CFL limit ----
	double rhod0 = 0.0, dt_max = dt;
	float vel_prev = 0.0f; double z_prev = 0.0;
	for (int k = 0; k < L; k++) {
		const double rho = x[(size_t)TMX_SLAB_R(L, k) * NS + col], rt = x[(size_t)TMX_SLAB_T(L, k) * NS + col];
		const double t0 = x[(size_t)TMX_SLAB_Q(L, 0, k) * NS + col], t1 = x[(size_t)TMX_SLAB_Q(L, 1, k) * NS + col], t2 = x[(size_t)TMX_SLAB_Q(L, 2, k) * NS + col];
		const double thv = rt / rho;
		const double rhod = rho - t0 - t1 - t2;
		const double pr = pscal * tmx_ref_exp(tmx_ref_log(rho * thv) * gamma);
		const double tv = pr / (rho * p.Rd);
		double qv = t0 / rho; if (qv < 0.0) qv = 0.0;
		double qc = t1 / rho; if (qc < 0.0) qc = 0.0;
		double qr = t2 / rho; if (qr < 0.0) qr = 0.0;
		const double th = thv / (1.0 + 0.61 * qv);
		const double pk = tv / thv;
		if (k == 0) rhod0 = rhod;
		const float r = (float)(0.001 * rhod);
		const float rhalf = (float)sqrt(rhod0 / rhod);
		const float pc = (float)(3.8 / (tmx_ref_pow(pk, inv_xk) * psl));
		const float vel = kes_velqr(qr, r, rhalf);
		const double z = zlev[(size_t)k * NS + col];
		if (k > 0 && vel_prev != 0.0f) { const double c = 0.8 * (z - z_prev) / (double)vel_prev; dt_max = (dt_max < c) ? dt_max : c; }
		vel_prev = vel; z_prev = z;
		WK(KES_TH, k) = th; WK(KES_QV, k) = qv; WK(KES_QC, k) = qc; WK(KES_QR, k) = qr;
		WK(KES_PK, k) = pk; WK(KES_RHOD, k) = rhod; WK(KES_PC, k) = (double)pc; WK(KES_VEL, k) = (double)vel;
	}
	const int rainsplit = (int)ceil(dt / dt_max);           // kessler.f90:126-127
	const double dt0 = dt / (double)rainsplit;
	double precl = 0.0;
	for (int nt = 1; nt <= rainsplit; nt++) {
		// level k of the sweep needs the PREVIOUS values of level k+1: they are read before level k+1 is updated
		double qr_k = WK(KES_QR, 0), rhod_k = WK(KES_RHOD, 0), z_k = zlev[col];
		float vel_k = (float)WK(KES_VEL, 0), r_k = (float)(0.001 * rhod_k);
		precl = precl + rhod_k * qr_k * (double)vel_k / rhoqr;          // :135
		double z_km = 0.0;
		for (int k = 0; k < L; k++) {
			float sed;
			double qr_n = 0.0, rhod_n = 0.0, z_n = 0.0; float vel_n = 0.0f, r_n = 0.0f;
			if (k < L - 1) {
				qr_n = WK(KES_QR, k + 1); rhod_n = WK(KES_RHOD, k + 1); z_n = zlev[(size_t)(k + 1) * NS + col];
				vel_n = (float)WK(KES_VEL, k + 1); r_n = (float)(0.001 * rhod_n);
				sed = (float)(dt0 * ((double)r_n * qr_n * (double)vel_n - (double)r_k * qr_k * (double)vel_k) / ((double)r_k * (z_n - z_k)));     // :139-141
			} else {
				sed = (float)(-dt0 * qr_k * (double)vel_k / ((double)0.5f * (z_k - z_km)));       // :142
			}
			double th = WK(KES_TH, k), qv = WK(KES_QV, k), qc = WK(KES_QC, k), qr = qr_k;
			const double pk = WK(KES_PK, k), pc = WK(KES_PC, k);
			// :147-165
			const double qrprod = qc - (qc - dt0 * kes_amax1((double)0.001f * (qc - .001), 0.0)) / (1.0 + dt0 * 2.2 * tmx_ref_pow(qr, (double)0.875f));
			qc = kes_amax1(qc - qrprod, 0.0);
			qr = kes_amax1(qr + qrprod + (double)sed, 0.0);
			const double pt = pk * th;
			const double qvs = pc * tmx_ref_exp(f2x * (pt - 273.0) / (pt - 36.0));
			const double pt36 = pt - 36.0;
			const double prod = (qv - qvs) / (1.0 + qvs * f5 / (pt36 * pt36));
			const double rq = (double)r_k * qr;
			const double dimv = (qvs - qv > 0.0) ? qvs - qv : 0.0;
			const double ern = kes_amin1(dt0 * (((1.6 + 124.9 * tmx_ref_pow(rq, (double)0.2046f)) * tmx_ref_pow(rq, (double)0.525f))
					/ (2550000.0 * pc / (3.8 * qvs) + 540000.0)) * (dimv / ((double)r_k * qvs)),
				kes_amax1(-prod - qc, 0.0), qr);
			const double mx = (prod > -qc) ? prod : -qc;
			th = th + 2500000.0 / (1003.0 * pk) * (kes_amax1(prod, -qc) - ern);
			qv = kes_amax1(qv - mx + ern, 0.0);
			qc = qc + mx;
			qr = qr - ern;
			WK(KES_TH, k) = th; WK(KES_QV, k) = qv; WK(KES_QC, k) = qc; WK(KES_QR, k) = qr;
			if (nt != rainsplit) {                                        // :168-172
				const float rhalf = (float)sqrt(rhod0 / rhod_k);
				WK(KES_VEL, k) = (double)kes_velqr(qr, r_k, rhalf);
			}
			z_km = z_k;
			qr_k = qr_n; rhod_k = rhod_n; z_k = z_n; vel_k = vel_n; r_k = r_n;
		}
	}
	precl = precl / (double)rainsplit;
	prect[col] += precl * dt;                                           // KesslerPhysics.cpp:234
	// ---- KesslerPhysics.cpp:237-271: back to densities ----
	for (int k = 0; k < L; k++) {
		const double qv = WK(KES_QV, k), qc = WK(KES_QC, k), qr = WK(KES_QR, k), th = WK(KES_TH, k);
		const double rho = WK(KES_RHOD, k) / (1.0 - qv - qc - qr);
		x[(size_t)TMX_SLAB_R(L, k) * NS + col] = rho;
		x[(size_t)TMX_SLAB_Q(L, 0, k) * NS + col] = qv * rho;
		x[(size_t)TMX_SLAB_Q(L, 1, k) * NS + col] = qc * rho;
		x[(size_t)TMX_SLAB_Q(L, 2, k) * NS + col] = qr * rho;
		x[(size_t)TMX_SLAB_T(L, k) * NS + col] = rho * th * (1.0 + 0.61 * qv);
	}
#undef WK
}

// The same subroutine, level-parallel.  Inside one sub-cycle of the rain loop level k reads, besides its own values, only the
// PREVIOUS qr, velqr and r of level k + 1 (kessler.f90:139-141), so the levels of a column update independently between two
// barriers.  A workgroup = 64 columns x KT wavefronts; wavefront ty owns the levels ty, ty + KT, ... (at most KES_NO of them)
// and keeps their th, qv, qc, qr, pk, pc, rhod in registers; qr, velqr, r of all levels sit in LDS for the neighbour above.
// The CFL limit (:117-123) is the reference's sequential minimum over the candidates every level leaves in LDS; the number
// of sub-cycles differs between columns, so the loop runs to the largest of the wavefront (all wavefronts of a workgroup
// hold the same 64 columns) with the finished lanes idle.  Every value comes from the expression of k_kessler above: the
// results are bit-identical.  No HBM work arrays (ne30 L40: 0.18 GB per call instead of 1.02 GB) and 6 480 wavefronts instead of
// 810 (one lane per column leaves most SIMDs without a wavefront): 397 -> 275 us per call there.  What is left is the fp64
// arithmetic of nine exp / log / pow evaluations per level (about 140 us at the chip's fp64 rate).
// one level of one sub-cycle (kessler.f90:147-172) and the inputs of one level (KesslerPhysics.cpp:147-220, kessler.f90:107-115)
// as functions that are NOT inlined: the level-parallel kernel below calls them for the KES_NO levels a thread owns, and
// inlined five times over the compiler interleaves the copies and runs out of registers (256 + spills)
struct KesLevel { double th, qv, qc, qr; float vel; };
__device__ __attribute__((noinline)) KesLevel kes_sweep_level(KesLevel s, double pk_, double pc_, double rhod_k, double rhod0, float sed, double dt0, int new_vel)
{
	const double f2x = 17.27;
	const double f5 = 237.3 * f2x * 2500000.0 / 1003.0;
	const float r_k = (float)(0.001 * rhod_k);
	double th_ = s.th, qv_ = s.qv, qc_ = s.qc, qr_ = s.qr;
	const double qrprod = qc_ - (qc_ - dt0 * kes_amax1((double)0.001f * (qc_ - .001), 0.0)) / (1.0 + dt0 * 2.2 * tmx_ref_pow(qr_, (double)0.875f));
	qc_ = kes_amax1(qc_ - qrprod, 0.0);
	qr_ = kes_amax1(qr_ + qrprod + (double)sed, 0.0);
	const double pt = pk_ * th_;
	const double qvs = pc_ * tmx_ref_exp(f2x * (pt - 273.0) / (pt - 36.0));
	const double pt36 = pt - 36.0;
	const double prod = (qv_ - qvs) / (1.0 + qvs * f5 / (pt36 * pt36));
	const double rq = (double)r_k * qr_;
	const double dimv = (qvs - qv_ > 0.0) ? qvs - qv_ : 0.0;
	const double ern = kes_amin1(dt0 * (((1.6 + 124.9 * tmx_ref_pow(rq, (double)0.2046f)) * tmx_ref_pow(rq, (double)0.525f))
			/ (2550000.0 * pc_ / (3.8 * qvs) + 540000.0)) * (dimv / ((double)r_k * qvs)),
		kes_amax1(-prod - qc_, 0.0), qr_);
	const double mx = (prod > -qc_) ? prod : -qc_;
	th_ = th_ + 2500000.0 / (1003.0 * pk_) * (kes_amax1(prod, -qc_) - ern);
	qv_ = kes_amax1(qv_ - mx + ern, 0.0);
	qc_ = qc_ + mx;
	qr_ = qr_ - ern;
	KesLevel o; o.th = th_; o.qv = qv_; o.qc = qc_; o.qr = qr_; o.vel = s.vel;
	if (new_vel) {                                        // :168-172
		const float rhalf = (float)sqrt(rhod0 / rhod_k);
		o.vel = kes_velqr(qr_, r_k, rhalf);
	}
	return o;
}
struct KesInputs { double th, qv, qc, qr, pk, rhod; };
__device__ __attribute__((noinline)) KesInputs kes_level_inputs(double rho, double rt, double t0, double t1, double t2, double gamma, double pscal, double Rd)
{
	KesInputs o;
	const double thv = rt / rho;
	o.rhod = rho - t0 - t1 - t2;
	const double pr = pscal * tmx_ref_exp(tmx_ref_log(rho * thv) * gamma);
	const double tv = pr / (rho * Rd);
	double a = t0 / rho; if (a < 0.0) a = 0.0; o.qv = a;
	a = t1 / rho; if (a < 0.0) a = 0.0; o.qc = a;
	a = t2 / rho; if (a < 0.0) a = 0.0; o.qr = a;
	o.th = thv / (1.0 + 0.61 * o.qv);
	o.pk = tv / thv;
	return o;
}
struct KesPcVel { float pc, vel; };
__device__ __attribute__((noinline)) KesPcVel kes_level_pc_vel(double pk_, double qr_, double rhod_k, double rhod0)
{
	const double xk = .2875, psl = 1000.0;
	const double inv_xk = (double)1.0f / xk;
	KesPcVel o;
	const float r_k = (float)(0.001 * rhod_k);
	const float rhalf = (float)sqrt(rhod0 / rhod_k);
	o.pc = (float)(3.8 / (tmx_ref_pow(pk_, inv_xk) * psl));
	o.vel = kes_velqr(qr_, r_k, rhalf);
	return o;
}

#define KES_NO 5
template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_kessler_tile(KParams p, double * __restrict__ x, const double * __restrict__ zlev,
	double * __restrict__ prect, double dt, double gamma, double pscal)
{
	extern __shared__ double ksm[];
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int KT = blockDim.y;
	double * s_qr = ksm;                                           // [L][64]
	double * s_c = ksm + (size_t)L * 64;                           // [L][64] CFL candidate of the level pair (k, k+1)
	float * s_vel = (float *)(ksm + (size_t)2 * L * 64);           // [L][64]
	float * s_r = s_vel + (size_t)L * 64;                          // [L][64]
	double * s_rhod0 = (double *)(s_r + (size_t)L * 64);           // [64]
	const int tx = threadIdx.x, ty = WAVE_UNIFORM(threadIdx.y);
	const bool valid = blockIdx.x * 64 + tx < p.ncol;
	const int col = valid ? blockIdx.x * 64 + tx : p.ncol - 1;     // lanes past the last column repeat it and store nothing
	const double rhoqr = 1000.0;
	double th[KES_NO], qv[KES_NO], qc[KES_NO], qr[KES_NO], pk[KES_NO], rhod[KES_NO], dz[KES_NO];      // dz: z(k+1) - z(k); top level: z(k) - z(k-1)
	float vel[KES_NO], pc[KES_NO];
	// ---- KesslerPhysics.cpp:147-220: column inputs ----
#pragma unroll
	for (int m = 0; m < KES_NO; m++) {
		const int k = ty + KT * m;
		th[m] = qv[m] = qc[m] = qr[m] = pk[m] = dz[m] = 0.0; rhod[m] = 1.0; vel[m] = pc[m] = 0.0f;
		if (k >= L) continue;
		const double rho = x[(size_t)TMX_SLAB_R(L, k) * NS + col], rt = x[(size_t)TMX_SLAB_T(L, k) * NS + col];
		const double t0 = x[(size_t)TMX_SLAB_Q(L, 0, k) * NS + col], t1 = x[(size_t)TMX_SLAB_Q(L, 1, k) * NS + col], t2 = x[(size_t)TMX_SLAB_Q(L, 2, k) * NS + col];
		{
			const double z_k = zlev[(size_t)k * NS + col];
			dz[m] = (k < L - 1) ? zlev[(size_t)(k + 1) * NS + col] - z_k : z_k - ((k > 0) ? zlev[(size_t)(k - 1) * NS + col] : 0.0);
		}
		const KesInputs in_ = kes_level_inputs(rho, rt, t0, t1, t2, gamma, pscal, p.Rd);
		th[m] = in_.th; qv[m] = in_.qv; qc[m] = in_.qc; qr[m] = in_.qr; pk[m] = in_.pk; rhod[m] = in_.rhod;
		if (k == 0) s_rhod0[tx] = rhod[m];
		__builtin_amdgcn_sched_barrier(0);
	}
	__syncthreads();
	const double rhod0 = s_rhod0[tx];
	// ---- kessler.f90:107-123: r, rhalf, pc, velqr, CFL limit ----
#pragma unroll
	for (int m = 0; m < KES_NO; m++) {
		const int k = ty + KT * m;
		if (k >= L) continue;
		const float r_k = (float)(0.001 * rhod[m]);
		const KesPcVel pv_ = kes_level_pc_vel(pk[m], qr[m], rhod[m], rhod0);
		pc[m] = pv_.pc; vel[m] = pv_.vel;
		s_qr[k * 64 + tx] = qr[m]; s_vel[k * 64 + tx] = vel[m]; s_r[k * 64 + tx] = r_k;
		// what the column loop compares at level k + 1 (skipped there when velqr(k) is zero: +inf leaves the minimum alone)
		s_c[k * 64 + tx] = (k < L - 1 && vel[m] != 0.0f) ? 0.8 * dz[m] / (double)vel[m] : __builtin_inf();
		__builtin_amdgcn_sched_barrier(0);      // one level at a time: interleaving the unrolled levels costs 100 registers
	}
	__syncthreads();
	double dt_max = dt;
	for (int k = 0; k < L - 1; k++) { const double c = s_c[k * 64 + tx]; dt_max = (dt_max < c) ? dt_max : c; }
	const int rainsplit = (int)ceil(dt / dt_max);           // kessler.f90:126-127
	const double dt0 = dt / (double)rainsplit;
	double precl = 0.0;
	for (int nt = 1; __any(valid && nt <= rainsplit); nt++) {
		const bool on = nt <= rainsplit;
		// the previous values of the level above, read before anybody updates them
		double qr_n[KES_NO]; float vel_n[KES_NO], r_n[KES_NO];
#pragma unroll
		for (int m = 0; m < KES_NO; m++) {
			const int k = ty + KT * m;
			qr_n[m] = 0.0; vel_n[m] = r_n[m] = 0.0f;
			if (k < L - 1) { qr_n[m] = s_qr[(k + 1) * 64 + tx]; vel_n[m] = s_vel[(k + 1) * 64 + tx]; r_n[m] = s_r[(k + 1) * 64 + tx]; }
		}
		__syncthreads();
#pragma unroll
		for (int m = 0; m < KES_NO; m++) {
			const int k = ty + KT * m;
			if (k >= L || !on) continue;
			const float r_k = (float)(0.001 * rhod[m]), vel_k = vel[m];
			const double qr_k = qr[m];
			if (k == 0) precl = precl + rhod[m] * qr_k * (double)vel_k / rhoqr;          // :135
			float sed;
			if (k < L - 1)
				sed = (float)(dt0 * ((double)r_n[m] * qr_n[m] * (double)vel_n[m] - (double)r_k * qr_k * (double)vel_k) / ((double)r_k * dz[m]));     // :139-141
			else
				sed = (float)(-dt0 * qr_k * (double)vel_k / ((double)0.5f * dz[m]));       // :142
			KesLevel lv; lv.th = th[m]; lv.qv = qv[m]; lv.qc = qc[m]; lv.qr = qr_k; lv.vel = vel_k;
			lv = kes_sweep_level(lv, pk[m], (double)pc[m], rhod[m], rhod0, sed, dt0, nt != rainsplit);
			th[m] = lv.th; qv[m] = lv.qv; qc[m] = lv.qc; qr[m] = lv.qr;
			s_qr[k * 64 + tx] = lv.qr;
			if (nt != rainsplit) { vel[m] = lv.vel; s_vel[k * 64 + tx] = lv.vel; }
			__builtin_amdgcn_sched_barrier(0);
		}
		__syncthreads();
	}
	precl = precl / (double)rainsplit;
	if (ty == 0 && valid) prect[col] += precl * dt;                     // KesslerPhysics.cpp:234
	// ---- KesslerPhysics.cpp:237-271: back to densities ----
#pragma unroll
	for (int m = 0; m < KES_NO; m++) {
		const int k = ty + KT * m;
		if (k >= L || !valid) continue;
		const double rho = rhod[m] / (1.0 - qv[m] - qc[m] - qr[m]);
		x[(size_t)TMX_SLAB_R(L, k) * NS + col] = rho;
		x[(size_t)TMX_SLAB_Q(L, 0, k) * NS + col] = qv[m] * rho;
		x[(size_t)TMX_SLAB_Q(L, 1, k) * NS + col] = qc[m] * rho;
		x[(size_t)TMX_SLAB_Q(L, 2, k) * NS + col] = qr[m] * rho;
		x[(size_t)TMX_SLAB_T(L, k) * NS + col] = rho * th[m] * (1.0 + 0.61 * qv[m]);
	}
}

void tmxk_kessler(tmx_engine * e, const KParams & p, double * x, double dt) {
	const double gamma = p.cp / (p.cp - p.Rd);
	const double pscal = p.p0 * pow(p.Rd / p.p0, gamma);
	// level-parallel form up to 8 x KES_NO = 40 levels (TMX_KESSLER_COLUMN=1 and taller columns: the one-lane-per-column kernel)
	const int kt = (p.L + KES_NO - 1) / KES_NO;
	const size_t lds = (size_t)p.L * 64 * (2 * sizeof(double) + 2 * sizeof(float)) + 64 * sizeof(double);
	if (kt <= 8 && !getenv("TMX_KESSLER_COLUMN")) {
		if (lds > 48 * 1024) hipFuncSetAttribute((const void *)k_kessler_tile<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(k_kessler_tile<512>, dim3((p.ncol + 63) / 64), dim3(64, kt), lds, e->stream, p, x, (const double *)e->d_zlev, e->d_prect, dt, gamma, pscal);
		return;
	}
	hipLaunchKernelGGL(k_kessler, dim3((p.ncol + 255) / 256), dim3(256), 0, e->stream, p, x, (const double *)e->d_zlev, e->d_kes, e->d_prect, dt, gamma, pscal);
}

// gather the copies other ranks need into the send buffer, one contiguous message per peer:
// sendbuf = [peer][slab][count_peer]; entry i of the plan lives at base[i] + slab * stride[i]
__global__ __launch_bounds__(256) void k_pack(int nslab, size_t NS, int nsend,
	const double * __restrict__ x, const int * __restrict__ cols, const int * __restrict__ base,
	const int * __restrict__ stride, double * __restrict__ buf)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const int sl = blockIdx.y;
	if (i >= nsend) return;
	buf[(size_t)base[i] + (size_t)sl * stride[i]] = x[(size_t)sl * NS + cols[i]];
}

void tmxk_pack(tmx_engine * e, const KParams & p, const double * x) {
	if (e->nsend == 0) return;
	dim3 blk(256), grd((e->nsend + 255) / 256, e->nslab);
	hipLaunchKernelGGL(k_pack, grd, blk, 0, e->stream, e->nslab, (size_t)p.NS, e->nsend,
		x, (const int *)e->d_send_cols, (const int *)e->d_send_base, (const int *)e->d_send_stride, e->d_sendbuf);
}

// Peer-to-peer transport (tmx_halo_p2p_connect): the same gather, written straight into the neighbour ranks' ghost buffers
// (mapped through HIP IPC; over xGMI these are posted writes).  dst[peer] = where this rank's message starts in the peer's
// buffer of the current parity; within a message the wire layout is the one above, [slab][count_peer].
__global__ __launch_bounds__(256) void k_pack_p2p(int nslab, size_t NS, int nsend,
	const double * __restrict__ x, const int * __restrict__ cols, const int * __restrict__ within,
	const int * __restrict__ stride, const int * __restrict__ peer, double * const * __restrict__ dst)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const int sl = blockIdx.y;
	if (i >= nsend) return;
	dst[peer[i]][(size_t)sl * stride[i] + within[i]] = x[(size_t)sl * NS + cols[i]];
}

// After the messages: raise this rank's arrival counter in every neighbour's block to seq (release, system scope: the
// preceding kernel's writes are complete at its end), then wait until every neighbour has raised its counter here.  A
// neighbour that never arrives (a rank died) sets the communication bit of the engine's failure flag after TMX_P2P_TIMEOUT_S
// (default 600 s) instead of hanging the GPU.
__global__ __launch_bounds__(64) void k_p2p_signal_wait(int npeers, const int * __restrict__ peers, unsigned long long * const * __restrict__ flag_dst,
	const unsigned long long * own, unsigned long long seq, int * __restrict__ fail, unsigned long long timeout_ticks)
{
	for (int q = threadIdx.x; q < npeers; q += 64) {
		__threadfence_system();
		__hip_atomic_store(flag_dst[peers[q]], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	// An exchange of this engine has already timed out: its state is lost anyway (tmx_sync will say so), so the exchanges
	// queued behind it do not wait out the time-out again, one after the other.
	const bool failed = (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TMX_FLAG_COMM) != 0;
	for (int q = threadIdx.x; q < npeers && !failed; q += 64) {
		const unsigned long long t0 = wall_clock64();
		while (__hip_atomic_load(own + peers[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
			__builtin_amdgcn_s_sleep(8);
			if (timeout_ticks && wall_clock64() - t0 > timeout_ticks) { atomicOr(fail, TMX_FLAG_COMM); break; }      // 100 MHz counter; 0 = wait for ever
		}
	}
	__threadfence_system();
}

void tmxk_pack_p2p(tmx_engine * e, const KParams & p, const double * x, int buf) {
	if (e->nsend == 0) return;
	dim3 blk(256), grd((e->nsend + 255) / 256, e->nslab);
	hipLaunchKernelGGL(k_pack_p2p, grd, blk, 0, e->stream, e->nslab, (size_t)p.NS, e->nsend,
		x, (const int *)e->d_send_cols, (const int *)e->d_send_within, (const int *)e->d_send_stride, (const int *)e->d_send_peer,
		(double * const *)(e->d_p2p_dst + (size_t)buf * e->cfg.n_ranks));
}

void tmxk_p2p_signal_wait(tmx_engine * e, hipStream_t s, int buf, unsigned long long seq) {
	// how long a neighbour may lag (it may be busy on its host: output, a slower node) before the exchange gives up
	// (TMX_P2P_TIMEOUT_S, default 600 s, 0 = never: under MPI a rank that writes output for minutes is slow, not dead)
	const unsigned long long timeout_ticks = 100000000ull * (unsigned long long)e->p2p_timeout_s;
	hipLaunchKernelGGL(k_p2p_signal_wait, dim3(1), dim3(64), 0, s, e->p2p_npeers, (const int *)e->d_p2p_peers,
		(unsigned long long * const *)(e->d_p2p_flag + (size_t)buf * e->cfg.n_ranks),
		(const unsigned long long *)e->p2p_block + (size_t)buf * e->cfg.n_ranks, seq, e->d_flag, timeout_ticks);
}

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

template <bool PULL>
__global__ __launch_bounds__(256) void k_hypervis(KParams p, const double * __restrict__ xsrc,
	const double * __restrict__ xbase, double * __restrict__ xout, double dt, double nu_s_in, double nu_d_in, double nu_v_in, int scale, DssPull pull)
{
	__shared__ double s[4][8][64];
	__shared__ double sD[16], sS[16];
	const int lane = threadIdx.x, ty = WAVE_UNIFORM(threadIdx.y);
	const int tid = ty * 64 + lane;
	if (tid < 16) { sD[tid] = p.dx[tid]; sS[tid] = p.stiff[tid]; }
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = (p.quads ? p.quads[blockIdx.x] : blockIdx.x) * 64 + lane;
	const int kk = blockIdx.y * 4 + ty;
	const bool actw = (kk <= L) && (col < p.ncol);     // interface level (W)
	const bool actn = (kk < L) && (col < p.ncol);      // model level (U,V,rho*theta,rho)
	const int q = lane & 15, i = q >> 2, j = q & 3, eb = lane & 48;

	double psiT = 0, psiR = 0, psiW = 0, ua = 0, ub = 0;
	double c2a0 = 0, c2a1 = 0, c2b1 = 0, j2d = 1, jn = 1, je = 1;
	if (actw) {
		c2a0 = p.g2d[G2_C2A0 * NS + col]; c2a1 = p.g2d[G2_C2A1 * NS + col]; c2b1 = p.g2d[G2_C2B1 * NS + col];
		j2d = p.g2d[G2_J2D * NS + col]; jn = p.g2d[G2_JN * NS + col]; je = p.g2d[G2_JE * NS + col];
		if (!PULL) psiW = xsrc[TMX_SLAB_W(L, kk) * NS + col];
	}
	// the base of the update (second pass: the instance the viscosity is applied to), loaded with the operands: at the point
	// of use its latency would follow three barriers and both contractions
	double bW = 0.0, bT = 0.0, bR = 0.0, bU = 0.0, bV = 0.0;
	if (xbase && actw) {
		bW = xbase[TMX_SLAB_W(L, kk) * NS + col];
		if (actn) {
			bT = xbase[TMX_SLAB_T(L, kk) * NS + col]; bR = xbase[TMX_SLAB_R(L, kk) * NS + col];
			bU = xbase[TMX_SLAB_U(L, kk) * NS + col]; bV = xbase[TMX_SLAB_V(L, kk) * NS + col];
		}
	}
	if (!PULL && actn) {
		psiT = xsrc[TMX_SLAB_T(L, kk) * NS + col];
		psiR = xsrc[TMX_SLAB_R(L, kk) * NS + col];
		ua = xsrc[TMX_SLAB_U(L, kk) * NS + col];
		ub = xsrc[TMX_SLAB_V(L, kk) * NS + col];
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
	s[ty][0][lane] = psiT; s[ty][1][lane] = psiR; s[ty][2][lane] = psiW;
	s[ty][3][lane] = ua;   s[ty][4][lane] = ub;
	s[ty][5][lane] = j2d * cua; s[ty][6][lane] = j2d * cub;
	__syncthreads();

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
	__syncthreads();
	s[ty][0][lane] = jn * (c2a0 * daT + c2a1 * dbT); s[ty][1][lane] = jn * (c2a1 * daT + c2b1 * dbT);
	s[ty][2][lane] = jn * (c2a0 * daR + c2a1 * dbR); s[ty][3][lane] = jn * (c2a1 * daR + c2b1 * dbR);
	s[ty][4][lane] = je * (c2a0 * daW + c2a1 * dbW); s[ty][5][lane] = je * (c2a1 * daW + c2b1 * dbW);
	s[ty][6][lane] = dv; s[ty][7][lane] = cl;
	__syncthreads();
	if (!actw) return;

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

	xout[TMX_SLAB_W(L, kk) * NS + col] = bW - dt * (1.0 / je) * nu_s * (uaW + ubW);
	if (actn) {
		const double invJ = 1.0 / jn;
		xout[TMX_SLAB_T(L, kk) * NS + col] = bT - dt * invJ * nu_s * (uaT + ubT);
		xout[TMX_SLAB_R(L, kk) * NS + col] = bR - dt * invJ * nu_s * (uaR + ubR);
		const double upa = nu_d * dad - nu_v * j2d * (c2a1 * dac + c2b1 * dbc);
		const double upb = nu_d * dbd + nu_v * j2d * (c2a0 * dac + c2a1 * dbc);
		xout[TMX_SLAB_U(L, kk) * NS + col] = bU - dt * upa;
		xout[TMX_SLAB_V(L, kk) * NS + col] = bV - dt * upb;
	}
}

void tmxk_hypervis(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout,
	double dt, double nu_s, double nu_d, double nu_v, int scale, bool pull_dss) {
	dim3 blk(64, 4), grd(p.quads ? e->launch_tiles : p.NS / 64, (p.L + 1 + 3) / 4);
	DssPull d = { (const int4 *)e->d_colref, (const double *)e->d_xmat, (const double *)e->d_ghost, (const int *)e->d_ghost_base, (const int *)e->d_ghost_stride };
	if (pull_dss) hipLaunchKernelGGL(k_hypervis<true>, grd, blk, 0, e->stream, p, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
	else hipLaunchKernelGGL(k_hypervis<false>, grd, blk, 0, e->stream, p, xsrc, xbase, xout, dt, nu_s, nu_d, nu_v, scale, d);
}

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
// Output interpolation: GridPatchCSGLL::InterpolateData (src/atm/GridPatchCSGLL.cpp:1365-1780) for DataType_State /
// DataType_Tracers.  One thread per (sample point, output level r); per field: the column value at source level l
// is the 4 x 4 Lagrange sum over the containing element (16 consecutive device columns), minus the same sum over the
// reference state when it is removed, then LinearColumnInterpFEM::Apply over the source levels with non-zero
// coefficient.  Same statements in the same order as the reference / the C oracle.
__device__ __forceinline__ double interp_point(const double * __restrict__ x, const double * __restrict__ xref, size_t NS,
	int slab0, int nlev, int col0, const double (&ca)[4], const double (&cb)[4], const double * __restrict__ oprow, bool divide, double dv)
{
	double acc = 0.0;
	for (int l = 0; l < nlev; l++) {
		const double oc = oprow[l];
		if (oc == 0.0) continue;
		const size_t so = (size_t)(slab0 + l) * NS + col0;
		double v = 0.0;
#pragma unroll
		for (int m = 0; m < 4; m++)
#pragma unroll
			for (int n = 0; n < 4; n++) {
				if (divide) v += ca[m] * cb[n] * x[so + m * 4 + n] / dv;
				else v += ca[m] * cb[n] * x[so + m * 4 + n];
			}
		if (xref) {
#pragma unroll
			for (int m = 0; m < 4; m++)
#pragma unroll
				for (int n = 0; n < 4; n++) v -= ca[m] * cb[n] * xref[so + m * 4 + n];
		}
		acc += oc * v;
	}
	return acc;
}

__global__ __launch_bounds__(256) void k_interp_state(KParams p, InterpArgs a, const double * __restrict__ x, const double * __restrict__ xref,
	int only_at, int primitive, double earth_radius, double * __restrict__ out)
{
	const int pt = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
	if (pt >= a.npts) return;
	const int L = p.L;
	const size_t NS = (size_t)p.NS, plane = (size_t)a.nreta * a.npts, o = (size_t)r * a.npts + pt;
	const int col0 = a.col0[pt];
	double res[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
	if (col0 >= 0) {
		double ca[4], cb[4];
#pragma unroll
		for (int m = 0; m < 4; m++) { ca[m] = a.ca[(size_t)pt * 4 + m]; cb[m] = a.cb[(size_t)pt * 4 + m]; }
		const double * on = a.opn + (size_t)r * L, * oe = a.ope + (size_t)r * (L + 1);
		if (only_at != 2) {
			res[0] = interp_point(x, xref, NS, TMX_SLAB_U(L, 0), L, col0, ca, cb, on, false, 1.0);
			res[1] = interp_point(x, xref, NS, TMX_SLAB_V(L, 0), L, col0, ca, cb, on, false, 1.0);
			res[2] = interp_point(x, xref, NS, TMX_SLAB_T(L, 0), L, col0, ca, cb, on, false, 1.0);
			res[4] = interp_point(x, xref, NS, TMX_SLAB_R(L, 0), L, col0, ca, cb, on, false, 1.0);
		}
		if (only_at != 1)
			res[3] = interp_point(x, xref, NS, TMX_SLAB_W(L, 0), L + 1, col0, ca, cb, oe, primitive != 0, p.g2d[G2_DRX * NS + col0]);
		if (primitive) {
			const double ua = res[0] / earth_radius, ub = res[1] / earth_radius;
			const double * M = a.rll + (size_t)pt * 4;
			res[0] = M[0] * ua + M[1] * ub;
			res[1] = M[2] * ua + M[3] * ub;
		}
	}
#pragma unroll
	for (int c = 0; c < 5; c++) out[(size_t)c * plane + o] = res[c];
}

__global__ __launch_bounds__(256) void k_interp_tracers(KParams p, InterpArgs a, int nt, const double * __restrict__ x, double * __restrict__ out)
{
	const int pt = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
	if (pt >= a.npts) return;
	const int L = p.L;
	const size_t NS = (size_t)p.NS, plane = (size_t)a.nreta * a.npts, o = (size_t)r * a.npts + pt;
	const int col0 = a.col0[pt];
	double ca[4] = { 0, 0, 0, 0 }, cb[4] = { 0, 0, 0, 0 };
	if (col0 >= 0) {
#pragma unroll
		for (int m = 0; m < 4; m++) { ca[m] = a.ca[(size_t)pt * 4 + m]; cb[m] = a.cb[(size_t)pt * 4 + m]; }
	}
	for (int c = 0; c < nt; c++)
		out[(size_t)c * plane + o] = (col0 >= 0) ? interp_point(x, nullptr, NS, TMX_SLAB_Q(L, c, 0), L, col0, ca, cb, a.opn + (size_t)r * L, false, 1.0) : 0.0;
}

void tmxk_interp_state(tmx_engine * e, const KParams & p, const InterpArgs & a, const double * x, const double * xref,
	int only_at, int primitive, double earth_radius, double * out) {
	hipLaunchKernelGGL(k_interp_state, dim3((a.npts + 255) / 256, a.nreta), dim3(256), 0, e->stream, p, a, x, xref, only_at, primitive, earth_radius, out);
}
void tmxk_interp_tracers(tmx_engine * e, const KParams & p, const InterpArgs & a, const double * x, double * out) {
	hipLaunchKernelGGL(k_interp_tracers, dim3((a.npts + 255) / 256, a.nreta), dim3(256), 0, e->stream, p, a, e->nt, x, out);
}

// ---------------------------------------------------------------------------------------------
// stage algebra: X_d <- c_d X_d + sum_m c_m X_m, accumulated in the reference's order
// (GridPatch::LinearCombineData, src/atm/GridPatch.cpp:1436-1508)


__global__ __launch_bounds__(256) void k_lincomb(size_t n, double * __restrict__ dst, LinTerms t) {
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	for (; i < n; i += stride) {
		double v = t.premul ? dst[i] * t.coef[0] : 0.0;
		for (int m = 1; m < t.n; m++) v += t.coef[m] * t.src[m][i];
		dst[i] = v;
	}
}

void tmxk_lincomb(tmx_engine * e, size_t n, double * dst, int nterms, const double * const * src, const double * coef, int premul) {
	LinTerms t;
	t.n = nterms; t.premul = premul;
	for (int m = 0; m < nterms; m++) { t.src[m] = src[m]; t.coef[m] = coef[m]; }
	const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
	hipLaunchKernelGGL(k_lincomb, dim3(blocks), dim3(256), 0, e->stream, n, dst, t);
}

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
	const int col = ucol[u];
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
	const int col = ucol[u];
	int dep[3];
#pragma unroll
	for (int t = 0; t < 3; t++) dep[t] = udep[u * 3 + t];
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
template <bool SPARSE, bool PRED, int TYPE>
__device__ __forceinline__ void lu_store_urow(const double (&piv)[TMX_BW], double pb, double * __restrict__ AB, double * __restrict__ RHS,
	int jcol, int NUS, int u, int * umask)
{
#ifdef TMX_ABL_NO_USTORE     // ablation builds: elimination without the U-row stream (one value kept alive)
	if (piv[0] == 12345.678 && pb == 1.0) AB[0] = piv[1];
	return;
#endif
#if TMX_USTREAM16
	auto val = [&](int e) -> double { return e == 9 ? pb : piv[e]; };
	double * row = AB + (size_t)jcol * TMX_RS(NUS);
#pragma unroll
	for (int sl = 0; sl < 3; sl++)
		*(double2 *)(row + sl * 128) = make_double2(val(TMX_UPAIR[TYPE][sl][0]), val(TMX_UPAIR[TYPE][sl][1]));
	int mask = 0;
	int one = 1;
	if (PRED) asm volatile("" : "+v"(one));
#pragma unroll
	for (int sl = 3; sl < 5; sl++) {
		const double a = val(TMX_UPAIR[TYPE][sl][0]), b = val(TMX_UPAIR[TYPE][sl][1]);
		const bool any = !SPARSE || __builtin_amdgcn_ballot_w64(a != 0.0 || b != 0.0) != 0;
		mask |= any ? (1 << sl) : 0;
		if (PRED ? (any && one) : any) *(double2 *)(row + sl * 128) = make_double2(a, b);
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

// Pivot step when every lane of the wavefront picked window row PV (the normal case: neighbouring columns have
// nearly identical matrices).  The interchange is a compile-time renaming and the rank-1 update writes straight
// into the slid position -- new row r-1, column c-1  <-  (row that sits at r after the swap)[c] - l_r * pivot[c] --
// so neither the 160 v_cndmask of the per-lane interchange nor the 36 moves of the slide are executed.  Only the
// pivot row and old row 0 are copied aside, because their slots are overwritten before their last use.
// Same multiplications and subtractions on the same operands as the generic path: bit-identical.
template <bool SPARSE, int PV, int IS, int TYPE>
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
#if TMX_TAIL == 1
	lu_store_urow<SPARSE, false, TYPE>(piv, pb, AB, RHS, jcol, NUS, u, umask);
	const double rcp = 1.0 / piv[0];
#pragma unroll
	for (int r = 1; r < 5; r++) {
		// after the interchange position r holds old row r, except position PV, which holds old row 0
		const double l = ((r == PV) ? t0[0] : R[r][0]) * rcp;
#pragma unroll
		for (int c = 1; c < TMX_BW; c++) {
			const double src = (r == PV) ? t0[c] : ((c == Z && r != 4) ? 0.0 : R[r][c]);
			R[r - 1][c - 1] = LU_UPD(src, l, piv[c]);
		}
		R[r - 1][TMX_BW - 1] = 0.0;
		b[r - 1] = LU_UPD((r == PV) ? tb0 : b[r], pb, l);
	}
#pragma unroll
	for (int c = 0; c < TMX_BW; c++) R[4][c] = inrow[c * IS];      // IS: stride of the incoming row (1 registers, 64 LDS ring)
	b[4] = inb;
#else
	// The reciprocal of the pivot and the four multipliers are started BEFORE the U-row stores: the IEEE division is a chain
	// of a dozen dependent instructions, and the (conditional) stores -- which nothing waits for -- cover its latency.  The
	// first entry of the incoming row is what the NEXT pivot search waits for: its LDS read is issued up front as well.
	const double in0 = inrow[0];
	const double rcp = 1.0 / piv[0];
	double l[5];
#pragma unroll
	for (int r = 1; r < 5; r++) l[r] = ((r == PV) ? t0[0] : R[r][0]) * rcp;
	lu_store_urow<SPARSE, TMX_STORE_PRED, TYPE>(piv, pb, AB, RHS, jcol, NUS, u, umask);
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
	for (int c = 1; c < TMX_BW; c++) R[4][c] = inrow[c * IS];      // IS: stride of the incoming row (1 registers, 64 LDS ring)
	b[4] = inb;
#endif
}

// One elimination step.  UNI: try the wave-uniform specialisations first (a ballot checks that all lanes agree on
// the pivot row); lanes that disagree -- and kernels without the register budget for five specialised tails -- take
// the per-lane select path below.
template <int TYPE, bool SPARSE, bool UNI = false, int IS = 1, int EXPECT = -1>
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
		if (__builtin_amdgcn_ballot_w64(pv != pv0) == 0) {
			if (n_uniform) (*n_uniform)++;      // wave-uniform counter (scalar add): pivot steps that took the renaming path
			// the usual pivot row of this kind of matrix row first (rho*theta / W / rho rows pivot 4 / 3 / 2 rows down almost
			// everywhere): one compare and branch instead of the switch's two or three
			if (EXPECT >= 0 && pv0 == EXPECT) { lu_pivot_tail_uniform<SPARSE, (EXPECT >= 0 ? EXPECT : 0), IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return; }
			switch (pv0) {
				case 0: lu_pivot_tail_uniform<SPARSE, 0, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				case 1: lu_pivot_tail_uniform<SPARSE, 1, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				case 2: lu_pivot_tail_uniform<SPARSE, 2, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				case 3: lu_pivot_tail_uniform<SPARSE, 3, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
				default: lu_pivot_tail_uniform<SPARSE, 4, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask); return;
			}
		}
#ifndef TMX_VI_SELECT_PATH
		// The lanes disagree (rare: neighbouring columns have nearly the same matrix): the same five tails, one pivot row at
		// a time under the lanes that chose it.  A lane executes exactly the tail of its own pivot row, i.e. the statements
		// of the per-lane select path below on the same operands; every U-row entry is stored (a fill-in entry that is zero
		// in the lanes of one tail need not be zero in the others, so the all-lanes ballot of the sparse stores does not
		// apply).  Unlike the select path -- whose 160 selects per step cost the kernel 30 spilled registers, reloaded in
		// the uniform tails behind an s_waitcnt vmcnt(0) that also drained the U-row stores -- this adds no live values.
		if (pv == 0) lu_pivot_tail_uniform<false, 0, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 1) lu_pivot_tail_uniform<false, 1, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 2) lu_pivot_tail_uniform<false, 2, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 3) lu_pivot_tail_uniform<false, 3, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
		if (pv == 4) lu_pivot_tail_uniform<false, 4, IS, TYPE>(R, b, singular, AB, RHS, jcol, NUS, u, inrow, inb, umask);
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
	lu_store_urow<SPARSE, false, TYPE>(R[0], b[0], AB, RHS, jcol, NUS, u, umask);
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
	for (int c = 0; c < TMX_BW; c++) R[4][c] = inrow[c * IS];      // IS: stride of the incoming row (1 registers, 64 LDS ring)
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

template <int COMP>
__device__ __forceinline__ void vi_load_urow(URow & d, int row, int L, size_t NS, int u, int col, int NUS,
	const double * __restrict__ AB, const double * __restrict__ RHS, const double * xin, const int * umask, const double * zp)
{
	const int rr = row < 0 ? 0 : row;
	const int mk = __builtin_amdgcn_readfirstlane(umask[rr]);
#if TMX_USTREAM16
	const double * rb = AB + (size_t)rr * TMX_RS(NUS);
	double2 v[5];
#pragma unroll
	for (int sl = 0; sl < 3; sl++) v[sl] = *(const double2 *)(rb + sl * 128);
#pragma unroll
	for (int sl = 3; sl < 5; sl++) {
		const double * bp = (mk & (1 << sl)) ? rb + sl * 128 : zp;
		v[sl] = *(const double2 *)bp;
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
#ifdef TMX_EXP_BRANCHY_BACK
		d.u[c] = 0.0;
		if (mk & (1 << c)) d.u[c] = rb[c * 64];
#else
		const double * bp = (mk & (1 << c)) ? rb + c * 64 : zp;
		d.u[c] = bp[0];
#endif
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

__device__ __forceinline__ void vi_back_substitute(int L, size_t NS, int n, int u, int col, int NUS,
	const double * __restrict__ AB, const double * __restrict__ RHS, const double * xin, double * xup,
	const int * __restrict__ udep, const int * umask, int zlane = -1)
{
	int dep[3];
#pragma unroll
	for (int t = 0; t < 3; t++) dep[t] = udep[u * 3 + t];
	double xw[TMX_BW - 1];
#pragma unroll
	for (int c = 0; c < TMX_BW - 1; c++) xw[c] = 0.0;
	// zero page behind the streams (one slot per lane); AB is the lane's stream pointer (vi_stream_base)
	const double * zp = RHS + ((zlane >= 0) ? zlane : (u & 63)) * TMX_USLOT;
	// two trips (six rows) of register sets: a set is refilled for the trip after next right after it is consumed,
	// so five rows = 55 loads are in flight (vmcnt holds 63) -- the back substitution of a full grid is HBM-bound
	URow A, B, C, D, E, F;      // A,D: rows 3k+2 (rho); B,E: 3k+1 (W); C,F: 3k (rho*theta)
#define LOADROW(S, COMP, row) vi_load_urow<COMP>(S, row, L, NS, u, col, NUS, AB, RHS, xin, umask, zp)
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
	const int col = ucol[u];
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
// elimination is sequential.  Wavefront 0 of the workgroup assembles block row t (compute_block, the same code)
// into a double-buffered LDS ring while wavefront 1 runs the three pivot steps of block row t-1; one barrier
// per block row.  The dependent chain of a column shrinks from assembly + LU to max(assembly, LU) per level
// (cycle counts in DESIGN.md), which is what bounds the kernel on small per-GPU grids.  Same statements on the
// same operands: bit-identical to k_vi_fused.
// BACK = false: the kernel ends with the elimination and leaves the fill-in masks of its U rows in `gmask` ([column group][3(L+1)]);
// k_vi_back does the back substitution as a launch of its own (one wavefront per workgroup, so that the HBM-bound half of the
// solve is spread evenly over the CUs whatever the residency of the pairs was).
template <bool SPARSE, bool CLOSED, int PAIRS, bool BACK = true>
__global__ __launch_bounds__(128 * PAIRS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vi_pair(KParams p, const double * xin, double * xup, double dt,
	int nunique, int NUS, const int * __restrict__ ucol, const int * __restrict__ udep,
	double * __restrict__ AB, double * __restrict__ RHS, int * __restrict__ flag, int cpw, unsigned long long * __restrict__ stats, int * __restrict__ gmask = nullptr,
	int stagger = 0)
{
	extern __shared__ double opsl_mem[];
	double * opsl = opsl_mem;
	const int L = p.L;
	constexpr int MM = CLOSED ? 1 : 2;
	constexpr int NQ = 3 * TMX_BW + 3;                                       // doubles per block row and column
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
#ifndef TMX_PAIR3_INTERLEAVED
	const int role = (PAIRS == 3) ? ((0x31 >> wv) & 1) : (wv & 1), pair = (PAIRS == 3) ? ((0x990 >> (2 * wv)) & 3) : (wv >> 1);
#else
	const int role = wv & 1, pair = wv >> 1;
#endif
	double * ring = etal + 2 * L + 1 + (size_t)pair * 2 * NQ * 64;           // [PAIRS][2][NQ][64]
	double * rmtab = etal + 2 * L + 1 + (size_t)PAIRS * 2 * NQ * 64;        // [512] exp / log tables (exner_from_rhotheta_lds)
	int * umask = (int *)(rmtab + TMX_RMTAB_DOUBLES) + pair * TMX_FTOT * (L + 1);   // [PAIRS][3(L+1)]
	for (int t = wv * 64 + lane; t < TMX_OP_COUNT * (L + 1) * TMX_OPW; t += 128 * PAIRS) opsl[t] = p.ops[t];
	if (CLOSED) for (int t = wv * 64 + lane; t < 2 * L + 1; t += 128 * PAIRS) etal[t] = p.eta[t];
	tmx_rm_tables_to_lds(rmtab, wv * 64 + lane, 128 * PAIRS);
	__syncthreads();
	const size_t NS = (size_t)p.NS;
	// cpw columns per wavefront (64, or fewer so that the column groups fill every SIMD of the chip evenly: tmxk_vi_fused).
	// Lanes beyond the group's columns do not return -- the block-row loops below are full of workgroup barriers -- they
	// redo the group's last column (the last column of the grid in a trailing, empty group): same values to the same
	// addresses of the state, a stream slot of their own.
	const int grp = blockIdx.x * PAIRS + pair;
	const int ulast = min(nunique, (grp + 1) * cpw) - 1;
	const int u = (grp * cpw < nunique) ? min(grp * cpw + min(lane, cpw - 1), ulast) : nunique - 1;
	const int n = TMX_FTOT * (L + 1);
	const int col = ucol[u];
	RHS = AB + (size_t)n * TMX_UW * NUS;      // zero page behind the streams (vi_back_substitute)
	AB = vi_stream_base_g(AB, n, grp, lane);  // this lane's slot in its wavefront's U-row stream

	if (role == 0) {
		// ---- producer: assemble block rows 0..L ----
		ColConst cc;
		cc.c2a0 = p.g2d[G2_C2A0 * NS + col]; cc.c2a1 = p.g2d[G2_C2A1 * NS + col]; cc.c2b1 = p.g2d[G2_C2B1 * NS + col];
		cc.jn = p.g2d[G2_JN * NS + col]; cc.je = p.g2d[G2_JE * NS + col]; cc.drx = p.g2d[G2_DRX * NS + col];
		cc.invdt = 1.0 / dt; cc.upc = 0.5 * (1.0 / (double)L); cc.cv = p.cp - p.Rd;
		const MetCol mcol = met_col(p, col);
		// Level window of block row t: nodes / interfaces t-1, t, t+1 in (A, B, C).  The loads of level t+4 are issued at the top
		// of iteration t into one of two staging sets (Y0 / Y1, alternating: the loop body is instantiated twice) and are first
		// touched -- by a register move into N1 -- at the end of iteration t+1, two block rows later: under load the memory latency
		// of this kernel is longer than one block row (measured: with the loads of level t+2 moved into place at the top of the
		// next iteration the producer stalled 1 000 - 2 700 cycles per block row on its prefetch), and vmcnt retires in order,
		// so a wait for the newest load cannot be hidden behind older ones.  The staging sets hold the five loaded values of a
		// level only; the closed-form metric of the level is evaluated when it enters the window (stored-metric engines, whose
		// levels carry six more loaded values, keep the one-row scheme: DEEP = false).
		constexpr bool DEEP = CLOSED;
		struct LevRaw { double un, vn, pn, rn, we; };
		auto load_raw = [&](int l) {
			LevRaw r;
#ifdef TMX_ABL_P_LEVEL0      // ablation builds: every level's loads hit the same five cache-resident rows
			const int lc = (l & 1), le = (l & 1);
#else
			const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l), le = l < 0 ? 0 : (l > L ? L : l);
#endif
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
		LevRaw rN1 = load_raw(2), rY0 = load_raw(3), rY1 = rY0;
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
		auto block_row = [&](int t, LevRaw & rYl, const LevRaw & rYm) {
			if (t <= L) {
				if (DEEP) rYl = load_raw(t + 4);
				else { /* issued below, after the shift */ }
#ifdef TMX_ABL_FAST_PRODUCER      // ablation builds: the block row is evaluated once, the ring is refilled with it every time
				if (t == 0)
#endif
#ifdef TMX_PAIR_TIMING
				PMARK(tp_seg[4]);      // prefetch issue
				compute_block<true>(p, opsl, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, 0.0, 0.0, rmtab, tp_seg);
				tp_mark = __builtin_readcyclecounter();
#else
				// Interior block rows whose Exner arguments are ordinary numbers in every lane can take the branch-free instantiation
				// (one basic block of 350 vector instructions per block row).  Measured: SLOWER -- the assembly alone 118 instead of
				// 98 us per launch, the kernel +3 % -- so it stays an experiment (-DTMX_INTERIOR_ROWS).
#ifdef TMX_INTERIOR_ROWS
				if (t >= 1 && t <= L - 2 && __builtin_amdgcn_ballot_w64(!exner_arg_in_main_range(p, nB.pn)) == 0)
					compute_block<true, false, true>(p, opsl, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, 0.0, 0.0, rmtab);
				else
#endif
				compute_block<true>(p, opsl, cc, t, nA, nB, nC, eA, eB, eC, rP, rW, rR, fP, fW, fR, cy, 0.0, 0.0, rmtab);
#endif
				double * q = ring + (size_t)(t & 1) * NQ * 64 + lane;
#pragma unroll
				for (int c = 0; c < TMX_BW; c++) { q[c * 64] = rP[c]; q[(TMX_BW + c) * 64] = rW[c]; q[(2 * TMX_BW + c) * 64] = rR[c]; }
				q[(3 * TMX_BW) * 64] = fP; q[(3 * TMX_BW + 1) * 64] = fW; q[(3 * TMX_BW + 2) * 64] = fR;
			}
			PMARK(tp_work);
			__syncthreads();
			PMARK(tp_wait);
			nA = nB; nB = nC; eA = eB; eB = eC;
			if (DEEP) { enter(rN1, t + 2, nC, eC); rN1 = rYm; }
			else { nC = nN; eC = eN; nN = load_node<MM>(p, mcol, xin, t + 3, col, etal); eN = load_edge<MM>(p, mcol, xin, t + 3, col, etal); }
#ifdef TMX_PAIR_TIMING
			PMARK(tp_seg[5]);      // window shift (DEEP: touches loads issued two block rows earlier)
#endif
		};
		for (int t = 0; t <= L + 1; t += 2) {
			block_row(t, rY0, rY1);
			if (t + 1 <= L + 1) block_row(t + 1, rY1, rY0);
		}
#ifdef TMX_PAIR_TIMING
		if (stats && lane == 0) {
			unsigned long long * q = stats + 2 + 8 * ((size_t)(blockIdx.x * PAIRS + pair) * 2 + role);
			q[0] = 1 + role; q[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
			q[3] = tp_work; q[4] = tp_wait; q[5] = (unsigned long long)((long long)__builtin_readcyclecounter() - tp_start); q[6] = 0;
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
#ifndef TMX_NO_SETPRIO
	__builtin_amdgcn_s_setprio(3);
#endif
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
	for (int t = 0; t <= L + 1; t++) {
		if (t >= 1) {
			const int m = t - 1;
			const double * q = ring + (size_t)(m & 1) * NQ * 64 + lane;
			const double * qP = q, * qW = q + TMX_BW * 64, * qR = q + 2 * TMX_BW * 64, * qf = q + 3 * TMX_BW * 64;
			if (m == 0) {
				// window row r holds matrix columns 0..8; band entry d of row r sits at column r - 4 + d
#pragma unroll
				for (int c = 0; c < TMX_BW; c++) {
					R[0][c] = (c + 4 < TMX_BW) ? qP[(c + 4) * 64] : 0.0;
					R[1][c] = (c + 3 < TMX_BW) ? qW[(c + 3) * 64] : 0.0;
					R[2][c] = (c + 2 < TMX_BW) ? qR[(c + 2) * 64] : 0.0;
				}
				b[0] = qf[0]; b[1] = qf[64]; b[2] = qf[128];
			} else if (m == 1) {
#pragma unroll
				for (int c = 0; c < TMX_BW; c++) {
					R[3][c] = (c + 1 < TMX_BW) ? qP[(c + 1) * 64] : 0.0;
					R[4][c] = qW[c * 64];
				}
				b[3] = qf[0]; b[4] = qf[64];
				lu_pivot_step<0, SPARSE, true, 64>(R, b, singular, AB, RHS, 0, NUS, u, qR, qf[128], umask, &nuni);      // pivot 0 takes in row 5
			} else
#ifdef TMX_ABL_NO_ELIM            // ablation builds: the consumer only keeps the barrier rhythm
			if (m < 0)
#endif
			{
				// matrix rows 3m-5 (W of level m-2), 3m-4 (rho), 3m-3 (rho*theta of level m-1)
				lu_pivot_step<1, SPARSE, true, 64, TMX_EXPECT_W>(R, b, singular, AB, RHS, 3 * m - 5, NUS, u, qP, qf[0], umask, &nuni);
				lu_pivot_step<2, SPARSE, true, 64, TMX_EXPECT_R>(R, b, singular, AB, RHS, 3 * m - 4, NUS, u, qW, qf[64], umask, &nuni);
				lu_pivot_step<0, SPARSE, true, 64, TMX_EXPECT_T>(R, b, singular, AB, RHS, 3 * m - 3, NUS, u, qR, qf[128], umask, &nuni);
			}
		}
		PMARK(tp_work);
		__syncthreads();
		PMARK(tp_wait);
	}
	{
		double z[TMX_BW];
#pragma unroll
		for (int c = 0; c < TMX_BW; c++) z[c] = 0.0;
		// the last five rows: n is a multiple of three, so their kinds are W, rho, rho*theta, W, rho
		lu_pivot_step<1, SPARSE, true>(R, b, singular, AB, RHS, n - 5, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<2, SPARSE, true>(R, b, singular, AB, RHS, n - 4, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<0, SPARSE, true>(R, b, singular, AB, RHS, n - 3, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<1, SPARSE, true>(R, b, singular, AB, RHS, n - 2, NUS, u, z, 0.0, umask, &nuni);
		lu_pivot_step<2, SPARSE, true>(R, b, singular, AB, RHS, n - 1, NUS, u, z, 0.0, umask, &nuni);
	}
#ifdef TMX_PAIR_TIMING
	const long long tp_fwd = __builtin_readcyclecounter();
#endif
	if (singular) atomicOr(flag, TMX_FLAG_SINGULAR);
	// statistics for bench.py (stats = {pivot steps on the wave-uniform path, all pivot steps}, per wavefront)
	if (stats && lane == 0) { atomicAdd(&stats[0], (unsigned long long)nuni); atomicAdd(&stats[1], (unsigned long long)n); }
#ifndef TMX_ABL_NO_BACK      // ablation builds (tools/ablate.sh): time of the forward phase alone
	if (BACK) vi_back_substitute(L, NS, n, u, col, NUS, AB, RHS, xin, xup, udep, umask, lane);
	else for (int j = lane; j < n; j += 64) gmask[(size_t)grp * n + j] = umask[j];      // this wavefront's own LDS writes: no barrier needed
#endif
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
	const int col = ucol[u];
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
	const int col = ucol[u];
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
			for (int t = 0; t < 3; t++) dep[t] = udep[u * 3 + t];
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
	const size_t lds = (size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW * sizeof(double) + (size_t)(2 * p.L + 1) * sizeof(double) + (size_t)TMX_FTOT * (p.L + 1) * sizeof(int);
	const bool sparse = !(getenv("TMX_VI_SPARSE") && atoi(getenv("TMX_VI_SPARSE")) == 0);      // A/B switch, default on
	// The pair kernel shortens the per-column dependent chain to max(assembly, elimination) and is built for two
	// wavefronts per SIMD (256 registers): the 1024 SIMDs hold 1024 pairs at once.  Measured per launch: ne30 on one
	// GPU (760 column groups) 197-211 us against 229 us for the one-wavefront kernel; ne60 (3038 groups, three rounds)
	// 840 us against 899 us; 190 groups 95 us with one pair per workgroup (every workgroup alone on a CU), 104 us with
	// two; 380 groups 125 us with two pairs per workgroup, 131 us with one.  So it is the production kernel at every
	// size; the one-wavefront kernel stays as TMX_VI_PAIR=0 (bit-identical, tested).
	// few columns for the chip (strong-scaling share): one column per 16-lane group
	if (e->vi_group == 1 || (e->vi_group < 0 && e->nunique <= e->vi_group_max)) {
		const int nwg = (e->nunique + 3) / 4;
		if (p.closed) hipLaunchKernelGGL((k_vi_group<true>), dim3(nwg), dim3(64), 0, e->stream, p, xin, xup, dt, e->nunique, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_flag);
		else hipLaunchKernelGGL((k_vi_group<false>), dim3(nwg), dim3(64), 0, e->stream, p, xin, xup, dt, e->nunique, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_flag);
		return;
	}
	const bool use_pair = (e->vi_pair == 1) || (e->vi_pair < 0 && e->vi_carry);
	const int cpw = use_pair ? e->vi_cpw : 64;
	const int ngrp = (e->nunique + cpw - 1) / cpw;
	if (use_pair) {
		const size_t lds_common = (size_t)TMX_OP_COUNT * (p.L + 1) * TMX_OPW * sizeof(double) + (size_t)(2 * p.L + 1) * sizeof(double) + TMX_RMTAB_DOUBLES * sizeof(double);
		const size_t lds_pair = (size_t)2 * (3 * TMX_BW + 3) * 64 * sizeof(double) + (size_t)TMX_FTOT * (p.L + 1) * sizeof(int);
		const int pairs = (e->vi_pair_wg > 0) ? e->vi_pair_wg : ((ngrp <= 256) ? 1 : 2);
		// back substitution as a launch of its own (TMX_VI_SPLIT_BACK=0: inside the pair kernel); the masks go through d_rhs,
		// which only the split cross-check kernels use otherwise
		const bool split_back = e->vi_split_back != 0;
		const int stagger = e->vi_stagger;
		const int ngrp_launched = ((ngrp + pairs - 1) / pairs) * pairs;
		const bool split = split_back && (size_t)ngrp_launched * TMX_FTOT * (p.L + 1) * sizeof(int) <= (size_t)TMX_FTOT * (p.L + 1) * e->NUS * sizeof(double);
#define LAUNCH_PAIR(CL_, NP_) do { if (split) hipLaunchKernelGGL((k_vi_pair<true, CL_, NP_, false>), dim3((ngrp + NP_ - 1) / NP_), dim3(64, 2 * NP_), lds_common + NP_ * lds_pair, \
			e->stream, p, xin, xup, dt, e->nunique, e->vi_stream_cols, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_rhs, e->d_flag, cpw, e->d_pivot_stats, (int *)e->d_rhs, 0); \
		else hipLaunchKernelGGL((k_vi_pair<true, CL_, NP_, true>), dim3((ngrp + NP_ - 1) / NP_), dim3(64, 2 * NP_), lds_common + NP_ * lds_pair, \
			e->stream, p, xin, xup, dt, e->nunique, e->vi_stream_cols, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_rhs, e->d_flag, cpw, e->d_pivot_stats, (int *)nullptr, stagger); } while (0)
		if (pairs == 1) { if (p.closed) LAUNCH_PAIR(true, 1); else LAUNCH_PAIR(false, 1); }
		else { if (p.closed) LAUNCH_PAIR(true, 2); else LAUNCH_PAIR(false, 2); }
#undef LAUNCH_PAIR
		if (split) {
			const int nsub = e->vi_back_sub;
#define LAUNCH_BACK(NS_) hipLaunchKernelGGL(k_vi_back<NS_>, dim3(ngrp), dim3(64 * NS_), (size_t)TMX_FTOT * (p.L + 1) * sizeof(int), e->stream, p, xin, xup, e->nunique, e->vi_stream_cols, \
				(const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, cpw, (const int *)e->d_rhs)
			if (nsub == 2) LAUNCH_BACK(2); else if (nsub == 4) LAUNCH_BACK(4); else LAUNCH_BACK(1);
#undef LAUNCH_BACK
		}
		return;
	}
#define LAUNCH_VI(C_, S_) do { if (p.closed) \
		hipLaunchKernelGGL((k_vi_fused<C_, S_, true, false>), dim3((e->nunique + 63) / 64), dim3(64), lds, e->stream, p, xin, xup, dt, \
			e->nunique, e->vi_stream_cols, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_rhs, e->d_flag); \
	else \
		hipLaunchKernelGGL((k_vi_fused<C_, S_, false, false>), dim3((e->nunique + 63) / 64), dim3(64), lds, e->stream, p, xin, xup, dt, \
			e->nunique, e->vi_stream_cols, (const int *)e->d_ucol, (const int *)e->d_udep, e->d_ab, e->d_rhs, e->d_flag); } while (0)
	if (!e->vi_carry) LAUNCH_VI(false, true);
	else if (sparse) LAUNCH_VI(true, true);
	else LAUNCH_VI(true, false);
#undef LAUNCH_VI
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

void tmxk_vi_terms_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_uv) {
	const int nt_ = NTILES(e, p), xm = e->xcd_vertical;
	dim3 blk(64, KT_VC), grd(xcd_column_grid(xm, nt_, (p.L + 1 + KT_VC - 1) / KT_VC));
	if (e->udiff && e->fully_explicit) {
		const double z2 = e->cfg.ztop * e->cfg.ztop;
		if (with_uv) hipLaunchKernelGGL((k_vi_terms_explicit<true, true>), grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)e->d_ref,
			e->cfg.uniform_diffusion_scalar / z2, e->cfg.uniform_diffusion_vector / z2, nt_, xm);
		else hipLaunchKernelGGL((k_vi_terms_explicit<true, false>), grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)e->d_ref,
			e->cfg.uniform_diffusion_scalar / z2, e->cfg.uniform_diffusion_vector / z2, nt_, xm);
	} else
		hipLaunchKernelGGL((k_vi_terms_explicit<false, false>), grd, blk, 0, e->stream, p, xin, xup, dt, (const double *)nullptr, 0.0, 0.0, nt_, xm);
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
