// tmx_device.h -- device-side helpers shared by the kernel translation units (tmx_k_*.hip): addressing, stage combinations,
// XCD-aware tile orders, the terrain-following metric, the Exner function.  gfx950 only.
#pragma once
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
// dmask (node-unique kernels instantiated with DM only): bit m set = term m is an ELEMENT-MAJOR instance whose copies of a node may differ
// (the model state behind a per-copy operation such as the Held-Suarez forcing): the thread reads its own copy, slab stride NS, the
// thread's own column; bit 31: the same for the stage's initial instance.  The kernels' results are node-unique either way.
struct LinTerms { const double * src[12]; double coef[12]; int n; int premul; unsigned dmask; };
// Where the U,V slabs of the instances named by a LinTerms (and of the stage's initial instance) live: the instance's own slot,
// or the slot of the instance it was copied from when the stepper left the CopyData of U,V out (tmx_step.hip: the implicit
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
// the same with the non-temporal hint (TMX_H_TERMS_NT: the raw terms of a stage combination -- up to seven whole instances read once per stage)
#ifndef TMX_H_TERMS_NT
#define TMX_H_TERMS_NT 0
#endif
__device__ __forceinline__ double ldu_term(const double * uniform_base, unsigned lane_byte_off) {
	if (TMX_H_TERMS_NT) return __builtin_nontemporal_load((const double *)((const char *)uniform_base + lane_byte_off));
	return *(const double *)((const char *)uniform_base + lane_byte_off);
}
__device__ __forceinline__ void stu(double * uniform_base, unsigned lane_byte_off, double v) {
	*(double *)((char *)uniform_base + lane_byte_off) = v;
}

// The same in two phases, for kernels that name several slabs of the combination: lin_load issues the loads of one slab's raw
// terms (no control flow: PM = "the destination's own coefficient is not zero" is a compile-time parameter, so the loads of all
// slabs can be issued back to back), lin_reduce combines them in the reference's order.  With base_value_u called slab after
// slab the compiler serialised load batch -> wait -> reduce per slab behind the run-time `premul` branches: five dependent
// memory round trips at the head of every wavefront of the explicit kernels.
template <int NT, bool PM>
__device__ __forceinline__ void lin_load(const double * const (&src)[12], size_t slab_off, unsigned colb, double (&x)[NT > 0 ? NT : 1]) {
	if (NT == 0) { x[0] = ldu(src[0] + slab_off, colb); return; }
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || PM) ? ldu(src[m] + slab_off, colb) : 0.0;
}
template <int NT, bool PM>
__device__ __forceinline__ double lin_reduce(const LinTerms & t, const double (&x)[NT > 0 ? NT : 1]) {
	if (NT == 0) return x[0];
	double v = PM ? x[0] * t.coef[0] : 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) v += t.coef[m] * x[m];
	return v;
}

// Prefix of a LATER stage's combination, formed from the terms this stage has loaded anyway.  The reference accumulates a
// combination in a fixed order -- the destination's own term first, then the other instances in ascending order
// (GridPatch::LinearCombineData) -- so the value after the first k terms is a well-defined intermediate; when the instances of
// those k terms are exactly the instances this stage combines, it can be formed here (same operands, same order, the later
// stage's coefficients) and stored, and the later stage starts from it (coefficient 1: exact) and reads k - 1 instances less.
// ARS343: the third stage combines instances 0..4, the fourth 1, 0, 2, 3, 4, 5, 6: seven source instances become three.
// coef[m] belongs to this stage's term m >= 1; first = this stage's term index of the later stage's destination (-1: it has none).
template <int NT>
__device__ __forceinline__ double emit_value(const EmitTerms & em, const double (&x)[NT > 0 ? NT : 1]) {
	double p = 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) if (em.first == m) p = x[m] * em.coef[m];
#pragma unroll
	for (int m = 1; m < NT; m++) { const double q = p + em.coef[m] * x[m]; p = (em.first == m) ? p : q; }
	return p;
}
// base_value_u / base_value_uv that also return that prefix (EM: compile-time switch)
template <int NT, bool EM>
__device__ __forceinline__ double base_value_u2(const LinTerms & t, const double * const (&src)[12], const EmitTerms & em, size_t slab_off, unsigned colb, double & pfx) {
	if (NT == 0) return ldu(src[0] + slab_off, colb);
	double x[NT > 0 ? NT : 1];
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? ldu_term(src[m] + slab_off, colb) : 0.0;
	double v = t.premul ? x[0] * t.coef[0] : 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) v += t.coef[m] * x[m];
	if (EM) pfx = emit_value<NT>(em, x);
	return v;
}

// base_value with the slab offset (uniform, in doubles) and the lane's byte offset kept apart
template <int NT>
__device__ __forceinline__ double base_value_u(const LinTerms & t, size_t slab_off, unsigned colb) {
	if (NT == 0) return ldu(t.src[0] + slab_off, colb);
	double x[NT > 0 ? NT : 1];
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? ldu_term(t.src[m] + slab_off, colb) : 0.0;
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
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? ldu_term(uv.src[m] + slab_off, colb) : 0.0;
	double v = t.premul ? x[0] * t.coef[0] : 0.0;
#pragma unroll
	for (int m = 1; m < NT; m++) v += t.coef[m] * x[m];
	return v;
}

// base_value_u / base_value_uv where some terms are element-major instances (LinTerms::dmask): slab index given, stride and lane offset per term
template <int NT>
__device__ __forceinline__ double base_value_dm(const LinTerms & t, const double * const (&src)[12], int slab, size_t SS, size_t NS, unsigned cb, unsigned gb) {
	auto ld = [&](int m) -> double { const bool d = (t.dmask >> m) & 1u; return ldu(src[m] + (size_t)slab * (d ? NS : SS), d ? gb : cb); };
	if (NT == 0) return ld(0);
	double x[NT > 0 ? NT : 1];
#pragma unroll
	for (int m = 0; m < NT; m++) x[m] = (m > 0 || t.premul) ? ld(m) : 0.0;
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


// DSS combination shared by k_dss (tmx_k_exchange.hip) and the kernels that average while loading (tmx_k_horizontal.hip)
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


// Ordering point for LDS rows that only ONE wavefront touches (the contraction rows s[ty][..] of the horizontal kernels, each
// wavefront's copy of the 4 x 4 operator tables): the LDS executes a wavefront's operations in order, so lanes see one another's
// writes without a workgroup barrier; the fences keep the compiler from moving reads above the writes they depend on.
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup -> (entry of the launch's tile list, level block).  Element-major kernels: the plain 2-D launch.  Node-unique kernels
// (UQ): a 1-D launch unfolded so that every XCD sweeps a contiguous range of tiles, level block by level block -- neighbouring
// tiles read the same rows of the unique slabs (the nodes on their common edge), which then come from that XCD's L2.
// Lane of a wavefront -> position e * 16 + i * 4 + j inside its tile of four elements.  Element-major kernels: the identity (a
// tile is 512 contiguous bytes of every slab).  Node-unique kernels: lane = j + 4 e + 16 i, so that 16 consecutive lanes hold one
// node row of the tile's four elements -- for elements that follow one another along beta that is ONE contiguous run of 13
// unique nodes in the slab (one or two 128-byte lines per 16 lanes); with the identity map a 16-lane group is one element = four
// runs of 32 bytes in four different rows, four times the L1 / L2 requests per load instruction (measured: 2.1 x the L2 requests
// of the element-major kernel and +12 % time for 23 % fewer bytes).
#ifndef TMX_UQ_ROWLANES
#define TMX_UQ_ROWLANES 0      // measured: with the LDS rows as they were the permuted lanes cost the contractions two-way bank conflicts (hyperviscosity pass 149 ->
                               // 196 us); with padded rows (lds_pos) the explicit stage is level (0.577 against 0.576 ms per step) and the hyperviscosity pass 4 % slower:
                               // halving the L2 requests buys nothing, so the identity map stays
#endif
// LDS position of tile position l in a contraction row.  With the row-major lane map a half-wavefront holds node rows i = 0, 1 of all
// four elements, whose 128-byte blocks would fall on the same banks two by two; eight doubles of padding in front of elements 2, 3
// (rows of 80 instead of 64 doubles) separate them.
#define TMX_UQ_ROWLEN(UQ_) (((UQ_) && TMX_UQ_ROWLANES) ? 80 : 64)
template <bool UQ> __device__ __forceinline__ int lds_pos(int l) { return (UQ && TMX_UQ_ROWLANES) ? l + ((l >> 5) << 3) : l; }
template <bool UQ> __device__ __forceinline__ int tile_lane() {
	const int h = threadIdx.x;
	return (UQ && TMX_UQ_ROWLANES) ? ((h & 3) | (((h >> 2) & 3) << 4) | ((h >> 4) << 2)) : h;
}
// unique-slab index of the node at position `lane` of tile `tile` (-1: padding), from the tile's eight wave-uniform table entries:
// no per-lane table load in front of the state loads (a dependent memory round trip at the head of every wavefront: the
// explicit kernels ran 12 % and the hyperviscosity pass 20 % longer with t_ucol[col] there)
__device__ __forceinline__ int tile_ucol(const KParams & p, int tile, int lane) {
	const int * ti = p.t_tinfo + (size_t)WAVE_UNIFORM(tile) * 8;
	const int e = lane >> 4;
	const int b0 = ti[0], b1 = ti[1], b2 = ti[2], b3 = ti[3], n0 = ti[4], n1 = ti[5], n2 = ti[6], n3 = ti[7];
	const int base = (e == 0) ? b0 : ((e == 1) ? b1 : ((e == 2) ? b2 : b3));
	const int nj = (e == 0) ? n0 : ((e == 1) ? n1 : ((e == 2) ? n2 : n3));
#ifdef TMX_ABL_UQ_DLOADS      // ablation builds (wrong results): contiguous addresses instead of the gather, everything else as it is
	return (base < 0) ? -1 : min(tile * 64 + lane, p.NSS - 1);
#endif
	return (base < 0) ? -1 : base + ((lane >> 2) & 3) * nj + (lane & 3);
}

template <bool UQ> __device__ __forceinline__ bool wg_tile(const KParams & p, int nyb, int & tile_ix, int & yb) {
	if (!UQ) { tile_ix = blockIdx.x; yb = blockIdx.y; return true; }
	if (p.u_xcd) return xcd_tile(p.u_ntiles, nyb, tile_ix, yb);
	tile_ix = blockIdx.x % p.u_ntiles; yb = blockIdx.x / p.u_ntiles;
	return yb < nyb;
}
static inline dim3 wg_grid(bool uq, const KParams & p, int ntiles, int nyb) {
	if (!uq) return dim3(ntiles, nyb);
	return dim3(p.u_xcd ? 8 * ((ntiles + 7) / 8) * nyb : ntiles * nyb);
}

// ---------------------------------------------------------------------------------------------
// Node-unique state layout (UniqueLayout, tmx_internal.h): what a thread = (element, node) does with its NV results.
//   sdst >= 0   the node belongs to this lane: store there -- after averaging with the copies other lanes of the wavefront hold,
//               in the reference's order (GridCSGLL::ApplyDSS, GridCSGLL.cpp:560-781: 0.5 (x + x'), four copies
//               0.5 (0.5 (x + x_alpha) + 0.5 (x_beta + x_diag)); inside a patch every copy would receive these very bits)
//   sdst == -1  nothing: another lane of the wavefront stores this node (or a padding lane)
//   sdst <= -2  the node's copies are not all in this wavefront (tile edge, patch edge): the per-element value goes to partial
//               slot -2 - sdst, k_dss_u finishes the node
// srow: NV rows of 64 doubles of LDS private to the wavefront.  LDS operations of one wavefront execute in order, so the
// exchange needs no workgroup barrier; the wavefront-scope fence keeps the compiler from moving the reads above the writes.
template <int NV, int RL>
__device__ __forceinline__ void seam_store(const KParams & p, double (*srow)[RL], int lane_, int sdst, int sred, const double (&v)[NV],
	double * __restrict__ xo, const int (&slab)[NV], int nv)
{
	const int lane = lds_pos<RL != 64>(lane_);
#ifdef TMX_ABL_UQ_DSTORES     // ablation builds (wrong results): every lane stores its own value to consecutive addresses, no exchange
	{
		const int c = min((int)(blockIdx.x & 1023) * 64 + lane, p.NSS - 1);
#pragma unroll
		for (int t = 0; t < NV; t++) if (t < nv) xo[(size_t)slab[t] * p.NSS + c] = v[t];
		return;
	}
#endif
#pragma unroll
	for (int t = 0; t < NV; t++) if (t < nv) srow[t][lane] = v[t];
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	if (sdst == -1) return;
	const int n = sred & 15;
	double o[NV];
#pragma unroll
	for (int t = 0; t < NV; t++) o[t] = v[t];
	if (n == 2) {
		const int l1 = lds_pos<RL != 64>((sred >> 4) & 63);
#pragma unroll
		for (int t = 0; t < NV; t++) if (t < nv) o[t] = 0.5 * (v[t] + srow[t][l1]);
	} else if (n == 4) {
		const int l1 = lds_pos<RL != 64>((sred >> 4) & 63), l2 = lds_pos<RL != 64>((sred >> 10) & 63), l3 = lds_pos<RL != 64>((sred >> 16) & 63);
#pragma unroll
		for (int t = 0; t < NV; t++) if (t < nv) {
			const double p0 = v[t] + srow[t][l1], p1 = srow[t][l2] + srow[t][l3];
			o[t] = 0.5 * (0.5 * p0 + 0.5 * p1);
		}
	}
	if (sdst >= 0) {
#pragma unroll
		for (int t = 0; t < NV; t++) if (t < nv) xo[(size_t)slab[t] * p.NSS + sdst] = o[t];
	} else {
		const int q = -2 - sdst;
#pragma unroll
		for (int t = 0; t < NV; t++) if (t < nv) p.part[(size_t)slab[t] * p.NP + q] = o[t];
	}
}

// Seam exchange of a BLOCK: the workgroup's (up to) four wavefronts hold a block of the thread order; a node whose copies all sit in the
// block is averaged here (reference order, as seam_store), whichever wavefronts hold them.  xrow: [6][256] doubles of LDS shared by the
// workgroup, one of two buffers alternating by level, so that ONE barrier per level orders everything: a wavefront that writes buffer b
// for level k + 2 has passed the barrier of level k + 1, which every wavefront reaches only after its reads of level k (buffer b).
// The barrier waits for LDS traffic only (s_waitcnt lgkmcnt(0)): __syncthreads() would also drain vmcnt, i.e. wait for the prefetched
// next level and the combination's terms in flight -- the very loads the walk issues early.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NV>
__device__ __forceinline__ void block_seam_store(const KParams & p, double (*xrow)[256], int pos, int sdst, int sred, const double (&v)[NV],
	double * __restrict__ xo, const int (&slab)[NV], unsigned vmask)
{
#pragma unroll
	for (int t = 0; t < NV; t++) if ((vmask >> t) & 1u) xrow[t][pos] = v[t];
	lds_barrier();
	if (sdst == -1) return;
	const int n = sred & 15;
	double o[NV];
#pragma unroll
	for (int t = 0; t < NV; t++) o[t] = v[t];
	if (n == 2) {
		const int l1 = (sred >> 4) & 255;
#pragma unroll
		for (int t = 0; t < NV; t++) if ((vmask >> t) & 1u) o[t] = 0.5 * (v[t] + xrow[t][l1]);
	} else if (n == 4) {
		const int l1 = (sred >> 4) & 255, l2 = (sred >> 12) & 255, l3 = (sred >> 20) & 255;
#pragma unroll
		for (int t = 0; t < NV; t++) if ((vmask >> t) & 1u) {
			const double p0 = v[t] + xrow[t][l1], p1 = xrow[t][l2] + xrow[t][l3];
			o[t] = 0.5 * (0.5 * p0 + 0.5 * p1);
		}
	}
	if (sdst >= 0) {
#pragma unroll
		for (int t = 0; t < NV; t++) if ((vmask >> t) & 1u) xo[(size_t)slab[t] * p.NSS + sdst] = o[t];
	} else {
		const int q = -2 - sdst;
#pragma unroll
		for (int t = 0; t < NV; t++) if ((vmask >> t) & 1u) p.part[(size_t)slab[t] * p.NP + q] = o[t];
	}
}
