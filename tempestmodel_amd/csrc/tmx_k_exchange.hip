// tmx_k_exchange.hip -- hand-written CDNA4 (gfx950) kernels of the spectral-element hot path (one translation unit per operator family;
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

// node-unique layout: what other ranks need are per-element values of patch-edge nodes, i.e. partial slots (tmx_unique.hip)
void tmxuk_pack(tmx_engine * e) {
	if (e->nsend == 0) return;
	dim3 blk(256), grd((e->nsend + 255) / 256, e->nslab);
	hipLaunchKernelGGL(k_pack, grd, blk, 0, e->stream, e->nslab, (size_t)e->u.NP, e->nsend,
		(const double *)e->u.d_part, (const int *)e->u.d_send_slots, (const int *)e->d_send_base, (const int *)e->d_send_stride, e->d_sendbuf);
}
void tmxuk_pack_p2p(tmx_engine * e, int buf) {
	if (e->nsend == 0) return;
	dim3 blk(256), grd((e->nsend + 255) / 256, e->nslab);
	hipLaunchKernelGGL(k_pack_p2p, grd, blk, 0, e->stream, e->nslab, (size_t)e->u.NP, e->nsend,
		(const double *)e->u.d_part, (const int *)e->u.d_send_slots, (const int *)e->d_send_within, (const int *)e->d_send_stride, (const int *)e->d_send_peer,
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

