// tmx_k_physics.hip -- hand-written CDNA4 (gfx950) kernels of the spectral-element hot path (one translation unit per operator family;
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

// FROM_U: the instance comes in node-unique form (xs, slab stride NSS, node of a stored column = ucd[col]) and leaves element-major (x): every
// stored copy is forced with its own inputs, rho and W -- which the forcing does not touch -- are copied across, so the conversion the
// element-major route would need first is this kernel's own load
template <bool FROM_U>
__global__ __launch_bounds__(256) void k_held_suarez(KParams p, double * __restrict__ x, const double * __restrict__ surf, double dt, double gamma, double kappa, double pscal,
	const double * __restrict__ xs, size_t NSS, const int * __restrict__ ucd)
{
	const int L = p.L;
	const size_t NS = (size_t)p.NS;
	const int col = blockIdx.x * 64 + threadIdx.x;
	const int k = blockIdx.y * 4 + WAVE_UNIFORM(threadIdx.y);
	if (k > (FROM_U ? L : L - 1) || col >= p.ncol) return;
	const unsigned cb = (unsigned)col * 8u;
	const unsigned sb = FROM_U ? (unsigned)ucd[col] * 8u : cb;      // where this copy's values come from
	const double * __restrict__ xr = FROM_U ? xs : x;
	const size_t SR = FROM_U ? NSS : NS;
	if (FROM_U) {
		stu(x + TMX_SLAB_W(L, k) * NS, cb, ldu(xr + TMX_SLAB_W(L, k) * SR, sb));
		if (k == L) return;
	}
	const double bsig = 0.7, kf = 1.0 / 86400.0, ka = (1.0 / 40.0) / 86400.0, ks = (1.0 / 4.0) / 86400.0;
	const double dty = 60.0, dthz = 10.0, tmin = 200.0, tmax = 315.0;
	const double ps = surf ? pscal * tmx_ref_exp(tmx_ref_log(ldu(surf, cb) * ldu(surf + NS, cb)) * gamma) : ldu(p.g2d + G2_PS * NS, cb);
	const double rho = ldu(xr + TMX_SLAB_R(L, k) * SR, sb), rt = ldu(xr + TMX_SLAB_T(L, k) * SR, sb);
	if (FROM_U) stu(x + TMX_SLAB_R(L, k) * NS, cb, rho);
	{
		const double pr = pscal * tmx_ref_exp(tmx_ref_log(rho * rt) * gamma);
		const double sigma = pr / ps;
		double bs = (sigma - bsig) / (1.0 - bsig);
		if (bs < 0.0) bs = 0.0;
		const double u = ldu(xr + TMX_SLAB_U(L, k) * SR, sb), v = ldu(xr + TMX_SLAB_V(L, k) * SR, sb);
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

// xs != null: the instance is read from its node-unique form (stride NSS, ucd = node of every stored column) and written element-major to x
void tmxk_held_suarez(tmx_engine * e, const KParams & p, double * x, const double * surf, double dt, const double * xs, size_t NSS, const int * ucd) {
	// PhysicalConstants::GetGamma / GetKappa / pressure scaling (src/atm/PhysicalConstants.h:355-376), host libm as in the reference
	const double gamma = p.cp / (p.cp - p.Rd), kappa = p.Rd / p.cp;
	const double pscal = p.p0 * pow(p.Rd / p.p0, gamma);
	dim3 blk(64, 4);
	if (xs) hipLaunchKernelGGL(k_held_suarez<true>, dim3(p.NS / 64, (p.L + 1 + 3) / 4), blk, 0, e->stream, p, x, surf, dt, gamma, kappa, pscal, xs, NSS, ucd);
	else hipLaunchKernelGGL(k_held_suarez<false>, dim3(p.NS / 64, (p.L + 3) / 4), blk, 0, e->stream, p, x, surf, dt, gamma, kappa, pscal, (const double *)nullptr, (size_t)0, (const int *)nullptr);
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
	if (kt <= 8 && !e->opt_kessler_column) {
		if (lds > 48 * 1024) hipFuncSetAttribute((const void *)k_kessler_tile<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(k_kessler_tile<512>, dim3((p.ncol + 63) / 64), dim3(64, kt), lds, e->stream, p, x, (const double *)e->d_zlev, e->d_prect, dt, gamma, pscal);
		return;
	}
	hipLaunchKernelGGL(k_kessler, dim3((p.ncol + 255) / 256), dim3(256), 0, e->stream, p, x, (const double *)e->d_zlev, e->d_kes, e->d_prect, dt, gamma, pscal);
}

// gather the copies other ranks need into the send buffer, one contiguous message per peer:
// sendbuf = [peer][slab][count_peer]; entry i of the plan lives at base[i] + slab * stride[i]
