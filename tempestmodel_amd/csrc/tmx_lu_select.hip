// tmx_lu_select.hip -- the launchers of the vertical operators, dispatched to one of the two compiled flavours of the band LU
// (tmx_k_column.hip, tmx_k_vertical.hip: namespaces lu_fma / lu_nofma) by the engine's option "lu_fma".
//
// Why two: LAPACK builds dgbtf2 / dgbtrs / dtbsv from dger and friends, and whether the update a - l * u is ONE rounding (a fused
// multiply-add: OpenBLAS, MKL on its AVX2 / AVX-512 code paths) or TWO (a netlib build without FMA contraction, MKL where it dispatches a
// non-FMA path) is a property of the BLAS the reference was linked to -- both are "the reference", and they differ in the last bit of the
// implicit step (tests/golden/lapack_spread.json).  Rounds 2-5 shipped two libraries and left the choice to the linker; a caller who linked
// the wrong one got a silent last-bit difference.  Now it is a run-time option of one library, and tmx_lu_flavour_from_dgbsv lets the adapter
// ask the LAPACK it is itself linked to.
#include "tmx_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>

#define TMX_LU_LAUNCHERS(NS_) namespace NS_ { \
	void tmxk_v_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_udiff_uv); \
	void tmxk_vi_assemble(tmx_engine * e, const KParams & p, const double * xin, double dt); \
	void tmxk_vi_solve(tmx_engine * e, const KParams & p, const double * xin, double * xup); \
	void tmxk_vi_fused(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt); \
	void tmxk_vi_terms_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_uv); \
	int tmxk_vi_tracers(tmx_engine * e, const KParams & p, const double * xin, const double * w0, const double * xbase, double * xup, double dt); \
	void tmxk_v_filter_tracers(tmx_engine * e, const KParams & p, double * x); \
	int tmxk_vi_tracers_all(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt); \
	int tmxk_vi_tracers_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt); }
TMX_LU_LAUNCHERS(lu_fma)
TMX_LU_LAUNCHERS(lu_nofma)
#define PICK(CALL_) (e->lu_fma ? lu_fma::CALL_ : lu_nofma::CALL_)

void tmxk_v_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_udiff_uv) { PICK(tmxk_v_explicit(e, p, xin, xup, dt, with_udiff_uv)); }
#if TMX_EXP      // the two-kernel cross-check path of the column solve: experiments flavour of the library only
void tmxk_vi_assemble(tmx_engine * e, const KParams & p, const double * xin, double dt) { PICK(tmxk_vi_assemble(e, p, xin, dt)); }
void tmxk_vi_solve(tmx_engine * e, const KParams & p, const double * xin, double * xup) { PICK(tmxk_vi_solve(e, p, xin, xup)); }
#endif
void tmxk_vi_fused(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) { PICK(tmxk_vi_fused(e, p, xin, xup, dt)); }
void tmxk_vi_terms_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_uv) { PICK(tmxk_vi_terms_explicit(e, p, xin, xup, dt, with_uv)); }
int tmxk_vi_tracers(tmx_engine * e, const KParams & p, const double * xin, const double * w0, const double * xbase, double * xup, double dt) { return PICK(tmxk_vi_tracers(e, p, xin, w0, xbase, xup, dt)); }
void tmxk_v_filter_tracers(tmx_engine * e, const KParams & p, double * x) { PICK(tmxk_v_filter_tracers(e, p, x)); }
int tmxk_vi_tracers_all(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) { return PICK(tmxk_vi_tracers_all(e, p, xin, xup, dt)); }
int tmxk_vi_tracers_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt) { return PICK(tmxk_vi_tracers_explicit(e, p, xin, xup, dt)); }

// ---------------------------------------------------------------------------------------------
// Which flavour does the CALLER'S LAPACK compute?  (LAPACK::DGBSV call site: src/base/LinearAlgebra.cpp:156-202.)
// A small band system of the column solve's shape (kl = ku = 4) is solved three ways: by the dgbsv the caller hands in -- the routine its
// reference is linked to -- and twice on the host by a plain restatement of LAPACK's algorithm for this band width (dgbtrf takes its
// unblocked branch dgbtf2 whenever the block size 32 exceeds kl; dgbtrs = the row interchanges with dger updates, then dtbsv), once with the
// update a - l u fused and once with multiply and subtract rounded separately.  Returns 1 / 0 for the flavour that reproduces the caller's
// bits (and sets the engine's "lu_fma" option accordingly when e is not NULL), -1 when neither does (option untouched).
namespace {
struct BandSys { int n, kl, ku, ld; std::vector<double> ab, b; };
// full-matrix entry (r, c) of LAPACK band storage with kl extra rows for the fill-in: ab[kl + ku + r - c][c]
inline double & at(BandSys & S, int r, int c) { return S.ab[(size_t)c * S.ld + (S.kl + S.ku + r - c)]; }
inline double upd(bool fused, double a, double l, double u) { return fused ? __builtin_fma(-l, u, a) : a - l * u; }
int host_band_solve(BandSys & S, bool fused) {
	const int n = S.n, kl = S.kl, kv = S.kl + S.ku;
	std::vector<int> piv(n);
	for (int c = 0; c < n; c++) for (int r = std::max(0, c - kv); r < c - S.ku; r++) at(S, r, c) = 0.0;      // fill-in diagonals start from zero
	int last = 0;      // last column the current pivot row reaches
	for (int c = 0; c < n; c++) {
		const int below = std::min(kl, n - 1 - c);
		int pr = c;      // first entry of largest magnitude (idamax)
		for (int r = c + 1; r <= c + below; r++) if (std::fabs(at(S, r, c)) > std::fabs(at(S, pr, c))) pr = r;
		piv[c] = pr;
		if (at(S, pr, c) == 0.0) return c + 1;
		last = std::max(last, std::min(c + S.ku + (pr - c), n - 1));
		if (pr != c) for (int q = c; q <= last; q++) std::swap(at(S, pr, q), at(S, c, q));
		if (!below) continue;
		const double rcp = 1.0 / at(S, c, c);      // dscal by the reciprocal
		for (int r = c + 1; r <= c + below; r++) at(S, r, c) *= rcp;
		for (int q = c + 1; q <= last; q++) {      // dger, column by column; a zero entry of the pivot row skips its column
			const double u = at(S, c, q);
			if (u != 0.0) for (int r = c + 1; r <= c + below; r++) at(S, r, q) = upd(fused, at(S, r, q), at(S, r, c), u);
		}
	}
	for (int c = 0; c + 1 < n; c++) {      // L y = P b
		if (piv[c] != c) std::swap(S.b[piv[c]], S.b[c]);
		for (int r = c + 1; r <= c + std::min(kl, n - 1 - c); r++) S.b[r] = upd(fused, S.b[r], S.b[c], at(S, r, c));
	}
	for (int c = n - 1; c >= 0; c--) {      // U x = y (dtbsv, column sweep)
		if (S.b[c] == 0.0) continue;
		S.b[c] /= at(S, c, c);
		for (int r = c - 1; r >= std::max(0, c - kv); r--) S.b[r] = upd(fused, S.b[r], S.b[c], at(S, r, c));
	}
	return 0;
}
}

extern "C" int tmx_lu_flavour_from_dgbsv(tmx_engine * e, tmx_dgbsv_fn dgbsv) {
	REQUIRE(dgbsv, TMX_ERR_INVALID, "tmx_lu_flavour_from_dgbsv: null routine");
	unsigned long long seed = 0x2545f4914f6cdd1dull;
	auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return (double)(seed >> 11) / 9007199254740992.0 - 0.5; };
	for (int attempt = 0; attempt < 8; attempt++) {
		BandSys S; S.n = 33; S.kl = 4; S.ku = 4; S.ld = 2 * S.kl + S.ku + 1;
		S.ab.assign((size_t)S.ld * S.n, 0.0); S.b.resize(S.n);
		for (int c = 0; c < S.n; c++) {
			for (int r = std::max(0, c - S.ku); r <= std::min(S.n - 1, c + S.kl); r++) at(S, r, c) = rnd() * ((r == c) ? 3.0 : 1.0);      // weakly dominant: some interchanges
			S.b[c] = rnd();
		}
		BandSys F = S, N = S, X = S;
		if (host_band_solve(F, true) || host_band_solve(N, false)) continue;
		if (memcmp(F.b.data(), N.b.data(), S.n * sizeof(double)) == 0) continue;      // this system cannot tell them apart
		int n = S.n, kl = S.kl, ku = S.ku, one = 1, ld = S.ld, info = 0;
		std::vector<int> ipiv(S.n);
		dgbsv(&n, &kl, &ku, &one, X.ab.data(), &ld, ipiv.data(), X.b.data(), &n, &info);
		REQUIRE(info == 0, TMX_ERR_INVALID, "tmx_lu_flavour_from_dgbsv: the caller's dgbsv returned info = %d", info);
		const bool isf = memcmp(X.b.data(), F.b.data(), S.n * sizeof(double)) == 0, isn = memcmp(X.b.data(), N.b.data(), S.n * sizeof(double)) == 0;
		if (isf || isn) { if (e) e->lu_fma = isf ? 1 : 0; return isf ? 1 : 0; }
		tmx_set_error("tmx_lu_flavour_from_dgbsv: the caller's dgbsv reproduces neither the fused nor the multiply + subtract band LU (option lu_fma left at %d)", e ? e->lu_fma : 1);
		return -1;
	}
	tmx_set_error("tmx_lu_flavour_from_dgbsv: no probe system separated the two flavours");
	return -1;
}
