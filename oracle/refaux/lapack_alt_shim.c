/* oracle/lapack_alt_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Second-LAPACK variant of the compiled reference (oracle/_ref/tmx_ref_dump_lapack2): the three banded routines on
 * the hot path -- dgbsv_ (src/base/LinearAlgebra.cpp:196-198 <- VerticalDynamicsFEM.cpp:1457), dgbtrf_
 * (LinearAlgebra.cpp:302 <- VerticalDynamicsFEM.cpp:4033) and dgbtrs_ (LinearAlgebra.cpp:360 <- :4260) -- are
 * defined HERE, in the executable, so they win symbol resolution over libmkl_rt, and forward to the netlib
 * reference LAPACK that ships inside scipy's OpenBLAS (symbol prefix scipy_, LP64).  Everything else of the
 * reference binary (objects, compiler, libm, MKL for the set-up-time dense routines) is unchanged, so the
 * difference between the two binaries is exactly "which LAPACK did `-llapack` resolve to" -- the reference does
 * not pin one (mk/system/*.make).  Used by tests/analysis/lapack_spread.py to measure the reference-vs-reference
 * spread that bounds any meaningful tolerance on W after 100 steps.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#ifndef TMX_ALT_LAPACK_SO
#define TMX_ALT_LAPACK_SO "/usr/local/lib/python3.10/dist-packages/scipy.libs/libscipy_openblas-68440149.so"
#endif

typedef void (*gbsv_fn)(int *, int *, int *, int *, double *, int *, int *, double *, int *, int *);
typedef void (*gbtrf_fn)(int *, int *, int *, int *, double *, int *, int *, int *);
typedef void (*gbtrs_fn)(char *, int *, int *, int *, int *, double *, int *, int *, double *, int *, int *, size_t);

static void * g_lib = NULL;
static gbsv_fn g_gbsv = NULL;
static gbtrf_fn g_gbtrf = NULL;
static gbtrs_fn g_gbtrs = NULL;

static void shim_init(void) {
	if (g_lib) return;
	const char * path = getenv("TMX_ALT_LAPACK_SO");
	if (!path || !*path) path = TMX_ALT_LAPACK_SO;
	g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
	if (!g_lib) { fprintf(stderr, "lapack_alt_shim: cannot open %s: %s\n", path, dlerror()); abort(); }
	g_gbsv = (gbsv_fn)dlsym(g_lib, "scipy_dgbsv_");
	g_gbtrf = (gbtrf_fn)dlsym(g_lib, "scipy_dgbtrf_");
	g_gbtrs = (gbtrs_fn)dlsym(g_lib, "scipy_dgbtrs_");
	if (!g_gbsv || !g_gbtrf || !g_gbtrs) { fprintf(stderr, "lapack_alt_shim: symbols missing in %s\n", path); abort(); }
	/* single thread: the reference's per-column solves are tiny */
	void (*setn)(int) = (void (*)(int))dlsym(g_lib, "scipy_openblas_set_num_threads");
	if (setn) setn(1);
	if (getenv("TMX_ALT_LAPACK_VERBOSE")) fprintf(stderr, "lapack_alt_shim: band LU from %s\n", path);
}

int dgbsv_(int * n, int * kl, int * ku, int * nrhs, double * ab, int * ldab, int * ipiv, double * b, int * ldb, int * info) {
	shim_init();
	g_gbsv(n, kl, ku, nrhs, ab, ldab, ipiv, b, ldb, info);
	return 0;
}

int dgbtrf_(int * m, int * n, int * kl, int * ku, double * ab, int * ldab, int * ipiv, int * info) {
	shim_init();
	g_gbtrf(m, n, kl, ku, ab, ldab, ipiv, info);
	return 0;
}

int dgbtrs_(char * trans, int * n, int * kl, int * ku, int * nrhs, double * ab, int * ldab, int * ipiv, double * b, int * ldb, int * info) {
	shim_init();
	g_gbtrs(trans, n, kl, ku, nrhs, ab, ldab, ipiv, b, ldb, info, (size_t)1);
	return 0;
}
