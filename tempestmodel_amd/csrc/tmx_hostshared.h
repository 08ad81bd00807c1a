// tmx_hostshared.h -- what the two host-side translation units share: tmx_host.hip (C ABI set-up: life cycle, patches, finalize and
// its plans, state transfer, restart image, communicator set-up, introspection, options) and tmx_step.hip (the operations on the
// resident state: stage algebra, dynamics entry points, exchange, stepper programs and their two interpreters, column physics).
#pragma once
#include "tmx_internal.h"
#include <pthread.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <numeric>
#include <functional>
#include <thread>

static inline bool plan_only(const tmx_engine * e) { return e->cfg.device == -2; }

// RCCL, resolved at run time so the library loads without it (single-GPU use, CPU symbol checks)
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_ncclGetUniqueId)(nccl_uid *);
typedef int (*fn_ncclCommInitRank)(void **, int, nccl_uid, int);
typedef int (*fn_ncclCommDestroy)(void *);
typedef int (*fn_ncclGroupStart)(void);
typedef int (*fn_ncclGroupEnd)(void);
typedef int (*fn_ncclSend)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_ncclRecv)(void *, size_t, int, int, void *, hipStream_t);
typedef const char * (*fn_ncclGetErrorString)(int);
typedef int (*fn_ncclCommCount)(void *, int *);

struct NcclApi {
	void * lib = nullptr;
	fn_ncclGetUniqueId GetUniqueId = nullptr;
	fn_ncclCommInitRank CommInitRank = nullptr;
	fn_ncclCommDestroy CommDestroy = nullptr;
	fn_ncclGroupStart GroupStart = nullptr;
	fn_ncclGroupEnd GroupEnd = nullptr;
	fn_ncclSend Send = nullptr;
	fn_ncclRecv Recv = nullptr;
	fn_ncclGetErrorString GetErrorString = nullptr;
	fn_ncclCommCount CommCount = nullptr;
};
extern NcclApi g_nccl;

int load_rccl();
#define NCCLCHK(call) do { int _r = (call); if (_r != 0) { \
	tmx_set_error("%s failed: %s", #call, g_nccl.GetErrorString ? g_nccl.GetErrorString(_r) : "?"); return TMX_ERR_COMM; } } while (0)

// ---------------------------------------------------------------------------------------------
// profiling helper: bracket a launch sequence with events on the engine's stream

struct ProfScope {
	tmx_engine * e; int id; hipEvent_t a = nullptr, b = nullptr;
	ProfScope(tmx_engine * e_, int id_) : e(e_), id(id_) {
		if (e->prof) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, e->stream); }
	}
	~ProfScope() {
		if (e->prof) { hipEventRecord(b, e->stream); e->prof_pending.push_back({ id, { a, b } }); }
	}
};

void prof_collect(tmx_engine * e);

// plan data kept on the host for introspection (tests of the N>1 logic)
struct PlanHost {
	std::vector<int> grp_cols, grp_n, grp_x, grp_type;
	std::vector<double> xmat;
	std::vector<int> send_nodes;   // triples (patch,i,j) in send order
	std::vector<int> recv_nodes;   // triples (patch,i,j) in ghost order
	std::vector<int> send_cols;
	std::vector<int> ucol, udep;
};
PlanHost & plan_of(tmx_engine * e);

// column index of local node (i,j) (reference patch-local indices, 1-based interior) of a local patch
static inline int col_of(const PatchInfo & P, int i, int j) {
	const int a = (i - 1) / TMX_NP, ii = (i - 1) % TMX_NP, b = (j - 1) / TMX_NP, jj = (j - 1) % TMX_NP;
	return (P.elem_base + a * P.neb + b) * TMX_NQ + ii * TMX_NP + jj;
}


static inline size_t p2p_header_bytes(int n_ranks) { return (((size_t)2 * n_ranks * sizeof(unsigned long long)) + 255) / 256 * 256; }

// defined in tmx_step.hip
KParams make_params(const tmx_engine * e);
int launch_check(const char * what);
int settle_instance(tmx_engine * e, int ix, bool read_only = false);
int check_ready(tmx_engine * e);
int check_inst(tmx_engine * e, int ix, bool read_only = false);
double * inst(tmx_engine * e, int ix);
void interp_orphan(tmx_engine * e);      // output-interpolation plans of an engine that is being destroyed
// defined in tmx_host.hip
int check_reference_state(tmx_engine * e);
int ensure_layout(tmx_engine * e);
struct LoopbackGroup;
