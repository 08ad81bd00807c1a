// tmx_host.hip -- host side of the engine, part 1: C ABI set-up (life cycle, operators, patches), finalize (HBM layout, DSS groups, exchange
// plan), state transfer, restart image, communicator and peer-to-peer set-up, introspection and options.  Part 2: tmx_step.hip.
#include "tmx_hostshared.h"

// ---------------------------------------------------------------------------------------------
// errors

static thread_local std::string g_err;

void tmx_set_error(const char * fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_err = buf;
}

extern "C" const char * tmx_last_error(void) { return g_err.c_str(); }



// ---------------------------------------------------------------------------------------------
// RCCL, resolved at run time so the library loads without it (single-GPU use, CPU symbol checks)

NcclApi g_nccl;

int load_rccl() {
	if (g_nccl.Send) return TMX_OK;
	// prefer an RCCL already in the process (torch ships its own); otherwise the ROCm one
	void * h = dlopen(nullptr, RTLD_NOW | RTLD_GLOBAL);
	if (!h || !dlsym(h, "ncclSend")) {
		const char * names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr };   // soname first: reuses an RCCL already loaded (torch bundles one)
		h = nullptr;
		for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
	}
	REQUIRE(h != nullptr, TMX_ERR_COMM, "RCCL not found (librccl.so)");
	g_nccl.lib = h;
	g_nccl.GetUniqueId = (fn_ncclGetUniqueId)dlsym(h, "ncclGetUniqueId");
	g_nccl.CommInitRank = (fn_ncclCommInitRank)dlsym(h, "ncclCommInitRank");
	g_nccl.CommDestroy = (fn_ncclCommDestroy)dlsym(h, "ncclCommDestroy");
	g_nccl.GroupStart = (fn_ncclGroupStart)dlsym(h, "ncclGroupStart");
	g_nccl.GroupEnd = (fn_ncclGroupEnd)dlsym(h, "ncclGroupEnd");
	g_nccl.Send = (fn_ncclSend)dlsym(h, "ncclSend");
	g_nccl.Recv = (fn_ncclRecv)dlsym(h, "ncclRecv");
	g_nccl.GetErrorString = (fn_ncclGetErrorString)dlsym(h, "ncclGetErrorString");
	g_nccl.CommCount = (fn_ncclCommCount)dlsym(h, "ncclCommCount");
	REQUIRE(g_nccl.GetUniqueId && g_nccl.CommInitRank && g_nccl.Send && g_nccl.Recv && g_nccl.GroupStart && g_nccl.GroupEnd,
		TMX_ERR_COMM, "RCCL symbols missing");
	return TMX_OK;
}

void prof_collect(tmx_engine * e) {
	for (auto & pe : e->prof_pending) {
		float ms = 0.f;
		hipEventSynchronize(pe.second.second);
		hipEventElapsedTime(&ms, pe.second.first, pe.second.second);
		e->prof_slots[pe.first].ms += ms;
		e->prof_slots[pe.first].n += 1;
		hipEventDestroy(pe.second.first);
		hipEventDestroy(pe.second.second);
	}
	e->prof_pending.clear();
}

// ---------------------------------------------------------------------------------------------
// life cycle

static void tmx_record_option_defaults(tmx_engine * e);
static bool options_table_consistent();
extern "C" int tmx_create(const tmx_config * cfg, tmx_engine ** out) {
	REQUIRE(cfg && out, TMX_ERR_INVALID, "tmx_create: null argument");
	{ static const bool ok = options_table_consistent(); if (!ok) return TMX_ERR_INVALID; }
	REQUIRE(cfg->abi_version == TMX_ABI_VERSION, TMX_ERR_INVALID, "tmx_create: ABI version %d, library is %d", cfg->abi_version, TMX_ABI_VERSION);
	REQUIRE(cfg->horizontal_order == TMX_NP, TMX_ERR_UNSUPPORTED, "horizontal order %d unsupported (np = 4 only)", cfg->horizontal_order);
	REQUIRE(cfg->vertical_order == 1, TMX_ERR_UNSUPPORTED, "vertical order %d unsupported (1 only)", cfg->vertical_order);
	REQUIRE(cfg->n_tracers >= 0 && cfg->n_tracers <= 16, TMX_ERR_UNSUPPORTED, "0..16 tracers supported, got %d", cfg->n_tracers);
	REQUIRE(cfg->n_tracers == 0 || cfg->equation_set == TMX_EQN_PRIMITIVE_NONHYDROSTATIC, TMX_ERR_UNSUPPORTED, "tracers are supported with the nonhydrostatic equation set only");
	REQUIRE(cfg->hypervis_order == 4 || cfg->hypervis_order == 2 || cfg->hypervis_order == 0, TMX_ERR_UNSUPPORTED, "hyperviscosity order %d unsupported (0, 2 or 4)", cfg->hypervis_order);
	REQUIRE((cfg->fully_explicit == 0 || cfg->fully_explicit == 1) && (cfg->uniform_diffusion == 0 || cfg->uniform_diffusion == 1), TMX_ERR_INVALID, "fully_explicit / uniform_diffusion must be 0 or 1");
	REQUIRE(!(cfg->fully_explicit || cfg->uniform_diffusion) || cfg->equation_set == TMX_EQN_PRIMITIVE_NONHYDROSTATIC, TMX_ERR_UNSUPPORTED,
		"fully explicit vertical dynamics / uniform diffusion need the nonhydrostatic equation set");
	// With implicit vertical dynamics the reference's StepExplicit diffuses whatever column the previous StepImplicit
	// left in m_dStateNode (SetupReferenceColumn is not called there, VerticalDynamicsFEM.cpp:745-812 vs :1059-1105)
	// and its tracer update throws "Not implemented" (:3914-3917): no stock case uses that combination.
	REQUIRE(!cfg->uniform_diffusion || cfg->fully_explicit, TMX_ERR_UNSUPPORTED, "uniform diffusion is supported with fully explicit vertical dynamics only (the supercell configuration)");
	REQUIRE(!cfg->uniform_diffusion || cfg->ztop > 0.0, TMX_ERR_INVALID, "uniform diffusion needs cfg.ztop > 0");
	REQUIRE(cfg->equation_set == TMX_EQN_PRIMITIVE_NONHYDROSTATIC || cfg->equation_set == TMX_EQN_SHALLOW_WATER,
		TMX_ERR_UNSUPPORTED, "equation set %d unsupported", cfg->equation_set);
	if (cfg->equation_set == TMX_EQN_SHALLOW_WATER) REQUIRE(cfg->levels == 1, TMX_ERR_INVALID, "shallow water needs levels == 1");
	else REQUIRE(cfg->levels >= 3, TMX_ERR_INVALID, "levels must be >= 3");
	REQUIRE(cfg->n_patches >= 1 && cfg->n_instances >= 1, TMX_ERR_INVALID, "bad patch / instance count");
	REQUIRE(cfg->n_instances <= 32, TMX_ERR_UNSUPPORTED, "%d data instances: the stepper programs track instances in 32-bit masks", cfg->n_instances);
	REQUIRE(cfg->n_ranks >= 1 && cfg->rank >= 0 && cfg->rank < cfg->n_ranks, TMX_ERR_INVALID, "bad rank %d of %d", cfg->rank, cfg->n_ranks);
	tmx_engine * e = new tmx_engine();
	tmx_record_option_defaults(e);      // what "the default" of every option is (the production flavour refuses any other value of an experiments option)
	e->cfg = *cfg;
	e->L = cfg->levels;
	e->patches.resize(cfg->n_patches);
	e->nt = cfg->n_tracers;
	e->nslab = 5 * e->L + 1 + e->nt * e->L;
	e->sw = (cfg->equation_set == TMX_EQN_SHALLOW_WATER);
	e->fully_explicit = cfg->fully_explicit != 0; e->udiff = cfg->uniform_diffusion != 0;
	e->imap.resize(cfg->n_instances);
	for (int k = 0; k < cfg->n_instances; k++) e->imap[k] = k;
	e->uvmap = e->imap;
	if (e->sw) { e->h_ops.assign((size_t)TMX_OP_COUNT * (e->L + 1) * TMX_OPW, 0.0); e->ops_set = true;
		for (int i = 0; i < 16; i++) { e->h_dx[i] = 0.0; e->h_stiff[i] = 0.0; } }
	// (no environment variable is read here: options arrive through tmx_set_option, or -- for test and bench plumbing that wants the
	// old TMX_* variables -- through an explicit tmx_options_from_environment call, which reports what it applied)
	if (!plan_only(e)) {
		int ndev = 0;
		hipError_t r = hipGetDeviceCount(&ndev);
		if (r != hipSuccess || ndev == 0) {
			delete e;
			tmx_set_error("tmx_create: no HIP device available (%s); this engine has no CPU path", hipGetErrorString(r));
			return TMX_ERR_DEVICE;
		}
		if (r == hipSuccess && cfg->device >= 0) r = hipSetDevice(cfg->device);
		if (r == hipSuccess) r = hipGetDevice(&e->device);
		if (r == hipSuccess) r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
		if (r != hipSuccess) {
			delete e;
			tmx_set_error("tmx_create: device %d: %s", cfg->device, hipGetErrorString(r));
			return TMX_ERR_DEVICE;
		}
	}
	*out = e;
	return TMX_OK;
}

static void free_dev(void * p) { if (p) hipFree(p); }
static void g_plans_erase(tmx_engine * e);


extern "C" void tmx_destroy(tmx_engine * e) {
	if (!e) return;
	interp_orphan(e);
	if (!plan_only(e)) {
		if (e->stream) hipStreamSynchronize(e->stream);
		for (auto & g : e->graphs) hipGraphExecDestroy(g.exec);
		if (e->h_stage) hipHostFree(e->h_stage);
		if (e->xstream) { hipStreamSynchronize(e->xstream); hipStreamDestroy(e->xstream); hipEventDestroy(e->ev_pack); hipEventDestroy(e->ev_recv); }
		prof_collect(e);
		if (e->comm && g_nccl.CommDestroy) g_nccl.CommDestroy(e->comm);
		free_dev(e->d_state); free_dev(e->d_ref); free_dev(e->d_refd); free_dev(e->d_area); free_dev(e->d_w0); free_dev(e->d_eta); free_dev(e->d_ray_nu); free_dev(e->d_ray_ref); free_dev(e->d_g2d); free_dev(e->d_g3n); free_dev(e->d_g3e); free_dev(e->d_ops);
		free_dev(e->d_scratch); free_dev(e->d_grp_cols); free_dev(e->d_colref); free_dev(e->d_hvblocks); free_dev(e->d_grp_n); free_dev(e->d_grp_x); free_dev(e->d_grp_type); free_dev(e->d_xmat); free_dev(e->d_zlev); free_dev(e->d_prect); free_dev(e->d_kes); free_dev(e->d_quads_early); free_dev(e->d_quads_late); free_dev(e->d_pivot_stats); free_dev(e->d_image);
		if (e->d_ghost_own) e->d_ghost = e->d_ghost_own;          // p2p mode pointed d_ghost into the shared block
		for (void * q : e->p2p_peer) if (q) hipIpcCloseMemHandle(q);
		free_dev(e->p2p_block); free_dev(e->d_p2p_dst); free_dev(e->d_p2p_flag); free_dev(e->d_send_peer); free_dev(e->d_send_within); free_dev(e->d_p2p_peers);
		free_dev(e->d_ghost); free_dev(e->d_sendbuf); free_dev(e->d_send_cols); free_dev(e->d_send_base); free_dev(e->d_send_stride); free_dev(e->d_ghost_base); free_dev(e->d_ghost_stride); free_dev(e->d_ucol); free_dev(e->d_udep);
		free_dev(e->d_ab); free_dev(e->d_rhs); free_dev(e->d_flag);
		tmxu_free(e);
		if (e->stream) hipStreamDestroy(e->stream);
	}
	g_plans_erase(e);
	delete e;
}

// ---------------------------------------------------------------------------------------------
// operators

extern "C" int tmx_set_operators(tmx_engine * e, const double * dx_basis, const double * stiffness,
	const double * const * coeff, const int * const * ix_begin, const int * const * ix_end,
	const int * n_out, const int * n_in)
{
	REQUIRE(e && dx_basis && stiffness, TMX_ERR_INVALID, "tmx_set_operators: null argument");
	REQUIRE(e->sw || (coeff && ix_begin && ix_end && n_out && n_in), TMX_ERR_INVALID, "tmx_set_operators: null argument");
	const int L = e->L;
	memcpy(e->h_dx, dx_basis, sizeof(double) * 16);
	memcpy(e->h_stiff, stiffness, sizeof(double) * 16);
	if (e->sw) { e->ops_set = true; return TMX_OK; }      // no column operators on one level
	// expected shapes (GridGLL.cpp:278-362)
	const int exp_out[TMX_OP_COUNT] = { L + 1, L, L, L + 1, L, L + 1, L, L + 1, L, L };
	const int exp_in[TMX_OP_COUNT]  = { L, L + 1, L, L, L + 1, L + 1, L, L + 1, L, L };
	// offsets each kernel is written for (vertical order 1 stencils)
	const int lo[TMX_OP_COUNT] = { -2, 0, -1, -1, 0, -1, -1, -1, 0, -1 };
	const int hi[TMX_OP_COUNT] = { 1, 1, 1, 0, 1, 1, 1, 1, 1, 0 };
	e->h_ops.assign((size_t)TMX_OP_COUNT * (L + 1) * TMX_OPW, 0.0);
	for (int op = 0; op < TMX_OP_COUNT; op++) {
		REQUIRE(n_out[op] == exp_out[op] && n_in[op] == exp_in[op], TMX_ERR_INVALID,
			"operator %d has shape %dx%d, expected %dx%d", op, n_out[op], n_in[op], exp_out[op], exp_in[op]);
		for (int k = 0; k < n_out[op]; k++) {
			for (int l = ix_begin[op][k]; l < ix_end[op][k]; l++) {
				const double c = coeff[op][(size_t)k * n_in[op] + l];
				if (c == 0.0) continue;
				const int off = l - k;
				REQUIRE(off >= lo[op] && off <= hi[op], TMX_ERR_UNSUPPORTED,
					"operator %d row %d has a coefficient at offset %d outside the vertical-order-1 stencil", op, k, off);
				if (op == TMX_OP_INTERP_NODE_TO_REDGE && k >= 1 && k <= L - 1)
					REQUIRE(off == -1 || off == 0, TMX_ERR_UNSUPPORTED, "interior row %d of InterpNodeToREdge is not a two-point stencil", k);
				e->h_ops[((size_t)op * (L + 1) + k) * TMX_OPW + (off + 2)] = c;
			}
		}
	}
	e->ops_set = true;
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// patches, halos, layout

extern "C" int tmx_define_patch(tmx_engine * e, int patch, int panel, int elems_a, int elems_b, int owner_rank,
	const int * neighbor_panels)
{
	REQUIRE(e && neighbor_panels, TMX_ERR_INVALID, "tmx_define_patch: null argument");
	REQUIRE(!e->finalized && e->ne_local == 0, TMX_ERR_INVALID, "tmx_define_patch after layout was fixed");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index %d out of range", patch);
	REQUIRE(elems_a > 0 && elems_b > 0 && panel >= 0 && panel < 6, TMX_ERR_INVALID, "bad patch box");
	REQUIRE(owner_rank >= 0 && owner_rank < e->cfg.n_ranks, TMX_ERR_INVALID, "owner rank %d out of range", owner_rank);
	PatchInfo & P = e->patches[patch];
	P.defined = true; P.panel = panel; P.nea = elems_a; P.neb = elems_b;
	P.na = TMX_NP * elems_a + 2; P.nb = TMX_NP * elems_b + 2; P.owner = owner_rank;
	memcpy(P.nbp, neighbor_panels, sizeof(int) * 8);
	return TMX_OK;
}

// GridPatchGLL::GetElementDeltaA / GetElementDeltaB of one patch.  The reference forms them as the difference of two
// element-edge coordinates of the patch (GridPatchGLL.cpp:67-75), so they differ from pi / (2 ne) -- and from patch to
// patch -- in the last bits, and every horizontal derivative is scaled with the patch's own 1 / delta (HorizontalDynamicsFEM.
// cpp:837-838, 1957-1958 ...).  Optional: without the call cfg.element_delta_a serves both directions of the patch.
extern "C" int tmx_set_patch_element_spacing(tmx_engine * e, int patch, double delta_a, double delta_b) {
	REQUIRE(e, TMX_ERR_INVALID, "tmx_set_patch_element_spacing: null engine");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches && e->patches[patch].defined, TMX_ERR_INVALID, "patch %d not defined", patch);
	REQUIRE(!e->finalized, TMX_ERR_INVALID, "tmx_set_patch_element_spacing after tmx_finalize");
	REQUIRE(delta_a > 0.0 && delta_b > 0.0, TMX_ERR_INVALID, "element spacing must be positive");
	e->patches[patch].da = delta_a; e->patches[patch].db = delta_b;
	return TMX_OK;
}

extern "C" int tmx_set_patch_halo(tmx_engine * e, int patch, int n, const int * halo_i, const int * halo_j,
	const int * src_patch, const int * src_i, const int * src_j, const int * src_panel, const double * trans)
{
	REQUIRE(e && halo_i && halo_j && src_patch && src_i && src_j && src_panel && trans, TMX_ERR_INVALID, "tmx_set_patch_halo: null argument");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches && e->patches[patch].defined, TMX_ERR_INVALID, "patch %d not defined", patch);
	REQUIRE(!e->finalized, TMX_ERR_INVALID, "tmx_set_patch_halo after tmx_finalize");
	PatchInfo & P = e->patches[patch];
	P.hi.assign(halo_i, halo_i + n); P.hj.assign(halo_j, halo_j + n);
	P.hsp.assign(src_patch, src_patch + n); P.hsi.assign(src_i, src_i + n); P.hsj.assign(src_j, src_j + n);
	P.hspanel.assign(src_panel, src_panel + n);
	P.htrans.assign(trans, trans + (size_t)4 * n);
	for (int m = 0; m < n; m++) {
		REQUIRE(P.hi[m] >= 0 && P.hi[m] < P.na && P.hj[m] >= 0 && P.hj[m] < P.nb, TMX_ERR_INVALID, "halo node out of range");
		REQUIRE(P.hsp[m] < e->cfg.n_patches, TMX_ERR_INVALID, "halo source patch out of range");
	}
	P.halo_set = true;
	return TMX_OK;
}

int ensure_layout(tmx_engine * e) {
	if (e->ne_local > 0) return TMX_OK;
	int ne = 0;
	e->local_patches.clear();
	for (int p = 0; p < e->cfg.n_patches; p++) {
		PatchInfo & P = e->patches[p];
		REQUIRE(P.defined, TMX_ERR_INVALID, "patch %d was never defined", p);
		if (P.owner == e->cfg.rank) { P.elem_base = ne; ne += P.nea * P.neb; e->local_patches.push_back(p); }
	}
	REQUIRE(ne > 0, TMX_ERR_INVALID, "rank %d owns no patch", e->cfg.rank);
	e->ne_local = ne;
	e->ncol = ne * TMX_NQ;
	e->NS = ((e->ncol + TMX_TILE - 1) / TMX_TILE) * TMX_TILE;
	// + 2 "surface slots" behind the exchanged slabs: the interface-level-0 entries of rho and rho*theta of the reference's
	// REdge array, which nothing but the stage algebra touches and HeldSuarezPhysics reads (see surface_copy below)
	e->inst_stride = (size_t)(e->nslab + 2) * e->NS;
	const size_t NS = e->NS; const int L = e->L;
	e->h_g2d.assign((size_t)G2_COUNT * NS, 0.0);
	e->h_g3n.assign((size_t)G3N_COUNT * L * NS, 0.0);
	e->h_g3e.assign((size_t)G3E_COUNT * (L + 1) * NS, 0.0);
	// benign values in the padding columns
	for (size_t c = e->ncol; c < NS; c++) { e->h_g2d[G2_J2D * NS + c] = 1.0; e->h_g2d[G2_JN * NS + c] = 1.0; e->h_g2d[G2_JE * NS + c] = 1.0; }
	return TMX_OK;
}

extern "C" int tmx_set_patch_geometry(tmx_engine * e, int patch, const tmx_patch_geometry * g) {
	REQUIRE(e && g, TMX_ERR_INVALID, "tmx_set_patch_geometry: null argument");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index out of range");
	REQUIRE(!e->finalized, TMX_ERR_INVALID, "tmx_set_patch_geometry after tmx_finalize");
	int r = ensure_layout(e);
	if (r) return r;
	PatchInfo & P = e->patches[patch];
	REQUIRE(P.owner == e->cfg.rank, TMX_ERR_INVALID, "patch %d is not owned by rank %d", patch, e->cfg.rank);
	const int L = e->L, nb = P.nb;
	const size_t NS = e->NS;
	for (int i = 1; i < P.na - 1; i++)
	for (int j = 1; j < P.nb - 1; j++) {
		const int c = col_of(P, i, j);
		const size_t ij = (size_t)i * nb + j;
		e->h_g2d[G2_J2D * NS + c] = g->jacobian2d[ij];
		e->h_g2d[G2_F * NS + c] = g->coriolis_f[ij];
		e->h_g2d[G2_C2A0 * NS + c] = g->contra_metric_2d_a[ij * 2 + 0];
		e->h_g2d[G2_C2A1 * NS + c] = g->contra_metric_2d_a[ij * 2 + 1];
		e->h_g2d[G2_C2B1 * NS + c] = g->contra_metric_2d_b[ij * 2 + 1];
		REQUIRE(g->contra_metric_2d_b[ij * 2 + 0] == g->contra_metric_2d_a[ij * 2 + 1], TMX_ERR_UNSUPPORTED, "2-D contravariant metric is not symmetric");
		e->h_g2d[G2_JN * NS + c] = g->jacobian[ij * L];
		if (e->nt > 0) {
			REQUIRE(g->element_area_node, TMX_ERR_INVALID, "tracers need tmx_patch_geometry.element_area_node (patch %d)", patch);
			if (e->h_area.empty()) e->h_area.assign((size_t)L * NS, 0.0);
			for (int k = 0; k < L; k++) e->h_area[(size_t)k * NS + c] = g->element_area_node[ij * L + k];
		}
		if (e->sw) {
			// 2-D equation set: only the 2-D metric, the (single-level) Jacobian and the topography are used
			e->h_g2d[G2_JE * NS + c] = g->jacobian[ij * L];
			e->h_g2d[G2_DRX * NS + c] = 1.0;
			e->h_g2d[G2_ZS * NS + c] = g->topography ? g->topography[ij] : 0.0;
			continue;
		}
		e->h_g2d[G2_JE * NS + c] = g->jacobian_redge[ij * (L + 1)];
		e->h_g2d[G2_DRX * NS + c] = g->deriv_r_node[(ij * L) * 3 + 2];
		for (int k = 0; k < L; k++) {
			const size_t o = (ij * L + k) * 3;
			// structure of the reference metric (GridPatchCSGLL.cpp:444-494): level-independent
			// Jacobian / d_xi R, horizontal block equal to the 2-D metric, symmetric cross terms
			REQUIRE(g->jacobian[ij * L + k] == g->jacobian[ij * L] &&
			        g->deriv_r_node[o + 2] == g->deriv_r_node[(ij * L) * 3 + 2] &&
			        g->contra_metric_a[o + 0] == g->contra_metric_2d_a[ij * 2 + 0] &&
			        g->contra_metric_a[o + 1] == g->contra_metric_2d_a[ij * 2 + 1] &&
			        g->contra_metric_b[o + 0] == g->contra_metric_2d_b[ij * 2 + 0] &&
			        g->contra_metric_b[o + 1] == g->contra_metric_2d_b[ij * 2 + 1] &&
			        g->contra_metric_xi[o + 0] == g->contra_metric_a[o + 2] &&
			        g->contra_metric_xi[o + 1] == g->contra_metric_b[o + 2],
			        TMX_ERR_UNSUPPORTED, "node metric of patch %d does not have the Gal-Chen structure the engine stores", patch);
			const size_t d = (size_t)k * NS + c, s3 = (size_t)L * NS;
			e->h_g3n[G3N_CA2 * s3 + d] = g->contra_metric_a[o + 2];
			e->h_g3n[G3N_CB2 * s3 + d] = g->contra_metric_b[o + 2];
			e->h_g3n[G3N_CX2 * s3 + d] = g->contra_metric_xi[o + 2];
			e->h_g3n[G3N_DRA * s3 + d] = g->deriv_r_node[o + 0];
			e->h_g3n[G3N_DRB * s3 + d] = g->deriv_r_node[o + 1];
		}
		for (int k = 0; k <= L; k++) {
			const size_t o = (ij * (L + 1) + k) * 3;
			REQUIRE(g->jacobian_redge[ij * (L + 1) + k] == g->jacobian_redge[ij * (L + 1)] &&
			        g->deriv_r_redge[o + 2] == g->deriv_r_node[(ij * L) * 3 + 2] &&
			        g->contra_metric_a_redge[o + 0] == g->contra_metric_2d_a[ij * 2 + 0] &&
			        g->contra_metric_a_redge[o + 1] == g->contra_metric_2d_a[ij * 2 + 1] &&
			        g->contra_metric_b_redge[o + 0] == g->contra_metric_2d_b[ij * 2 + 0] &&
			        g->contra_metric_b_redge[o + 1] == g->contra_metric_2d_b[ij * 2 + 1] &&
			        g->contra_metric_xi_redge[o + 0] == g->contra_metric_a_redge[o + 2] &&
			        g->contra_metric_xi_redge[o + 1] == g->contra_metric_b_redge[o + 2],
			        TMX_ERR_UNSUPPORTED, "interface metric of patch %d does not have the Gal-Chen structure the engine stores", patch);
			const size_t d = (size_t)k * NS + c, s3 = (size_t)(L + 1) * NS;
			e->h_g3e[G3E_CX0 * s3 + d] = g->contra_metric_xi_redge[o + 0];
			e->h_g3e[G3E_CX1 * s3 + d] = g->contra_metric_xi_redge[o + 1];
			e->h_g3e[G3E_CX2 * s3 + d] = g->contra_metric_xi_redge[o + 2];
		}
	}
	P.geom_set = true;
	return TMX_OK;
}

// Closed form of the 3-D metric (GridPatchCSGLL.cpp:370-568).  The factors are accepted only if they
// reproduce every stored value bit for bit, so the kernels' in-register evaluation cannot change results.
extern "C" int tmx_set_patch_metric_factors(tmx_engine * e, int patch, const double * x_node, const double * y_node,
	const double * topography_deriv, double earth_radius, const double * reta_levels, const double * reta_interfaces)
{
	REQUIRE(e && x_node && y_node && topography_deriv && reta_levels && reta_interfaces, TMX_ERR_INVALID, "tmx_set_patch_metric_factors: null argument");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index out of range");
	REQUIRE(!e->finalized, TMX_ERR_INVALID, "tmx_set_patch_metric_factors after tmx_finalize");
	PatchInfo & P = e->patches[patch];
	REQUIRE(P.geom_set, TMX_ERR_INVALID, "tmx_set_patch_metric_factors before tmx_set_patch_geometry (patch %d)", patch);
	P.metric_ok = false;
	if (e->sw) return TMX_OK;                 // 2-D equation set: there is no 3-D metric
	const int L = e->L, nb = P.nb;
	const size_t NS = e->NS;
	std::vector<double> eta(2 * L + 1);
	for (int k = 0; k < L; k++) eta[k] = 1.0 - reta_levels[k];
	for (int k = 0; k <= L; k++) eta[L + k] = 1.0 - reta_interfaces[k];
	if (e->h_eta.empty()) e->h_eta = eta;
	else if (memcmp(e->h_eta.data(), eta.data(), eta.size() * sizeof(double)) != 0) return TMX_OK;   // patches disagree
	const size_t s3n = (size_t)L * NS, s3e = (size_t)(L + 1) * NS;
	bool ok = true;
	for (int i = 1; i < P.na - 1 && ok; i++)
	for (int j = 1; j < P.nb - 1 && ok; j++) {
		const int c = col_of(P, i, j);
		const size_t ij = (size_t)i * nb + j;
		const double dX = x_node[i], dY = y_node[j];
		const double dDelta2 = (1.0 + dX * dX + dY * dY);
		const double sc = dDelta2 / (1.0 + dX * dX) / (1.0 + dY * dY) / (earth_radius * earth_radius);
		const double dxr = e->h_g2d[G2_DRX * NS + c];
		const double mp = -sc / dxr, ma = (1.0 + dY * dY), mb = dX * dY, mc = (1.0 + dX * dX);
		const double daz = topography_deriv[ij * 2 + 0], dbz = topography_deriv[ij * 2 + 1];
		const double idx = 1.0 / dxr, idx2 = 1.0 / (dxr * dxr);
		// the 2-D metric must come from the same factors
		ok = ok && e->h_g2d[G2_C2A0 * NS + c] == sc * ma && e->h_g2d[G2_C2A1 * NS + c] == sc * dX * dY &&
		     e->h_g2d[G2_C2B1 * NS + c] == sc * mc;
		for (int k = 0; k < 2 * L + 1 && ok; k++) {
			const double dar = eta[k] * daz, dbr = eta[k] * dbz;
			const double c0 = mp * (ma * dar + mb * dbr);
			const double c1 = mp * (mb * dar + mc * dbr);
			const double c2 = idx2 - idx * (c0 * dar + c1 * dbr);
			if (k < L) {
				const size_t d = (size_t)k * NS + c;
				ok = e->h_g3n[G3N_CA2 * s3n + d] == c0 && e->h_g3n[G3N_CB2 * s3n + d] == c1 && e->h_g3n[G3N_CX2 * s3n + d] == c2 &&
				     e->h_g3n[G3N_DRA * s3n + d] == dar && e->h_g3n[G3N_DRB * s3n + d] == dbr;
			} else {
				const size_t d = (size_t)(k - L) * NS + c;
				ok = e->h_g3e[G3E_CX0 * s3e + d] == c0 && e->h_g3e[G3E_CX1 * s3e + d] == c1 && e->h_g3e[G3E_CX2 * s3e + d] == c2;
			}
		}
		e->h_g2d[G2_MP * NS + c] = mp; e->h_g2d[G2_MA * NS + c] = ma; e->h_g2d[G2_MB * NS + c] = mb; e->h_g2d[G2_MC * NS + c] = mc;
		e->h_g2d[G2_DAZ * NS + c] = daz; e->h_g2d[G2_DBZ * NS + c] = dbz; e->h_g2d[G2_IDX * NS + c] = idx; e->h_g2d[G2_IDX2 * NS + c] = idx2;
	}
	P.metric_ok = ok;
	return TMX_OK;
}

// Column physics inputs (latitude, Held-Suarez surface pressure) into two 2-D slabs
extern "C" int tmx_set_patch_physics_inputs(tmx_engine * e, int patch, const double * latitude, const double * surface_pressure) {
	REQUIRE(e && latitude, TMX_ERR_INVALID, "tmx_set_patch_physics_inputs: null argument");
	// surface_pressure == NULL: the forcing forms it from the tracked surface slots, as the reference does (see surface_copy)
	REQUIRE(e->patches_with_physics == 0 || e->track_surface == (surface_pressure == nullptr), TMX_ERR_INVALID,
		"tmx_set_patch_physics_inputs: every patch must use the same surface-pressure mode");
	e->track_surface = (surface_pressure == nullptr);
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index out of range");
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "column physics with the shallow-water equation set is not supported");
	int r = ensure_layout(e);
	if (r) return r;
	PatchInfo & P = e->patches[patch];
	REQUIRE(P.owner == e->cfg.rank, TMX_ERR_INVALID, "patch %d is not owned by rank %d", patch, e->cfg.rank);
	const size_t NS = e->NS;
	std::vector<double> lat((size_t)P.nea * P.neb * TMX_NQ), ps(lat.size()), sl(lat.size()), cl(lat.size());
	const int c0 = P.elem_base * TMX_NQ;
	for (int i = 1; i < P.na - 1; i++)
	for (int j = 1; j < P.nb - 1; j++) {
		const int c = col_of(P, i, j);
		lat[c - c0] = latitude[(size_t)i * P.nb + j];
		ps[c - c0] = surface_pressure ? surface_pressure[(size_t)i * P.nb + j] : 0.0;
		sl[c - c0] = sin(lat[c - c0]); cl[c - c0] = cos(lat[c - c0]);      // host libm: the values the reference computes per call
		if (!e->finalized) { e->h_g2d[G2_LAT * NS + c] = lat[c - c0]; e->h_g2d[G2_PS * NS + c] = ps[c - c0];
			e->h_g2d[G2_SINLAT * NS + c] = sl[c - c0]; e->h_g2d[G2_COSLAT * NS + c] = cl[c - c0]; }
	}
	if (e->finalized && !plan_only(e)) {       // may be refreshed at any time (the caller owns the source arrays)
		HIPCHK(hipStreamSynchronize(e->stream));
		HIPCHK(hipMemcpy(e->d_g2d + G2_LAT * NS + c0, lat.data(), lat.size() * sizeof(double), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(e->d_g2d + G2_PS * NS + c0, ps.data(), ps.size() * sizeof(double), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(e->d_g2d + G2_SINLAT * NS + c0, sl.data(), sl.size() * sizeof(double), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(e->d_g2d + G2_COSLAT * NS + c0, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
	}
	if (!P.physics_set) e->patches_with_physics++;
	P.physics_set = true;
	e->u.physics_dirty = true;
	return TMX_OK;
}

// Rayleigh friction inputs (GridPatch::GetRayleighStrength / GetReferenceState), gathered into the device layout
extern "C" int tmx_set_patch_rayleigh(tmx_engine * e, int patch, const double * strength_node, const double * strength_redge,
	const double * ref_node, const double * ref_redge)
{
	REQUIRE(e && strength_node && strength_redge && ref_node && ref_redge, TMX_ERR_INVALID, "tmx_set_patch_rayleigh: null argument");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index out of range");
	REQUIRE(!e->finalized, TMX_ERR_INVALID, "tmx_set_patch_rayleigh after tmx_finalize");
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "Rayleigh friction with the shallow-water equation set is not supported");
	int r = ensure_layout(e);
	if (r) return r;
	PatchInfo & P = e->patches[patch];
	REQUIRE(P.owner == e->cfg.rank, TMX_ERR_INVALID, "patch %d is not owned by rank %d", patch, e->cfg.rank);
	const int L = e->L, na = P.na, nb = P.nb;
	const size_t NS = e->NS;
	if (e->h_ray_nu.empty()) { e->h_ray_nu.assign((size_t)(2 * L + 1) * NS, 0.0); e->h_ray_ref.assign((size_t)(4 * L + 1) * NS, 0.0); }
	for (int i = 1; i < na - 1; i++)
	for (int j = 1; j < nb - 1; j++) {
		const int c = col_of(P, i, j);
		const size_t ij = (size_t)i * nb + j;
		for (int k = 0; k < L; k++) {
			e->h_ray_nu[(size_t)k * NS + c] = strength_node[ij * L + k];
			for (int v = 0; v < 3; v++)      // U, V, rho*theta
				e->h_ray_ref[(size_t)(v * L + k) * NS + c] = ref_node[(((size_t)v * na + i) * nb + j) * L + k];
		}
		for (int k = 0; k <= L; k++) {
			e->h_ray_nu[(size_t)(L + k) * NS + c] = strength_redge[ij * (L + 1) + k];
			e->h_ray_ref[(size_t)(3 * L + k) * NS + c] = ref_redge[(((size_t)3 * na + i) * nb + j) * (L + 1) + k];
		}
	}
	P.rayleigh_set = true;
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// finalize: DSS groups, unique columns, exchange plan, device buffers

struct NodeRef { int patch, i, j; };

static int uf_find(std::vector<int> & par, int x) {
	while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; }
	return x;
}

template <class T> static int dev_upload(T ** d, const std::vector<T> & h, size_t * bytes) {
	const size_t n = h.size() ? h.size() : 1;
	HIPCHK(hipMalloc((void **)d, n * sizeof(T)));
	if (h.size()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
	*bytes += n * sizeof(T);
	return TMX_OK;
}

// One plan per engine.  Engines may be driven from different host threads (the loopback tests do): the map is only
// touched under g_plans_mutex; a std::map never moves its values, so the reference handed out stays valid until the
// engine's own tmx_destroy erases it.
static std::map<tmx_engine *, PlanHost> g_plans;
static std::mutex g_plans_mutex;
PlanHost & plan_of(tmx_engine * e) { std::lock_guard<std::mutex> lk(g_plans_mutex); return g_plans[e]; }
static void g_plans_erase(tmx_engine * e) { std::lock_guard<std::mutex> lk(g_plans_mutex); g_plans.erase(e); }

extern "C" int tmx_finalize(tmx_engine * e) {
	REQUIRE(e, TMX_ERR_INVALID, "tmx_finalize: null engine");
	REQUIRE(!e->finalized, TMX_ERR_INVALID, "tmx_finalize called twice");
	REQUIRE(e->ops_set, TMX_ERR_INVALID, "tmx_set_operators must precede tmx_finalize");
	int r = ensure_layout(e);
	if (r) return r;
	const int np = e->cfg.n_patches, me = e->cfg.rank, NR = e->cfg.n_ranks;
	for (int p = 0; p < np; p++) {
		REQUIRE(e->patches[p].halo_set, TMX_ERR_INVALID, "halo of patch %d not set", p);
		if (e->patches[p].owner == me && !plan_only(e))
			REQUIRE(e->patches[p].geom_set, TMX_ERR_INVALID, "geometry of local patch %d not set", p);
	}
	PlanHost & plan = plan_of(e);

	// ---- union-find over all interior nodes of all patches
	std::vector<size_t> poff(np + 1, 0);
	for (int p = 0; p < np; p++) poff[p + 1] = poff[p] + (size_t)e->patches[p].na * e->patches[p].nb;
	std::vector<int> par(poff[np]);
	std::iota(par.begin(), par.end(), 0);
	auto nid = [&](int p, int i, int j) { return (int)(poff[p] + (size_t)i * e->patches[p].nb + j); };
	auto unite = [&](int a, int b) { a = uf_find(par, a); b = uf_find(par, b); if (a != b) par[std::max(a, b)] = std::min(a, b); };
	for (int p = 0; p < np; p++) {
		const PatchInfo & P = e->patches[p];
		for (int a = 1; a < P.nea; a++) for (int j = 1; j < P.nb - 1; j++) unite(nid(p, a * TMX_NP, j), nid(p, a * TMX_NP + 1, j));
		for (int b = 1; b < P.neb; b++) for (int i = 1; i < P.na - 1; i++) unite(nid(p, i, b * TMX_NP), nid(p, i, b * TMX_NP + 1));
		for (size_t m = 0; m < P.hi.size(); m++) {
			if (P.hsp[m] < 0) continue;
			const int xi = std::min(std::max(P.hi[m], 1), P.na - 2), xj = std::min(std::max(P.hj[m], 1), P.nb - 2);
			const PatchInfo & Q = e->patches[P.hsp[m]];
			REQUIRE(P.hsi[m] >= 1 && P.hsi[m] < Q.na - 1 && P.hsj[m] >= 1 && P.hsj[m] < Q.nb - 1, TMX_ERR_INVALID, "halo source is not an interior node");
			unite(nid(p, xi, xj), nid(P.hsp[m], P.hsi[m], P.hsj[m]));
		}
	}
	// members per root
	std::map<int, std::vector<int>> comps;
	for (int p = 0; p < np; p++) {
		const PatchInfo & P = e->patches[p];
		for (int i = 1; i < P.na - 1; i++) for (int j = 1; j < P.nb - 1; j++) {
			const bool edge = ((i - 1) % TMX_NP == 0) || ((i - 1) % TMX_NP == TMX_NP - 1) || ((j - 1) % TMX_NP == 0) || ((j - 1) % TMX_NP == TMX_NP - 1);
			if (!edge) continue;
			comps[uf_find(par, nid(p, i, j))].push_back(nid(p, i, j));
		}
	}
	auto node_of = [&](int id) {
		int p = (int)(std::upper_bound(poff.begin(), poff.end(), (size_t)id) - poff.begin()) - 1;
		const int loc = id - (int)poff[p];
		return NodeRef{ p, loc / e->patches[p].nb, loc % e->patches[p].nb };
	};
	// ---- exchange lists: (owner s -> needer r) node ids
	std::vector<std::vector<int>> send_to(NR), recv_from(NR);
	struct Grp { std::vector<int> ids; };
	std::vector<Grp> groups;
	for (auto & kv : comps) {
		std::vector<int> & ids = kv.second;
		if (ids.size() < 2) continue;
		REQUIRE(ids.size() <= 4, TMX_ERR_INVALID, "DSS group with %d members (connectivity is inconsistent)", (int)ids.size());
		std::sort(ids.begin(), ids.end());
		bool local = false;
		for (int id : ids) if (e->patches[node_of(id).patch].owner == me) local = true;
		if (!local) continue;
		groups.push_back(Grp{ ids });
		for (int a : ids) for (int b : ids) {
			const int oa = e->patches[node_of(a).patch].owner, ob = e->patches[node_of(b).patch].owner;
			if (oa == me && ob != me) send_to[ob].push_back(a);
			if (oa != me && ob == me) recv_from[oa].push_back(a);
		}
	}
	// order the groups by the device column of their first local member: consecutive lanes of the
	// DSS kernel then touch the same 128-byte element rows (element-major locality)
	{
		std::vector<std::pair<int, int>> key(groups.size());
		for (size_t g = 0; g < groups.size(); g++) {
			int best = 0x7fffffff;
			for (int id : groups[g].ids) {
				NodeRef nr = node_of(id);
				const PatchInfo & P = e->patches[nr.patch];
				if (P.owner == me) best = std::min(best, col_of(P, nr.i, nr.j));
			}
			key[g] = { best, (int)g };
		}
		// groups with a member on another rank go last: the DSS of the others does not wait for the exchange; groups whose
		// copies all lie in ONE patch go first: the fused hyperviscosity kernel averages those itself (k_hypervis_block)
		std::vector<char> remote(groups.size(), 0);      // 0: one patch, 1: several patches of this rank, 2: a member on another rank
		for (size_t g = 0; g < groups.size(); g++) {
			const int p0 = node_of(groups[g].ids[0]).patch;
			for (int id : groups[g].ids) {
				if (e->patches[node_of(id).patch].owner != me) remote[g] = 2;
				// (the in-patch class only with the fused hyperviscosity kernel: taking the patch-edge groups out of the column
				// order costs k_dss 10 % -- 0.45 instead of 0.41 ms per step at ne30 -- when it still averages all of them)
				else if (e->hvis_block && node_of(id).patch != p0 && remote[g] < 1) remote[g] = 1;
			}
		}
		std::sort(key.begin(), key.end(), [&](const std::pair<int, int> & a, const std::pair<int, int> & b) {
			if (remote[a.second] != remote[b.second]) return remote[a.second] < remote[b.second];
			return a < b;
		});
		std::vector<Grp> sorted(groups.size());
		e->ngroups_local = 0; e->ngroups_inpatch = 0;
		for (size_t g = 0; g < groups.size(); g++) {
			sorted[g] = groups[key[g].second];
			if (remote[key[g].second] < 2) e->ngroups_local++;
			if (e->hvis_block && remote[key[g].second] == 0) e->ngroups_inpatch++;
		}
		groups.swap(sorted);
	}
	e->send_rank_off.assign(NR + 1, 0); e->recv_rank_off.assign(NR + 1, 0);
	std::map<int, int> ghost_index;
	for (int rk = 0; rk < NR; rk++) {
		auto uniq = [](std::vector<int> & v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
		uniq(send_to[rk]); uniq(recv_from[rk]);
		e->send_rank_off[rk + 1] = e->send_rank_off[rk] + (int)send_to[rk].size();
		e->recv_rank_off[rk + 1] = e->recv_rank_off[rk] + (int)recv_from[rk].size();
		for (size_t t = 0; t < recv_from[rk].size(); t++) {
			ghost_index[recv_from[rk][t]] = e->recv_rank_off[rk] + (int)t;
			NodeRef nr = node_of(recv_from[rk][t]);
			plan.recv_nodes.insert(plan.recv_nodes.end(), { nr.patch, nr.i, nr.j });
		}
		for (int id : send_to[rk]) {
			NodeRef nr = node_of(id);
			plan.send_nodes.insert(plan.send_nodes.end(), { nr.patch, nr.i, nr.j });
			plan.send_cols.push_back(col_of(e->patches[nr.patch], nr.i, nr.j));
		}
	}
	e->nsend = e->send_rank_off[NR];
	e->nghost = e->recv_rank_off[NR];
	e->nghost_pad = std::max(e->nghost, 1);

	// ---- device group tables, in the reference's own averaging order
	// GridCSGLL::ApplyDSS (GridCSGLL.cpp:560-781) averages inside every patch (halo ring included) first across the
	// alpha seams, then across the beta seams: a node shared by four copies becomes
	//     0.5 * (0.5 * (x + x_alpha) + 0.5 * (x_beta + x_diag)),
	// with the partners named in the frame of the node's OWN patch, halo values first rotated into that frame
	// (TransformHaloVelocities, GridPatchCSGLL.cpp:1783-1924).  The sums commute but do not associate, so the copies
	// of a node on patches whose alpha axes are not parallel (panel edges towards panels 4 / 5) receive results that
	// differ in the last bit -- in the reference, and therefore here.  Per group the members are stored as
	// [m0, alpha partner, beta partner, diagonal] of member m0, and every member gets a 2-bit pairing type relative to
	// that order (0: {01|23}, 1: {02|13}, 2: {03|12}); cube corners (three copies, (1/3) * ((x + x_alpha) + x_beta),
	// :735-781) get the order of their two partners (0: next, previous; 1: previous, next).  The covector matrices are
	// kept per (member, partner): exactly the matrix of the ring entry through which the member's patch sees the partner.
	std::vector<std::vector<int>> ring(np);
	for (int p = 0; p < np; p++) {
		const PatchInfo & P = e->patches[p];
		ring[p].assign((size_t)P.na * P.nb, -1);
		for (size_t h = 0; h < P.hi.size(); h++) ring[p][(size_t)P.hi[h] * P.nb + P.hj[h]] = (int)h;
	}
	// node id seen by patch p at extended (ring included) position (i, j); h = ring entry or -1
	auto ext = [&](int p, int i, int j, int & h) -> int {
		const PatchInfo & P = e->patches[p];
		h = -1;
		if (i >= 1 && i < P.na - 1 && j >= 1 && j < P.nb - 1) return nid(p, i, j);
		h = ring[p][(size_t)i * P.nb + j];
		if (h < 0 || P.hsp[h] < 0) { h = -1; return -1; }
		return nid(P.hsp[h], P.hsi[h], P.hsj[h]);
	};
	auto seam = [&](int c) -> int { const int q = (c - 1) % TMX_NP; return (q == 0) ? c - 1 : ((q == TMX_NP - 1) ? c + 1 : -1); };
	struct Roles { int id[3]; int h[3]; };      // alpha partner, beta partner, diagonal: node id (-1 none) and ring entry
	auto roles_of = [&](int id) -> Roles {
		const NodeRef nr = node_of(id);
		Roles r;
		const int ia = seam(nr.i), jb = seam(nr.j);
		r.id[0] = (ia >= 0) ? ext(nr.patch, ia, nr.j, r.h[0]) : (r.h[0] = -1, -1);
		r.id[1] = (jb >= 0) ? ext(nr.patch, nr.i, jb, r.h[1]) : (r.h[1] = -1, -1);
		r.id[2] = (ia >= 0 && jb >= 0) ? ext(nr.patch, ia, jb, r.h[2]) : (r.h[2] = -1, -1);
		return r;
	};
	e->ngroups = (int)groups.size();
	plan.grp_cols.assign((size_t)e->ngroups * 4, -1);
	plan.grp_n.assign(e->ngroups, 0);
	plan.grp_x.assign(e->ngroups, -1);
	plan.grp_type.assign(e->ngroups, 0);
	for (int g = 0; g < e->ngroups; g++) {
		std::vector<int> ids = groups[g].ids;
		const int n = (int)ids.size();
		// order: [m0, alpha partner, beta partner, diagonal] of the first member
		{
			const Roles r0 = roles_of(ids[0]);
			std::vector<int> ord(1, ids[0]);
			if (n == 4) { ord.push_back(r0.id[0]); ord.push_back(r0.id[1]); ord.push_back(r0.id[2]); }
			else if (n == 3) { ord.push_back(r0.id[0]); ord.push_back(r0.id[1]); }
			else ord.push_back((r0.id[0] >= 0) ? r0.id[0] : r0.id[1]);
			std::vector<int> chk = ord; std::sort(chk.begin(), chk.end());
			REQUIRE(chk == ids, TMX_ERR_INVALID, "DSS group %d: the seam / halo partners of a node are not the group's members", g);
			ids = ord;
		}
		plan.grp_n[g] = n;
		auto pos = [&](int id) { for (int t = 0; t < n; t++) if (ids[t] == id) return t; return -1; };
		double M[64];
		for (int t = 0; t < 16; t++) { M[4 * t] = 1; M[4 * t + 1] = 0; M[4 * t + 2] = 0; M[4 * t + 3] = 1; }
		bool cross = false;
		int type = 0;
		for (int m = 0; m < n; m++) {
			const NodeRef nr = node_of(ids[m]);
			const PatchInfo & P = e->patches[nr.patch];
			plan.grp_cols[(size_t)g * 4 + m] = (P.owner == me) ? col_of(P, nr.i, nr.j) : e->NS + ghost_index[ids[m]];
			const Roles r = roles_of(ids[m]);
			int ty = 0;
			if (n == 4) {
				const int pa = pos(r.id[0]), pb = pos(r.id[1]), pd = pos(r.id[2]);
				REQUIRE(pa >= 0 && pb >= 0 && pd >= 0 && pa != pb && pa != pd && pb != pd && pa != m && pb != m && pd != m,
					TMX_ERR_INVALID, "DSS group %d: inconsistent partners of member %d", g, m);
				const int lo = std::min(m, pa), hi = std::max(m, pa);
				ty = ((lo == 0 && hi == 1) || (lo == 2 && hi == 3)) ? 0 : (((lo == 0 && hi == 2) || (lo == 1 && hi == 3)) ? 1 : 2);
			} else if (n == 3) {
				const int pa = pos(r.id[0]), pb = pos(r.id[1]);
				REQUIRE(pa >= 0 && pb >= 0 && pa != pb && pa != m && pb != m && r.id[2] < 0, TMX_ERR_INVALID, "DSS group %d: inconsistent cube-corner partners", g);
				ty = (pa == (m + 1) % 3) ? 0 : 1;
			} else {
				const int pp = pos((r.id[0] >= 0) ? r.id[0] : r.id[1]);
				REQUIRE(pp == 1 - m && (r.id[0] < 0 || r.id[1] < 0), TMX_ERR_INVALID, "DSS group %d: inconsistent edge partners", g);
			}
			type |= ty << (2 * m);
			// covector matrices: partner seen through a ring entry whose source lies on another panel
			for (int t = 0; t < 3; t++) {
				if (r.id[t] < 0 || r.h[t] < 0) continue;
				if (P.hspanel[r.h[t]] == P.panel) continue;
				REQUIRE(!P.htrans.empty(), TMX_ERR_INVALID, "covector transforms of patch %d not set", nr.patch);
				memcpy(M + (m * 4 + pos(r.id[t])) * 4, &P.htrans[4 * (size_t)r.h[t]], 4 * sizeof(double));
				cross = true;
			}
		}
		plan.grp_type[g] = type;
		if (cross) {
			plan.grp_x[g] = (int)(plan.xmat.size() / 64);
			plan.xmat.insert(plan.xmat.end(), M, M + 64);
		}
	}
	e->nxgroups = (int)(plan.xmat.size() / 64);

	// ---- unique columns of the implicit solve and their in-patch duplicates (VerticalDynamicsFEM.cpp:1315-1337, 1543-1633)
	for (int p : e->local_patches) {
		const PatchInfo & P = e->patches[p];
		for (int a = 0; a < P.nea; a++) for (int b = 0; b < P.neb; b++)
		for (int ii = 0; ii < TMX_NP; ii++) for (int jj = 0; jj < TMX_NP; jj++) {
			const bool ua = (ii < TMX_NP - 1) || (a == P.nea - 1), ub = (jj < TMX_NP - 1) || (b == P.neb - 1);
			if (!ua || !ub) continue;
			const int i = 1 + a * TMX_NP + ii, j = 1 + b * TMX_NP + jj;
			plan.ucol.push_back(col_of(P, i, j));
			const bool da = (ii == 0 && a > 0), db = (jj == 0 && b > 0);
			plan.udep.push_back(da ? col_of(P, i - 1, j) : -1);
			plan.udep.push_back(db ? col_of(P, i, j - 1) : -1);
			plan.udep.push_back((da && db) ? col_of(P, i - 1, j - 1) : -1);
		}
	}
	e->nunique = (int)plan.ucol.size();
	e->NUS = ((e->nunique + 63) / 64) * 64;
	e->finalized = true;
	if (plan_only(e)) return TMX_OK;

	// ---- device allocation + upload
	HIPCHK(hipSetDevice(e->device));
	size_t bytes = 0;
	const size_t NS = e->NS; const int L = e->L;
	const size_t state_bytes = (size_t)e->cfg.n_instances * e->inst_stride * sizeof(double);
	HIPCHK(hipMalloc((void **)&e->d_state, state_bytes)); bytes += state_bytes;
	HIPCHK(hipMemset(e->d_state, 0, state_bytes));
	// closed-form 3-D metric only if every owned patch delivered factors that reproduce its arrays exactly
	e->metric_closed = !e->sw && !e->h_eta.empty();
	for (int lp : e->local_patches) e->metric_closed = e->metric_closed && e->patches[lp].metric_ok;
	if (e->opt_metric_stored) e->metric_closed = false;
	// per-column element spacing and local hyperviscosity scale of the column's patch
	for (int lp : e->local_patches) {
		const PatchInfo & P = e->patches[lp];
		const double da = (P.da > 0.0) ? P.da : e->cfg.element_delta_a, db = (P.db > 0.0) ? P.db : e->cfg.element_delta_a;
		const double ida = 1.0 / da, idb = 1.0 / db;
		const double nus = (e->cfg.reference_length != 0.0) ? pow(da / e->cfg.reference_length, 3.2) : 1.0;
		for (int c = P.elem_base * TMX_NQ; c < (P.elem_base + P.nea * P.neb) * TMX_NQ; c++) {
			e->h_g2d[G2_IDA * e->NS + c] = ida; e->h_g2d[G2_IDB * e->NS + c] = idb; e->h_g2d[G2_NUS * e->NS + c] = nus;
		}
	}
	if ((r = dev_upload(&e->d_g2d, e->h_g2d, &bytes))) return r;
	if (e->metric_closed) {
		if ((r = dev_upload(&e->d_eta, e->h_eta, &bytes))) return r;
	} else {
		if ((r = dev_upload(&e->d_g3n, e->h_g3n, &bytes))) return r;
		if ((r = dev_upload(&e->d_g3e, e->h_g3e, &bytes))) return r;
	}
	if ((r = dev_upload(&e->d_ops, e->h_ops, &bytes))) return r;
	if (e->nt > 0) {
		if ((r = dev_upload(&e->d_area, e->h_area, &bytes))) return r;
		std::vector<double>().swap(e->h_area);
		HIPCHK(hipMalloc((void **)&e->d_w0, (size_t)(L + 1) * NS * sizeof(double))); bytes += (size_t)(L + 1) * NS * sizeof(double);
	}
	if (e->udiff) {
		// reference state in the layout of a state instance (filled by tmx_set_patch_reference_state)
		const size_t rb = (size_t)e->nslab * NS * sizeof(double);
		HIPCHK(hipMalloc((void **)&e->d_ref, rb)); HIPCHK(hipMemset(e->d_ref, 0, rb)); bytes += rb;
	}
	{
		int nset = 0;
		for (int lp : e->local_patches) nset += e->patches[lp].rayleigh_set ? 1 : 0;
		REQUIRE(nset == 0 || nset == (int)e->local_patches.size(), TMX_ERR_INVALID, "tmx_set_patch_rayleigh was called for %d of %d owned patches", nset, (int)e->local_patches.size());
		e->rayleigh = nset > 0;
		if (e->rayleigh) {
			if ((r = dev_upload(&e->d_ray_nu, e->h_ray_nu, &bytes))) return r;
			if ((r = dev_upload(&e->d_ray_ref, e->h_ray_ref, &bytes))) return r;
		}
		std::vector<double>().swap(e->h_ray_nu); std::vector<double>().swap(e->h_ray_ref);
	}
	HIPCHK(hipMalloc((void **)&e->d_scratch, (size_t)(L + 4) * NS * sizeof(double))); bytes += (size_t)(L + 4) * NS * sizeof(double);
	HIPCHK(hipMemset(e->d_scratch, 0, (size_t)(L + 4) * NS * sizeof(double)));
	if ((r = dev_upload(&e->d_grp_cols, plan.grp_cols, &bytes))) return r;
	{
		// column -> its other copies: the group table inverted, for kernels that average while loading (k_hypervis<PULL>):
		// {the three other members in group order, n | me << 3 | type << 6 | (matrix index + 1) << 8}; all zero / -1: no copies
		std::vector<int> colref((size_t)e->NS * 4, 0);
		for (size_t c = 0; c < (size_t)e->NS; c++) { colref[c * 4] = colref[c * 4 + 1] = colref[c * 4 + 2] = -1; }
		for (int g = 0; g < e->ngroups; g++)
			for (int m = 0; m < plan.grp_n[g]; m++) {
				const int c = plan.grp_cols[(size_t)g * 4 + m];
				if (c < 0 || c >= e->NS) continue;
				int q = 0;
				for (int o = 0; o < 4; o++) if (o != m) colref[(size_t)c * 4 + q++] = (o < plan.grp_n[g]) ? plan.grp_cols[(size_t)g * 4 + o] : -1;
				colref[(size_t)c * 4 + 3] = plan.grp_n[g] | (m << 3) | (((plan.grp_type[g] >> (2 * m)) & 3) << 6) | ((plan.grp_x[g] + 1) << 8);
			}
		if ((r = dev_upload(&e->d_colref, colref, &bytes))) return r;
	}
	{
		// element blocks of the fused hyperviscosity kernel: 5 x 5 inner elements (kernels: TMX_HB_E) per block, patch by patch
		std::vector<int> blocks;
		for (int pp : e->local_patches) {
			const PatchInfo & P = e->patches[pp];
			for (int a0 = 0; a0 < P.nea; a0 += 5) for (int b0 = 0; b0 < P.neb; b0 += 5)
				blocks.insert(blocks.end(), { P.elem_base, P.nea, P.neb, a0, b0 });
		}
		e->n_hvblocks = (int)(blocks.size() / 5);
		if (e->n_hvblocks && (r = dev_upload(&e->d_hvblocks, blocks, &bytes))) return r;
	}
	if ((r = dev_upload(&e->d_grp_n, plan.grp_n, &bytes))) return r;
	if ((r = dev_upload(&e->d_grp_x, plan.grp_x, &bytes))) return r;
	if ((r = dev_upload(&e->d_grp_type, plan.grp_type, &bytes))) return r;
	if ((r = dev_upload(&e->d_xmat, plan.xmat, &bytes))) return r;
	if ((r = dev_upload(&e->d_send_cols, plan.send_cols, &bytes))) return r;
	// Boundary-first launches (north-star: "exchange overlapped with interior-element updates").  A 64-column tile (four
	// elements) is EARLY when it holds a column some other rank needs; the kernels that feed an exchange run on the early
	// tiles first, the pack + grouped send/recv starts on the exchange stream, and the remaining tiles -- three quarters
	// of a 15 x 15-element patch -- are updated while the wire is busy (hv_stage_split).  TMX_NO_SPLIT=1 switches it off.
	e->split_stage = false;
	if (NR > 1 && !plan.send_cols.empty() && !e->opt_no_split) {
		std::vector<char> early(e->NS / 64, 0);
		for (int c : plan.send_cols) early[c / 64] = 1;
		std::vector<int> qe, ql;
		for (int t = 0; t < (int)early.size(); t++) (early[t] ? qe : ql).push_back(t);
		if (!qe.empty() && !ql.empty()) {
			if ((r = dev_upload(&e->d_quads_early, qe, &bytes)) || (r = dev_upload(&e->d_quads_late, ql, &bytes))) return r;
			e->n_quads_early = (int)qe.size(); e->n_quads_late = (int)ql.size();
			e->split_stage = true;
		}
	}
	{
		// wire layout [peer][slab][count_peer]: element t of a peer's segment at nslab*off + slab*count + (t - off)
		std::vector<int> sb(e->nsend), ss(e->nsend), gb(e->nghost), gs(e->nghost);
		for (int rk = 0; rk < NR; rk++) {
			const int so = e->send_rank_off[rk], sc = e->send_rank_off[rk + 1] - so;
			for (int t = 0; t < sc; t++) { sb[so + t] = e->nslab * so + t; ss[so + t] = sc; }
			const int ro = e->recv_rank_off[rk], rc = e->recv_rank_off[rk + 1] - ro;
			for (int t = 0; t < rc; t++) { gb[ro + t] = e->nslab * ro + t; gs[ro + t] = rc; }
		}
		REQUIRE((long long)e->nslab * std::max(e->nsend, e->nghost) < 0x7fffffffLL, TMX_ERR_UNSUPPORTED, "exchange buffer exceeds 2^31 doubles");
		if ((r = dev_upload(&e->d_send_base, sb, &bytes))) return r;
		if ((r = dev_upload(&e->d_send_stride, ss, &bytes))) return r;
		if ((r = dev_upload(&e->d_ghost_base, gb, &bytes))) return r;
		if ((r = dev_upload(&e->d_ghost_stride, gs, &bytes))) return r;
	}
	if ((r = dev_upload(&e->d_ucol, plan.ucol, &bytes))) return r;
	if ((r = dev_upload(&e->d_udep, plan.udep, &bytes))) return r;
	const size_t gb = (size_t)e->nslab * e->nghost_pad * sizeof(double), sb = (size_t)e->nslab * std::max(e->nsend, 1) * sizeof(double);
	HIPCHK(hipMalloc((void **)&e->d_ghost, gb)); HIPCHK(hipMemset(e->d_ghost, 0, gb)); bytes += gb;
	HIPCHK(hipMalloc((void **)&e->d_sendbuf, sb)); bytes += sb;
	const int n = TMX_FTOT * (L + 1);
	// column-solve scratch: per-wavefront U-row streams [NUS/64][n][9 + 1][64] plus a zero page (fused / pair kernels);
	// the split cross-check kernels keep the band matrix [n][9][NUS] and the right-hand sides [n][NUS] in the same buffers
	const size_t zpage = 128 * sizeof(double);      // one 16-byte slot per lane
	// Columns per wavefront of the two-wavefront column kernel.  A pair (assembly + elimination wavefront) is the unit of
	// residency: 512 pairs give every SIMD of the 256 CUs one wavefront, 1024 two.  With 64 columns per pair a grid
	// such as ne30 (760 groups) loads 124 CUs twice and 132 once and the kernel runs as long as the doubly loaded ones;
	// with ceil(nunique / (512 m)) columns per pair (48 at ne30) every SIMD carries the same number of wavefronts.
	{
		const int ngrp64 = (e->nunique + 63) / 64;
		e->vi_cpw = 64;
		// (measured at ne30: 56 columns per wavefront = 64; 48 and 40, which would load every SIMD evenly, are 70 % SLOWER --
		// the kernel is not bound by the doubly loaded CUs; the knob stays for experiments, the default is 64)
		(void)ngrp64;
		if (e->opt_vi_cpw >= 1 && e->opt_vi_cpw <= 64) e->vi_cpw = e->opt_vi_cpw;
		const int ngrp = (e->nunique + e->vi_cpw - 1) / e->vi_cpw + 2;
		e->vi_stream_cols = std::max(e->NUS, ngrp * 64);
		// the lane-group kernel (k_vi_group) streams 16 doubles per row and column instead of 10
		if (e->vi_group == 1 || (e->vi_group < 0 && e->nunique <= e->vi_group_max))
			e->vi_stream_cols = std::max(e->vi_stream_cols, (int)(((size_t)16 * (e->nunique + 4) + 9) / 10) + 64);
	}
	const size_t abb = (size_t)n * (TMX_BW + 1) * e->vi_stream_cols * sizeof(double) + zpage, rb = (size_t)n * e->NUS * sizeof(double);
	HIPCHK(hipMalloc((void **)&e->d_ab, abb)); HIPCHK(hipMemset((char *)e->d_ab + abb - zpage, 0, zpage)); bytes += abb;
	HIPCHK(hipMalloc((void **)&e->d_rhs, rb)); bytes += rb;
	// (behind the flag word: the order slots of the column solve's workgroups, one int per CU -- 16 XCC ids x 256 CU / SH / SE ids; k_vi_pair)
	HIPCHK(hipMalloc((void **)&e->d_flag, (64 + 4096) * sizeof(int))); HIPCHK(hipMemset(e->d_flag, 0, (64 + 4096) * sizeof(int)));
	// Node-unique state layout (tmx_unique.hip): for the configurations all of whose step kernels have the U form -- the
	// nonhydrostatic set with implicit vertical dynamics, no tracers, no uniform diffusion, closed-form metric; a Rayleigh layer (its
	// strength is stored per copy of a node) since round 5: the relaxation at the end of StepAfterSubCycle reads node-unique and writes
	// element-major, and the next step reads that copy by copy ("unique_mixed").  Every other configuration, and every entry point other
	// than tmx_step, works on the element-major layout as before.
	if (e->u.option != 0 && !e->sw && !e->fully_explicit && !e->udiff && e->nt == 0 && (!e->rayleigh || (e->u.mixed_option && e->u.tile_shape == 0)) && e->metric_closed &&
	    !e->hvis_pull && !e->hvis_block && !e->use_graph && !e->use_mfma && e->vi_mode == 0) {
		UniquePlanInput in = { &plan.grp_cols, &plan.grp_n, &plan.grp_x, &plan.grp_type, &plan.send_cols };
		if ((r = tmxu_build(e, in, &bytes))) return r;
		if (e->u.NTS > NS) {      // the block thread order pads: the level-parallel kernels' hand-over arrays are indexed by thread
			hipFree(e->d_scratch); e->d_scratch = nullptr;
			HIPCHK(hipMalloc((void **)&e->d_scratch, (size_t)(L + 4) * e->u.NTS * sizeof(double))); bytes += (size_t)(L + 4) * (e->u.NTS - NS) * sizeof(double);
			HIPCHK(hipMemset(e->d_scratch, 0, (size_t)(L + 4) * e->u.NTS * sizeof(double)));
		}
	}
	e->hbm_bytes = bytes;
	// host staging no longer needed
	std::vector<double>().swap(e->h_g2d); std::vector<double>().swap(e->h_g3n); std::vector<double>().swap(e->h_g3e);
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// state transfer

static int check_state_args(tmx_engine * e, int patch, int instance, bool read_only = false) {
	REQUIRE(e && e->finalized && !plan_only(e), TMX_ERR_INVALID, "engine not finalized");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches && e->patches[patch].owner == e->cfg.rank, TMX_ERR_INVALID, "patch %d is not local", patch);
	REQUIRE(instance >= 0 && instance < e->cfg.n_instances, TMX_ERR_INVALID, "instance %d out of range", instance);
	return settle_instance(e, instance, read_only);
}

// ---------------------------------------------------------------------------------------------
// Host <-> slab transposition at the ABI boundary.  The reference keeps a column contiguous ([var][i][j][k]); the
// device keeps a level contiguous over columns ([slab][column]).  The staging buffer is walked in tiles of 64
// device columns (= 4 elements: 64 x L doubles per variable stay in L1 while all levels of the tile are moved) and
// the tiles are dealt to a few host threads -- the naive column-at-a-time loop read one double per cache line
// (measured at ne30 L30: download 45 -> see DESIGN.md section 5).
static void host_parallel(int ntiles, const std::function<void(int, int)> & fn) {
	int nthr = (int)std::thread::hardware_concurrency();
	if (nthr > 8) nthr = 8;
	if (nthr < 1 || ntiles < 4 * nthr) { fn(0, ntiles); return; }
	std::vector<std::thread> th;
	const int per = (ntiles + nthr - 1) / nthr;
	for (int t = 0; t < nthr; t++) {
		const int a = t * per, b = std::min(ntiles, a + per);
		if (a < b) th.emplace_back(fn, a, b);
	}
	for (auto & x : th) x.join();
}

// pinned staging buffer of the engine (grown on demand): no per-call allocation / zero fill, DMA at full PCIe rate
static double * stage_buffer(tmx_engine * e, size_t n) {
	if (e->h_stage_n < n) {
		if (e->h_stage) hipHostFree(e->h_stage);
		e->h_stage = nullptr; e->h_stage_n = 0;
		if (hipHostMalloc((void **)&e->h_stage, n * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
		e->h_stage_n = n;
	}
	return e->h_stage;
}

// host node offset (i * nb + j) of every device column of the patch, device order
static const std::vector<int> & host_offsets(tmx_engine * e, int patch) {
	PatchInfo & P = e->patches[patch];
	if (P.hoff.empty()) {
		const int c0 = P.elem_base * TMX_NQ;
		P.hoff.assign((size_t)P.nea * P.neb * TMX_NQ, 0);
		for (int i = 1; i < P.na - 1; i++) for (int j = 1; j < P.nb - 1; j++) P.hoff[col_of(P, i, j) - c0] = i * P.nb + j;
	}
	return P.hoff;
}

// one variable: host[(var_off + hoff[c]) * nlev + k]  <->  buf[(slab0 + k) * ncp + c], columns [ca, cb)
static inline void xpose_to_buf(const double * host, size_t var_off, int nlev, double * buf, int slab0, int ncp, const int * hoff, int ca, int cb) {
	for (int k = 0; k < nlev; k++) {
		double * row = buf + (size_t)(slab0 + k) * ncp;
		for (int c = ca; c < cb; c++) row[c] = host[(var_off + hoff[c]) * nlev + k];
	}
}
static inline void xpose_to_host(double * host, size_t var_off, int nlev, const double * buf, int slab0, int ncp, const int * hoff, int ca, int cb) {
	for (int k = 0; k < nlev; k++) {
		const double * row = buf + (size_t)(slab0 + k) * ncp;
		for (int c = ca; c < cb; c++) host[(var_off + hoff[c]) * nlev + k] = row[c];
	}
}

// GridPatch::GetReferenceState / GetReferenceTracers -> the reference "instance" of the uniform diffusion
extern "C" int tmx_set_patch_reference_state(tmx_engine * e, int patch, const double * ref_node, const double * ref_redge, const double * ref_tracers) {
	int r = check_state_args(e, patch, 0);
	if (r) return r;
	if (!e->d_ref) {      // engines without uniform diffusion keep a reference state only if the caller provides one (tmx_interp_state)
		const size_t rb = (size_t)e->nslab * e->NS * sizeof(double);
		HIPCHK(hipMalloc((void **)&e->d_ref, rb)); HIPCHK(hipMemset(e->d_ref, 0, rb)); e->hbm_bytes += rb;
	}
	REQUIRE(ref_node && ref_redge && (ref_tracers || e->nt == 0), TMX_ERR_INVALID, "tmx_set_patch_reference_state: null array");
	PatchInfo & P = e->patches[patch];
	const int L = e->L, na = P.na, nb = P.nb;
	const int ncp = P.nea * P.neb * TMX_NQ, c0 = P.elem_base * TMX_NQ;
	std::vector<double> buf((size_t)e->nslab * ncp);
	const int nodevar[4] = { 0, 1, 2, 4 };
	for (int i = 1; i < na - 1; i++) for (int j = 1; j < nb - 1; j++) {
		const int c = col_of(P, i, j) - c0;
		for (int v = 0; v < 4; v++) for (int k = 0; k < L; k++)
			buf[(size_t)(v * L + k) * ncp + c] = ref_node[(((size_t)nodevar[v] * na + i) * nb + j) * L + k];
		for (int k = 0; k <= L; k++)
			buf[(size_t)(4 * L + k) * ncp + c] = ref_redge[(((size_t)3 * na + i) * nb + j) * (L + 1) + k];
		for (int v = 0; v < e->nt; v++) for (int k = 0; k < L; k++)
			buf[(size_t)TMX_SLAB_Q(L, v, k) * ncp + c] = ref_tracers[(((size_t)v * na + i) * nb + j) * L + k];
	}
	HIPCHK(hipStreamSynchronize(e->stream));
	HIPCHK(hipMemcpy2D(e->d_ref + c0, (size_t)e->NS * sizeof(double), buf.data(), (size_t)ncp * sizeof(double),
		(size_t)ncp * sizeof(double), e->nslab, hipMemcpyHostToDevice));
	P.ref_set = true;
	e->refd_valid = false;
	return TMX_OK;
}

int check_reference_state(tmx_engine * e) {
	if (!e->udiff) return TMX_OK;
	for (int lp : e->local_patches)
		REQUIRE(e->patches[lp].ref_set, TMX_ERR_INVALID, "uniform diffusion: tmx_set_patch_reference_state was not called for patch %d", lp);
	return TMX_OK;
}

extern "C" int tmx_upload_state(tmx_engine * e, int patch, int instance, const double * node, const double * redge) {
	int r = check_state_args(e, patch, instance);
	if (r) return r;
	REQUIRE(node && (redge || e->sw), TMX_ERR_INVALID, "tmx_upload_state: null array");
	const PatchInfo & P = e->patches[patch];
	const int L = e->L, na = P.na, nb = P.nb;
	const int ncp = P.nea * P.neb * TMX_NQ, c0 = P.elem_base * TMX_NQ;
	const int nstate = 5 * L + 1;          // state slabs; tracer slabs are moved by tmx_upload_tracers
	HIPCHK(hipStreamSynchronize(e->stream));      // the staging buffer may still feed an earlier copy
	double * bufp = stage_buffer(e, (size_t)nstate * ncp);
	REQUIRE(bufp, TMX_ERR_DEVICE, "cannot allocate the pinned staging buffer");
	struct { double * p; double * data() const { return p; } } buf = { bufp };
	const int nodevar[4] = { 0, 1, 2, 4 };
	const int * hoff = host_offsets(e, patch).data();
	const size_t nn = (size_t)na * nb;
	const bool sw = e->sw;
	host_parallel((ncp + 63) / 64, [&](int ta, int tb) {
		for (int t = ta; t < tb; t++) {
			const int ca = t * 64, cb = std::min(ncp, ca + 64);
			if (sw) {        // node is [3][na][nb][1]: U, V, H -> slabs U, V, "rho*theta"
				for (int v = 0; v < 3; v++) xpose_to_buf(node, (size_t)v * nn, 1, buf.data(), v, ncp, hoff, ca, cb);
				continue;
			}
			for (int v = 0; v < 4; v++) xpose_to_buf(node, (size_t)nodevar[v] * nn, L, buf.data(), v * L, ncp, hoff, ca, cb);
			xpose_to_buf(redge, (size_t)3 * nn, L + 1, buf.data(), 4 * L, ncp, hoff, ca, cb);
		}
	});
	HIPCHK(hipStreamSynchronize(e->stream));
	HIPCHK(hipMemcpy2D(e->d_state + (size_t)instance * e->inst_stride + c0, (size_t)e->NS * sizeof(double),
		buf.data(), (size_t)ncp * sizeof(double), (size_t)ncp * sizeof(double), nstate, hipMemcpyHostToDevice));
	if (!sw) {
		// surface slots: interface-level-0 entries of rho (REdge component 4) and rho*theta (component 2), see surface_copy
		std::vector<double> ss((size_t)2 * ncp);
		for (int c = 0; c < ncp; c++) {
			ss[c] = redge[((size_t)4 * nn + hoff[c]) * (L + 1)];
			ss[(size_t)ncp + c] = redge[((size_t)2 * nn + hoff[c]) * (L + 1)];
		}
		HIPCHK(hipMemcpy2D(e->d_state + (size_t)instance * e->inst_stride + (size_t)e->nslab * e->NS + c0, (size_t)e->NS * sizeof(double),
			ss.data(), (size_t)ncp * sizeof(double), (size_t)ncp * sizeof(double), 2, hipMemcpyHostToDevice));
	}
	return TMX_OK;
}

// GridPatch::GetDataTracers(instance) [nt][na][nb][L] <-> tracer slabs
extern "C" int tmx_upload_tracers(tmx_engine * e, int patch, int instance, const double * tracers) {
	int r = check_state_args(e, patch, instance);
	if (r) return r;
	REQUIRE(e->nt > 0 && tracers, TMX_ERR_INVALID, "tmx_upload_tracers: engine has no tracers / null array");
	const PatchInfo & P = e->patches[patch];
	const int L = e->L, na = P.na, nb = P.nb, nq = e->nt * L;
	const int ncp = P.nea * P.neb * TMX_NQ, c0 = P.elem_base * TMX_NQ;
	std::vector<double> buf((size_t)nq * ncp);
	for (int i = 1; i < na - 1; i++) for (int j = 1; j < nb - 1; j++) {
		const int c = col_of(P, i, j) - c0;
		for (int v = 0; v < e->nt; v++) for (int k = 0; k < L; k++)
			buf[(size_t)(v * L + k) * ncp + c] = tracers[(((size_t)v * na + i) * nb + j) * L + k];
	}
	HIPCHK(hipStreamSynchronize(e->stream));
	HIPCHK(hipMemcpy2D(e->d_state + (size_t)instance * e->inst_stride + (size_t)(5 * L + 1) * e->NS + c0, (size_t)e->NS * sizeof(double),
		buf.data(), (size_t)ncp * sizeof(double), (size_t)ncp * sizeof(double), nq, hipMemcpyHostToDevice));
	return TMX_OK;
}

extern "C" int tmx_download_tracers(tmx_engine * e, int patch, int instance, double * tracers) {
	int r = check_state_args(e, patch, instance, true);
	if (r) return r;
	REQUIRE(e->nt > 0 && tracers, TMX_ERR_INVALID, "tmx_download_tracers: engine has no tracers / null array");
	const PatchInfo & P = e->patches[patch];
	const int L = e->L, na = P.na, nb = P.nb, nq = e->nt * L;
	const int ncp = P.nea * P.neb * TMX_NQ, c0 = P.elem_base * TMX_NQ;
	std::vector<double> buf((size_t)nq * ncp);
	HIPCHK(hipStreamSynchronize(e->stream));
	HIPCHK(hipMemcpy2D(buf.data(), (size_t)ncp * sizeof(double), e->d_state + (size_t)instance * e->inst_stride + (size_t)(5 * L + 1) * e->NS + c0,
		(size_t)e->NS * sizeof(double), (size_t)ncp * sizeof(double), nq, hipMemcpyDeviceToHost));
	for (int i = 1; i < na - 1; i++) for (int j = 1; j < nb - 1; j++) {
		const int c = col_of(P, i, j) - c0;
		for (int v = 0; v < e->nt; v++) for (int k = 0; k < L; k++)
			tracers[(((size_t)v * na + i) * nb + j) * L + k] = buf[(size_t)(v * L + k) * ncp + c];
	}
	return TMX_OK;
}

extern "C" int tmx_download_state(tmx_engine * e, int patch, int instance, double * node, double * redge) {
	int r = check_state_args(e, patch, instance, true);
	if (r) return r;
	REQUIRE(node && (redge || e->sw), TMX_ERR_INVALID, "tmx_download_state: null array");
	const PatchInfo & P = e->patches[patch];
	const int L = e->L, na = P.na, nb = P.nb;
	const int ncp = P.nea * P.neb * TMX_NQ, c0 = P.elem_base * TMX_NQ;
	const int nstate = 5 * L + 1;
	double * bufp = stage_buffer(e, (size_t)nstate * ncp);
	REQUIRE(bufp, TMX_ERR_DEVICE, "cannot allocate the pinned staging buffer");
	struct { double * p; double * data() const { return p; } } buf = { bufp };
	HIPCHK(hipStreamSynchronize(e->stream));
	HIPCHK(hipMemcpy2D(buf.data(), (size_t)ncp * sizeof(double), e->d_state + (size_t)instance * e->inst_stride + c0,
		(size_t)e->NS * sizeof(double), (size_t)ncp * sizeof(double), nstate, hipMemcpyDeviceToHost));
	const int nodevar[4] = { 0, 1, 2, 4 };
	const double * ops = e->h_ops.data();
	auto opc = [&](int op, int k, int off) { return ops[((size_t)op * (L + 1) + k) * TMX_OPW + (off + 2)]; };
	const int * hoff = host_offsets(e, patch).data();
	const size_t nn = (size_t)na * nb;
	const bool sw = e->sw;
	if (!sw && e->track_surface) {
		// tracked surface slots back into the interface-level-0 entries of rho and rho*theta (see surface_copy)
		std::vector<double> ss((size_t)2 * ncp);
		HIPCHK(hipMemcpy2D(ss.data(), (size_t)ncp * sizeof(double), e->d_state + (size_t)instance * e->inst_stride + (size_t)e->nslab * e->NS + c0,
			(size_t)e->NS * sizeof(double), (size_t)ncp * sizeof(double), 2, hipMemcpyDeviceToHost));
		for (int c = 0; c < ncp; c++) {
			redge[((size_t)4 * nn + hoff[c]) * (L + 1)] = ss[c];
			redge[((size_t)2 * nn + hoff[c]) * (L + 1)] = ss[(size_t)ncp + c];
		}
	}
	host_parallel((ncp + 63) / 64, [&](int ta, int tb) {
		for (int t = ta; t < tb; t++) {
			const int ca = t * 64, cb = std::min(ncp, ca + 64);
			if (sw) {
				for (int v = 0; v < 3; v++) xpose_to_host(node, (size_t)v * nn, 1, buf.data(), v, ncp, hoff, ca, cb);
				continue;
			}
			for (int v = 0; v < 4; v++) xpose_to_host(node, (size_t)nodevar[v] * nn, L, buf.data(), v * L, ncp, hoff, ca, cb);
			xpose_to_host(redge, (size_t)3 * nn, L + 1, buf.data(), 4 * L, ncp, hoff, ca, cb);
			// derived slots the reference keeps beside the prognostic ones (HorizontalDynamicsFEM.cpp:817-831):
			// W on levels, U and V on interfaces, from the columns just written (contiguous on the host)
			for (int c = ca; c < cb; c++) {
				const size_t ho = hoff[c];
				const double * colU = node + ((size_t)0 * nn + ho) * L, * colV = node + ((size_t)1 * nn + ho) * L;
				const double * colW = redge + ((size_t)3 * nn + ho) * (L + 1);
				double * wn = node + ((size_t)3 * nn + ho) * L;
				double * ue_ = redge + ((size_t)0 * nn + ho) * (L + 1), * ve_ = redge + ((size_t)1 * nn + ho) * (L + 1);
				for (int k = 0; k < L; k++) {
					double w = 0.0;
					w += opc(TMX_OP_INTERP_REDGE_TO_NODE, k, 0) * colW[k];
					w += opc(TMX_OP_INTERP_REDGE_TO_NODE, k, 1) * colW[k + 1];
					wn[k] = w;
				}
				for (int k = 0; k <= L; k++) {
					double ue = 0.0, ve = 0.0;
					for (int off = -2; off <= 1; off++) {
						const int l = k + off;
						if (l < 0 || l >= L) continue;
						const double cc = opc(TMX_OP_INTERP_NODE_TO_REDGE, k, off);
						if (cc == 0.0) continue;
						ue += cc * colU[l]; ve += cc * colV[l];
					}
					ue_[k] = ue; ve_[k] = ve;
				}
			}
		}
	});
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// Restart image (second half): the bytes of GridPatch::GetDataContainerActiveState() (GridPatch.cpp:359-361,387),
// which is what OutputManagerComposite::Output writes per patch (OutputManagerComposite.cpp:331-349) and what a restart reads
// back.  The transposition from the slab layout runs on the device; the host sees ONE contiguous copy per patch.

static size_t active_state_doubles(const tmx_engine * e, const PatchInfo & P) {
	const size_t nn = (size_t)P.na * P.nb;
	return 5 * nn * e->L + 5 * nn * (e->L + 1) + (size_t)e->nt * nn * e->L;
}

extern "C" long long tmx_active_state_bytes(tmx_engine * e, int patch) {
	if (!e || patch < 0 || patch >= e->cfg.n_patches || e->sw) return -1;
	return (long long)(sizeof(double) * (1 + active_state_doubles(e, e->patches[patch])));      // 8-byte DataArray1D<int> chunk in front
}

static int image_buffer(tmx_engine * e, size_t n) {
	if (e->image_n < n) {
		if (e->d_image) { hipFree(e->d_image); e->hbm_bytes -= e->image_n * sizeof(double); }
		e->d_image = nullptr; e->image_n = 0;
		HIPCHK(hipMalloc((void **)&e->d_image, n * sizeof(double)));
		e->image_n = n; e->hbm_bytes += n * sizeof(double);
	}
	return TMX_OK;
}

extern "C" int tmx_pack_active_state(tmx_engine * e, int patch, int instance, void * dst, size_t dst_bytes) {
	int r = check_state_args(e, patch, instance, true);
	if (r) return r;
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "restart image with the shallow-water equation set is not supported");
	const PatchInfo & P = e->patches[patch];
	const size_t nd = active_state_doubles(e, P);
	REQUIRE(dst && dst_bytes == sizeof(double) * (1 + nd), TMX_ERR_INVALID, "tmx_pack_active_state: %zu bytes given, the container of patch %d has %zu",
		dst_bytes, patch, sizeof(double) * (1 + nd));
	if ((r = image_buffer(e, nd))) return r;
	HIPCHK(hipMemsetAsync(e->d_image, 0, nd * sizeof(double), e->stream));       // halo ring and the entries nothing on the path reads
	tmxk_active_state_image(e, make_params(e), true, P.elem_base * TMX_NQ, P.nea * P.neb * TMX_NQ, P.na, P.nb, P.neb,
		e->d_state + (size_t)instance * e->inst_stride, e->track_surface ? e->d_state + (size_t)instance * e->inst_stride + (size_t)e->nslab * e->NS : nullptr, e->d_image);
	if ((r = launch_check("pack_active_state"))) return r;
	unsigned char * out = (unsigned char *)dst;
	HIPCHK(hipMemcpyAsync(out + sizeof(double), e->d_image, nd * sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	memset(out, 0, sizeof(double));
	const int ix = patch; memcpy(out, &ix, sizeof(int));       // m_iActiveStatePatchIx (GridPatch.cpp:359)
	return TMX_OK;
}

extern "C" int tmx_unpack_active_state(tmx_engine * e, int patch, int instance, const void * src, size_t src_bytes) {
	int r = check_state_args(e, patch, instance);
	if (r) return r;
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "restart image with the shallow-water equation set is not supported");
	const PatchInfo & P = e->patches[patch];
	const size_t nd = active_state_doubles(e, P);
	REQUIRE(src && src_bytes == sizeof(double) * (1 + nd), TMX_ERR_INVALID, "tmx_unpack_active_state: %zu bytes given, the container of patch %d has %zu",
		src_bytes, patch, sizeof(double) * (1 + nd));
	int ix = -1; memcpy(&ix, src, sizeof(int));
	REQUIRE(ix == patch, TMX_ERR_INVALID, "tmx_unpack_active_state: the image is of patch %d, not %d", ix, patch);
	if ((r = image_buffer(e, nd))) return r;
	HIPCHK(hipStreamSynchronize(e->stream));
	HIPCHK(hipMemcpyAsync(e->d_image, (const unsigned char *)src + sizeof(double), nd * sizeof(double), hipMemcpyHostToDevice, e->stream));
	tmxk_active_state_image(e, make_params(e), false, P.elem_base * TMX_NQ, P.nea * P.neb * TMX_NQ, P.na, P.nb, P.neb,
		e->d_state + (size_t)instance * e->inst_stride, e->d_state + (size_t)instance * e->inst_stride + (size_t)e->nslab * e->NS, e->d_image);
	if ((r = launch_check("unpack_active_state"))) return r;
	HIPCHK(hipStreamSynchronize(e->stream));      // the caller may reuse src
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU

extern "C" int tmx_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

extern "C" int tmx_comm_unique_id(unsigned char id[128]) {
	int r = load_rccl();
	if (r) return r;
	nccl_uid u;
	NCCLCHK(g_nccl.GetUniqueId(&u));
	memcpy(id, u.internal, 128);
	return TMX_OK;
}

extern "C" int tmx_comm_init(tmx_engine * e, const unsigned char id[128]) {
	if (e && !e->xstream && e->cfg.device != -2 && !e->opt_no_exchange_overlap) {
		// second stream + events for the exchange / interior-DSS overlap
		if (hipStreamCreateWithFlags(&e->xstream, hipStreamNonBlocking) != hipSuccess) e->xstream = nullptr;
		else { hipEventCreateWithFlags(&e->ev_pack, hipEventDisableTiming); hipEventCreateWithFlags(&e->ev_recv, hipEventDisableTiming); }
	}
	REQUIRE(e && id && !plan_only(e), TMX_ERR_INVALID, "tmx_comm_init: bad argument");
	int r = load_rccl();
	if (r) return r;
	HIPCHK(hipSetDevice(e->device));
	nccl_uid u;
	memcpy(u.internal, id, 128);
	{
		const int nr = g_nccl.CommInitRank(&e->comm, e->cfg.n_ranks, u, e->cfg.rank);
		if (nr != 0) {
			// the usual cause: several ranks of one node on the same HIP device (RCCL: "Duplicate GPU detected")
			tmx_set_error("ncclCommInitRank failed (RCCL error %d) for rank %d of %d on HIP device %d: every rank needs a GPU of its own "
				"-- set tmx_config.device per rank (tmx_device_count(), node-local rank) or restrict HIP_VISIBLE_DEVICES per rank",
				nr, e->cfg.rank, e->cfg.n_ranks, e->device);
			e->comm = nullptr;
			return TMX_ERR_DEVICE;
		}
	}
	return TMX_OK;
}

// Peer-to-peer halo transport: instead of RCCL send/recv, every rank maps the ghost buffers of its neighbour ranks (HIP IPC)
// and its gather kernel writes the boundary columns straight into them; an arrival counter per (parity, source rank) in the
// same block orders the two sides.  Set-up: every rank exports a blob, the caller all-gathers the blobs (any transport: the
// reference's MPI, torch.distributed) and hands all of them to tmx_halo_p2p_connect; before tmx_destroy the caller lets every
// rank finish (a barrier), since neighbours write into this rank's block.  The reference's exchange this replaces:
// Connectivity.cpp:928-1120 (ExchangeBuffer pack / MPI_Isend / MPI_Irecv / unpack), Grid.cpp:627-685.
struct p2p_blob_head { hipIpcMemHandle_t handle; unsigned long long ghost_doubles; int n_ranks, rank; char pci[16]; };      // pci: bus id of the exporting device

extern "C" int tmx_halo_p2p_blob_bytes(tmx_engine * e) {
	if (!e || !e->finalized) return -1;
	return (int)(sizeof(p2p_blob_head) + sizeof(int) * (e->cfg.n_ranks + 1));
}

extern "C" int tmx_halo_p2p_export(tmx_engine * e, unsigned char * blob) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(blob, TMX_ERR_INVALID, "tmx_halo_p2p_export: null blob");
	REQUIRE(!plan_only(e) && e->cfg.n_ranks > 1 && !e->lb, TMX_ERR_INVALID, "tmx_halo_p2p_export needs a device engine of a multi-rank layout");
	HIPCHK(hipSetDevice(e->device));
	const size_t gd = (size_t)e->nslab * e->nghost_pad;
	if (!e->p2p_block) {
		e->p2p_block_bytes = p2p_header_bytes(e->cfg.n_ranks) + 2 * gd * sizeof(double);
		// fine-grained: writes of other devices and the counters are visible without a cache flush on this side
		HIPCHK(hipExtMallocWithFlags(&e->p2p_block, e->p2p_block_bytes, hipDeviceMallocFinegrained));
		HIPCHK(hipMemset(e->p2p_block, 0, e->p2p_block_bytes));
		HIPCHK(hipDeviceSynchronize());
		e->hbm_bytes += e->p2p_block_bytes;
	}
	p2p_blob_head h;
	memset(&h, 0, sizeof(h));
	HIPCHK(hipIpcGetMemHandle(&h.handle, e->p2p_block));
	h.ghost_doubles = gd; h.n_ranks = e->cfg.n_ranks; h.rank = e->cfg.rank;
	if (hipDeviceGetPCIBusId(h.pci, (int)sizeof(h.pci), e->device) != hipSuccess) { (void)hipGetLastError(); h.pci[0] = 0; }
	h.pci[sizeof(h.pci) - 1] = 0;
	memcpy(blob, &h, sizeof(h));
	memcpy(blob + sizeof(h), e->recv_rank_off.data(), sizeof(int) * (e->cfg.n_ranks + 1));
	return TMX_OK;
}

extern "C" int tmx_halo_p2p_connect(tmx_engine * e, const unsigned char * blobs) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(blobs && e->p2p_block, TMX_ERR_INVALID, "tmx_halo_p2p_connect: call tmx_halo_p2p_export first and pass the blobs of all ranks");
	REQUIRE(!e->p2p_connected, TMX_ERR_INVALID, "tmx_halo_p2p_connect: already connected");
	const int NR = e->cfg.n_ranks, me = e->cfg.rank;
	const size_t bb = (size_t)tmx_halo_p2p_blob_bytes(e), hb = p2p_header_bytes(NR);
	HIPCHK(hipSetDevice(e->device));
	e->p2p_peer.assign(NR, nullptr);
	std::vector<double *> dst(2 * (size_t)NR, nullptr);
	std::vector<unsigned long long *> flg(2 * (size_t)NR, nullptr);
	std::vector<int> peers;
	for (int rk = 0; rk < NR; rk++) {
		p2p_blob_head h;
		memcpy(&h, blobs + bb * rk, sizeof(h));
		const int * roff = (const int *)(blobs + bb * rk + sizeof(h));
		REQUIRE(h.n_ranks == NR && h.rank == rk, TMX_ERR_INVALID, "tmx_halo_p2p_connect: blob %d is from rank %d of %d", rk, h.rank, h.n_ranks);
		const int ns = e->send_rank_off[rk + 1] - e->send_rank_off[rk], nr = e->recv_rank_off[rk + 1] - e->recv_rank_off[rk];
		const int their = roff[me + 1] - roff[me];
		REQUIRE(rk == me || their == ns, TMX_ERR_INVALID, "rank %d sends %d columns to rank %d which expects %d", me, ns, rk, their);
		REQUIRE(rk == me || (ns > 0) == (nr > 0), TMX_ERR_INVALID, "halo between ranks %d and %d is one-sided (%d out, %d in)", me, rk, ns, nr);
		if (rk == me || ns == 0) continue;
		// a neighbour on another device of this node: the gather kernel will write into its memory, which needs peer access
		// (where this process can see that device at all; otherwise the mapping below is the test)
		if (h.pci[0]) {
			int their_dev = -1;
			if (hipDeviceGetByPCIBusId(&their_dev, h.pci) == hipSuccess && their_dev >= 0 && their_dev != e->device) {
				int can = 0;
				if (hipDeviceCanAccessPeer(&can, e->device, their_dev) != hipSuccess) { (void)hipGetLastError(); can = 0; }
				REQUIRE(can, TMX_ERR_COMM, "device %d (this rank) has no peer access to device %d (%s, rank %d): no peer-to-peer halo between them", e->device, their_dev, h.pci, rk);
			} else (void)hipGetLastError();
		}
		void * base = nullptr;
		const hipError_t oe = hipIpcOpenMemHandle(&base, h.handle, hipIpcMemLazyEnablePeerAccess);
		REQUIRE(oe == hipSuccess, TMX_ERR_COMM, "hipIpcOpenMemHandle of rank %d's halo block failed: %s (ranks of one process cannot map each other; "
			"HSA_ENABLE_IPC_MODE_LEGACY=0 is needed where the driver only supports dmabuf IPC)", rk, hipGetErrorString(oe));
		e->p2p_peer[rk] = base;
		peers.push_back(rk);
		for (int b = 0; b < 2; b++) {
			dst[(size_t)b * NR + rk] = (double *)((char *)base + hb) + (size_t)b * h.ghost_doubles + (size_t)e->nslab * roff[me];
			flg[(size_t)b * NR + rk] = (unsigned long long *)base + (size_t)b * NR + me;
		}
	}
	std::vector<int> sp(e->nsend), sw(e->nsend);
	for (int rk = 0; rk < NR; rk++)
		for (int t = e->send_rank_off[rk]; t < e->send_rank_off[rk + 1]; t++) { sp[t] = rk; sw[t] = t - e->send_rank_off[rk]; }
	size_t bytes = 0;
	if ((r = dev_upload(&e->d_p2p_dst, dst, &bytes)) || (r = dev_upload(&e->d_p2p_flag, flg, &bytes)) || (r = dev_upload(&e->d_send_peer, sp, &bytes))
		|| (r = dev_upload(&e->d_send_within, sw, &bytes)) || (r = dev_upload(&e->d_p2p_peers, peers, &bytes))) return r;
	e->hbm_bytes += bytes;
	e->p2p_npeers = (int)peers.size();
	if (!e->xstream && !e->opt_no_exchange_overlap) {
		if (hipStreamCreateWithFlags(&e->xstream, hipStreamNonBlocking) != hipSuccess) e->xstream = nullptr;
		else { hipEventCreateWithFlags(&e->ev_pack, hipEventDisableTiming); hipEventCreateWithFlags(&e->ev_recv, hipEventDisableTiming); }
	}
	e->d_ghost_own = e->d_ghost;
	e->p2p = true;
	e->p2p_connected = true;
	return TMX_OK;
}

// Switch between the two transports of a connected engine (both set up: tmx_comm_init and tmx_halo_p2p_connect), e.g. to time
// them against each other at start-up.  Every rank must switch at the same point of the program, with no exchange in flight.
extern "C" int tmx_halo_p2p_enable(tmx_engine * e, int on) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(e->p2p_connected, TMX_ERR_INVALID, "tmx_halo_p2p_enable: tmx_halo_p2p_connect first");
	REQUIRE(on || e->comm, TMX_ERR_INVALID, "tmx_halo_p2p_enable(0): no RCCL communicator to fall back to (tmx_comm_init)");
	HIPCHK(hipStreamSynchronize(e->stream));
	if (e->xstream) HIPCHK(hipStreamSynchronize(e->xstream));
	e->p2p = on != 0;
	if (!e->p2p) e->d_ghost = e->d_ghost_own;
	return TMX_OK;
}

// After a failed exchange (tmx_sync returned TMX_ERR_COMM) the ranks' exchange counters disagree.  Once EVERY rank has
// returned from tmx_sync and the caller has put a barrier behind that (nobody is inside an exchange, nobody writes into a
// neighbour's block), each rank calls this: its arrival counters and its own exchange count start from zero again.  A second
// barrier, then the ranks may go on -- with either transport -- from a state they upload again.
extern "C" int tmx_halo_p2p_reset(tmx_engine * e) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(e->p2p_connected, TMX_ERR_INVALID, "tmx_halo_p2p_reset: tmx_halo_p2p_connect first");
	HIPCHK(hipStreamSynchronize(e->stream));
	if (e->xstream) HIPCHK(hipStreamSynchronize(e->xstream));
	HIPCHK(hipMemset(e->p2p_block, 0, p2p_header_bytes(e->cfg.n_ranks)));
	HIPCHK(hipMemset(e->d_flag, 0, sizeof(int)));
	HIPCHK(hipDeviceSynchronize());
	e->p2p_seq = 0;
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// introspection

// Options that change how (never what) the engine computes -- the one exception, "contraction_mfma", is named as such -- by name.
// The library reads NO environment variable on its own: a stray TMX_* in a job script cannot change a run.  Test and bench plumbing
// that wants the historical variables calls tmx_options_from_environment, which turns them into these options, prints ONE line
// naming what it applied, and leaves them queryable (tmx_get_option, tmx_options_report).
// kind 0: any time, 1: before tmx_finalize; +2: selects an archived experiment or a cross-check kernel that only the experiments flavour of the
// library holds (-DTMX_EXPERIMENTS, libtempest_mi355x_exp.so): the production library refuses every value but the default
struct OptionDef { const char * name; const char * env; int kind; const char * help; };
static int * option_slot(tmx_engine * e, const std::string & n) {
#define OPT(NAME_, FIELD_) if (n == NAME_) return &(FIELD_)
	OPT("unique_layout", e->u.option); OPT("unique_tile_shape", e->u.tile_shape); OPT("unique_blocks", e->u.block_option); OPT("unique_xcd_order", e->u.xcd_order); OPT("unique_mixed", e->u.mixed_option); OPT("unique_prefix", e->u.prefix_option);
	OPT("share_copies", e->share_copies); OPT("xcd_vertical", e->xcd_vertical);
	OPT("vi_carry", e->vi_carry); OPT("vi_pair", e->vi_pair); OPT("vi_group", e->vi_group); OPT("vi_group_max", e->vi_group_max);
	OPT("vi_pair_workgroup", e->vi_pair_wg); OPT("vi_producers", e->vi_producers); OPT("vi_ring_depth", e->vi_ring_depth); OPT("vi_split_back", e->vi_split_back); OPT("vi_back_sub", e->vi_back_sub);
	OPT("vi_stagger", e->vi_stagger); OPT("vi_split_kernels", e->vi_mode); OPT("vi_sparse", e->opt_vi_sparse); OPT("vi_columns_per_wavefront", e->opt_vi_cpw);
	OPT("contraction_mfma", e->use_mfma); OPT("step_graph", e->use_graph); OPT("p2p_timeout_s", e->p2p_timeout_s);
	OPT("hvis_pull", e->hvis_pull); OPT("hvis_block", e->hvis_block);
	OPT("split_stage_off", e->opt_no_split); OPT("metric_stored", e->opt_metric_stored); OPT("tracer_lincomb_pass", e->opt_tracer_lincomb_pass);
	OPT("udv_separate", e->opt_udv_separate); OPT("vx_fused", e->opt_vx_fused); OPT("debug_skip_exchange", e->opt_skip_exchange);
	OPT("exchange_overlap_off", e->opt_no_exchange_overlap); OPT("kessler_column", e->opt_kessler_column);
	OPT("vt_column", e->opt_vt_column); OPT("vt_explicit_v1", e->opt_vt_explicit_v1); OPT("vt_explicit_walk", e->opt_vt_walk); OPT("vite_walk", e->opt_vite_walk); OPT("vx_walk", e->opt_vx_walk); OPT("vt_lanes", e->opt_vt_lanes); OPT("vt_lw8", e->opt_vt_lw8);
	OPT("vt_row_lanes", e->opt_vt_nr); OPT("vt_rows", e->opt_vt_rows); OPT("h_walk", e->opt_h_walk); OPT("hv_walk", e->opt_hv_walk); OPT("h_walk_udiff", e->opt_h_walk_udiff); OPT("lu_fma", e->lu_fma);
#undef OPT
	return nullptr;
}
static const OptionDef g_options[] = {
	{ "unique_layout", "TMX_UNIQUE", 1, "node-unique state layout inside tmx_step: -1 default (= 1), 0 off, 1 on where eligible" },
	{ "unique_tile_shape", "TMX_UNIQUE_TILE", 1, "elements of a wavefront on that layout: 0 (default, measured fastest) the element-major order = 1 x 4 strips that wrap around patch rows, 1 = 2 x 2 blocks, 2 = strips that stay inside a patch row; experiments flavour only: 3 = a generalised Hilbert curve through every patch, 4 = 4 x 4 element blocks of 1 x 4 strips" },
	{ "unique_blocks", "TMX_UNIQUE_BLOCKS", 2, "archived experiment (round 6: the DSS loses 0.05 ms per step, the producers pay 0.10): block kernels on that layout -- a workgroup of four wavefronts averages the seams between them through LDS, the DSS kernel finishes fewer nodes: -1 (default) on with unique_tile_shape 4, 0 off, 1 on with any thread order" },
	{ "unique_xcd_order", "TMX_UNIQUE_XCD", 3, "A/B switch, experiments flavour only: 1 (default): every XCD sweeps a contiguous range of tiles" },
	{ "unique_mixed", "TMX_UNIQUE_MIXED", 1, "1 (default): the explicit stages read a live-in element-major instance copy by copy (no check, no conversion); 0: check the copies, convert or run the step element-major" },
	{ "unique_prefix", "TMX_UNIQUE_PREFIX", 0, "1 (default): an explicit stage also stores the leading partial sum of a later stage's combination over the instances both read (ARS343: the fourth stage reads 3 instances instead of 7); 0: every stage reads all its terms" },
	{ "share_copies", "TMX_SHARE_COPIES", 0, "1 (default): stage copies that stay identical to their source share its slot instead of being made" },
	{ "xcd_vertical", "TMX_XCD_VERTICAL", 2, "A/B switch, experiments flavour only: 1 (default): level blocks of a column tile on one XCD in the vertical stencil kernels" },
	{ "vi_carry", "TMX_VI_CARRY", 2, "column solve: carry shared sub-expressions between block rows (1)" },
	{ "vi_pair", "TMX_VI_PAIR", 2, "column solve: two-wavefront kernel (-1 auto, 0 one-wavefront kernel, 1 on)" },
	{ "vi_group", "TMX_VI_GROUP", 1, "column solve: one column per 16-lane group (-1 auto: up to vi_group_max unique columns, 0, 1)" },
	{ "vi_group_max", "TMX_VI_GROUP_MAX", 1, "largest rank share (unique columns) served by the lane-group kernel (6400)" },
	{ "vi_pair_workgroup", "TMX_VI_PAIR_WG", 0, "pairs per workgroup of the two-wavefront kernel (0 auto)" },
	{ "vi_ring_depth", "TMX_VI_RING_DEPTH", 0, "block rows in the LDS ring between assembly and elimination: 0 auto (3, or 2 where only that fits two workgroups per CU: more than ~38 levels), 2, 3" },
	{ "vi_producers", "TMX_VI_PRODUCERS", 0, "assembly wavefronts per column group of that kernel: 0 auto (2 on grids that leave every wavefront a SIMD of its own), 1, 2" },
	{ "vi_split_back", "TMX_VI_SPLIT_BACK", 2, "back substitution as a launch of its own (0)" },
	{ "vi_back_sub", "TMX_VI_BACK_SUB", 2, "wavefronts per column group of that launch (1)" },
	{ "vi_stagger", "TMX_VI_STAGGER_NS", 2, "A/B switch, experiments flavour only: staggered workgroup starts, window in units of 10 ns (0)" },
	{ "vi_split_kernels", "TMX_VI_MODE", 3, "1: assemble + solve as two kernels (cross-check path)" },
	{ "vi_sparse", "TMX_VI_SPARSE", 2, "1 (default): U-row entries that are zero in all 64 columns are not stored" },
	{ "vi_columns_per_wavefront", "TMX_VI_CPW", 1, "columns per wavefront of the two-wavefront kernel (0 = 64)" },
	{ "contraction_mfma", "TMX_MFMA", 1, "1: the 4 x 4 contractions of the fused explicit kernel on the matrix unit -- NOT bit-exact (1e-15 per call, W 1.5e-10 after 100 steps)" },
	{ "step_graph", "TMX_GRAPH", 1, "1: single-rank steps replayed from a captured hipGraph" },
	{ "p2p_timeout_s", "TMX_P2P_TIMEOUT_S", 0, "peer-to-peer halo: seconds a neighbour's message may take (600; 0 = for ever)" },
	{ "hvis_pull", "TMX_HVIS_PULL", 3, "archived experiment: DSS pulled into the second hyperviscosity pass" },
	{ "hvis_block", "TMX_HVIS_BLOCK", 3, "archived experiment: hyperviscosity passes fused with the in-patch DSS" },
	{ "split_stage_off", "TMX_NO_SPLIT", 1, "1: no boundary-first stages on several ranks" },
	{ "metric_stored", "TMX_METRIC", 1, "1: stream the stored 3-D metric arrays even where the closed form is verified" },
	{ "tracer_lincomb_pass", "TMX_TRACER_LINCOMB_PASS", 2, "A/B switch, experiments flavour only: 1: tracer stage combination by a separate pass (A/B)" },
	{ "udv_separate", "TMX_UDV_SEPARATE", 2, "A/B switch, experiments flavour only: 1: vertical diffusion of U,V as a pass of its own (A/B)" },
	{ "vx_fused", "TMX_VX_FUSED", 2, "1: V.StepExplicit's U,V update inside the explicitly-treated-terms kernel (A/B)" },
	{ "debug_skip_exchange", "TMX_DEBUG_SKIP_EXCHANGE", 2, "TIMING AID, WRONG RESULTS at rank boundaries: a lone rank engine of an N-rank layout with the wire left out" },
	{ "exchange_overlap_off", "TMX_NO_EXCHANGE_OVERLAP", 1, "1: the exchange runs on the engine's stream" },
	{ "kessler_column", "TMX_KESSLER_COLUMN", 0, "1: one-lane-per-column Kessler kernel (cross-check)" },
	{ "vt_column", "TMX_VT_COLUMN", 0, "1: one-lane-per-column explicit tracer update (cross-check)" },
	{ "vt_explicit_v1", "TMX_VT_EXPLICIT_V1", 2, "1: level-parallel explicit tracer update without LDS staging (cross-check)" },
	{ "vt_explicit_walk", "TMX_VT_WALK", 0, "explicit tracer update: -1000 (default) a sliding register window over column segments, their number chosen from the grid size; -n = n segments; 0 = the LDS-tiled level-parallel kernel; 4, 5, 6, 8, 10 = that many levels per thread held in registers (experiments build)" },
	{ "vite_walk", "TMX_VITE_WALK", 0, "explicitly evaluated implicit terms (StepImplicitTermsExplicitly): -1000 (default) a sliding register window over column segments, their number chosen from the grid size; -n = n segments; 0 = the level-parallel kernel" },
	{ "vx_walk", "TMX_VX_WALK", 0, "V.StepExplicit's U,V update: -1000 (default) a sliding register window over column segments, their number chosen from the grid size; -n = n segments; 0 = the level-parallel kernel" },
	{ "vt_lanes", "TMX_VT_LANES", 2, "A/B switch, experiments flavour only: columns per workgroup of the one-lane tracer column kernel (16)" },
	{ "vt_lw8", "TMX_VT_LW8", 2, "A/B switch, experiments flavour only: row-parallel tracer kernel: 8 columns per workgroup (-1 auto)" },
	{ "vt_row_lanes", "TMX_VT_NR", 2, "A/B switch, experiments flavour only: row lanes of that kernel (0 auto)" },
	{ "vt_rows", "TMX_VT_ROWS", 0, "1 (default): row-parallel implicit tracer update" },
	{ "lu_fma", "TMX_LU_FMA", 0, "band LU of the column solves: 1 (default) updates a - l u as ONE rounding (fused multiply-add: a reference linked to OpenBLAS, or to MKL on its FMA code paths), 0 = multiply and subtract rounded separately (a BLAS without fused multiply-adds); tmx_lu_flavour_from_dgbsv asks the caller's own LAPACK" },
	{ "h_walk_udiff", "TMX_H_WALK_UDIFF", 0, "uniform-diffusion configurations: the explicit stage's walk applies the horizontal uniform diffusion to its results in registers (1) and V.StepExplicit's U,V part behind it (2, default); 0 = k_uniform_diffusion and k_v_explicit as passes of their own (bit-identical)" },
	{ "hv_walk", "TMX_HV_WALK", 0, "hyperviscosity pass on the node-unique layout: 1 (default) a wavefront walks a segment of levels (k_hv_walk; -n: n segments per column), 0 = the level-parallel k_hypervis (bit-identical)" },
	{ "h_walk", "TMX_H_WALK", 0, "explicit stage on the node-unique layout: -1000 (default) a wavefront walks a column segment with a sliding register window, segments per column from the grid size; -n = n segments; 0 = the level-parallel kernel + k_h_w_update" },
};
static void tmx_record_option_defaults(tmx_engine * e) {
	e->opt_default.clear();
	for (const OptionDef & d : g_options) { int * slot = option_slot(e, d.name); e->opt_default.push_back(slot ? *slot : 0); }
}
// every row of g_options has its OPT() line (a row without one used to crash tmx_create): checked once per process
static bool options_table_consistent() {
	tmx_engine probe;
	for (const OptionDef & d : g_options) if (!option_slot(&probe, d.name)) { tmx_set_error("internal: option '%s' has no slot", d.name); return false; }
	return true;
}
extern "C" int tmx_set_option(tmx_engine * e, const char * name, double value) {
	REQUIRE(e && name, TMX_ERR_INVALID, "tmx_set_option: null argument");
	const std::string n(name);
	int * slot = option_slot(e, n);
	const OptionDef * def = nullptr;
	for (const OptionDef & d : g_options) if (n == d.name) def = &d;
	REQUIRE(slot && def, TMX_ERR_INVALID, "tmx_set_option: unknown option '%s'", name);
	REQUIRE(!((def->kind & 1) && e->finalized), TMX_ERR_INVALID, "tmx_set_option(%s) after tmx_finalize", name);
	const int iv = (int)value;
	REQUIRE(TMX_EXP || n != "vt_explicit_walk" || iv <= 0, TMX_ERR_UNSUPPORTED, "tmx_set_option(vt_explicit_walk = %d): the register-held segments are compiled into the experiments flavour of the library only", iv);
	int dflt = *slot;
	for (size_t k = 0; k < e->opt_default.size() && k < sizeof(g_options) / sizeof(g_options[0]); k++) if (n == g_options[k].name) dflt = e->opt_default[k];
	REQUIRE(TMX_EXP || !(def->kind & 2) || iv == dflt || (n == "vi_pair" && iv != 0), TMX_ERR_UNSUPPORTED,      // (vi_pair: -1 and 1 both mean the production kernel)
		"tmx_set_option(%s = %d): an archived experiment / cross-check kernel, compiled into the experiments flavour of the library only (libtempest_mi355x_exp.so)", name, iv);
	if (n == "unique_layout") REQUIRE(iv >= -1 && iv <= 1, TMX_ERR_INVALID, "unique_layout: -1 (default), 0 or 1");
	if (n == "unique_tile_shape") {
		REQUIRE(iv >= 0 && iv <= 4, TMX_ERR_INVALID, "unique_tile_shape: 0, 1 or 2 (3, 4: experiments flavour of the library)");
		REQUIRE(TMX_EXP || iv <= 2, TMX_ERR_UNSUPPORTED, "tmx_set_option(unique_tile_shape = %d): an archived experiment, compiled into the experiments flavour of the library only (libtempest_mi355x_exp.so)", iv);
	}
	if (n == "unique_blocks") REQUIRE(iv >= -1 && iv <= 3, TMX_ERR_INVALID, "unique_blocks: -1 (auto), 0, 1, 2 (the walk only) or 3 (the hyperviscosity pass only)");
	if (n == "p2p_timeout_s") REQUIRE(iv >= 0, TMX_ERR_INVALID, "p2p_timeout_s must not be negative");
	// integer options that select a kernel shape: only the values a launch knows (everything else used to fall through to a default silently)
	if (n == "vi_group") REQUIRE(iv >= -1 && iv <= 1, TMX_ERR_INVALID, "vi_group: -1 (auto), 0 or 1");
	if (n == "vi_group_max") REQUIRE(iv >= 0, TMX_ERR_INVALID, "vi_group_max must not be negative");
	if (n == "vi_pair_workgroup") REQUIRE(iv >= 0 && iv <= 2, TMX_ERR_INVALID, "vi_pair_workgroup: 0 (auto), 1 or 2");
	if (n == "vi_ring_depth") REQUIRE(iv == 0 || iv == 2 || iv == 3, TMX_ERR_INVALID, "vi_ring_depth: 0 (auto), 2 or 3");
	if (n == "vi_producers") REQUIRE(iv >= 0 && iv <= 2, TMX_ERR_INVALID, "vi_producers: 0 (auto), 1 or 2");
	if (n == "vi_stagger") REQUIRE(iv >= 0, TMX_ERR_INVALID, "vi_stagger must not be negative");
	if (n == "vi_columns_per_wavefront") REQUIRE(iv >= 0 && iv <= 64, TMX_ERR_INVALID, "vi_columns_per_wavefront: 0 (= 64) .. 64");
	if (n == "vi_back_sub") REQUIRE(iv == 1 || iv == 2 || iv == 4, TMX_ERR_INVALID, "vi_back_sub: 1, 2 or 4");
	if (n == "vt_lanes") REQUIRE(iv == 8 || iv == 16 || iv == 32 || iv == 64, TMX_ERR_INVALID, "vt_lanes: 8, 16, 32 or 64");
	if (n == "vt_row_lanes") REQUIRE(iv == 0 || iv == 4 || iv == 8 || iv == 16 || iv == 32, TMX_ERR_INVALID, "vt_row_lanes: 0 (auto), 4, 8, 16 or 32");
	if (n == "hv_walk") REQUIRE(iv <= 1 && iv >= -64, TMX_ERR_INVALID, "hv_walk: 0, 1 or -n (n segments)");
	if (n == "h_walk_udiff") REQUIRE(iv >= 0 && iv <= 2, TMX_ERR_INVALID, "h_walk_udiff: 0, 1 or 2");
	if (n == "vt_lw8") REQUIRE(iv >= -1 && iv <= 1, TMX_ERR_INVALID, "vt_lw8: -1 (auto), 0 or 1");
	if (n == "unique_xcd_order" || n == "unique_mixed" || n == "unique_prefix" || n == "lu_fma" || n == "share_copies" || n == "xcd_vertical" || n == "vt_rows" || n == "vi_sparse" || n == "vi_carry" || n == "step_graph" || n == "contraction_mfma")
		REQUIRE(iv == 0 || iv == 1 || ((n == "vi_carry") && iv == -1), TMX_ERR_INVALID, "%s: 0 or 1", name);
	*slot = iv;
	return TMX_OK;
}
extern "C" int tmx_get_option(tmx_engine * e, const char * name, double * value) {
	REQUIRE(e && name && value, TMX_ERR_INVALID, "tmx_get_option: null argument");
	int * slot = option_slot(e, std::string(name));
	REQUIRE(slot, TMX_ERR_INVALID, "tmx_get_option: unknown option '%s'", name);
	*value = *slot;
	return TMX_OK;
}
// "name=value" of every option, one per line, defaults included; returns the length needed (buf may be null)
extern "C" int tmx_options_report(tmx_engine * e, char * buf, int cap) {
	if (!e) return -1;
	std::string out;
	for (const OptionDef & d : g_options) { char line[160]; snprintf(line, sizeof(line), "%s=%d\n", d.name, *option_slot(e, d.name)); out += line; }
	if (!e->env_applied.empty()) { out += "from_environment="; for (const std::string & v : e->env_applied) out += v + " "; out += "\n"; }
	if (buf && cap > 0) { strncpy(buf, out.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
	return (int)out.size() + 1;
}
// The historical TMX_* variables -> options (test / bench plumbing calls this right after tmx_create; the library never does).
// Returns the number of variables applied and says so on stderr, ONE line, unless TMX_QUIET is set.
extern "C" int tmx_options_from_environment(tmx_engine * e) {
	REQUIRE(e, TMX_ERR_INVALID, "tmx_options_from_environment: null engine");
	int n = 0;
	std::string refused;
	for (const OptionDef & d : g_options) {
		const char * ev = getenv(d.env);
		if (!ev) continue;
		int v = atoi(ev);
		const std::string nm(d.name);
		if (nm == "metric_stored") v = (strcmp(ev, "stored") == 0) ? 1 : 0;
		else if (nm == "vi_split_kernels") v = (strcmp(ev, "split") == 0) ? 1 : 0;
		else if (nm == "vi_stagger") v = atoi(ev) / 10;
		else if (nm == "debug_skip_exchange" || nm == "split_stage_off" || nm == "exchange_overlap_off" || nm == "tracer_lincomb_pass" || nm == "udv_separate" ||
		         nm == "vx_fused" || nm == "vt_column" || nm == "vt_explicit_v1") v = 1;      // (variables whose presence alone was the switch)
		else if (nm == "unique_layout" || nm == "vi_pair" || nm == "vi_group" || nm == "contraction_mfma" || nm == "step_graph" || nm == "vi_carry" || nm == "xcd_vertical") v = v ? 1 : 0;
		if ((d.kind & 1) && e->finalized) continue;
		if (tmx_set_option(e, d.name, v) != TMX_OK) { refused += std::string(" ") + d.env + "=" + ev; continue; }
		e->env_applied.push_back(std::string(d.env) + "=" + ev);
		n++;
	}
	if (n && !getenv("TMX_QUIET")) {
		std::string l = "tempest_mi355x: options taken from the environment:";
		for (const std::string & v : e->env_applied) l += " " + v;
		fprintf(stderr, "%s\n", l.c_str());
	}
	// a variable that would have changed the run and cannot: an error, not a silent default
	REQUIRE(refused.empty(), TMX_ERR_UNSUPPORTED, "tmx_options_from_environment: refused by this build of the library (out of range, or an experiments-only option):%s", refused.c_str());
	return n;
}

extern "C" long long tmx_info(tmx_engine * e, int what) {
	if (!e) return -1;
	switch (what) {
		case TMX_INFO_LOCAL_COLUMNS: return e->ncol;
		case TMX_INFO_UNIQUE_COLUMNS: return e->nunique;
		case TMX_INFO_DSS_GROUPS: return e->ngroups;
		case TMX_INFO_LOCAL_ELEMENTS: return e->ne_local;
		case TMX_INFO_GHOST_COLUMNS: return e->nghost;
		case TMX_INFO_HBM_BYTES: return (long long)e->hbm_bytes;
		case TMX_INFO_METRIC_CLOSED_FORM: return e->metric_closed ? 1 : 0;
		case TMX_INFO_EARLY_TILES: return e->split_stage ? e->n_quads_early : 0;
		case TMX_INFO_LATE_TILES: return e->split_stage ? e->n_quads_late : 0;
		case TMX_INFO_UNIQUE_LAYOUT: return e->u.built ? 1 : 0;
		case TMX_INFO_UNIQUE_INSTANCES: return e->u.n_uform;
		case TMX_INFO_UNIQUE_CONVERSIONS: return e->u.conversions;
		case TMX_INFO_PARTIAL_SLOTS: return e->u.built ? e->u.nslots : 0;
		case TMX_INFO_UNIQUE_DSS_GROUPS: return e->u.built ? e->u.ngroups : 0;
		case TMX_INFO_PREFIX_STAGES: return e->u.prefix_stages;
		case TMX_INFO_EXPERIMENTS_BUILD: return TMX_EXP;
		case TMX_INFO_MIXED_STEPS: return e->u.mixed_steps;
		case TMX_INFO_COLUMN_KERNEL: return e->vi_kernel_launched;
		case TMX_INFO_COMM_RANKS: {      // what RCCL itself reports for the communicator (0: no communicator)
			int n = 0;
			if (e->comm && g_nccl.CommCount && g_nccl.CommCount(e->comm, &n) == 0) return n;
			return 0;
		}
		case TMX_INFO_SEND_COLUMNS: return e->nsend;
		case TMX_INFO_HALO_TRANSPORT: return (e->cfg.n_ranks == 1) ? 0 : (e->lb ? 3 : (e->p2p ? 2 : (e->comm ? 1 : 0)));
	}
	return -1;
}

// exchange / DSS plan for host-side tests: what = 0 send nodes (patch,i,j,dest_rank), 1 recv nodes
// (patch,i,j,src_rank), 2 groups (n, then 4 x column-or-NS+ghost).  Returns the number of ints written
// (or needed when out == nullptr).
extern "C" int tmx_plan_get(tmx_engine * e, int what, int * out, int cap) {
	if (!e || !e->finalized) return -1;
	PlanHost & plan = plan_of(e);
	std::vector<int> v;
	if (what == 0 || what == 1) {
		const std::vector<int> & nodes = what ? plan.recv_nodes : plan.send_nodes;
		const std::vector<int> & off = what ? e->recv_rank_off : e->send_rank_off;
		for (int rk = 0; rk < e->cfg.n_ranks; rk++)
			for (int t = off[rk]; t < off[rk + 1]; t++) { v.push_back(nodes[3 * t]); v.push_back(nodes[3 * t + 1]); v.push_back(nodes[3 * t + 2]); v.push_back(rk); }
	} else if (what == 2) {
		for (int g = 0; g < e->ngroups; g++) { v.push_back(plan.grp_n[g]); for (int m = 0; m < 4; m++) v.push_back(plan.grp_cols[(size_t)g * 4 + m]); }
	} else if (what == 3) {
		v.push_back(e->NS); v.push_back(e->ncol); v.push_back(e->nunique); v.push_back(e->ngroups); v.push_back(e->nxgroups);
	} else if (what == 4) {
		v = plan.grp_x;
	} else if (what == 5) {
		v = plan.grp_type;
	} else return -1;
	if (out) { if ((int)v.size() > cap) return -1; if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int)); }      // (an empty table has no data pointer to hand to memcpy: found by the sanitized build)
	return (int)v.size();
}

// cross-panel covector matrices of the DSS groups: [n_cross][member m][partner q][2x2], the matrix that rotates q's
// (U,V) into the frame of m's patch (identity where q == m or both lie on one panel)
extern "C" int tmx_plan_get_matrices(tmx_engine * e, double * out, int cap) {
	if (!e || !e->finalized) return -1;
	PlanHost & plan = plan_of(e);
	if (out) { if ((int)plan.xmat.size() > cap) return -1; if (!plan.xmat.empty()) memcpy(out, plan.xmat.data(), plan.xmat.size() * sizeof(double)); }
	return (int)plan.xmat.size();
}

// Host-side tables of the node-unique layout for the tile shape given, built on the spot (works on plan-only engines: no device),
// for the CPU tests of that logic.  what = 0: (NU, NUS, NTS, slots, groups left to the DSS kernel, of them without remote member,
// early tiles, late tiles); 1 t_dcol; 2 t_ucol; 3 t_sdst; 4 t_sred; 5 gsrc; 6 gdst; 7 gn; 8 slot_ucol; 9 send_slots; 10 u_rep;
// 11 ucol_of_dcol; 12 per-tile info.  Returns the number of ints written (needed, when out == NULL), -1 on error.
extern "C" int tmx_debug_unique_tables(tmx_engine * e, int tile_shape, int what, int * out, int cap) {
	if (!e || !e->finalized || tile_shape < 0 || tile_shape > 4) return -1;
	PlanHost & plan = plan_of(e);
	UniqueLayout keep = e->u;
	e->u = UniqueLayout(); e->u.tile_shape = tile_shape;
	UniquePlanInput in = { &plan.grp_cols, &plan.grp_n, &plan.grp_x, &plan.grp_type, &plan.send_cols };
	UniqueTables T;
	const int r = tmxu_tables(e, in, T);
	const UniqueLayout u = e->u;
	e->u = keep;
	if (r) return -1;
	std::vector<int> v;
	switch (what) {
		case 0: v = { u.NU, u.NUS, u.NTS, u.nslots, u.ngroups, u.ngroups_local, u.n_early, u.n_late, u.b_ngroups, u.b_ngroups_local, u.nblocks, u.nb_early, u.nb_late }; break;
		case 13: v = T.b_sdst; break; case 14: v = T.b_sred; break; case 15: v = T.blk_info; break; case 16: v = T.b_gsrc; break; case 17: v = T.b_gdst; break; case 18: v = T.b_gn; break;
		case 19: v = T.blks_early; break; case 20: v = T.blks_late; break;
		case 1: v = T.t_dcol; break; case 2: v = T.t_ucol; break; case 3: v = T.t_sdst; break; case 4: v = T.t_sred; break;
		case 5: v = T.gsrc; break; case 6: v = T.gdst; break; case 7: v = T.gn; break; case 8: v = T.slot_ucol; break;
		case 9: v = T.send_slots; break; case 10: v = T.u_rep; break; case 11: v = T.ucol_of_dcol; break; case 12: v = T.tinfo; break;
		default: return -1;
	}
	if (out) { if ((int)v.size() > cap) return -1; if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int)); }
	return (int)v.size();
}

// Diagnostic builds (-DTMX_H_TIMING): shader cycles per wavefront of the fused explicit kernel by phase, [16][8] (tmx_k_horizontal.hip)
extern "C" int tmx_debug_h_timing(tmx_engine * e, unsigned long long * out) {
	int r; if ((r = check_ready(e))) return r;
	HIPCHK(hipStreamSynchronize(e->stream));
	tmxk_h_timing_read(out);
	return TMX_OK;
}
// the same for the column-segment walk (tmx_k_hwalk.hip), [8][8]; padded to [16][8] like the above
extern "C" int tmx_debug_h_walk_timing(tmx_engine * e, unsigned long long * out) {
	int r; if ((r = check_ready(e))) return r;
	HIPCHK(hipStreamSynchronize(e->stream));
	tmxk_h_walk_timing_read(out);
	return TMX_OK;
}

// Statistics of the two-wavefront column kernel: how many pivot steps found the same pivot row in all 64 columns of a
// wavefront (the renaming path) out of all pivot steps.  enable = 1 starts counting (zeroed), 0 stops; out (may be NULL)
// receives {uniform, total} accumulated so far.
extern "C" int tmx_debug_pivot_stats(tmx_engine * e, int enable, unsigned long long out[2]) {
	int r; if ((r = check_ready(e))) return r;
	HIPCHK(hipStreamSynchronize(e->stream));
	if (out) {
		out[0] = out[1] = 0;
		if (e->d_pivot_stats) HIPCHK(hipMemcpy(out, e->d_pivot_stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	}
#ifdef TMX_PAIR_TIMING
	// diagnostic builds (tools/vi_timing.py): the column kernel leaves one record of cycle counts per wavefront behind the two counters
	const size_t stat_words = 2 + 8 * 4096;
	if (e->d_pivot_stats && !enable) {
		std::vector<unsigned long long> rec(stat_words);
		HIPCHK(hipMemcpy(rec.data(), e->d_pivot_stats, stat_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		for (size_t w = 0; w < 4096; w++) {
			const unsigned long long * q = &rec[2 + 8 * w];
			if (q[0]) fprintf(stderr, "pairtiming wave %zu role %llu hw_id %llx xcc %llu work %llu wait %llu forward %llu back %llu\n", w, q[0] - 1, q[1], q[2], q[3], q[4], q[5], q[6]);
		}
		if (e->vi_split_back) {      // fill-in masks of three column groups (which U-row entries exist), one line per group
			const int n = TMX_FTOT * (e->L + 1);
			for (int g : { 0, 100, 400 }) {
				std::vector<int> m(n);
				HIPCHK(hipMemcpy(m.data(), (const int *)e->d_rhs + (size_t)g * n, n * sizeof(int), hipMemcpyDeviceToHost));
				fprintf(stderr, "umask group %d:", g);
				for (int j = 0; j < n; j++) fprintf(stderr, " %x", m[j] >> 4);
				fprintf(stderr, "\n");
			}
		}
	}
#else
	const size_t stat_words = 2;
#endif
	if (enable && !e->d_pivot_stats) { HIPCHK(hipMalloc((void **)&e->d_pivot_stats, stat_words * sizeof(unsigned long long))); HIPCHK(hipMemset(e->d_pivot_stats, 0, stat_words * sizeof(unsigned long long))); }
	if (!enable && e->d_pivot_stats) { (void)hipFree(e->d_pivot_stats); e->d_pivot_stats = nullptr; }
	return TMX_OK;
}

extern "C" int tmx_profile_enable(tmx_engine * e, int on) {
	REQUIRE(e, TMX_ERR_INVALID, "null engine");
	e->prof = (on != 0);
	return TMX_OK;
}

extern "C" int tmx_profile_get(tmx_engine * e, int kernel, double * total_ms, long long * launches) {
	REQUIRE(e && kernel >= 0 && kernel < TMX_K_COUNT, TMX_ERR_INVALID, "bad kernel id");
	if (!plan_only(e) && e->stream) { hipStreamSynchronize(e->stream); prof_collect(e); }
	if (total_ms) *total_ms = e->prof_slots[kernel].ms;
	if (launches) *launches = e->prof_slots[kernel].n;
	return TMX_OK;
}

extern "C" int tmx_profile_reset(tmx_engine * e) {
	REQUIRE(e, TMX_ERR_INVALID, "null engine");
	if (!plan_only(e) && e->stream) { hipStreamSynchronize(e->stream); prof_collect(e); }
	for (int i = 0; i < TMX_K_COUNT; i++) e->prof_slots[i] = ProfSlot();
	return TMX_OK;
}

