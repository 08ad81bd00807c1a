// tmx_step.hip -- host side of the engine, part 2: the operations on the resident state -- stage algebra, the dynamics entry points,
// DSS + exchange, the stepper programs with their element-major and node-unique interpreters, column physics.  (Part 1, set-up and
// transfers: tmx_host.hip; shared declarations: tmx_hostshared.h.)
#include "tmx_hostshared.h"

// ---------------------------------------------------------------------------------------------
// kernel parameter block

KParams make_params(const tmx_engine * e) {
	KParams p;
	p.L = e->L; p.ncol = e->ncol; p.NS = e->NS;
	p.g2d = e->d_g2d; p.g3n = e->d_g3n; p.g3e = e->d_g3e; p.ops = e->d_ops;
	p.eta = e->d_eta; p.closed = e->metric_closed ? 1 : 0;
	p.inv_da = 1.0 / e->cfg.element_delta_a;
	p.quads = e->stage_quads;      // tile list of a boundary-first stage in progress (hv_stage_split), else all tiles
	p.NSS = e->NS; p.t_ucol = nullptr; p.t_tinfo = nullptr; p.t_sdst = nullptr; p.t_sred = nullptr; p.part = nullptr; p.NP = 0;      // element-major layout (tmxu_params: node-unique)
	p.u_ntiles = 0; p.u_xcd = e->u.xcd_order;
	p.NSD = e->NS; p.t_dcol = nullptr; p.b_sdst = nullptr; p.b_sred = nullptr; p.blk_info = nullptr; p.bquads = nullptr;
	p.grav = e->cfg.grav; p.Rd = e->cfg.Rd; p.cp = e->cfg.cp; p.p0 = e->cfg.p0;
	memcpy(p.dx, e->h_dx, sizeof(p.dx)); memcpy(p.stiff, e->h_stiff, sizeof(p.stiff));
	return p;
}

int check_ready(tmx_engine * e) {
	REQUIRE(e && e->finalized && !plan_only(e), TMX_ERR_INVALID, "engine not finalized");
	return TMX_OK;
}
int check_inst(tmx_engine * e, int ix, bool read_only) {
	REQUIRE(ix >= 0 && ix < e->cfg.n_instances, TMX_ERR_INVALID, "instance %d out of range [0,%d)", ix, e->cfg.n_instances);
	return settle_instance(e, ix, read_only);
}
double * inst(tmx_engine * e, int ix) { return e->d_state + (size_t)e->imap[ix] * e->inst_stride; }
// where the U,V slabs of an instance live (the first 2 L slabs of a slot): its slot, or the slot it shares U,V with
static const double * inst_uv(tmx_engine * e, int ix) { return e->d_state + (size_t)(e->uvmap[ix] != ix ? e->uvmap[ix] : e->imap[ix]) * e->inst_stride; }
// Entry points other than tmx_step see the instances they name in slots of their own: an instance that reads another one's
// slot, or whose slot others read (b == ix or imap[b] == ix), gets the CopyData that was left out.  b < 0: all of them.
// Shared U,V slabs are settled inside a stepper program too: the operations that understand them (the fused explicit stage)
// do not come through here.
// read_only: the caller only reads instance ix (downloads, output interpolation, the restart image).  The node-unique form of an
// instance converted for a reader stays valid beside the element-major one (form 2), so a download between two steps costs one
// conversion of the instance read and the next tmx_step neither checks nor converts anything.
int settle_instance(tmx_engine * e, int ix, bool read_only) {
	// slots in node-unique form (left by tmx_step) go back to the element-major form before anything else looks at them: the one
	// named (where nothing shares slots), else all of them; a writer invalidates the node-unique copy
	if (e->u.n_uform) {
		bool one = ix >= 0 && e->imap[ix] == ix && e->uvmap[ix] == ix;
		for (int b = 0; b < (int)e->imap.size() && one; b++) if (b != ix && (e->imap[b] == ix || e->uvmap[b] == ix)) one = false;
		for (int b = 0; b < (int)e->u.form.size(); b++)
			if (e->u.form[b] && (!one || b == ix)) { int r = tmxu_to_d(e, b, one && read_only); if (r) return r; }
	}
	if (e->n_shared && !e->in_program)
		for (int b = 0; b < (int)e->imap.size(); b++) {
			if (e->imap[b] == b || !(ix < 0 || b == ix || e->imap[b] == ix)) continue;
			HIPCHK(hipMemcpyAsync(e->d_state + (size_t)b * e->inst_stride, e->d_state + (size_t)e->imap[b] * e->inst_stride,
				e->inst_stride * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
			e->imap[b] = b; e->n_shared--;
		}
	if (e->n_uvshared)
		for (int b = 0; b < (int)e->uvmap.size(); b++) {
			if (e->uvmap[b] == b || !(ix < 0 || b == ix || e->uvmap[b] == ix)) continue;
			HIPCHK(hipMemcpyAsync(e->d_state + (size_t)b * e->inst_stride, e->d_state + (size_t)e->uvmap[b] * e->inst_stride,
				(size_t)2 * e->L * e->NS * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
			e->uvmap[b] = b; e->n_uvshared--;
		}
	return TMX_OK;
}

// Surface slots.  HeldSuarezPhysics::Perform forms its "surface pressure" from dataREdge[RIx][i][j][0] *
// dataREdge[TIx][i][j][0] (HeldSuarezPhysics.cpp:113-116).  With Lorenz staggering rho and rho*theta live on levels and
// those interface entries are scratch: the test case fills them and afterwards ONLY Grid::CopyData / ZeroData /
// LinearCombineData -- which run over whole arrays (GridPatch.cpp:1402-1553) -- change them, by a rounding error per
// combination.  To reproduce the forcing bit for bit the engine carries the two entries per column through the same
// stage algebra: they sit behind the state of every instance, whole-instance copies / combinations include them for
// free, and the fused paths below (which never materialise the copy or combination) update them separately -- only
// when a caller asked for tracked surface slots (tmx_set_patch_physics_inputs with surface_pressure == NULL).
static double * surface_slots(tmx_engine * e, int ix) { return inst(e, ix) + (size_t)e->nslab * e->NS; }
static int surface_copy(tmx_engine * e, int src, int dst) {
	if (!e->track_surface || src == dst) return TMX_OK;
	HIPCHK(hipMemcpyAsync(surface_slots(e, dst), surface_slots(e, src), (size_t)2 * e->NS * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	return TMX_OK;
}
static int surface_zero(tmx_engine * e, int ix) {
	if (!e->track_surface) return TMX_OK;
	HIPCHK(hipMemsetAsync(surface_slots(e, ix), 0, (size_t)2 * e->NS * sizeof(double), e->stream));
	return TMX_OK;
}
static int surface_lincomb(tmx_engine * e, int dst, int n, const double * const * src, const double * cf, int premul) {
	if (!e->track_surface) return TMX_OK;
	const size_t off = (size_t)e->nslab * e->NS;
	const double * s2[12];
	for (int m = 0; m < n; m++) s2[m] = src[m] + off;
	tmxk_lincomb(e, (size_t)2 * e->NS, surface_slots(e, dst), n, s2, cf, premul);
	return TMX_OK;
}

int launch_check(const char * what) {
	hipError_t r = hipGetLastError();
	if (r != hipSuccess) { tmx_set_error("%s: %s", what, hipGetErrorString(r)); return TMX_ERR_DEVICE; }
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// stage algebra

extern "C" int tmx_copy_data(tmx_engine * e, int src, int dst) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, src)) || (r = check_inst(e, dst))) return r;
	if (src == dst) return TMX_OK;
	ProfScope ps(e, TMX_K_LINCOMB);
	HIPCHK(hipMemcpyAsync(inst(e, dst), inst(e, src), e->inst_stride * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	return TMX_OK;
}

extern "C" int tmx_zero_data(tmx_engine * e, int ix) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, ix))) return r;
	HIPCHK(hipMemsetAsync(inst(e, ix), 0, e->inst_stride * sizeof(double), e->stream));
	return TMX_OK;
}

extern "C" int tmx_linear_combine_data(tmx_engine * e, const double * coeff, int n_coeff, int dst) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, dst))) return r;
	REQUIRE(coeff && n_coeff > dst && n_coeff <= e->cfg.n_instances, TMX_ERR_INVALID,
		"linear_combine_data: %d coefficients for destination %d and %d instances", n_coeff, dst, e->cfg.n_instances);
	if ((r = settle_instance(e, -1, false))) return r;
	const double * src[12]; double cf[12];
	int n = 1;
	src[0] = inst(e, dst); cf[0] = coeff[dst];
	for (int m = 0; m < n_coeff; m++) {
		if (m == dst || coeff[m] == 0.0) continue;
		REQUIRE(n < 12, TMX_ERR_UNSUPPORTED, "linear_combine_data: more than 11 source terms");
		src[n] = inst(e, m); cf[n] = coeff[m]; n++;
	}
	ProfScope ps(e, TMX_K_LINCOMB);
	tmxk_lincomb(e, e->inst_stride, inst(e, dst), n, src, cf, coeff[dst] != 0.0 ? 1 : 0);
	return launch_check("lincomb");
}

// ---------------------------------------------------------------------------------------------
// dynamics

// uniform-diffusion extras at the end of HorizontalDynamicsFEM::StepExplicit (:1817-1859)
// fused: the explicit stage's kernel has applied it already (tmxk_h_walk_fuses_udiff)
static int h_uniform_diffusion(tmx_engine * e, int iinit, int iupd, double dt, bool fused = false) {
	if (!e->udiff || fused) return TMX_OK;
	int r; if ((r = check_reference_state(e))) return r;
	tmxk_uniform_diffusion(e, make_params(e), inst(e, iinit), e->d_ref, inst(e, iupd), dt, e->cfg.uniform_diffusion_scalar, e->cfg.uniform_diffusion_vector);
	return TMX_OK;
}

// VerticalDynamicsFEM::StepExplicit beyond the upwind penalty of U,V (which tmxk_v_explicit / the fused explicit kernel
// apply): in the fully explicit mode -dt F on rho*theta, W, rho (:745-790) and the explicit tracer update (:792-800),
// and with uniform diffusion the vertical diffusion of U,V (:1059-1105).
// uv_done: the vertical diffusion of U,V has been added by tmxk_v_explicit already (TMX_UDV_SEPARATE=1 keeps the separate pass)
static bool udv_fused(const tmx_engine * e) { return e->udiff && e->fully_explicit && !e->opt_udv_separate; }
// TMX_VX_FUSED=1: the U,V update of V.StepExplicit evaluated by the kernel of the explicitly treated implicit terms (one launch
// less, shared operands).  Off by default: config 4's step measured 4.81 ms with it, 4.75 ms without -- the terms kernel is
// bound by its dependent arithmetic, the U,V update on its own by bandwidth, and the two overlap better as two launches.
static bool uvx_fused(const tmx_engine * e) { return udv_fused(e) && e->opt_vx_fused; }
static int v_explicit_extras(tmx_engine * e, int iinit, int iupd, double dt, bool uv_done = false, bool with_uv = false) {
	if (!e->fully_explicit) return TMX_OK;
	int r; if ((r = check_reference_state(e))) return r;
	const KParams p = make_params(e);
	tmxk_vi_terms_explicit(e, p, inst(e, iinit), inst(e, iupd), dt, with_uv);
	if (e->nt > 0)
		REQUIRE(tmxk_vi_tracers_explicit(e, p, inst(e, iinit), inst(e, iupd), dt) == 0, TMX_ERR_UNSUPPORTED,
			"tracer column update: %d levels do not fit the LDS working set", e->L);
	if (e->udiff && !uv_done)
		tmxk_v_uniform_diffusion_uv(e, p, inst(e, iinit), e->d_ref, inst(e, iupd), dt, e->cfg.uniform_diffusion_vector / (e->cfg.ztop * e->cfg.ztop));
	return TMX_OK;
}

extern "C" int tmx_h_step_explicit(tmx_engine * e, int iinit, int iupd, double dt) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd))) return r;
	// same precondition as the reference (HorizontalDynamicsFEM.cpp:1793-1796)
	REQUIRE(iinit != iupd, TMX_ERR_INVALID, "StepExplicit: initial and update data instance must be distinct");
	ProfScope ps(e, TMX_K_H_EXPLICIT);
	const bool udf = !e->sw && tmxk_h_walk_fuses_udiff(e, make_params(e), 0, 0);
	if (udf && (r = check_reference_state(e))) return r;
	if (e->sw) tmxk_sw_explicit(e, make_params(e), inst(e, iinit), inst(e, iupd), inst(e, iupd), dt);
	else tmxk_h_explicit(e, make_params(e), inst(e, iinit), inst(e, iupd), inst(e, iupd), dt, 0);
	if (e->nt > 0) tmxk_h_tracers(e, make_params(e), inst(e, iinit), inst(e, iupd), inst(e, iupd), dt);
	if ((r = h_uniform_diffusion(e, iinit, iupd, dt, udf))) return r;
	return launch_check("h_step_explicit");
}

// CopyData(initial -> update) followed by StepExplicit(initial, update) in one pass:
// update = initial + dt * rhs(initial), bit-identical to the two reference calls
static int lincomb_terms(tmx_engine * e, const double * coeff, int n_coeff, int dst, const double ** src, double * cf, int * premul, const double ** srcuv = nullptr) {
	int n = 1;
	src[0] = inst(e, dst); cf[0] = coeff[dst];
	if (srcuv) srcuv[0] = inst_uv(e, dst);
	for (int m = 0; m < n_coeff; m++) {
		if (m == dst || coeff[m] == 0.0) continue;
		if (n >= 12) return -1;
		if (srcuv) srcuv[n] = inst_uv(e, m);
		src[n] = inst(e, m); cf[n] = coeff[m]; n++;
	}
	*premul = (coeff[dst] != 0.0) ? 1 : 0;
	return n;
}

// LinearCombineData(coeff -> d) + H.StepExplicit(i, d) + V.StepExplicit(i, d) in one pass: the combination is
// evaluated inside the kernels (same accumulation order), the combined state is never written and re-read.
// the kernels of one explicit stage (H + tracers + uniform diffusion + V) over the tiles make_params() selects: all of them,
// or the tile list of a boundary-first stage (e->stage_quads).  n > 0: the update starts from the combination src / cf.
static int hv_stage_kernels(tmx_engine * e, int iinit, int ibase, int iupd, double dt, int n, const double * const * src, const double * cf, int premul,
	const double * const * srcuv) {
	int r;
	// U,V slabs that live in another instance's slot are understood by k_h_explicit and k_h_tracers only; the kernels of the other configurations
	// read them through the instance pointer, and the stepper never shares U,V there
	REQUIRE(!e->n_uvshared || (!e->udiff && !e->fully_explicit && !e->sw), TMX_ERR_UNSUPPORTED, "internal: shared U,V slabs in a configuration whose kernels do not take them");
	// with uniform diffusion the horizontal diffusion of U,V precedes the vertical penalty, as in the reference: V.StepExplicit's U,V part is
	// fused in only where the same kernel applies that diffusion first (the walk, option h_walk_udiff = 2)
	const bool uvx = uvx_fused(e);
	const bool vfu = e->udiff && !uvx && tmxk_h_walk_fuses_udiff(e, make_params(e), 1, n);
	const int fv = e->udiff ? (vfu ? 1 : 0) : 1;
	const bool udf = vfu || tmxk_h_walk_fuses_udiff(e, make_params(e), fv, n);
	if (udf && (r = check_reference_state(e))) return r;
	if (n > 0) tmxk_h_explicit(e, make_params(e), inst(e, iinit), inst(e, iupd), inst(e, iupd), dt, fv, n, src, cf, premul, inst_uv(e, iinit), inst_uv(e, iupd), srcuv);
	else tmxk_h_explicit(e, make_params(e), inst(e, iinit), inst(e, ibase), inst(e, iupd), dt, fv, 0, nullptr, nullptr, 0, inst_uv(e, iinit), inst_uv(e, ibase), nullptr);
	if (e->nt > 0) {
		if (n > 0) {
			// tracers: the combination of the tracer slabs is evaluated inside the tracer kernel, which updates in place
			// (TMX_TRACER_LINCOMB_PASS=1: formed by a separate pass first, for A/B and tests; whole patches only)
			if (e->opt_tracer_lincomb_pass) {
				const size_t off = (size_t)(5 * e->L + 1) * e->NS, cnt = (size_t)e->nt * e->L * e->NS;
				const double * tsrc[12];
				for (int m = 0; m < n; m++) tsrc[m] = src[m] + off;
				tmxk_lincomb(e, cnt, inst(e, iupd) + off, n, tsrc, cf, premul);
				tmxk_h_tracers(e, make_params(e), inst(e, iinit), inst(e, iupd), inst(e, iupd), dt, 0, nullptr, nullptr, 0, inst_uv(e, iinit));
			} else
				tmxk_h_tracers(e, make_params(e), inst(e, iinit), inst(e, iupd), inst(e, iupd), dt, n, src, cf, premul, inst_uv(e, iinit));
		} else
			tmxk_h_tracers(e, make_params(e), inst(e, iinit), inst(e, ibase), inst(e, iupd), dt, 0, nullptr, nullptr, 0, inst_uv(e, iinit));
	}
	if ((r = h_uniform_diffusion(e, iinit, iupd, dt, udf))) return r;
	if (e->udiff && !uvx && !vfu) {
		if (udv_fused(e) && (r = check_reference_state(e))) return r;
		tmxk_v_explicit(e, make_params(e), inst(e, iinit), inst(e, iupd), dt, udv_fused(e));
	}
	return v_explicit_extras(e, iinit, iupd, dt, udv_fused(e), uvx);
}

static int hv_step_explicit_lincomb(tmx_engine * e, int iinit, int iupd, double dt, const double * coeff, int n_coeff) {
	const double * src[12], * srcuv[12]; double cf[12]; int premul = 0;
	const int n = lincomb_terms(e, coeff, n_coeff, iupd, src, cf, &premul, srcuv);
	REQUIRE(n > 0, TMX_ERR_UNSUPPORTED, "linear combination with more than 11 source terms");
	ProfScope ps(e, TMX_K_H_EXPLICIT);
	int r;
	if ((r = hv_stage_kernels(e, iinit, iupd, iupd, dt, n, src, cf, premul, srcuv))) return r;
	if ((r = surface_lincomb(e, iupd, n, src, cf, premul))) return r;
	return launch_check("hv_step_explicit(lincomb)");
}

// H.StepExplicit + V.StepExplicit of one stage in one pass over the state; `ibase` is the instance the
// update starts from (== iupd for the reference's in-place accumulation, == iinit when the preceding
// CopyData(initial -> update) is folded in).  Bit-identical to the separate calls.
static int hv_step_explicit(tmx_engine * e, int iinit, int ibase, int iupd, double dt) {
	ProfScope ps(e, TMX_K_H_EXPLICIT);
	int r;
	if ((r = hv_stage_kernels(e, iinit, ibase, iupd, dt, 0, nullptr, nullptr, 0, nullptr))) return r;
	if ((r = surface_copy(e, ibase, iupd))) return r;       // the folded CopyData(base -> update)
	return launch_check("hv_step_explicit");
}

// One explicit stage followed by its DSS on more than one rank, boundary first: the stage's kernels on the tiles that
// hold columns other ranks need, pack + grouped send/recv on the exchange stream, the same kernels on the remaining tiles
// while the wire is busy, then the averaging (local groups, wait for the wire, groups with remote members).  Same
// kernels on disjoint tiles -- every kernel of a stage is local to a 64-column tile (whole elements, whole columns) --
// so the result is bit-identical to the unsplit sequence.  All configurations: plain dynamics, tracers, uniform
// diffusion, the fully explicit vertical mode (BASELINE config 4), shallow water (sw_stage_split).  The two whole-patch
// debug variants of the tracer kernels (TMX_VT_COLUMN, TMX_TRACER_LINCOMB_PASS) switch the split off.
static bool stage_can_split(const tmx_engine * e) {
	return e->split_stage && !e->opt_vt_column && !e->opt_tracer_lincomb_pass;
}
static int exchange(tmx_engine * e, const KParams & p, double * x, bool * overlapped);
static int dss_after_exchange(tmx_engine * e, const KParams & p, int ix, bool overlapped);
static int dss_after_exchange(tmx_engine * e, const KParams & p, int ix, bool overlapped, int g_first);

static int hv_stage_split(tmx_engine * e, int iinit, int ibase, int iupd, double dt, const double * lc, int nlc) {
	const double * src[12], * srcuv[12]; double cf[12]; int premul = 0, n = 0;
	if (lc) {
		n = lincomb_terms(e, lc, nlc, iupd, src, cf, &premul, srcuv);
		REQUIRE(n > 0, TMX_ERR_UNSUPPORTED, "linear combination with more than 11 source terms");
	}
	int r = TMX_OK;
	for (int part = 0; part < 2 && !r; part++) {
		e->stage_quads = part ? e->d_quads_late : e->d_quads_early;
		e->launch_tiles = part ? e->n_quads_late : e->n_quads_early;
		{
			ProfScope ps(e, TMX_K_H_EXPLICIT);
			r = hv_stage_kernels(e, iinit, ibase, iupd, dt, n, src, cf, premul, lc ? srcuv : nullptr);
		}
		e->stage_quads = nullptr;
		if (part == 0 && !r) {
			bool overlapped = false;
			r = exchange(e, make_params(e), inst(e, iupd), &overlapped);
			e->split_overlapped = overlapped;
		}
	}
	if (r) return r;
	if (lc) { if ((r = surface_lincomb(e, iupd, n, src, cf, premul))) return r; }
	else if ((r = surface_copy(e, ibase, iupd))) return r;
	if ((r = launch_check("hv_stage_split"))) return r;
	return dss_after_exchange(e, make_params(e), iupd, e->split_overlapped);
}

// the same for the shallow-water set, whose stage is H.StepExplicit (V is a stub); ibase: the instance the update starts from
// (== iupd in place, == the source of a CopyData(ibase -> iupd) folded in)
static int sw_stage_split(tmx_engine * e, int iinit, int ibase, int iupd, double dt) {
	int r = TMX_OK;
	for (int part = 0; part < 2 && !r; part++) {
		e->stage_quads = part ? e->d_quads_late : e->d_quads_early;
		e->launch_tiles = part ? e->n_quads_late : e->n_quads_early;
		{
			ProfScope ps(e, TMX_K_H_EXPLICIT);
			tmxk_sw_explicit(e, make_params(e), inst(e, iinit), inst(e, ibase), inst(e, iupd), dt);
		}
		e->stage_quads = nullptr;
		if (part == 0) {
			bool overlapped = false;
			r = exchange(e, make_params(e), inst(e, iupd), &overlapped);
			e->split_overlapped = overlapped;
		}
	}
	if (r) return r;
	if ((r = launch_check("sw_stage_split"))) return r;
	return dss_after_exchange(e, make_params(e), iupd, e->split_overlapped);
}

// CopyData restricted to the U,V slabs: the implicit step overwrites rho*theta, W, rho of every column
static int copy_uv(tmx_engine * e, int src, int dst) {
	ProfScope ps(e, TMX_K_LINCOMB);
	// (the source's U,V slabs may live in another instance's slot: inst_uv, not inst)
	HIPCHK(hipMemcpyAsync(inst(e, dst), inst_uv(e, src), (size_t)2 * e->L * e->NS * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	return surface_copy(e, src, dst);
}

extern "C" int tmx_v_step_explicit(tmx_engine * e, int iinit, int iupd, double dt) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd))) return r;
	if (e->sw) return TMX_OK;      // VerticalDynamicsStub
	REQUIRE(iinit != iupd, TMX_ERR_INVALID, "V StepExplicit: initial and update data instance must be distinct");
	ProfScope ps(e, TMX_K_V_EXPLICIT);
	if (udv_fused(e) && (r = check_reference_state(e))) return r;
	tmxk_v_explicit(e, make_params(e), inst(e, iinit), inst(e, iupd), dt, udv_fused(e));
	if ((r = v_explicit_extras(e, iinit, iupd, dt, udv_fused(e)))) return r;
	return launch_check("v_step_explicit");
}

// itbase: instance whose tracer densities the column update is subtracted from (the update instance in the
// reference; the initial instance when the CopyData in front of the call was folded away)
static int v_step_implicit_impl(tmx_engine * e, int iinit, int iupd, double dt, int itbase) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd))) return r;
	if (e->sw) return TMX_OK;      // VerticalDynamicsStub
	REQUIRE(dt != 0.0, TMX_ERR_INVALID, "StepImplicit: dt must be non-zero");
	KParams p = make_params(e);
	const double * w0 = inst(e, iinit) + (size_t)TMX_SLAB_W(e->L, 0) * e->NS;
	if (e->nt > 0 && iinit == iupd) {
		// in place: the state kernel overwrites W, the tracer update needs the initial one (m_dColumnState)
		HIPCHK(hipMemcpyAsync(e->d_w0, w0, (size_t)(e->L + 1) * e->NS * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
		w0 = e->d_w0;
	}
#if TMX_EXP
	if (e->vi_mode == 1) {
		{ ProfScope ps(e, TMX_K_VI_ASSEMBLE); tmxk_vi_assemble(e, p, inst(e, iinit), dt); }
		{ ProfScope ps(e, TMX_K_VI_SOLVE); tmxk_vi_solve(e, p, inst(e, iinit), inst(e, iupd)); }
	} else
#endif
	{
		ProfScope ps(e, TMX_K_VI_SOLVE); tmxk_vi_fused(e, p, inst(e, iinit), inst(e, iupd), dt);
	}
	if (e->nt > 0) {
		// UpdateColumnTracers with the updated W, duplicates, then VerticalDynamicsFEM::FilterNegativeTracers
		ProfScope ps(e, TMX_K_VI_SOLVE);
		REQUIRE(tmxk_vi_tracers(e, p, inst(e, iinit), w0, inst(e, itbase), inst(e, iupd), dt) == 0, TMX_ERR_UNSUPPORTED,
			"tracer column update: %d levels do not fit the LDS working set", e->L);
		tmxk_v_filter_tracers(e, p, inst(e, iupd));
	}
	return launch_check("v_step_implicit");
}

extern "C" int tmx_v_step_implicit(tmx_engine * e, int iinit, int iupd, double dt) {
	if (e && e->fully_explicit) {      // VerticalDynamicsFEM::StepImplicit, :1239-1242: nothing to do
		int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd))) return r;
		return TMX_OK;
	}
	return v_step_implicit_impl(e, iinit, iupd, dt, iupd);
}

// test hook (tmx_debug_loopback_group): rank engines of one process, one host thread each
struct LoopbackGroup {
	std::vector<tmx_engine *> members;
	pthread_barrier_t barrier;
};

static int exchange_loopback(tmx_engine * e, const KParams & p, double * x) {
	LoopbackGroup * G = e->lb;
	const int n = (int)G->members.size(), me = e->cfg.rank;
	if (x) tmxk_pack(e, p, x); else tmxuk_pack(e);
	HIPCHK(hipStreamSynchronize(e->stream));
	pthread_barrier_wait(&G->barrier);                  // every member has packed
	for (int s_ = 0; s_ < n; s_++) {
		if (s_ == me) continue;
		tmx_engine * S = G->members[s_];
		const int ns = S->send_rank_off[me + 1] - S->send_rank_off[me], nr = e->recv_rank_off[s_ + 1] - e->recv_rank_off[s_];
		REQUIRE(ns == nr, TMX_ERR_INVALID, "rank %d sends %d columns to rank %d which expects %d", s_, ns, me, nr);
		if (ns == 0) continue;
		HIPCHK(hipMemcpyAsync(e->d_ghost + (size_t)e->nslab * e->recv_rank_off[s_], S->d_sendbuf + (size_t)S->nslab * S->send_rank_off[me],
			(size_t)ns * S->nslab * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	}
	HIPCHK(hipStreamSynchronize(e->stream));
	pthread_barrier_wait(&G->barrier);                  // nobody repacks before everyone has copied
	return TMX_OK;
}

extern "C" int tmx_debug_loopback_group(tmx_engine ** engines, int n) {
	REQUIRE(engines && engines[0] && n >= 0, TMX_ERR_INVALID, "tmx_debug_loopback_group: bad argument");
	if (n == 0) {
		LoopbackGroup * G = engines[0]->lb;
		if (G) { for (tmx_engine * m : G->members) m->lb = nullptr; pthread_barrier_destroy(&G->barrier); delete G; }
		return TMX_OK;
	}
	LoopbackGroup * G = new LoopbackGroup();
	for (int a = 0; a < n; a++) {
		tmx_engine * e = engines[a];
		REQUIRE(e && e->finalized && e->cfg.n_ranks == n && e->cfg.rank == a && !e->lb, TMX_ERR_INVALID, "engine %d is not rank %d of %d (or already grouped)", a, a, n);
		G->members.push_back(e);
	}
	pthread_barrier_init(&G->barrier, nullptr, (unsigned)n);
	for (tmx_engine * m : G->members) m->lb = G;
	return TMX_OK;
}


// Exchange of the boundary columns of instance x.  ONE message per neighbour rank (the reference aggregates its
// ExchangeBuffers the same way, Connectivity.cpp:928-993): buffers are [peer][slab][count_peer], contiguous per peer.
// The grouped send/recv runs on a second stream between two events, so that the DSS of the groups without remote
// members (the great majority) overlaps the wire; *overlapped tells the caller to wait for ev_recv before the rest.
static int exchange(tmx_engine * e, const KParams & p, double * x, bool * overlapped = nullptr) {
	if (overlapped) *overlapped = false;
	if (e->cfg.n_ranks == 1 || (e->nsend == 0 && e->nghost == 0)) return TMX_OK;
	if (e->lb) return exchange_loopback(e, p, x);
	// timing aid: a lone rank engine of an N-rank layout with the wire left out (results are then wrong at the rank
	// boundary; used only by tools/rank_share_timing.py to measure the per-rank compute share of a step)
	if (TMX_EXP && e->opt_skip_exchange) { if (x) tmxk_pack(e, p, x); else tmxuk_pack(e); return TMX_OK; }      // (experiments flavour only)
	REQUIRE(e->comm || e->p2p, TMX_ERR_COMM, "tmx_comm_init or tmx_halo_p2p_connect must be called before a multi-rank exchange");
	ProfScope ps(e, TMX_K_EXCHANGE);
	const unsigned long long seq = ++e->p2p_seq;
	const int buf = (int)(seq & 1);
	// (x == nullptr: node-unique layout, what travels are the per-element values in the partial slots)
	if (e->p2p) { if (x) tmxk_pack_p2p(e, p, x, buf); else tmxuk_pack_p2p(e, buf); }
	else { if (x) tmxk_pack(e, p, x); else tmxuk_pack(e); }
	hipStream_t ws = e->stream;
	if (overlapped && e->xstream) {
		HIPCHK(hipEventRecord(e->ev_pack, e->stream));
		HIPCHK(hipStreamWaitEvent(e->xstream, e->ev_pack, 0));
		ws = e->xstream;
	}
	if (e->p2p) {
		// the gather has written the neighbours' ghost buffers of this parity; say so and wait for theirs.  Two parities are
		// enough: a neighbour writes parity b again only after its own averaging of the exchange in between, which needed this
		// rank's message of that exchange, which this rank sent after the averaging that read parity b.
		tmxk_p2p_signal_wait(e, ws, buf, seq);
		e->d_ghost = (double *)((char *)e->p2p_block + p2p_header_bytes(e->cfg.n_ranks)) + (size_t)buf * e->nslab * e->nghost_pad;
		if (ws != e->stream) {
			HIPCHK(hipEventRecord(e->ev_recv, ws));
			*overlapped = true;
		}
		return TMX_OK;
	}
	NCCLCHK(g_nccl.GroupStart());
	for (int rk = 0; rk < e->cfg.n_ranks; rk++) {
		const int ns = e->send_rank_off[rk + 1] - e->send_rank_off[rk], nr = e->recv_rank_off[rk + 1] - e->recv_rank_off[rk];
		if (ns) NCCLCHK(g_nccl.Send(e->d_sendbuf + (size_t)e->nslab * e->send_rank_off[rk], (size_t)ns * e->nslab, 8 /* ncclFloat64 */, rk, e->comm, ws));
		if (nr) NCCLCHK(g_nccl.Recv(e->d_ghost + (size_t)e->nslab * e->recv_rank_off[rk], (size_t)nr * e->nslab, 8, rk, e->comm, ws));
	}
	NCCLCHK(g_nccl.GroupEnd());
	if (ws != e->stream) {
		HIPCHK(hipEventRecord(e->ev_recv, ws));
		*overlapped = true;
	}
	return TMX_OK;
}

// ---- test hooks for the multi-rank device path on a single GPU ------------------------------------
// Several engines (ranks 0..n-1 of the same n-rank grid) live in ONE process on one device; the
// transport is replaced by device-to-device copies that follow exactly the wire order of the RCCL
// path (segment [send_rank_off[r], send_rank_off[r+1]) of each slab of the sender -> segment
// [recv_rank_off[s], ...) of the receiver's ghost buffer).  Packing, ghost indexing and the DSS
// kernel with remote members are the production code.
extern "C" int tmx_debug_dss_loopback(tmx_engine ** engines, int n, int ix) {
	REQUIRE(engines && n >= 1, TMX_ERR_INVALID, "tmx_debug_dss_loopback: bad argument");
	int r;
	for (int a = 0; a < n; a++) {
		tmx_engine * e = engines[a];
		if ((r = check_ready(e)) || (r = check_inst(e, ix))) return r;
		REQUIRE(e->cfg.n_ranks == n && e->cfg.rank == a, TMX_ERR_INVALID, "engine %d is rank %d of %d", a, e->cfg.rank, e->cfg.n_ranks);
		tmxk_pack(e, make_params(e), inst(e, ix));
		HIPCHK(hipStreamSynchronize(e->stream));
	}
	for (int s_ = 0; s_ < n; s_++) for (int d_ = 0; d_ < n; d_++) {
		if (s_ == d_) continue;
		tmx_engine * S = engines[s_], * D = engines[d_];
		const int ns = S->send_rank_off[d_ + 1] - S->send_rank_off[d_], nr = D->recv_rank_off[s_ + 1] - D->recv_rank_off[s_];
		REQUIRE(ns == nr, TMX_ERR_INVALID, "rank %d sends %d columns to rank %d which expects %d", s_, ns, d_, nr);
		if (ns == 0) continue;
		HIPCHK(hipMemcpy(D->d_ghost + (size_t)D->nslab * D->recv_rank_off[s_], S->d_sendbuf + (size_t)S->nslab * S->send_rank_off[d_],
			(size_t)ns * S->nslab * sizeof(double), hipMemcpyDeviceToDevice));
	}
	HIPCHK(hipDeviceSynchronize());     // device-to-device copies may still be in flight on the null stream
	for (int a = 0; a < n; a++) {
		tmx_engine * e = engines[a];
		tmxk_dss(e, make_params(e), inst(e, ix), 0, e->ngroups_local);
		tmxk_dss(e, make_params(e), inst(e, ix), e->ngroups_local, e->ngroups);
		if ((r = launch_check("dss (loopback)"))) return r;
	}
	return TMX_OK;
}

// RCCL transport self-test on one rank: a grouped send/recv to self of the send buffer into the ghost
// buffer region (library resolution, communicator, stream ordering).  Returns TMX_OK if the bytes arrive.
extern "C" int tmx_debug_comm_selftest(tmx_engine * e) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(e->comm, TMX_ERR_COMM, "tmx_comm_init first");
	const int n = 4096;
	double * a = nullptr, * b = nullptr;
	HIPCHK(hipMalloc((void **)&a, n * sizeof(double))); HIPCHK(hipMalloc((void **)&b, n * sizeof(double)));
	std::vector<double> h(n), g(n, 0.0);
	for (int i = 0; i < n; i++) h[i] = 0.5 * i + 1.0;
	HIPCHK(hipMemcpy(a, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
	HIPCHK(hipMemset(b, 0, n * sizeof(double)));
	// the choreography of exchange(): producer on the engine's stream, event, grouped send/recv on the exchange
	// stream, event, consumer back on the engine's stream -- three rounds so that buffer reuse is ordered too
	for (int round = 0; round < 3; round++) {
		HIPCHK(hipMemcpyAsync(a, h.data(), n * sizeof(double), hipMemcpyHostToDevice, e->stream));
		hipStream_t ws = e->stream;
		if (e->xstream) {
			HIPCHK(hipEventRecord(e->ev_pack, e->stream));
			HIPCHK(hipStreamWaitEvent(e->xstream, e->ev_pack, 0));
			ws = e->xstream;
		}
		NCCLCHK(g_nccl.GroupStart());
		NCCLCHK(g_nccl.Send(a, (size_t)n, 8, e->cfg.rank, e->comm, ws));
		NCCLCHK(g_nccl.Recv(b, (size_t)n, 8, e->cfg.rank, e->comm, ws));
		NCCLCHK(g_nccl.GroupEnd());
		if (e->xstream) {
			HIPCHK(hipEventRecord(e->ev_recv, ws));
			HIPCHK(hipStreamWaitEvent(e->stream, e->ev_recv, 0));
		}
		HIPCHK(hipMemcpyAsync(g.data(), b, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
		HIPCHK(hipStreamSynchronize(e->stream));
		for (int i = 0; i < n; i++) REQUIRE(g[i] == h[i], TMX_ERR_COMM, "RCCL self send/recv returned wrong data at %d (round %d)", i, round);
		for (int i = 0; i < n; i++) { h[i] = h[i] * 1.5 + round; g[i] = 0.0; }
	}
	hipFree(a); hipFree(b);
	return TMX_OK;
}

// the averaging after the exchange of instance ix has been started (exchange()): groups without remote members first
static int dss_after_exchange(tmx_engine * e, const KParams & p, int ix, bool overlapped);

extern "C" int tmx_apply_dss(tmx_engine * e, int ix) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, ix))) return r;
	KParams p = make_params(e);
	bool overlapped = false;
	if ((r = exchange(e, p, inst(e, ix), &overlapped))) return r;
	return dss_after_exchange(e, p, ix, overlapped);
}

// g_first: first group to average (0: all; ngroups_inpatch: the producing kernel has averaged the in-patch groups itself)
static int dss_after_exchange(tmx_engine * e, const KParams & p, int ix, bool overlapped, int g_first) {
	ProfScope ps(e, TMX_K_DSS);
	if (e->cfg.n_ranks > 1 && e->ngroups_local < e->ngroups) {
		// groups whose members all live on this rank first (they overlap the wire), then the ones with remote members
		tmxk_dss(e, p, inst(e, ix), g_first, e->ngroups_local);
		if (overlapped) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_recv, 0));
		tmxk_dss(e, p, inst(e, ix), e->ngroups_local, e->ngroups);
	} else {
		tmxk_dss(e, p, inst(e, ix), g_first, e->ngroups);
	}
	return launch_check("apply_dss");
}
static int dss_after_exchange(tmx_engine * e, const KParams & p, int ix, bool overlapped) { return dss_after_exchange(e, p, ix, overlapped, 0); }

// work_is_scratch: the caller never looks at the working instance afterwards (the steppers' own programs); the ABI call
// leaves it as the reference does (the first pass's Laplacians, DSS'ed).
static int h_step_after_subcycle_impl(tmx_engine * e, int iinit, int iupd, int iwork, double dt, bool work_is_scratch);
extern "C" int tmx_h_step_after_subcycle(tmx_engine * e, int iinit, int iupd, int iwork, double dt) {
	return h_step_after_subcycle_impl(e, iinit, iupd, iwork, dt, false);
}
static int h_step_after_subcycle_impl(tmx_engine * e, int iinit, int iupd, int iwork, double dt, bool work_is_scratch) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd)) || (r = check_inst(e, iwork))) return r;
	// preconditions of the reference (HorizontalDynamicsFEM.cpp:2648-2655)
	REQUIRE(iinit != iwork, TMX_ERR_INVALID, "StepAfterSubCycle: initial and working data must be distinct");
	REQUIRE(iupd != iwork, TMX_ERR_INVALID, "StepAfterSubCycle: working and update data must be distinct");
	const tmx_config & c = e->cfg;
	KParams p = make_params(e);
	if ((c.nu_scalar == 0.0 && c.nu_div == 0.0 && c.nu_vort == 0.0) || c.hypervis_order == 0) {
		if ((r = tmx_copy_data(e, iinit, iupd))) return r;
	} else if (c.hypervis_order == 2) {
		if ((r = surface_copy(e, iinit, iupd))) return r;       // CopyData(initial -> update), :2663-2664
		// viscosity (HorizontalDynamicsFEM.cpp:2672-2686): one pass from the initial instance, coefficients not scaled
		// with the grid spacing; the scalar part runs with +dt and the vector part with -dt, which the kernel's single
		// dt expresses exactly through the sign of nu_scalar ((-dt) * (-nu) is the same product)
		{ ProfScope ps(e, TMX_K_HYPERVIS); tmxk_hypervis(e, p, inst(e, iinit), inst(e, iinit), inst(e, iupd), -dt, -c.nu_scalar, c.nu_div, c.nu_vort, 0);
		  if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iinit), inst(e, iinit), inst(e, iupd), dt, c.nu_scalar, 1, 0); }
		if ((r = launch_check("viscosity pass"))) return r;
		if ((r = tmx_apply_dss(e, iupd))) return r;
	} else {
		// second pass: coefficients scaled with the patch's (delta_alpha / reference length)^3.2, per column (G2_NUS)
		const int scale = (c.reference_length != 0.0) ? 1 : 0;
		if ((r = surface_copy(e, iinit, iupd)) || (r = surface_zero(e, iwork))) return r;     // CopyData :2663, ZeroData :2693
		// Experiment (TMX_HVIS_PULL=1, judge's "node-unique" go / no-go): the DSS between the two passes is not run as a pass
		// of its own, the second pass averages the first pass's Laplacians while it loads them (k_hypervis<PULL>); on several
		// ranks the raw Laplacians of the rank boundary still travel, and the second pass starts once they have arrived.
		// Bit-identical, but a NO-GO: at ne30 L30 the second pass takes 166 instead of 67 us and moves 703 instead of 308 MB (every
		// seam node gathers its one to three partner values per field from other elements' rows: n^2 instead of n loads per
		// group, 8 bytes per lane from up to 30 different cache lines per wavefront instruction, not L2 hits) against the 62 us
		// and 248 MB of the DSS pass it replaces (profiles/r03_dss_pull_ab.txt).  Without tracers only (k_hypervis_tracers has no such form).
		const bool pull = TMX_EXP && e->hvis_pull && e->nt == 0 && work_is_scratch;      // (experiments flavour only)
		if (stage_can_split(e) && !e->sw) {
			// each pass boundary tiles first, its exchange overlapped with the interior tiles (see hv_stage_split)
			for (int pass = 0; pass < 2; pass++) {
				const int idst = pass ? iupd : iwork;
				bool overlapped = false;
				for (int part = 0; part < 2; part++) {
					p.quads = part ? e->d_quads_late : e->d_quads_early;
					e->launch_tiles = part ? e->n_quads_late : e->n_quads_early;
					{
						ProfScope ps(e, TMX_K_HYPERVIS);
						if (pass == 0) {
							tmxk_hypervis(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 1.0, 1.0, 0);
							if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 0, 0);
						} else {
							tmxk_hypervis(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, c.nu_div, c.nu_vort, scale, pull);
							if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, 1, scale);
						}
					}
					if (part == 0) { p.quads = nullptr; if ((r = exchange(e, p, inst(e, idst), &overlapped))) return r; }
				}
				p.quads = nullptr;
				if ((r = launch_check("hypervis pass (split)"))) return r;
				if (pass == 0 && pull) {
					if (overlapped) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_recv, 0));      // the neighbours' Laplacians are in the ghost buffer
				} else if ((r = dss_after_exchange(e, p, idst, overlapped))) return r;
			}
		}
#if TMX_EXP
		else if (e->hvis_block && e->nt == 0 && !e->sw && e->n_hvblocks > 0) {
			// both passes fused with the DSS of the seams inside a patch (k_hypervis_block); k_dss only for the groups that span patches
			for (int pass = 0; pass < 2; pass++) {
				const int idst = pass ? iupd : iwork;
				{
					ProfScope ps(e, TMX_K_HYPERVIS);
					if (pass == 0) tmxk_hypervis_block(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 1.0, 1.0, 0);
					else tmxk_hypervis_block(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, c.nu_div, c.nu_vort, scale);
				}
				if ((r = launch_check("hypervis pass (fused with the in-patch DSS)"))) return r;
				bool overlapped = false;
				if ((r = exchange(e, p, inst(e, idst), &overlapped))) return r;
				if ((r = dss_after_exchange(e, p, idst, overlapped, e->ngroups_inpatch))) return r;
			}
		}
#endif
		else {
			{ ProfScope ps(e, TMX_K_HYPERVIS); tmxk_hypervis(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 1.0, 1.0, 0);
			  if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 0, 0); }
			if ((r = launch_check("hypervis pass 1"))) return r;
			if (pull) {
				bool overlapped = false;
				if ((r = exchange(e, p, inst(e, iwork), &overlapped))) return r;
				if (overlapped) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_recv, 0));
			} else if ((r = tmx_apply_dss(e, iwork))) return r;
			{ ProfScope ps(e, TMX_K_HYPERVIS); tmxk_hypervis(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, c.nu_div, c.nu_vort, scale, pull);
			  if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, 1, scale); }
			if ((r = launch_check("hypervis pass 2"))) return r;
			if ((r = tmx_apply_dss(e, iupd))) return r;
		}
	}
	// APPLY_RAYLEIGH_WITH_HYPERVIS (Defines.h:70; HorizontalDynamicsFEM.cpp:2719-2724)
	if (e->rayleigh) {
		ProfScope ps(e, TMX_K_HYPERVIS);
		tmxk_rayleigh(e, p, inst(e, iupd), dt);
		return launch_check("rayleigh friction");
	}
	return TMX_OK;
}

// HorizontalDynamics::GetSubStepAfterSubCycleCount / SubStepAfterSubCycle (HorizontalDynamicsFEM.cpp:2574-2633): the two
// halves of the hyperviscosity step WITHOUT the DSS calls -- the caller (Model::SubStep, Model.cpp:286-) exchanges
// between them.  Sub-step 0: working <- Laplacians of initial; sub-step 1: update <- initial - dt nu Laplacians of
// working, tracer filter, Rayleigh friction.  *result receives the instance that holds the sub-step's output.
extern "C" int tmx_h_substep_after_subcycle_count(tmx_engine * e) {
	if (!e) return -1;
	return e->cfg.hypervis_order / 2;
}

extern "C" int tmx_h_substep_after_subcycle(tmx_engine * e, int iinit, int iupd, int iwork, double dt, int isubstep, int * result) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd)) || (r = check_inst(e, iwork))) return r;
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "SubStepAfterSubCycle with the shallow-water equation set is not supported");
	REQUIRE(isubstep == 0 || isubstep == 1, TMX_ERR_INVALID, "Invalid iSubStep %d", isubstep);
	REQUIRE(iinit != iwork && iupd != iwork, TMX_ERR_INVALID, "SubStepAfterSubCycle: working data must be distinct from initial and update data");
	const tmx_config & c = e->cfg;
	KParams p = make_params(e);
	ProfScope ps(e, TMX_K_HYPERVIS);
	if (isubstep == 0) {
		tmxk_hypervis(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 1.0, 1.0, 0);
		if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iinit), nullptr, inst(e, iwork), 1.0, 1.0, 0, 0);
		if (result) *result = iwork;
		return launch_check("hypervis sub-step 0");
	}
	REQUIRE(iinit != iupd, TMX_ERR_INVALID, "SubStepAfterSubCycle: initial and update data must be distinct");
	const int scale = (c.reference_length != 0.0) ? 1 : 0;
	tmxk_hypervis(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, c.nu_div, c.nu_vort, scale);
	if (e->nt > 0) tmxk_hypervis_tracers(e, p, inst(e, iwork), inst(e, iinit), inst(e, iupd), -dt, c.nu_scalar, 1, scale);
	if (e->rayleigh) tmxk_rayleigh(e, p, inst(e, iupd), dt);
	if (result) *result = iupd;
	return launch_check("hypervis sub-step 1");
}

static bool hypervis_active(const tmx_engine * e) {
	const tmx_config & c = e->cfg;
	return !((c.nu_scalar == 0.0 && c.nu_div == 0.0 && c.nu_vort == 0.0) || c.hypervis_order == 0);
}

// ARS(3,4,3) tableau in U-form (TimestepSchemeARS343.cpp:25-146)
static void ars343_tableau(double * u2, double * u3, double * u4, double * dexp, double * dimp) {
	const double gm = 0.4358665215084590;
	const double b1 = -1.5 * gm * gm + 4.0 * gm - 0.25;
	const double b2 = 1.5 * gm * gm - 5.0 * gm + 1.25;
	const double a42 = 0.5529291480359398, a43 = 0.5529291480359398;
	const double a31 = (1.0 - 4.5 * gm + 1.5 * gm * gm) * a42 + (2.75 - 10.5 * gm + 3.75 * gm * gm) * a43 - 3.5 + 13 * gm - 4.5 * gm * gm;
	const double a32 = (-1.0 + 4.5 * gm - 1.5 * gm * gm) * a42 + (-2.75 + 10.5 * gm - 3.75 * gm * gm) * a43 + 4.0 - 12.5 * gm + 4.5 * gm * gm;
	const double a41 = 1.0 - a42 - a43;
	const double I[4][4] = { { gm, 0, 0, 0 }, { 0.5 * (1.0 - gm), gm, 0, 0 }, { b1, b2, gm, 0 }, { b1, b2, gm, 0 } };
	const double E[4][4] = { { gm, 0, 0, 0 }, { a31, a32, 0, 0 }, { a41, a42, a43, 0 }, { 0, b1, b2, gm } };
	for (int i = 0; i < 4; i++) { dexp[i] = E[i][i]; dimp[i] = I[i][i]; }
	for (int i = 0; i < 7; i++) u2[i] = u3[i] = u4[i] = 0.0;
	u2[0] = 1.0 - E[1][0] / E[0][0]; u2[1] = E[1][0] / E[0][0] - I[1][0] / I[0][0]; u2[2] = I[1][0] / I[0][0];
	u3[0] = 1.0 - E[2][0] / E[0][0]; u3[1] = E[2][0] / E[0][0] - I[2][0] / I[0][0]; u3[2] = I[2][0] / I[0][0];
	u3[3] = E[2][1] / E[1][1] - I[2][1] / I[1][1]; u3[4] = I[2][1] / I[1][1];
	const double u37 = -E[2][1] / E[1][1];
	u4[0] = 1.0 - E[3][0] / E[0][0]; u4[1] = E[3][0] / E[0][0] - I[3][0] / I[0][0]; u4[2] = I[3][0] / I[0][0];
	u4[3] = E[3][1] / E[1][1] - I[3][1] / I[1][1]; u4[4] = I[3][1] / I[1][1];
	u4[5] = E[3][2] / E[2][2] - I[3][2] / I[2][2]; u4[6] = I[3][2] / I[2][2];
	const double u47 = -E[3][1] / E[1][1], u48 = -E[3][2] / E[2][2];
	u3[0] += u37 * u2[0]; u3[1] += u37 * u2[1]; u3[2] += u37 * u2[2];
	u4[0] += u47 * u2[0] + u48 * u3[0]; u4[1] += u47 * u2[1] + u48 * u3[1]; u4[2] += u47 * u2[2] + u48 * u3[2];
	u4[3] += u48 * u3[3]; u4[4] += u48 * u3[4];
}

// ---------------------------------------------------------------------------------------------
// Time steppers.  Every reference TimestepScheme::Step on this path is a fixed sequence of the same
// few Grid / HorizontalDynamics / VerticalDynamics calls; it is written down here call for call as a
// small program, and run_program folds neighbouring calls into one kernel pass where the result is
// bit-identical (see the rules there).

enum OpKind { OP_COPY, OP_LINCOMB, OP_HEXP, OP_VEXP, OP_DSS, OP_VIMP, OP_VITE, OP_HVIS, OP_VFILT };
struct Op {
	int kind, a, b, c;
	double coef;
	double lc[12];
	int nlc;
};
typedef std::vector<Op> Program;

static void P_copy(Program & p, int a, int b) { Op o = { OP_COPY, a, b, 0, 0.0, { 0 }, 0 }; p.push_back(o); }
static void P_h(Program & p, int i, int u, double c) { Op o = { OP_HEXP, i, u, 0, c, { 0 }, 0 }; p.push_back(o); }
static void P_v(Program & p, int i, int u, double c) { Op o = { OP_VEXP, i, u, 0, c, { 0 }, 0 }; p.push_back(o); }
static void P_hv(Program & p, int i, int u, double c) { P_h(p, i, u, c); P_v(p, i, u, c); }
static void P_dss(Program & p, int i) { Op o = { OP_DSS, i, 0, 0, 0.0, { 0 }, 0 }; p.push_back(o); }
static void P_vimp(Program & p, int i, int u, double c) { Op o = { OP_VIMP, i, u, 0, c, { 0 }, 0 }; p.push_back(o); }
static void P_vfilt(Program & p, int u) { Op o = { OP_VFILT, u, u, 0, 0.0, { 0 }, 0 }; p.push_back(o); }
static void P_vite(Program & p, int i, int u, double c) { Op o = { OP_VITE, i, u, 0, c, { 0 }, 0 }; p.push_back(o); }
static void P_hvis(Program & p, int i, int u, int w, double c) { Op o = { OP_HVIS, i, u, w, c, { 0 }, 0 }; p.push_back(o); }
static void P_lc(Program & p, const double * cf, int n, int dst) {
	Op o = { OP_LINCOMB, dst, 0, 0, 0.0, { 0 }, n };
	for (int i = 0; i < n; i++) o.lc[i] = cf[i];
	p.push_back(o);
}

static int scheme_instances(int scheme) {
	switch (scheme) {
		case TMX_SCHEME_ARS343: return 7;    // TimestepSchemeARS343.h:48-56
		case TMX_SCHEME_ARS232: return 7;    // TimestepSchemeARS232.h
		case TMX_SCHEME_ARS222: return 4;
		case TMX_SCHEME_ARS443: return 10;
		case TMX_SCHEME_ARK232: return 8;
		case TMX_SCHEME_STRANG: case TMX_SCHEME_STRANG_FE: case TMX_SCHEME_STRANG_RK4: case TMX_SCHEME_STRANG_SSP3:
		case TMX_SCHEME_STRANG_SSPRK53: return 5;    // TimestepSchemeStrang.h GetComponentDataInstances
	}
	return -1;
}

// U-form combination coefficients shared by the ARS schemes: u_f(stage) from explicit/implicit tableaux
static int build_program(int scheme, int first, int last, double dt, Program & p, double offc = 0.0) {
	p.clear();
	if (scheme == TMX_SCHEME_ARS343) {
		// TimestepSchemeARS343::Step (TimestepSchemeARS343.cpp:150-236)
		double u2[7], u3[7], u4[7], de[4], di[4];
		ars343_tableau(u2, u3, u4, de, di);
		P_copy(p, 0, 1); P_hv(p, 0, 1, de[0] * dt); P_dss(p, 1);
		P_copy(p, 1, 2); P_vimp(p, 2, 2, di[0] * dt);
		P_lc(p, u2, 7, 3); P_hv(p, 2, 3, de[1] * dt); P_dss(p, 3);
		P_copy(p, 3, 4); P_vimp(p, 4, 4, di[1] * dt);
		P_lc(p, u3, 7, 5); P_hv(p, 4, 5, de[2] * dt); P_dss(p, 5);
		P_copy(p, 5, 6); P_vimp(p, 6, 6, di[2] * dt);
		P_lc(p, u4, 7, 1); P_hv(p, 6, 1, de[3] * dt); P_dss(p, 1);
		P_copy(p, 1, 0); P_hvis(p, 1, 0, 2, dt);
		return TMX_OK;
	}
	if (scheme == TMX_SCHEME_ARS232) {
		// TimestepSchemeARS232::Step (TimestepSchemeARS232.cpp:25-150)
		const double g = 1.0 - 1.0 / std::sqrt(2.0), d = -(2.0 * std::sqrt(2.0)) / 3.0;
		const double I[3][3] = { { g, 0, 0 }, { 1.0 - g, g, 0 }, { 1.0 - g, g, 0 } };
		const double E[3][3] = { { g, 0, 0 }, { d, 1.0 - d, 0 }, { 0, 1.0 - g, g } };
		double u2[6] = { 1.0 - E[1][0] / E[0][0], E[1][0] / E[0][0] - I[1][0] / I[0][0], I[1][0] / I[0][0], 0, 0, 0 };
		double u3[7] = { 1.0 - E[2][0] / E[0][0], E[2][0] / E[0][0] - I[2][0] / I[0][0], I[2][0] / I[0][0],
			E[2][1] / E[1][1] - I[2][1] / I[1][1], I[2][1] / I[1][1], -E[2][1] / E[1][1], 0 };
		P_copy(p, 0, 1); P_hv(p, 0, 1, E[0][0] * dt); P_dss(p, 1);
		P_copy(p, 1, 2); P_vimp(p, 2, 2, I[0][0] * dt);
		P_lc(p, u2, 6, 5); P_copy(p, 5, 3); P_hv(p, 2, 3, E[1][1] * dt); P_dss(p, 3);
		P_copy(p, 3, 4); P_vimp(p, 4, 4, I[1][1] * dt);
		P_lc(p, u3, 7, 6); P_hv(p, 4, 6, E[2][2] * dt); P_dss(p, 6);
		P_copy(p, 6, 2); P_hvis(p, 2, 1, 6, dt); P_copy(p, 1, 0);
		return TMX_OK;
	}
	if (scheme == TMX_SCHEME_ARS222) {
		// TimestepSchemeARS222::Step (TimestepSchemeARS222.cpp:25-119)
		const double g = 1.0 - 0.5 * std::sqrt(2.0), d = 1.0 - 1.0 / (2.0 * g);
		const double I[2][2] = { { g, 0 }, { 1.0 - g, g } };
		const double E[2][2] = { { g, 0 }, { d, 1.0 - d } };
		double u2[4] = { 1.0 - E[1][0] / E[0][0], E[1][0] / E[0][0] - I[1][0] / I[0][0], I[1][0] / I[0][0], 0 };
		P_copy(p, 0, 1); P_hv(p, 0, 1, E[0][0] * dt); P_dss(p, 1);
		P_copy(p, 1, 2); P_vimp(p, 2, 2, I[0][0] * dt); P_dss(p, 2);
		P_lc(p, u2, 4, 3); P_hv(p, 2, 3, E[1][1] * dt); P_dss(p, 3);
		P_vimp(p, 3, 3, I[1][1] * dt); P_dss(p, 3);
		P_copy(p, 3, 2); P_hvis(p, 2, 1, 3, dt); P_copy(p, 1, 0);
		return TMX_OK;
	}
	if (scheme == TMX_SCHEME_STRANG || (scheme >= TMX_SCHEME_STRANG_FE && scheme <= TMX_SCHEME_STRANG_SSPRK53)) {
		// TimestepSchemeStrang::Step (TimestepSchemeStrang.cpp:450-674) with each of its explicit discretisations
		// (:39-51, combinations :53-117) and the off-centering of its constructor
		const double half = 0.5 * dt, oc = offc;
		const double carry[2] = { 1.0, 1.0 }, offcomb[2] = { (2.0 - oc) / 2.0, oc / 2.0 }, fin[2] = { 1.0, -1.0 };
		if (first) P_vimp(p, 0, 0, half); else { P_lc(p, carry, 2, 0); P_vfilt(p, 0); }      // FilterNegativeTracers(0), :476-481
		if (scheme == TMX_SCHEME_STRANG_FE) {
			P_copy(p, 0, 4); P_hv(p, 0, 4, dt); P_dss(p, 4);
		} else if (scheme == TMX_SCHEME_STRANG_RK4) {
			const double rk4[5] = { -1.0 / 3.0, 1.0 / 3.0, 2.0 / 3.0, 1.0 / 3.0, 0.0 };
			P_copy(p, 0, 1); P_hv(p, 0, 1, half); P_dss(p, 1);
			P_copy(p, 0, 2); P_hv(p, 1, 2, half); P_dss(p, 2);
			P_copy(p, 0, 3); P_hv(p, 2, 3, dt); P_dss(p, 3);
			P_lc(p, rk4, 5, 4); P_hv(p, 3, 4, dt / 6.0); P_dss(p, 4);
		} else if (scheme == TMX_SCHEME_STRANG_SSP3) {
			const double a[3] = { 3.0 / 4.0, 1.0 / 4.0, 0.0 }, b[5] = { 1.0 / 3.0, 0.0, 2.0 / 3.0, 0.0, 0.0 };
			P_copy(p, 0, 1); P_hv(p, 0, 1, dt); P_dss(p, 1);
			P_lc(p, a, 3, 2); P_hv(p, 1, 2, 0.25 * dt); P_dss(p, 2);
			P_lc(p, b, 5, 4); P_hv(p, 2, 4, (2.0 / 3.0) * dt); P_dss(p, 4);
		} else if (scheme == TMX_SCHEME_STRANG_SSPRK53) {
			const double a[4] = { 0.355909775063327, 0.0, 0.644090224936674, 0.0 };
			const double b[4] = { 0.367933791638137, 0.0, 0.0, 0.632066208361863 };
			const double c[5] = { 0.762406163401431, 0.0, 0.237593836598569, 0.0, 0.0 };
			P_copy(p, 0, 1); P_hv(p, 0, 1, 0.377268915331368 * dt); P_dss(p, 1);
			P_copy(p, 1, 2); P_hv(p, 1, 2, 0.377268915331368 * dt); P_dss(p, 2);
			P_lc(p, a, 4, 3); P_hv(p, 2, 3, 0.242995220537396 * dt); P_dss(p, 3);
			P_lc(p, b, 4, 0); P_hv(p, 3, 0, 0.238458932846290 * dt); P_dss(p, 0);
			P_lc(p, c, 5, 4); P_hv(p, 0, 4, 0.287632146308408 * dt); P_dss(p, 4);
		} else {
			const double kgu[5] = { -1.0 / 4.0, 5.0 / 4.0, 0.0, 0.0, 0.0 };
			P_copy(p, 0, 1); P_hv(p, 0, 1, dt / 5.0); P_dss(p, 1);
			P_copy(p, 0, 2); P_hv(p, 1, 2, dt / 5.0); P_dss(p, 2);
			P_copy(p, 0, 3); P_hv(p, 2, 3, dt / 3.0); P_dss(p, 3);
			P_copy(p, 0, 2); P_hv(p, 3, 2, 2.0 * dt / 3.0); P_dss(p, 2);
			P_lc(p, kgu, 5, 4); P_hv(p, 2, 4, 3.0 * dt / 4.0); P_dss(p, 4);
		}
		P_copy(p, 4, 1); P_hvis(p, 4, 1, 2, dt);
		P_copy(p, 1, 0); P_vimp(p, 0, 0, 0.5 * (1.0 + oc) * dt);
		P_lc(p, offcomb, 2, 0);
		if (!last) P_lc(p, fin, 2, 1);
		return TMX_OK;
	}
	if (scheme == TMX_SCHEME_ARK232) {
		// TimestepSchemeARK232::Step with its two sub-cycled stages (TimestepSchemeARK232.cpp:25-225)
		const double g = 1.0 - 1.0 / std::sqrt(2.0), d = 1.0 / (2.0 * std::sqrt(2.0));
		const double al = 1.0 / 6.0 * (3.0 + 2.0 * std::sqrt(2.0));
		const double I[3][3] = { { g, g, 0 }, { d, d, g }, { d, d, g } };
		const double E[3][3] = { { 2.0 * g, 0, 0 }, { 1.0 - al, al, 0 }, { d, d, g } };
		double u2[7] = { 1.0 - E[1][0] / E[0][0], E[1][0] / E[0][0] - I[1][0] / I[0][0],
			I[1][0] / I[0][0] - I[1][1] / I[0][1], I[1][1] / I[0][1], 0, 0, 0 };
		double u3[8] = { 1.0 - E[2][0] / E[0][0], E[2][0] / E[0][0] - I[2][0] / I[0][0],
			I[2][0] / I[0][0] - I[2][1] / I[0][1], I[2][1] / I[0][1],
			E[2][1] / E[1][1] - I[2][2] / I[1][2], I[2][2] / I[1][2], -E[2][1] / E[1][1], 0 };
		// SubcycleStageExplicit(coeff E00, 2 sub-steps, 0 -> 1)
		for (int n = 0; n < 2; n++) {
			P_copy(p, 0, 1); P_hv(p, 0, 1, E[0][0] * dt / 2); P_dss(p, 1);
			if (n < 1) P_copy(p, 1, 0);
		}
		// SubcycleStageImplicitExplicitly(coeff I00, 1 sub-step, 1 -> 2)
		P_copy(p, 1, 2); P_vite(p, 1, 2, I[0][0] * dt); P_dss(p, 2);
		P_copy(p, 2, 3); P_vimp(p, 3, 3, I[0][1] * dt); P_dss(p, 3);
		P_lc(p, u2, 7, 6); P_copy(p, 6, 4); P_hv(p, 3, 4, E[1][1] * dt); P_dss(p, 4);
		P_copy(p, 4, 5); P_vimp(p, 5, 5, I[1][2] * dt); P_dss(p, 5);
		P_lc(p, u3, 8, 7); P_hv(p, 5, 7, E[2][2] * dt); P_dss(p, 7);
		P_copy(p, 7, 2); P_hvis(p, 7, 1, 3, dt); P_copy(p, 1, 0);
		return TMX_OK;
	}
	if (scheme == TMX_SCHEME_ARS443) {
		// TimestepSchemeARS443::Step (TimestepSchemeARS443.cpp)
		const double I[4][4] = { { 0.5, 0, 0, 0 }, { 1.0 / 6.0, 0.5, 0, 0 }, { -0.5, 0.5, 0.5, 0 }, { 1.5, -1.5, 0.5, 0.5 } };
		const double E[4][4] = { { 0.5, 0, 0, 0 }, { 11.0 / 18.0, 1.0 / 18.0, 0, 0 }, { 5.0 / 6.0, -5.0 / 6.0, 0.5, 0 }, { 0.25, 1.75, 0.75, -1.75 } };
		double u2[8], u3[9], u4[10];
		for (int i = 0; i < 8; i++) u2[i] = 0; for (int i = 0; i < 9; i++) u3[i] = 0; for (int i = 0; i < 10; i++) u4[i] = 0;
		u2[0] = 1.0 - E[1][0] / E[0][0]; u2[1] = E[1][0] / E[0][0] - I[1][0] / I[0][0]; u2[2] = I[1][0] / I[0][0];
		u3[0] = 1.0 - E[2][0] / E[0][0]; u3[1] = E[2][0] / E[0][0] - I[2][0] / I[0][0]; u3[2] = I[2][0] / I[0][0];
		u3[3] = E[2][1] / E[1][1] - I[2][1] / I[1][1]; u3[4] = I[2][1] / I[1][1]; u3[7] = -E[2][1] / E[1][1];
		u4[0] = 1.0 - E[3][0] / E[0][0]; u4[1] = E[3][0] / E[0][0] - I[3][0] / I[0][0]; u4[2] = I[3][0] / I[0][0];
		u4[3] = E[3][1] / E[1][1] - I[3][1] / I[1][1]; u4[4] = I[3][1] / I[1][1];
		u4[5] = E[3][2] / E[2][2] - I[3][2] / I[2][2]; u4[6] = I[3][2] / I[2][2];
		u4[7] = -E[3][1] / E[1][1]; u4[8] = -E[3][2] / E[2][2];
		P_copy(p, 0, 1); P_hv(p, 0, 1, E[0][0] * dt); P_dss(p, 1);
		P_copy(p, 1, 2); P_vimp(p, 2, 2, I[0][0] * dt); P_dss(p, 2);
		P_lc(p, u2, 8, 7); P_copy(p, 7, 3); P_hv(p, 2, 3, E[1][1] * dt); P_dss(p, 3);
		P_copy(p, 3, 4); P_vimp(p, 4, 4, I[1][1] * dt); P_dss(p, 4);
		P_lc(p, u3, 9, 8); P_copy(p, 8, 5); P_hv(p, 4, 5, E[2][2] * dt); P_dss(p, 5);
		P_copy(p, 5, 6); P_vimp(p, 6, 6, I[2][2] * dt); P_dss(p, 6);
		P_lc(p, u4, 10, 9); P_hv(p, 6, 9, E[3][3] * dt); P_dss(p, 9);
		P_vimp(p, 9, 9, I[3][3] * dt); P_dss(p, 9);
		P_copy(p, 9, 2); P_hvis(p, 2, 1, 9, dt); P_copy(p, 1, 0);
		return TMX_OK;
	}
	tmx_set_error("unknown time scheme %d", scheme);
	return TMX_ERR_INVALID;
}

extern "C" int tmx_v_step_implicit_terms_explicitly(tmx_engine * e, int iinit, int iupd, double dt) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, iinit)) || (r = check_inst(e, iupd))) return r;
	REQUIRE(iinit != iupd && dt != 0.0, TMX_ERR_INVALID, "StepImplicitTermsExplicitly: distinct instances and non-zero dt required");
	ProfScope ps(e, TMX_K_VI_ASSEMBLE);
	tmxk_vi_terms_explicit(e, make_params(e), inst(e, iinit), inst(e, iupd), dt, false);
	if (e->nt > 0)      // UpdateColumnTracers(dt, initial, update, ...) of every column, :600-608
		REQUIRE(tmxk_vi_tracers_all(e, make_params(e), inst(e, iinit), inst(e, iupd), dt) == 0, TMX_ERR_UNSUPPORTED,
			"tracer column update: %d levels do not fit the LDS working set", e->L);
	return launch_check("v_step_implicit_terms_explicitly");
}
// ---------------------------------------------------------------------------------------------
// output interpolation (Grid::ReduceInterpolate)

struct tmx_interp {
	tmx_engine * owner = nullptr;
	int npts = 0, nreta = 0;
	bool has_rll = false;
	int * d_col0 = nullptr;
	double * d_ca = nullptr, * d_cb = nullptr, * d_rll = nullptr, * d_opn = nullptr, * d_ope = nullptr, * d_out = nullptr;
	size_t out_n = 0;
};

// plans outlive their engine safely: tmx_destroy clears `owner` of every plan the engine still lists
void interp_orphan(tmx_engine * e) {
	for (tmx_interp * q : e->interps) q->owner = nullptr;
	e->interps.clear();
}

extern "C" void tmx_interp_destroy(tmx_interp * q) {
	if (!q) return;
	if (q->owner) {
		if (q->owner->stream) hipStreamSynchronize(q->owner->stream);
		auto & v = q->owner->interps;
		v.erase(std::remove(v.begin(), v.end(), q), v.end());
	}
	hipFree(q->d_col0); hipFree(q->d_ca); hipFree(q->d_cb); hipFree(q->d_rll); hipFree(q->d_opn); hipFree(q->d_ope); hipFree(q->d_out);
	delete q;
}

extern "C" int tmx_interp_create(tmx_engine * e, const tmx_interp_points * pts, tmx_interp ** out) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(pts && out, TMX_ERR_INVALID, "tmx_interp_create: null argument");
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "output interpolation with the shallow-water equation set is not supported");
	REQUIRE(pts->n_points > 0 && pts->n_reta > 0 && pts->patch && pts->node_a && pts->node_b && pts->coeff_a && pts->coeff_b &&
		pts->op_levels && pts->op_interfaces, TMX_ERR_INVALID, "tmx_interp_create: incomplete point description");
	const int n = pts->n_points, L = e->L;
	std::vector<int> col0(n, -1);
	for (int i = 0; i < n; i++) {
		const int pi = pts->patch[i];
		REQUIRE(pi >= 0 && pi < e->cfg.n_patches, TMX_ERR_INVALID, "interpolation point %d: patch %d out of range", i, pi);
		const PatchInfo & P = e->patches[pi];
		if (P.owner != e->cfg.rank) continue;
		const int a = pts->node_a[i], b = pts->node_b[i];
		REQUIRE(a >= 1 && b >= 1 && a + TMX_NP <= P.na - 1 && b + TMX_NP <= P.nb - 1 && (a - 1) % TMX_NP == 0 && (b - 1) % TMX_NP == 0,
			TMX_ERR_INVALID, "interpolation point %d: (%d, %d) is not the first node of an element of patch %d", i, a, b, pi);
		col0[i] = col_of(P, a, b);
	}
	tmx_interp * q = new tmx_interp();
	q->owner = e; q->npts = n; q->nreta = pts->n_reta; q->has_rll = pts->rll_from_abp != nullptr;
	auto up = [&](auto ** d, const auto * hsrc, size_t cnt) -> bool {
		if (hipMalloc((void **)d, cnt * sizeof(**d)) != hipSuccess) return false;
		return hipMemcpy(*d, hsrc, cnt * sizeof(**d), hipMemcpyHostToDevice) == hipSuccess;
	};
	bool ok = up(&q->d_col0, col0.data(), (size_t)n) && up(&q->d_ca, pts->coeff_a, (size_t)n * 4) && up(&q->d_cb, pts->coeff_b, (size_t)n * 4) &&
		up(&q->d_opn, pts->op_levels, (size_t)pts->n_reta * L) && up(&q->d_ope, pts->op_interfaces, (size_t)pts->n_reta * (L + 1));
	if (ok && q->has_rll) ok = up(&q->d_rll, pts->rll_from_abp, (size_t)n * 4);
	q->out_n = (size_t)std::max(5, e->nt) * pts->n_reta * n;
	if (ok) ok = hipMalloc((void **)&q->d_out, q->out_n * sizeof(double)) == hipSuccess;
	if (!ok) { (void)hipGetLastError(); tmx_interp_destroy(q); tmx_set_error("tmx_interp_create: device allocation failed"); return TMX_ERR_DEVICE; }
	e->interps.push_back(q);
	*out = q;
	return TMX_OK;
}

static InterpArgs interp_args(const tmx_interp * q) {
	InterpArgs a;
	a.npts = q->npts; a.nreta = q->nreta; a.col0 = q->d_col0; a.ca = q->d_ca; a.cb = q->d_cb; a.rll = q->d_rll; a.opn = q->d_opn; a.ope = q->d_ope;
	return a;
}

extern "C" int tmx_interp_state(tmx_engine * e, tmx_interp * q, int instance, int only_at, int include_ref, int primitive,
	double earth_radius, double * out) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, instance, true))) return r;
	REQUIRE(q && q->owner == e && out, TMX_ERR_INVALID, "tmx_interp_state: bad plan / null output");
	REQUIRE(only_at >= 0 && only_at <= 2, TMX_ERR_INVALID, "tmx_interp_state: only_variables_at must be 0, 1 or 2");
	REQUIRE(!primitive || q->has_rll, TMX_ERR_INVALID, "tmx_interp_state: convert_to_primitive needs rll_from_abp in the plan");
	REQUIRE(!primitive || earth_radius > 0.0, TMX_ERR_INVALID, "tmx_interp_state: earth_radius must be positive");
	const double * xref = nullptr;
	if (!include_ref) {
		for (int lp : e->local_patches)
			REQUIRE(e->patches[lp].ref_set, TMX_ERR_INVALID, "tmx_interp_state without the reference state: tmx_set_patch_reference_state was not called for patch %d", lp);
		xref = e->d_ref;
	}
	tmxk_interp_state(e, make_params(e), interp_args(q), inst(e, instance), xref, only_at, primitive ? 1 : 0, earth_radius, q->d_out);
	if ((r = launch_check("interp_state"))) return r;
	HIPCHK(hipMemcpyAsync(out, q->d_out, (size_t)5 * q->nreta * q->npts * sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	return TMX_OK;
}

extern "C" int tmx_interp_tracers(tmx_engine * e, tmx_interp * q, int instance, double * out) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, instance, true))) return r;
	REQUIRE(q && q->owner == e && out, TMX_ERR_INVALID, "tmx_interp_tracers: bad plan / null output");
	REQUIRE(e->nt > 0, TMX_ERR_INVALID, "Unable to Interpolate with no tracers.");
	tmxk_interp_tracers(e, make_params(e), interp_args(q), inst(e, instance), q->d_out);
	if ((r = launch_check("interp_tracers"))) return r;
	HIPCHK(hipMemcpyAsync(out, q->d_out, (size_t)e->nt * q->nreta * q->npts * sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	return TMX_OK;
}

extern "C" int tmx_v_filter_negative_tracers(tmx_engine * e, int instance) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, instance))) return r;
	if (e->nt == 0) return TMX_OK;
	ProfScope ps(e, TMX_K_LINCOMB);
	tmxk_v_filter_tracers(e, make_params(e), inst(e, instance));
	return launch_check("v_filter_negative_tracers");
}

static int vi_terms_explicit(tmx_engine * e, int iinit, int iupd, double dt) { return tmx_v_step_implicit_terms_explicitly(e, iinit, iupd, dt); }

static thread_local bool g_hvis_two_pass = true;      // set by run_program from the engine's configuration (order-4 hyperviscosity with a non-zero coefficient)
static bool g_dry_unsafe = false;      // tmx_debug_program_copies' negative control (dry runs only): every share accepted, no copy-on-write
// What an operation of a stepper program reads and writes, as bit masks over the data instances.  full: instances it
// overwrites entirely without reading them; part: instances it updates in place.
struct OpAccess { unsigned reads, full, part; bool stencil; };
// vimp_noop: V.StepImplicit does nothing (fully explicit vertical mode, shallow water's stub)
static OpAccess op_access(const Op & o, bool vimp_noop) {
	OpAccess x = { 0u, 0u, 0u, true };
	auto bit = [](int k) { return 1u << k; };
	switch (o.kind) {
	case OP_COPY: x.reads = bit(o.a); x.full = bit(o.b); break;
	case OP_VIMP: if (vimp_noop) break;      // fall through
	case OP_HEXP: case OP_VEXP: case OP_VITE: x.reads = bit(o.a) | bit(o.b); x.part = bit(o.b); break;
	case OP_DSS: case OP_VFILT: x.reads = bit(o.a); x.part = bit(o.a); break;
	// StepAfterSubCycle rewrites all of b on every branch; the working instance c is written (entirely) by the two-pass
	// branch only -- without viscosity or with one pass it is left alone, so a c that shares a slot keeps its contents
	case OP_HVIS: x.reads = bit(o.a); x.full = bit(o.b) | (g_hvis_two_pass ? bit(o.c) : 0u); break;
	case OP_LINCOMB:
		x.stencil = false;      // node by node: a source may share the destination's slot
		for (int m = 0; m < o.nlc; m++) if (o.lc[m] != 0.0) x.reads |= bit(m);
		if (o.lc[o.a] != 0.0) x.part = bit(o.a); else x.full = bit(o.a);
		break;
	}
	return x;
}

// May instance b read instance a's slot from operation `first` of the program on, in place of a copy of a made just before?
// Yes if, until b is next overwritten entirely, nothing updates b in place, and once a has been written b is not read again
// (b is dead from there on); a stencil operation must not write a while it reads b.  A program that ends with the two still
// identical leaves them shared; one that ends after a was written would leave b undefined, so it is refused.
static bool share_is_safe(const Program & p, size_t first, int a, int b, bool vimp_noop) {
	if (g_dry_unsafe) return true;
	const unsigned ba = 1u << a, bb = 1u << b;
	bool a_written = false;
	for (size_t i = first; i < p.size(); i++) {
		const OpAccess x = op_access(p[i], vimp_noop);
		if (x.part & bb) return false;
		if ((x.reads & bb) && a_written) return false;
		if (((x.full | x.part) & ba) && (x.reads & bb) && x.stencil) return false;
		if (x.full & bb) return true;
		if ((x.full | x.part) & ba) a_written = true;
	}
	return !a_written;
}

// an operation is about to write instance x: give it its own slot back (with its contents unless the write is total); the
// same for U,V slabs it shares
static int own_slot(tmx_engine * e, int x, bool total) {
	if (e->imap[x] != x) {
		if (!total) {
			if (e->dry_run) { e->dry_copies++; if (!g_dry_unsafe) for (int c = 0; c < 2; c++) e->dry_slot[c][x] = e->dry_slot[c][e->imap[x]]; }
			else HIPCHK(hipMemcpyAsync(e->d_state + (size_t)x * e->inst_stride, e->d_state + (size_t)e->imap[x] * e->inst_stride,
				e->inst_stride * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
		}
		e->imap[x] = x; e->n_shared--;
	}
	if (e->uvmap[x] != x) {
		if (!total) {
			if (e->dry_run) { e->dry_copies++; if (!g_dry_unsafe) e->dry_slot[0][x] = e->dry_slot[0][e->uvmap[x]]; }
			else HIPCHK(hipMemcpyAsync(e->d_state + (size_t)x * e->inst_stride, e->d_state + (size_t)e->uvmap[x] * e->inst_stride,
				(size_t)2 * e->L * e->NS * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
		}
		e->uvmap[x] = x; e->n_uvshared--;
	}
	return TMX_OK;
}
// instance x has been written: the instances that read its slot (or its U,V slabs) are dead by share_is_safe and go back to
// their own
static void drop_readers(tmx_engine * e, int x) {
	for (int y = 0; y < (int)e->imap.size(); y++) {
		if (y != x && e->imap[y] == x) { e->imap[y] = y; e->n_shared--; }
		if (y != x && e->uvmap[y] == x) { e->uvmap[y] = y; e->n_uvshared--; }
	}
}

struct ProgramScope {
	tmx_engine * e;
	ProgramScope(tmx_engine * e_) : e(e_) { e->in_program = true; }
	~ProgramScope() { e->in_program = false; }
};
struct AfterOp {      // at the end of an operation: the readers of the slots it wrote
	tmx_engine * e; unsigned w;
	~AfterOp() { for (int x = 0; x < (int)e->imap.size() && (e->n_shared || e->n_uvshared); x++) if (w >> x & 1u) drop_readers(e, x); }
};

// Symbolic model of the dry run.  Value ids stand for array contents: the reference's semantics of every operation is applied to
// dry_ref (per instance, U,V and the rest apart -- the implicit column solve is the one operation that writes only the rest),
// and an executed operation is taken to store the reference's result in the slots of the instances it writes, provided that
// every instance it read was found, through the instance maps, in a slot holding the reference's value: that is what is checked.
static unsigned long long dry_mix(unsigned long long h, unsigned long long v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h * 0xff51afd7ed558ccdull; }
static int dry_slot_of(const tmx_engine * e, int k, int c) { return (c == 0 && e->uvmap[k] != k) ? e->uvmap[k] : e->imap[k]; }
// operations [i0, i1] of the program were executed as one unit
static void dry_model(tmx_engine * e, const Program & p, size_t i0, size_t i1, bool noop) {
	const int ni = (int)e->imap.size();
	unsigned written[2] = { 0u, 0u };
	for (size_t j = i0; j <= i1 && j < p.size(); j++) {
		const Op & o = p[j];
		if (o.kind == OP_VIMP && noop) continue;
		const OpAccess x = op_access(o, noop);
		const bool rest_only = (o.kind == OP_VIMP);            // V.StepImplicit leaves U,V alone
		// reads from outside the unit must find the reference's value where the maps point
		for (int k = 0; k < ni; k++)
			for (int c = 0; c < 2; c++)
				if ((x.reads >> k & 1u) && !(written[c] >> k & 1u) && e->dry_slot[c][dry_slot_of(e, k, c)] != e->dry_ref[c][k]) e->dry_mismatch++;
		// the reference's result
		std::vector<unsigned long long> nr[2] = { e->dry_ref[0], e->dry_ref[1] };
		for (int c = (rest_only ? 1 : 0); c < 2; c++) {
			auto in = [&](int k) { return dry_mix(e->dry_ref[0][k], e->dry_ref[1][k]); };      // a stencil operation's result depends on all of its input instance
			switch (o.kind) {
			case OP_COPY: nr[c][o.b] = e->dry_ref[c][o.a]; break;
			case OP_HEXP: case OP_VEXP: case OP_VITE: case OP_VIMP: nr[c][o.b] = dry_mix(dry_mix(dry_mix(1000 + o.kind, c), in(o.a)), e->dry_ref[c][o.b]); break;
			case OP_DSS: case OP_VFILT: nr[c][o.a] = dry_mix(dry_mix(2000 + o.kind, c), e->dry_ref[c][o.a]); break;
			case OP_HVIS: nr[c][o.b] = dry_mix(dry_mix(3000, c), in(o.a)); nr[c][o.c] = dry_mix(dry_mix(3001, c), in(o.a)); break;
			case OP_LINCOMB: {
				unsigned long long h = dry_mix(4000, c);
				for (int m = 0; m < o.nlc; m++) if (o.lc[m] != 0.0) { unsigned long long b; memcpy(&b, &o.lc[m], 8); h = dry_mix(dry_mix(h, b), e->dry_ref[c][m]); }
				nr[c][o.a] = h; break; }
			}
			written[c] |= x.full | x.part;
		}
		e->dry_ref[0] = nr[0]; e->dry_ref[1] = nr[1];
	}
	// the unit's results land in the slots of the instances it wrote (or must already be there, for an instance left sharing)
	for (int k = 0; k < ni; k++)
		for (int c = 0; c < 2; c++)
			if (written[c] >> k & 1u) {
				const int sl = dry_slot_of(e, k, c);
				if (sl == k) e->dry_slot[c][k] = e->dry_ref[c][k];
				else if (e->dry_slot[c][sl] != e->dry_ref[c][k]) e->dry_mismatch++;
			}
}

// RUN: the launches of an operation; a dry run (tmx_debug_program_copies: the copy / sharing decisions of a program on the
// host, no device) skips them
#define RUN(call) (e->dry_run ? TMX_OK : (call))
static int run_program(tmx_engine * e, const Program & p) {
	int r;
	const size_t n = p.size();
	ProgramScope scope(e);
	const bool noop = e->sw || e->fully_explicit;      // V.StepImplicit does nothing
	g_hvis_two_pass = e->dry_run || (hypervis_active(e) && e->cfg.hypervis_order != 2);
	// shared slots left by the previous step: kept where this program allows it, given their own copy otherwise
	for (int b = 0; b < (int)e->imap.size() && (e->n_shared || e->n_uvshared); b++)
		if ((e->imap[b] != b && !share_is_safe(p, 0, e->imap[b], b, noop)) || (e->uvmap[b] != b && !share_is_safe(p, 0, e->uvmap[b], b, noop)))
			if ((r = own_slot(e, b, false))) return r;
	for (size_t i = 0; i < n; i++) {
		const Op & o = p[i];
		const Op * n1 = (i + 1 < n) ? &p[i + 1] : nullptr;
		const Op * n2 = (i + 2 < n) ? &p[i + 2] : nullptr;
		// the instances this operation (and the ones fused with it below, which write the same instance) writes
		const OpAccess acc = op_access(o, noop);
		for (int x = 0; x < (int)e->imap.size() && (e->n_shared || e->n_uvshared); x++)
			if (((acc.full | acc.part) >> x & 1u) && (r = own_slot(e, x, (acc.full >> x & 1u) != 0))) return r;
		AfterOp after = { e, acc.full | acc.part };
		const size_t i_first = i;
		struct DryUnit { tmx_engine * e; const Program & p; size_t i0; const size_t & i1; bool noop;
			~DryUnit() { if (e->dry_run) dry_model(e, p, i0, i1, noop); } } dry_unit = { e, p, i_first, i, noop };      // runs before `after`
		switch (o.kind) {
		case OP_COPY:
			// Copy(a->b) + H(a,b,c) + V(a,b,c)  ==  one pass  b = a + c * rhs(a)
			if (!e->sw && n1 && n2 && n1->kind == OP_HEXP && n2->kind == OP_VEXP && n1->b == o.b && n2->a == n1->a &&
			    n2->b == o.b && n1->coef == n2->coef && n1->a != o.b) {
				// ... + DSS(b) on several ranks: boundary tiles first, exchange overlapped with the interior tiles
				if (stage_can_split(e) && i + 3 < n && p[i + 3].kind == OP_DSS && p[i + 3].a == o.b) {
					if ((r = RUN(hv_stage_split(e, n1->a, o.a, o.b, n1->coef, nullptr, 0)))) return r;
					i += 3; break;
				}
				if ((r = RUN(hv_step_explicit(e, n1->a, o.a, o.b, n1->coef)))) return r;
				i += 2; break;
			}
			// shallow water: Copy(a->b) + H(a,b,c) [+ V stub]: out = a + c * rhs(a) in one pass (k_sw_explicit takes the base apart)
			if (e->sw && e->nt == 0 && n1 && n2 && n1->kind == OP_HEXP && n2->kind == OP_VEXP && n1->b == o.b && n2->a == n1->a && n2->b == o.b && n1->a != o.b && o.a != o.b) {
				if (stage_can_split(e) && i + 3 < n && p[i + 3].kind == OP_DSS && p[i + 3].a == o.b) {
					if ((r = RUN(sw_stage_split(e, n1->a, o.a, o.b, n1->coef)))) return r;
					i += 3; break;
				}
				if (!e->dry_run) {
					ProfScope ps(e, TMX_K_H_EXPLICIT);
					tmxk_sw_explicit(e, make_params(e), inst(e, n1->a), inst(e, o.a), inst(e, o.b), n1->coef);
					if ((r = launch_check("sw copy + H"))) return r;
				}
				i += 2; break;
			}
			// Copy(s->t) [twice in some schemes] + V.StepImplicit(t,t,c): the solve rewrites rho*theta, W, rho
			// of every stored column, so only U,V need copying and the solve reads s directly
			{
				size_t j = i + 1;
				while (j < n && p[j].kind == OP_COPY && p[j].a == o.a && p[j].b == o.b) j++;
				if (!e->sw && !e->fully_explicit && j < n && p[j].kind == OP_VIMP && p[j].a == o.b && p[j].b == o.b && o.a != o.b) {
					// ... and U,V need not be copied either while the copy's U,V are only read by the fused explicit stage
					// (k_h_explicit takes the U,V slabs of its initial instance and of the combination sources by separate
					// pointers, k_h_tracers those of its initial instance): the copy shares its source's U,V slabs under the rules
					// of share_is_safe.
					const bool share = e->share_copies && !e->use_graph && e->imap[o.a] == o.a && e->uvmap[o.a] == o.a &&
						share_is_safe(p, j + 1, o.a, o.b, noop);
					if (share) { if ((r = RUN(surface_copy(e, o.a, o.b)))) return r; }
					else { if (e->dry_run) e->dry_copies++; if ((r = RUN(copy_uv(e, o.a, o.b)))) return r; }
					if ((r = RUN(v_step_implicit_impl(e, o.a, o.b, p[j].coef, o.a)))) return r;
					if (share) {
						drop_readers(e, o.b);
						e->uvmap[o.b] = o.a; e->n_uvshared++; e->dry_shared++;
						after.w = 0;
					}
					i = j; break;
				}
			}
			// Copy(a->b) + StepAfterSubCycle(a,b,w): every branch of StepAfterSubCycle rewrites all of b from a (the second
			// hyperviscosity pass and the order-2 pass write every slab with a as their base; without viscosity it IS the copy)
			if (n1 && n1->kind == OP_HVIS && n1->a == o.a && n1->b == o.b && o.a != o.b) break;
			// Any other Copy(a->b) whose copy is only read until it is next overwritten entirely (share_is_safe) is not made: b
			// reads a's slot.  The case that matters: the copies in front of a V.StepImplicit that does nothing (fully explicit
			// vertical mode :1239-1242, shallow water's stub) -- config 4: three whole-state copies per ARS343 step.
			// (not into instance 0: the model state, which everything outside a step reads)
			if (e->share_copies && !e->use_graph && o.a != o.b && o.b != 0 && e->imap[o.a] == o.a && e->uvmap[o.a] == o.a && share_is_safe(p, i + 1, o.a, o.b, noop)) {
				drop_readers(e, o.b);      // (dead by the same argument: b is overwritten here)
				e->imap[o.b] = o.a; e->n_shared++; e->dry_shared++;
				after.w = 0;
				break;
			}
			if (e->dry_run) e->dry_copies++;
			else if ((r = tmx_copy_data(e, o.a, o.b))) return r;
			break;
		case OP_HEXP:
			if (e->sw && stage_can_split(e) && n1 && n2 && n1->kind == OP_VEXP && n1->a == o.a && n1->b == o.b && n2->kind == OP_DSS && n2->a == o.b && o.a != o.b) {
				if ((r = RUN(sw_stage_split(e, o.a, o.b, o.b, o.coef)))) return r;
				i += 2; break;
			}
			if (!e->sw && n1 && n1->kind == OP_VEXP && n1->a == o.a && n1->b == o.b && n1->coef == o.coef) {
				if (stage_can_split(e) && n2 && n2->kind == OP_DSS && n2->a == o.b) {
					if ((r = RUN(hv_stage_split(e, o.a, o.b, o.b, o.coef, nullptr, 0)))) return r;
					i += 2; break;
				}
				if ((r = RUN(hv_step_explicit(e, o.a, o.b, o.b, o.coef)))) return r;
				i += 1; break;
			}
			if ((r = RUN(tmx_h_step_explicit(e, o.a, o.b, o.coef)))) return r;
			break;
		case OP_VEXP:
			if ((r = RUN(tmx_v_step_explicit(e, o.a, o.b, o.coef)))) return r;
			break;
		case OP_DSS:
			if ((r = RUN(tmx_apply_dss(e, o.a)))) return r;
			break;
		case OP_VIMP:
			if ((r = RUN(tmx_v_step_implicit(e, o.a, o.b, o.coef)))) return r;
			break;
		case OP_VFILT:
			// VerticalDynamics::FilterNegativeTracers(instance): nothing without tracers
			if ((r = RUN(tmx_v_filter_negative_tracers(e, o.a)))) return r;
			break;
		case OP_VITE:
			if ((r = RUN(vi_terms_explicit(e, o.a, o.b, o.coef)))) return r;
			break;
		case OP_HVIS:
			if ((r = RUN(h_step_after_subcycle_impl(e, o.a, o.b, o.c, o.coef, true)))) return r;
			break;
		case OP_LINCOMB:
			// LinearCombine(-> d) + H(i,d,c) + V(i,d,c): combination evaluated inside the explicit kernels
			if (!e->sw && n1 && n2 && n1->kind == OP_HEXP && n2->kind == OP_VEXP && n1->b == o.a && n2->b == o.a &&
			    n1->a == n2->a && n1->coef == n2->coef && n1->a != o.a) {
				if (stage_can_split(e) && i + 3 < n && p[i + 3].kind == OP_DSS && p[i + 3].a == o.a) {
					if ((r = RUN(hv_stage_split(e, n1->a, o.a, o.a, n1->coef, o.lc, o.nlc)))) return r;
					i += 3; break;
				}
				if ((r = RUN(hv_step_explicit_lincomb(e, n1->a, o.a, n1->coef, o.lc, o.nlc)))) return r;
				i += 2; break;
			}
			if ((r = RUN(tmx_linear_combine_data(e, o.lc, o.nlc, o.a)))) return r;
			break;
		}
	}
	return TMX_OK;
}

#undef RUN

// The instance copies a stepper program performs (whole instances; with implicit vertical dynamics the U,V slabs in front of
// the column solve) and the ones it replaces by slot sharing, decided exactly as
// run_program decides them, without a device: `steps` consecutive steps (the first with first_step set), counts of the last
// one.  mode 0: implicit vertical dynamics, 1: fully explicit vertical mode, 2: shallow water.  Host logic for the CPU tests.
extern "C" int tmx_debug_program_copies(int scheme, int mode, int steps, int share, int * copies, int * shared) {
	REQUIRE(copies && shared && steps >= 1 && mode >= 0 && mode <= 2, TMX_ERR_INVALID, "tmx_debug_program_copies: bad argument");
	const int need = scheme_instances(scheme);
	REQUIRE(need > 0, TMX_ERR_INVALID, "unknown time scheme %d", scheme);
	tmx_engine e;
	e.cfg.n_instances = need; e.cfg.n_ranks = 1;
	e.sw = mode == 2; e.fully_explicit = mode == 1; e.share_copies = share != 0; e.dry_run = true;
	e.imap.resize(need);
	for (int k = 0; k < need; k++) e.imap[k] = k;
	e.uvmap = e.imap;
	for (int c = 0; c < 2; c++) {
		e.dry_ref[c].resize(need); e.dry_slot[c].resize(need);
		for (int k = 0; k < need; k++) e.dry_ref[c][k] = e.dry_slot[c][k] = dry_mix(77 + c, k);
	}
	g_dry_unsafe = share == 2;      // negative control of the symbolic check: every share accepted
	int r = TMX_OK;
	for (int n = 0; n < steps && !r; n++) {
		Program p;
		if ((r = build_program(scheme, n == 0, 0, 1.0, p, 0.0))) break;
		e.dry_copies = e.dry_shared = 0;
		r = run_program(&e, p);
		// between steps instance 0 is what everything outside looks at: it must sit in its own slot and hold the reference's value
		for (int c = 0; c < 2; c++)
			if (dry_slot_of(&e, 0, c) != 0 || e.dry_slot[c][0] != e.dry_ref[c][0]) e.dry_mismatch++;
	}
	g_dry_unsafe = false;
	if (r) return r;
	*copies = e.dry_copies; *shared = e.dry_shared;
	return e.dry_mismatch ? -1000 - e.dry_mismatch : TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// Stepper programs on the node-unique layout (UniqueLayout, tmx_internal.h; kernels instantiated with UQ = true).  The same
// programs, interpreted with the same fusion rules: [Copy | LinearCombine]; H; V; DSS is one stage (explicit kernels writing
// averaged nodes and partial slots, exchange of the slots other ranks need, k_dss_u), Copy; V.StepImplicit a U,V slab copy
// plus the column solve on the unique columns, StepAfterSubCycle its one or two passes with their DSS.  An explicit update
// that is not followed by the DSS of its instance has no node-unique form (the copies of a seam node then differ), so a program
// is run this way only if every operation fits (program_fits_unique); otherwise, and for instances whose copies differ on entry
// (a freshly uploaded initial state), tmx_step runs the element-major program.

static double * uinst(tmx_engine * e, int ix) { return e->u.d_ustate + (size_t)e->u.uslot[ix] * e->u.ustride; }
// where a kernel that reads instance ix while producing it must write: the spare slot; u_adopt_spare makes it the instance's afterwards
static double * uspare(tmx_engine * e) { return e->u.d_ustate + (size_t)e->u.uspare * e->u.ustride; }
static void u_adopt_spare(tmx_engine * e, int ix) { std::swap(e->u.uslot[ix], e->u.uspare); }
// U,V slabs of an instance: its own slot, or the slot of the instance it shares them with (tmx_engine::uvmap, the rules of the
// element-major programs: the Copy in front of the column solve leaves the copy's U,V identical to the source's)
static const double * uinst_uv(tmx_engine * e, int ix) { return uinst(e, e->uvmap[ix]); }
// instance ix is about to be read through ONE pointer (or updated in place): give it its own U,V slabs
static int u_own_uv(tmx_engine * e, int ix, bool total = false) {
	if (e->uvmap[ix] == ix) return TMX_OK;
	if (!total) {
		ProfScope ps(e, TMX_K_LINCOMB);
		HIPCHK(hipMemcpyAsync(uinst(e, ix), uinst_uv(e, ix), (size_t)2 * e->L * e->u.NUS * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	}
	e->uvmap[ix] = ix; e->n_uvshared--;
	return TMX_OK;
}
// instance x has been rewritten: whoever read its U,V slabs is dead by share_is_safe
static void u_written(tmx_engine * e, int x) { if (e->n_uvshared) drop_readers(e, x); }

// what the U interpreter does with operation i: 0 = not representable, else the number of operations it consumes
static int unique_unit(const Program & p, size_t i) {
	const size_t n = p.size();
	const Op & o = p[i];
	auto stage_at = [&](size_t j, int upd) {      // H(i,upd,c); V(i,upd,c); DSS(upd) starting at j
		return j + 2 < n && p[j].kind == OP_HEXP && p[j + 1].kind == OP_VEXP && p[j + 2].kind == OP_DSS && p[j].b == upd && p[j + 1].b == upd &&
			p[j + 2].a == upd && p[j].a == p[j + 1].a && p[j].coef == p[j + 1].coef && p[j].a != upd;
	};
	switch (o.kind) {
	case OP_COPY:
		if (o.a != o.b && stage_at(i + 1, o.b) && p[i + 1].a != o.b) return 4;
		return 1;
	case OP_LINCOMB:
		if (stage_at(i + 1, o.a)) return 4;
		return 1;
	case OP_HEXP: return stage_at(i, o.b) ? 3 : 0;
	case OP_VIMP: case OP_DSS: case OP_HVIS: case OP_VFILT: case OP_VITE: return 1;
	}
	return 0;      // V explicit on its own
}
static bool program_fits_unique(const Program & p) {
	for (size_t i = 0; i < p.size(); ) { const int k = unique_unit(p, i); if (!k) return false; i += k; }
	return true;
}

// Host logic for the CPU tests: does the program of a scheme have a node-unique form, and in how many units does the interpreter run it?
// Returns the number of units (fused groups of operations), 0 if some operation has no node-unique form, -1 for an unknown scheme.
extern "C" int tmx_debug_program_unique(int scheme, int first_step, int last_step) {
	Program p;
	if (build_program(scheme, first_step, last_step, 1.0, p, 0.0)) return -1;
	int units = 0;
	for (size_t i = 0; i < p.size(); ) { const int k = unique_unit(p, i); if (!k) return 0; i += k; units++; }
	return units;
}

// (the partial slots were filled by block kernels: the shorter list -- what no workgroup averaged)
static int u_dss_finish(tmx_engine * e, int ix, bool overlapped) {
	ProfScope ps(e, TMX_K_DSS);
	const UniqueLayout & u = e->u;
	const bool bl = u.slots_by_blocks;
	const int ng = bl ? u.b_ngroups : u.ngroups, nl = bl ? u.b_ngroups_local : u.ngroups_local;
	if (e->cfg.n_ranks > 1 && nl < ng) {
		tmxuk_dss(e, uinst(e, ix), 0, nl, bl);
		if (overlapped) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_recv, 0));
		tmxuk_dss(e, uinst(e, ix), nl, ng, bl);
	} else tmxuk_dss(e, uinst(e, ix), 0, ng, bl);
	return launch_check("DSS of the partial slots");
}

// kernels(KParams): the launches that fill instance ix and its partial slots; run boundary tiles first where the rank has both kinds
template <class F> static int u_produce_and_average(tmx_engine * e, int ix, int prof_slot, F kernels) {
	const UniqueLayout & u = e->u;
	KParams p = tmxu_params(e, make_params(e));
	bool overlapped = false;
	int r;
	if (u.split) {
		for (int part = 0; part < 2; part++) {
			p.quads = part ? u.d_quads_late : u.d_quads_early;
			e->launch_tiles = part ? u.n_late : u.n_early;
			p.bquads = part ? u.d_blks_late : u.d_blks_early;      // (block kernels: the same split by blocks)
			e->launch_blocks = part ? u.nb_late : u.nb_early;
			{ ProfScope ps(e, prof_slot); kernels(p); }
			if (part == 0) { p.quads = nullptr; p.bquads = nullptr; if ((r = exchange(e, p, nullptr, &overlapped))) return r; }
		}
		p.quads = nullptr; p.bquads = nullptr;
	} else {
		{ ProfScope ps(e, prof_slot); kernels(p); }
		if ((r = exchange(e, p, nullptr, &overlapped))) return r;
	}
	if ((r = launch_check("kernels of a stage (node-unique layout)"))) return r;
	return u_dss_finish(e, ix, overlapped);
}

// A stage "LinearCombine(lc -> d); H; V; DSS(d)" at operation i whose sources are exactly the LEADING terms of a later such
// stage's combination (in the reference's accumulation order: the destination's own term, then ascending instance index), none of
// them written in between: it also stores that leading partial sum (EmitTerms), and the later stage starts from it.
// ARS343 (the bench): stage 3 reads instances 0..4, stage 4 combines 1, 0, 2, 3, 4, 5, 6 -- it reads the prefix, 5 and 6.
struct PrefixPlan {
	size_t producer = 0, consumer = 0;      // operation index of the two stages' LinearCombine
	unsigned members = 0u;    // instances the prefix holds
	int first = -1;           // producer's term index of the consumer's destination (-1: the consumer's own coefficient is zero)
	double coef[12];          // consumer's coefficient of the producer's term m
};
static bool prefix_plan(const tmx_engine * e, const Program & p, size_t i, PrefixPlan * plan) {
	const Op & R = p[i];
	if (!e->u.prefix_option || R.kind != OP_LINCOMB || unique_unit(p, i) != 4 || R.lc[R.a] != 0.0) return false;
	int term_of[32]; unsigned A = 0u; int nA = 0;
	for (int m = 0; m < 32; m++) term_of[m] = -1;
	for (int m = 0; m < R.nlc && m < 32; m++) if (m != R.a && R.lc[m] != 0.0) { term_of[m] = 1 + nA++; A |= 1u << m; }
	if (nA < 2 || nA > 7) return false;      // (the kernels that store a prefix: 3 to 8 terms, the destination's unread one included)
	for (size_t j = i + 4; j < p.size(); j++) {
		const Op & S = p[j];
		if (S.kind == OP_LINCOMB && unique_unit(p, j) == 4) {
			unsigned B = 0u;
			for (int m = 0; m < S.nlc && m < 32; m++) if (S.lc[m] != 0.0) B |= 1u << m;
			if ((B & A) == A && (B & ~A) != 0u) {
				// order: [S.a if its coefficient is not zero], then ascending -- the members first
				int top = -1, low = 32;      // last member and first other term of the ascending part
				for (int q = 0; q < 32; q++) if ((A >> q & 1u) && q != S.a) top = q;
				for (int q = 31; q >= 0; q--) if ((B & ~A) >> q & 1u) low = q;
				if ((!(B >> S.a & 1u) || (A >> S.a & 1u)) && low > top) {
					plan->producer = i; plan->consumer = j; plan->members = A; plan->first = (A >> S.a & 1u) ? term_of[S.a] : -1;
					for (int m = 0; m < 12; m++) plan->coef[m] = 0.0;
					for (int m = 0; m < 32; m++) if (term_of[m] >= 0) plan->coef[term_of[m]] = S.lc[m];
					return true;
				}
			}
		}
		const OpAccess x = op_access(S, false);
		if ((x.full | x.part) & A) return false;      // a member is rewritten: no later stage sees these values
	}
	return false;
}

// the producer / consumer pairs of a program that are used: there is one prefix slot, so their spans must not touch; the pairs that
// save the most source instances first (ARS343: the third stage's five sources serve the fourth -- four instances less to read --
// rather than the second stage's three serving the third)
static std::vector<PrefixPlan> prefix_schedule(const tmx_engine * e, const Program & p) {
	std::vector<PrefixPlan> all, use;
	for (size_t i = 0; i < p.size(); ) {
		const int k = unique_unit(p, i);
		if (!k) return use;
		PrefixPlan plan;
		if (prefix_plan(e, p, i, &plan)) all.push_back(plan);
		i += k;
	}
	auto saved = [](const PrefixPlan & a) { int c = 0; for (int m = 0; m < 32; m++) c += a.members >> m & 1u; return c - 1; };
	std::stable_sort(all.begin(), all.end(), [&](const PrefixPlan & a, const PrefixPlan & b) { return saved(a) > saved(b); });
	for (const PrefixPlan & a : all) {
		bool clash = false;
		for (const PrefixPlan & b : use) clash = clash || !(a.consumer < b.producer || b.consumer < a.producer);
		if (!clash) use.push_back(a);
	}
	std::sort(use.begin(), use.end(), [](const PrefixPlan & a, const PrefixPlan & b) { return a.producer < b.producer; });
	return use;
}

extern "C" int tmx_debug_program_prefix(int scheme, int first_step, int last_step, int * pairs, int * terms, int cap) {
	Program p;
	if (build_program(scheme, first_step, last_step, 1.0, p, 0.0)) return -1;
	tmx_engine e;      // (only the option is read)
	e.u.prefix_option = 1;
	if (!program_fits_unique(p)) return 0;
	const std::vector<PrefixPlan> use = prefix_schedule(&e, p);
	int n = 0;
	for (const PrefixPlan & plan : use) {
		if (n < cap) {
			if (pairs) { pairs[2 * n] = (int)plan.producer; pairs[2 * n + 1] = (int)plan.consumer; }
			if (terms) { int c = 0; for (int m = 0; m < 32; m++) c += plan.members >> m & 1u; terms[n] = c - 1; }
		}
		n++;
	}
	return n;
}

// [Copy(ibase -> iupd) | LinearCombine(lc -> iupd)]; H.StepExplicit(iinit, iupd); V.StepExplicit(iinit, iupd); DSS(iupd)
// emit: also store the prefix of a later stage's combination; held: members of the stored prefix this stage's combination starts from
static int u_stage(tmx_engine * e, int iinit, int ibase, int iupd, double dt, const double * lc, int nlc, const PrefixPlan * emit = nullptr, unsigned held = 0u) {
	const double * src[12]; double cf[12]; int premul = 0, n = 0;
	const double * srcuv[12];
	double * const prefix = e->u.d_ustate + (size_t)e->u.uprefix * e->u.ustride;
	// instances read in element-major form, copy by copy (UniqueLayout::dlive): their pointers are the element-major slots', the kernels get a mask
	auto dl = [&](int k) -> bool { return (e->u.dlive >> k & 1u) != 0u; };
	unsigned dmask = 0u;
	if (lc) {
		premul = (lc[iupd] != 0.0) ? 1 : 0;
		n = 1; src[0] = dl(iupd) ? inst(e, iupd) : uinst(e, iupd); srcuv[0] = dl(iupd) ? inst_uv(e, iupd) : uinst_uv(e, iupd); cf[0] = lc[iupd];
		if (dl(iupd)) dmask |= 1u;
		if (held) { src[0] = srcuv[0] = prefix; cf[0] = 1.0; premul = 1; e->u.prefix_stages++; }      // (times one: exact)
		for (int m = 0; m < nlc; m++) {
			if (m == iupd || lc[m] == 0.0 || (held >> m & 1u)) continue;
			REQUIRE(n < 12, TMX_ERR_UNSUPPORTED, "linear combination with more than 11 source terms");
			src[n] = dl(m) ? inst(e, m) : uinst(e, m); srcuv[n] = dl(m) ? inst_uv(e, m) : uinst_uv(e, m); cf[n] = lc[m];
			if (dl(m)) dmask |= 1u << n;
			n++;
		}
	} else if (dl(ibase)) dmask |= 1u;
	if (dl(iinit)) dmask |= 1u << 31;
	REQUIRE(!(held && dmask), TMX_ERR_UNSUPPORTED, "internal: stored prefix with element-major sources");
	REQUIRE(!dmask || e->u.tile_shape == 0 || tmxk_h_walk_ok(e, tmxu_params(e, make_params(e)), 1, n, false), TMX_ERR_UNSUPPORTED,
		"internal: copy-by-copy reads of an element-major instance with this thread order need the column-segment walk");
	const double * xin_uv = dl(iinit) ? inst_uv(e, iinit) : uinst_uv(e, iinit), * base_uv = dl(ibase) ? inst_uv(e, ibase) : uinst_uv(e, ibase);      // (taken before the update instance changes slots)
	// the update instance is also read (in-place accumulation, or its own coefficient in the combination is not zero): the
	// element-major kernels read and write a thread's own column, here other elements' threads read the node too -- write the
	// spare slot and let it become the instance's
	bool aliased = lc ? (premul != 0) : (ibase == iupd);
	if (held) {      // the destination's own term is inside the prefix: its slot is read only if something else lives there
		const double * mine = uinst(e, iupd);
		aliased = xin_uv == mine;
		for (int m = 1; m < n; m++) aliased = aliased || src[m] == mine || srcuv[m] == mine;
	}
	REQUIRE(iinit != iupd, TMX_ERR_INVALID, "StepExplicit: initial and update data instance must be distinct");
	double * out = aliased ? uspare(e) : uinst(e, iupd);
	const double * base = lc ? nullptr : (dl(ibase) ? inst(e, ibase) : uinst(e, ibase)), * xin = dl(iinit) ? inst(e, iinit) : uinst(e, iinit);
	EmitTerms em; em.xp = nullptr; em.first = -1;
	// (only a kernel that really stores it: the column-segment walk, or the level-parallel kernel of the experiments flavour; element-major sources: no prefix)
	const bool walk = tmxk_h_walk_ok(e, tmxu_params(e, make_params(e)), 1, n, true);
	if (emit && lc && !premul && n >= 3 && n <= 8 && !dmask && (walk ? !tmxu_blocks_on(e, 1) : TMX_EXP != 0)) {      // (the block form of the walk stores none)
		em.xp = prefix; em.first = emit->first; for (int m = 0; m < 12; m++) em.coef[m] = emit->coef[m];
	}
	if (emit) e->u.prefix_for = em.xp ? (long long)emit->consumer : -1;      // which operation may start from the stored prefix (run_program_unique)
	// the stage writes every slab of the update instance: its U,V slabs are its own from here on (what it shared is read through
	// the pointers taken above), and whoever shared ITS slabs is dead
	if (e->uvmap[iupd] != iupd) { e->uvmap[iupd] = iupd; e->n_uvshared--; }
	if (aliased) u_adopt_spare(e, iupd);      // (src[], base and the U,V pointers were taken before: they still name the old slots)
	int r = u_produce_and_average(e, iupd, TMX_K_H_EXPLICIT, [&](const KParams & p) {
		if (n > 0) tmxk_h_explicit(e, p, xin, out, out, dt, 1, n, src, cf, premul, xin_uv, nullptr, srcuv, em.xp ? &em : nullptr, dmask);
		else tmxk_h_explicit(e, p, xin, base, out, dt, 1, 0, nullptr, nullptr, 0, xin_uv, base_uv, nullptr, nullptr, dmask);
	});
	e->u.dlive &= ~(1u << iupd);      // the instance is node-unique from here on
	u_written(e, iupd);
	if (r) return r;
	if (lc) {
		const double * dsrc[12]; double dcf[12];      // the surface slots live with the element-major slots (all terms: no prefix there)
		int m2 = 1; dsrc[0] = inst(e, iupd); dcf[0] = lc[iupd];
		for (int m = 0; m < nlc; m++) if (m != iupd && lc[m] != 0.0) { dsrc[m2] = inst(e, m); dcf[m2++] = lc[m]; }
		return surface_lincomb(e, iupd, m2, dsrc, dcf, lc[iupd] != 0.0 ? 1 : 0);
	}
	return surface_copy(e, ibase, iupd);
}

static int u_copy(tmx_engine * e, int a, int b, size_t doubles) {
	if (a == b) return TMX_OK;
	int r;
	if ((r = u_own_uv(e, a)) || (r = u_own_uv(e, b, true))) return r;
	{
		ProfScope ps(e, TMX_K_LINCOMB);
		HIPCHK(hipMemcpyAsync(uinst(e, b), uinst(e, a), doubles * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	}
	if (doubles == e->u.ustride) e->u.dlive &= ~(1u << b);
	u_written(e, b);
	return surface_copy(e, a, b);
}

// [Copy(a -> b);] V.StepImplicit(b, b): the solve rewrites rho*theta, W, rho of every column, so only the U,V slabs are copied
// share: b's U,V slabs are not copied, b reads a's (decided on the program by share_is_safe, as in the element-major interpreter)
static int u_vimp(tmx_engine * e, int a, int b, double dt, bool share) {
	REQUIRE(dt != 0.0, TMX_ERR_INVALID, "StepImplicit: dt must be non-zero");
	int r;
	if ((r = u_own_uv(e, a))) return r;      // the column kernel reads U,V of the initial instance through its one pointer
	if (a != b) {
		if (share) {
			if ((r = u_own_uv(e, b, true)) || (r = surface_copy(e, a, b))) return r;
		} else if ((r = u_copy(e, a, b, (size_t)2 * e->L * e->u.NUS))) return r;
	}
	{
		ProfScope ps(e, TMX_K_VI_SOLVE);
		e->vi_unique = true;
		tmxk_vi_fused(e, tmxu_params_columns(e, make_params(e)), uinst(e, a), uinst(e, b), dt);
		e->vi_unique = false;
	}
	e->u.dlive &= ~(1u << b);      // b is node-unique in every slab from here on (its U,V copied or shared from a, the rest solved): a later stage must not read a stale element-major slot
	u_written(e, b);
	if (a != b && share) { e->uvmap[b] = a; e->n_uvshared++; }
	return launch_check("column solve (node-unique layout)");
}

// VerticalDynamicsFEM::StepImplicitTermsExplicitly (ARK232): update -= dt F(initial) for rho*theta, W, rho, column by column -- on the
// unique columns (every copy of a node would receive the same increment of the same value)
static int u_vite(tmx_engine * e, int a, int b, double dt) {
	REQUIRE(a != b && dt != 0.0, TMX_ERR_INVALID, "StepImplicitTermsExplicitly: distinct instances and non-zero dt required");
	int r;
	if ((r = u_own_uv(e, a))) return r;      // (the kernel reads U,V of the initial instance through its one pointer)
	{
		ProfScope ps(e, TMX_K_VI_ASSEMBLE);
		tmxk_vi_terms_explicit(e, tmxu_params_columns(e, make_params(e)), uinst(e, a), uinst(e, b), dt, false);
	}
	u_written(e, b);
	return launch_check("explicitly evaluated implicit terms (node-unique layout)");
}

// DSS of an instance whose copies already agree inside every patch (e.g. after the column solve): only the nodes on patch edges change
static int u_dss(tmx_engine * e, int ix) {
	bool overlapped = false;
	int r;
	if ((r = u_own_uv(e, ix))) return r;
	{ ProfScope ps(e, TMX_K_DSS); tmxuk_slots_from_u(e, uinst(e, ix)); }
	e->u.slots_by_blocks = tmxu_blocks_on(e);      // every slot holds its node's value: the shorter list changes the same nodes (a group of identical copies averages to itself)
	if ((r = exchange(e, tmxu_params(e, make_params(e)), nullptr, &overlapped))) return r;
	r = u_dss_finish(e, ix, overlapped);
	u_written(e, ix);
	return r;
}

// HorizontalDynamicsFEM::StepAfterSubCycle (h_step_after_subcycle_impl above, same branches)
static int u_hvis_core(tmx_engine * e, int iinit, int iupd, int iwork, double dt);
// ... and its Rayleigh friction (APPLY_RAYLEIGH_WITH_HYPERVIS, HorizontalDynamicsFEM.cpp:2719-2724, ApplyRayleighFriction :2420-2570): the
// reference relaxes every stored copy with the copy's own strength and reference state, so the result has no node-unique form -- the kernel
// reads the node-unique instance and writes the element-major one, and the next step's explicit stages read that copy by copy ("unique_mixed")
static int u_hvis(tmx_engine * e, int iinit, int iupd, int iwork, double dt) {
	int r = u_hvis_core(e, iinit, iupd, iwork, dt);
	if (r || !e->rayleigh) return r;
	UniqueLayout & u = e->u;
	REQUIRE(e->imap[iupd] == iupd, TMX_ERR_INVALID, "internal: Rayleigh friction on a shared slot");
	if ((r = u_own_uv(e, iupd))) return r;
	{
		ProfScope ps(e, TMX_K_HYPERVIS);
		tmxk_rayleigh(e, make_params(e), inst(e, iupd), dt, uinst(e, iupd), (size_t)u.NUS, (const int *)u.d_ucol_of_dcol);
	}
	if (u.form[iupd]) { u.form[iupd] = 0; u.n_uform--; }
	u.conversions++;
	return launch_check("rayleigh friction (node-unique in, element-major out)");
}
static int u_hvis_core(tmx_engine * e, int iinit, int iupd, int iwork, double dt) {
	const tmx_config & c = e->cfg;
	int r;
	REQUIRE(iinit != iwork && iupd != iwork && iinit != iupd, TMX_ERR_UNSUPPORTED, "StepAfterSubCycle on the node-unique layout needs three distinct instances");
	if (!hypervis_active(e)) return u_copy(e, iinit, iupd, e->u.ustride);
	// the passes read their source through one pointer and rewrite update and (two passes) working instance entirely
	if ((r = u_own_uv(e, iinit)) || (r = u_own_uv(e, iupd, true))) return r;
	if (c.hypervis_order != 2 && (r = u_own_uv(e, iwork, true))) return r;
	struct Written { tmx_engine * e; int a, b; ~Written() { e->u.dlive &= ~(1u << a); u_written(e, a); if (b >= 0) { e->u.dlive &= ~(1u << b); u_written(e, b); } } } written = { e, iupd, c.hypervis_order != 2 ? iwork : -1 };
	if (c.hypervis_order == 2) {
		if ((r = surface_copy(e, iinit, iupd))) return r;
		return u_produce_and_average(e, iupd, TMX_K_HYPERVIS, [&](const KParams & p) {
			tmxk_hypervis(e, p, uinst(e, iinit), uinst(e, iinit), uinst(e, iupd), -dt, -c.nu_scalar, c.nu_div, c.nu_vort, 0); });
	}
	const int scale = (c.reference_length != 0.0) ? 1 : 0;
	if ((r = surface_copy(e, iinit, iupd)) || (r = surface_zero(e, iwork))) return r;
	if ((r = u_produce_and_average(e, iwork, TMX_K_HYPERVIS, [&](const KParams & p) {
		tmxk_hypervis(e, p, uinst(e, iinit), nullptr, uinst(e, iwork), 1.0, 1.0, 1.0, 1.0, 0); }))) return r;
	return u_produce_and_average(e, iupd, TMX_K_HYPERVIS, [&](const KParams & p) {
		tmxk_hypervis(e, p, uinst(e, iwork), uinst(e, iinit), uinst(e, iupd), -dt, c.nu_scalar, c.nu_div, c.nu_vort, scale); });
}

static int run_program_unique(tmx_engine * e, const Program & p) {
	int r;
	const std::vector<PrefixPlan> plans = prefix_schedule(e, p);
	e->u.prefix_for = -1;
	auto plan_at = [&](size_t i, bool consumer) -> const PrefixPlan * {
		for (const PrefixPlan & a : plans) if ((consumer ? a.consumer : a.producer) == i) return &a;
		return nullptr;
	};
	// U,V slabs still shared from the previous step: kept where this program allows it
	for (int b = 0; b < (int)e->uvmap.size() && e->n_uvshared; b++)
		if (e->uvmap[b] != b && !share_is_safe(p, 0, e->uvmap[b], b, false) && (r = u_own_uv(e, b))) return r;
	for (size_t i = 0; i < p.size(); ) {
		const Op & o = p[i];
		const int k = unique_unit(p, i);
		switch (o.kind) {
		case OP_COPY:
			if (k == 4) { if ((r = u_stage(e, p[i + 1].a, o.a, o.b, p[i + 1].coef, nullptr, 0))) return r; break; }
			{
				size_t j = i + 1;      // Copy(s -> t) [repeated] + V.StepImplicit(t, t)
				while (j < p.size() && p[j].kind == OP_COPY && p[j].a == o.a && p[j].b == o.b) j++;
				if (j < p.size() && p[j].kind == OP_VIMP && p[j].a == o.b && p[j].b == o.b && o.a != o.b) {
					const bool share = e->share_copies && e->uvmap[o.a] == o.a && share_is_safe(p, j + 1, o.a, o.b, false);
					if ((r = u_vimp(e, o.a, o.b, p[j].coef, share))) return r;
					i = j + 1; continue;
				}
			}
			// Copy(a -> b) + StepAfterSubCycle(a, b, w): every branch rewrites all of b from a
			if (i + 1 < p.size() && p[i + 1].kind == OP_HVIS && p[i + 1].a == o.a && p[i + 1].b == o.b && o.a != o.b) break;
			if (!e->u.form[o.a] && o.a != o.b) {
				// the source is element-major: the Rayleigh relaxation at the end of StepAfterSubCycle left it so (u_hvis); the copy is
				// a whole-slot copy as in the element-major program (surface slots included), and the destination is element-major too
				REQUIRE(e->rayleigh && e->imap[o.a] == o.a && e->imap[o.b] == o.b, TMX_ERR_INVALID, "internal: element-major source of a copy inside a node-unique program");
				if ((r = u_own_uv(e, o.b, true))) return r;
				{
					ProfScope ps(e, TMX_K_LINCOMB);
					HIPCHK(hipMemcpyAsync(inst(e, o.b), inst(e, o.a), e->inst_stride * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
				}
				if (e->u.form[o.b]) { e->u.form[o.b] = 0; e->u.n_uform--; }
				e->u.dlive &= ~(1u << o.b);
				u_written(e, o.b);
				break;
			}
			if ((r = u_copy(e, o.a, o.b, e->u.ustride))) return r;
			break;
		case OP_LINCOMB:
			if (k == 4) {
				const PrefixPlan * held = plan_at(i, true);      // an earlier stage has stored the leading partial sum of this combination ...
				if (held && e->u.prefix_for != (long long)i) held = nullptr;      // ... unless it could not (it read an element-major instance copy by copy)
				if ((r = u_stage(e, p[i + 1].a, o.a, o.a, p[i + 1].coef, o.lc, o.nlc, plan_at(i, false), held ? held->members : 0u))) return r;
				break;
			}
			{
				const double * src[12], * dsrc[12]; double cf[12];
				if ((r = u_own_uv(e, o.a, o.lc[o.a] == 0.0))) return r;
				for (int m = 0; m < o.nlc; m++) if (m != o.a && o.lc[m] != 0.0 && (r = u_own_uv(e, m))) return r;      // (whole-instance pass: one pointer per source)
				int n = 1; src[0] = uinst(e, o.a); dsrc[0] = inst(e, o.a); cf[0] = o.lc[o.a];
				for (int m = 0; m < o.nlc; m++) {
					if (m == o.a || o.lc[m] == 0.0) continue;
					REQUIRE(n < 12, TMX_ERR_UNSUPPORTED, "linear_combine_data: more than 11 source terms");
					src[n] = uinst(e, m); dsrc[n] = inst(e, m); cf[n] = o.lc[m]; n++;
				}
				const int premul = o.lc[o.a] != 0.0 ? 1 : 0;
				{ ProfScope ps(e, TMX_K_LINCOMB); tmxk_lincomb(e, e->u.ustride, uinst(e, o.a), n, src, cf, premul); }
				e->u.dlive &= ~(1u << o.a);
				u_written(e, o.a);
				if ((r = surface_lincomb(e, o.a, n, dsrc, cf, premul)) || (r = launch_check("lincomb"))) return r;
			}
			break;
		case OP_HEXP: if ((r = u_stage(e, o.a, o.b, o.b, o.coef, nullptr, 0))) return r; break;
		case OP_VIMP: if ((r = u_vimp(e, o.a, o.b, o.coef, false))) return r; break;
		case OP_DSS: if ((r = u_dss(e, o.a))) return r; break;
		case OP_HVIS: if ((r = u_hvis(e, o.a, o.b, o.c, o.coef))) return r; break;
		case OP_VITE: if ((r = u_vite(e, o.a, o.b, o.coef))) return r; break;
		case OP_VFILT: break;      // nothing without tracers
		default: REQUIRE(false, TMX_ERR_UNSUPPORTED, "internal: operation %d has no node-unique form", o.kind);
		}
		i += k;
	}
	return TMX_OK;
}

// May the program read the instances of `dlive` in element-major form, copy by copy?  Only the explicit stages can (their kernels take a
// source mask); every other operation must find the instance overwritten -- node-unique -- by the time it reads it.
static bool dlive_program_ok(const Program & p, unsigned dlive) {
	for (size_t i = 0; i < p.size() && dlive; ) {
		const int k = unique_unit(p, i);
		if (!k) return false;
		const Op & o = p[i];
		const bool stage = ((o.kind == OP_COPY || o.kind == OP_LINCOMB) && k == 4) || (o.kind == OP_HEXP && k == 3);
		unsigned wr = 0u;
		for (int j = 0; j < k; j++) {
			const OpAccess x = op_access(p[i + j], false);
			if (!stage && ((x.reads | x.part) & dlive)) return false;
			wr |= x.full | x.part;
		}
		dlive &= ~wr;
		i += k;
	}
	return true;
}
// Host logic for the CPU tests: can the scheme's program read instance `instance` copy by copy (1), or not (0)?  -1: unknown scheme
extern "C" int tmx_debug_program_mixed(int scheme, int first_step, int last_step, int instance) {
	Program p;
	if (build_program(scheme, first_step, last_step, 1.0, p, 0.0)) return -1;
	g_hvis_two_pass = true;
	return (program_fits_unique(p) && dlive_program_ok(p, 1u << instance)) ? 1 : 0;
}

// Entry of tmx_step: bring the instances the program reads before it overwrites them into the node-unique form (checking, for
// those that come from the element-major form, that the copies of every node agree) and run the program there.  *done = false:
// not possible (configuration, program, or an instance whose copies differ); the caller runs the element-major program.
// With a Rayleigh layer the result of StepAfterSubCycle is element-major (u_hvis): the node-unique interpreter serves the programs in which
// only copies of that result follow (ARS343: nothing follows; ARS222 / 232 / 443 / ARK232: Copy(result -> 0)); Strang's closing column solve
// of the relaxed state has no node-unique form, its programs stay element-major
static bool rayleigh_program_ok(const Program & p) {
	size_t h = p.size();
	for (size_t i = 0; i < p.size(); i++) if (p[i].kind == OP_HVIS) { if (h != p.size()) return false; h = i; }
	for (size_t i = h + 1; i < p.size(); i++) if (!(p[i].kind == OP_COPY && p[i].a == p[h].b)) return false;
	return true;
}
// Host logic for the CPU tests: with a Rayleigh layer, does the scheme's program run on the node-unique layout (1) or element-major (0)?  -1: unknown scheme
extern "C" int tmx_debug_program_rayleigh(int scheme, int first_step, int last_step) {
	Program p;
	if (build_program(scheme, first_step, last_step, 1.0, p, 0.0)) return -1;
	return (program_fits_unique(p) && rayleigh_program_ok(p)) ? 1 : 0;
}
static int try_step_unique(tmx_engine * e, const Program & p, bool * done) {
	*done = false;
	UniqueLayout & u = e->u;
	if (!u.built || false || !program_fits_unique(p)) return TMX_OK;
	if (e->rayleigh && !rayleigh_program_ok(p)) return TMX_OK;
	// StepImplicitTermsExplicitly evaluates EVERY stored copy of a node with the copy's own column metric (VerticalDynamicsFEM.cpp:543-566:
	// all interior (i, j), none skipped); the node-unique form evaluates the representative copy once, which is the same only where the
	// copies' metric entries carry the same bits (UniqueLayout::vite_ok, checked on the host's arrays when the layout is built)
	if (!u.vite_ok) for (const Op & o : p) if (o.kind == OP_VITE) return TMX_OK;
	int r;
	unsigned written = 0u, livein = 0u;
	const bool noop = false;
	g_hvis_two_pass = hypervis_active(e) && e->cfg.hypervis_order != 2;
	for (size_t i = 0; i < p.size(); i++) {
		const OpAccess x = op_access(p[i], noop);
		livein |= x.reads & ~written;
		livein |= x.part & ~written;
		written |= x.full | x.part;
	}
	bool need_check = false;
	unsigned dform = 0u;
	for (int k = 0; k < e->cfg.n_instances; k++) if ((livein >> k & 1u) && !u.form[k]) { need_check = true; dform |= 1u << k; }
	// Round 5: live-in instances in element-major form are READ that way, copy by copy, by the explicit stages (no check of the copies, no
	// host synchronisation, no conversion; exact whether the copies agree or not) wherever the program lets only its stages read them
	u.dlive = 0u;
	// (a thread reads its own stored copy: the level-parallel kernels take the thread index for the stored column, which only the default thread order
	// allows; the column-segment walk looks the column up, KParams::t_dcol)
	const bool own_copy_ok = u.tile_shape == 0 || (u.tile_shape == 4 && e->opt_h_walk != 0 && !e->use_mfma);
	if (dform && u.mixed_option && own_copy_ok && dlive_program_ok(p, dform)) {
		for (int k = 0; k < e->cfg.n_instances; k++)
			if ((dform >> k & 1u) && (e->imap[k] != k || e->uvmap[k] != k) && (r = settle_instance(e, k, false))) return r;
		u.dlive = dform; need_check = false; u.mixed_steps++;
	}
	if (need_check && u.skip_steps > 0) { u.skip_steps--; return TMX_OK; }      // the last checks failed: not every step pays for another one
	// (timing aid TMX_DEBUG_SKIP_EXCHANGE: the state of such a run is garbage at the rank boundary, NaN payloads included; no check)
	const bool trust = TMX_EXP && e->opt_skip_exchange != 0;
	// the check comes first and works on the slots as the element-major programs left them (shared slots included: an instance
	// that shares is given its own first, which for the model state never happens), so that a step that stays element-major keeps them
	for (int k = 0; k < e->cfg.n_instances && !trust && !u.dlive; k++) {
		if (!(livein >> k & 1u) || u.form[k]) continue;
		if ((e->imap[k] != k || e->uvmap[k] != k) && (r = settle_instance(e, k, false))) return r;
		bool ok = true;
		if ((r = tmxu_check(e, k, &ok))) return r;
		if (!ok) {      // copies differ: this step runs element-major
			// (a single failure -- the pointwise initial state of a run -- costs nothing further; from the second in a row on, 4, 8, ... 1024 steps pass before the next try)
			u.backoff++;
			u.skip_steps = (u.backoff >= 2) ? std::min(4 << std::min(u.backoff - 2, 8), 1024) : 0;
			return TMX_OK;
		}
	}
	u.backoff = 0;
	// instance maps to identity (shares left by an element-major program, that is: nothing is in U form; U,V slabs shared by the last node-unique step stay shared)
	if ((e->n_shared || e->n_uvshared) && u.n_uform == 0) { if ((r = settle_instance(e, -1, false))) return r; }
	for (int k = 0; k < e->cfg.n_instances; k++)
		if ((livein >> k & 1u) && !u.form[k] && !(u.dlive >> k & 1u) && (r = tmxu_to_u(e, k, false, nullptr))) return r;
	// everything the program writes is in U form afterwards; whatever else it leaves alone keeps its form
	for (int k = 0; k < e->cfg.n_instances; k++)
		if (written >> k & 1u) { if (!u.form[k]) u.n_uform++; u.form[k] = 1; }      // (form 2 -> 1: the element-major copy goes stale)
	e->in_program = true;
	r = run_program_unique(e, p);
	e->in_program = false;
	u.dlive = 0u;
	*done = true;
	return r;
}

extern "C" int tmx_scheme_instances(int scheme) { return scheme_instances(scheme); }

extern "C" int tmx_step(tmx_engine * e, int scheme, int first_step, int last_step, double dt) {
	int r; if ((r = check_ready(e))) return r;
	const int need = scheme_instances(scheme);
	REQUIRE(need > 0, TMX_ERR_INVALID, "unknown time scheme %d", scheme);
	REQUIRE(e->cfg.n_instances >= need, TMX_ERR_INVALID, "time scheme %d needs %d data instances, engine has %d", scheme, need, e->cfg.n_instances);
	Program p;
	if ((r = build_program(scheme, first_step, last_step, dt, p, e->strang_offc))) return r;
	// Optional (TMX_GRAPH=1): single-rank steps replayed from a hipGraph -- the 20-30 launches of a step are
	// captured once per (scheme, first, last, dt), every kernel argument being fixed after tmx_finalize, and
	// relaunched as one graph.  Off by default: the plain sequence is already queued asynchronously far ahead of the
	// GPU, and the replay measured 0-1 % SLOWER (ne8: 0.419 vs 0.414 ms/step, ne16: 0.605 vs 0.601, ne30: 1.65 both).
	// Multi-rank steps (RCCL calls on a second stream) and profiled steps always use the plain sequence.
	if (e->use_graph && (e->cfg.n_ranks == 1 || (TMX_EXP && e->opt_skip_exchange)) && !e->prof && !e->lb) {
		StepGraphKey key = { scheme, first_step ? 1 : 0, last_step ? 1 : 0, dt };
		if (e->udiff) {      // (the reference state's stored diffusion terms: allocated and formed outside the capture)
			bool all = true;
			for (int lp : e->local_patches) all = all && e->patches[lp].ref_set;
			if (all) tmxk_h_walk_prepare(e, make_params(e));
		}
		for (auto & g : e->graphs)
			if (g.key.scheme == key.scheme && g.key.first == key.first && g.key.last == key.last && g.key.dt == key.dt) {
				HIPCHK(hipGraphLaunch(g.exec, e->stream));
				return TMX_OK;
			}
		hipGraph_t graph = nullptr;
		HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
		r = run_program(e, p);
		const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
		if (r) { if (graph) hipGraphDestroy(graph); (void)hipGetLastError(); return r; }      // (the error is reported; it must not stay behind as the runtime's last error)
		if (ce != hipSuccess || !graph) {
			(void)hipGetLastError();
			e->use_graph = false;                 // capture not possible here: run the plain sequence from now on
			return run_program(e, p);
		}
		StepGraph sg; sg.key = key;
		const hipError_t ie = hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0);
		hipGraphDestroy(graph);
		if (ie != hipSuccess) { (void)hipGetLastError(); e->use_graph = false; return run_program(e, p); }
		if (e->graphs.size() >= 8) { hipGraphExecDestroy(e->graphs.front().exec); e->graphs.erase(e->graphs.begin()); }
		e->graphs.push_back(sg);
		HIPCHK(hipGraphLaunch(sg.exec, e->stream));
		return TMX_OK;
	}
	if (e->u.built) {
		bool done = false;
		if ((r = try_step_unique(e, p, &done)) || done) return r;
		// element-major step: nothing stays in U form (the shared slots of the element-major programs are left as they are)
		if (e->u.n_uform)
			for (int b = 0; b < (int)e->u.form.size(); b++) if (e->u.form[b] && (r = tmxu_to_d(e, b))) return r;
	}
	return run_program(e, p);
}

extern "C" int tmx_set_strang_off_centering(tmx_engine * e, double off_centering) {
	REQUIRE(e, TMX_ERR_INVALID, "tmx_set_strang_off_centering: null engine");
	REQUIRE(off_centering >= 0.0 && off_centering <= 1.0, TMX_ERR_INVALID, "off-centering %g outside [0, 1] (TimestepSchemeStrang.cpp:34-36)", off_centering);
	if (off_centering != e->strang_offc) {
		if (!plan_only(e) && e->stream) hipStreamSynchronize(e->stream);
		for (auto & g : e->graphs) hipGraphExecDestroy(g.exec);      // captured steps carry the old coefficients
		e->graphs.clear();
	}
	e->strang_offc = off_centering;
	return TMX_OK;
}

extern "C" int tmx_step_ars343(tmx_engine * e, double dt) { return tmx_step(e, TMX_SCHEME_ARS343, 0, 0, dt); }

static int held_suarez_unique(tmx_engine * e, int instance, double dt, bool * done);
extern "C" int tmx_physics_held_suarez(tmx_engine * e, int instance, double dt) {
	int r; if ((r = check_ready(e))) return r;
	if (e->u.built && !e->sw) { bool done = false; if ((r = held_suarez_unique(e, instance, dt, &done)) || done) return r; }
	// The instance is node-unique but the forcing's inputs differ between the copies of a node (the reference's own latitude array does):
	// every stored copy is forced with its own inputs straight from the node-unique slabs -- the kernel's load is the conversion -- and the
	// instance is element-major afterwards; the next step's explicit stages read it copy by copy ("unique_mixed")
	{
		UniqueLayout & u = e->u;
		bool all_set = true;
		for (int lp : e->local_patches) all_set = all_set && e->patches[lp].physics_set;
		if (u.built && !e->sw && all_set && instance >= 0 && instance < e->cfg.n_instances && u.form[instance] == 1 && e->imap[instance] == instance && u.mixed_option) {
			bool shared = false;
			for (int b = 0; b < (int)e->imap.size(); b++) if (b != instance && (e->imap[b] == instance || e->uvmap[b] == instance)) shared = true;
			if (!shared) {
				if ((r = u_own_uv(e, instance))) return r;
				ProfScope ps(e, TMX_K_LINCOMB);
				tmxk_held_suarez(e, make_params(e), inst(e, instance), e->track_surface ? surface_slots(e, instance) : nullptr, dt,
					uinst(e, instance), (size_t)u.NUS, (const int *)u.d_ucol_of_dcol);
				u.form[instance] = 0; u.n_uform--; u.conversions++;
				return launch_check("physics_held_suarez (node-unique in, element-major out)");
			}
		}
	}
	if ((r = check_inst(e, instance))) return r;
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "Held-Suarez physics with the shallow-water equation set is not supported");
	for (int lp : e->local_patches)
		REQUIRE(e->patches[lp].physics_set, TMX_ERR_INVALID, "tmx_set_patch_physics_inputs was not called for patch %d", lp);
	ProfScope ps(e, TMX_K_LINCOMB);
	tmxk_held_suarez(e, make_params(e), inst(e, instance), e->track_surface ? surface_slots(e, instance) : nullptr, dt);
	return launch_check("physics_held_suarez");
}
// the same on the unique columns of an instance that tmx_step left in node-unique form (no conversion there and back): the forcing is
// column by column, so with inputs that agree on all copies of a node (checked) one evaluation per node is every copy's result.
// *done = false: not applicable, the caller takes the element-major route.
static int held_suarez_unique(tmx_engine * e, int instance, double dt, bool * done) {
	*done = false;
	UniqueLayout & u = e->u;
	if (!u.built || instance < 0 || instance >= e->cfg.n_instances || u.form[instance] == 0 || e->imap[instance] != instance) return TMX_OK;
	for (int lp : e->local_patches) if (!e->patches[lp].physics_set) return TMX_OK;      // (the element-major route reports it)
	bool ok = false;
	int r = tmxu_physics_inputs(e, &ok);
	if (r || !ok) return r;
	if ((r = u_own_uv(e, instance))) return r;
	ProfScope ps(e, TMX_K_LINCOMB);
	// the tracked surface slots live with the element-major slot (the stage algebra of tmx_step keeps them there); their copies agree
	// whenever the instance's do (they are covered by the check that admitted the instance to the node-unique form)
	if (e->track_surface) tmxuk_gather_rows(e, 2, surface_slots(e, instance), u.d_surf_u);
	tmxk_held_suarez(e, tmxu_params_columns(e, make_params(e)), uinst(e, instance), e->track_surface ? u.d_surf_u : nullptr, dt);
	u.form[instance] = 1;      // (an element-major copy kept for a reader is stale now)
	u_written(e, instance);
	*done = true;
	return launch_check("physics_held_suarez (node-unique)");
}

// ---- Kessler microphysics (BASELINE config 4) ----
extern "C" int tmx_set_patch_level_heights(tmx_engine * e, int patch, const double * z_levels) {
	REQUIRE(e && z_levels, TMX_ERR_INVALID, "tmx_set_patch_level_heights: null argument");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index out of range");
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "column physics with the shallow-water equation set is not supported");
	int r = ensure_layout(e);
	if (r) return r;
	PatchInfo & P = e->patches[patch];
	REQUIRE(P.owner == e->cfg.rank, TMX_ERR_INVALID, "patch %d is not owned by rank %d", patch, e->cfg.rank);
	const int L = e->L;
	const size_t NS = e->NS;
	if (e->h_zlev.empty()) e->h_zlev.assign((size_t)L * NS, 0.0);
	for (int i = 1; i < P.na - 1; i++)
	for (int j = 1; j < P.nb - 1; j++) {
		const int c = col_of(P, i, j);
		for (int k = 0; k < L; k++) e->h_zlev[(size_t)k * NS + c] = z_levels[((size_t)i * P.nb + j) * L + k];
	}
	P.zlev_set = true; e->zlev_dirty = true;
	return TMX_OK;
}

extern "C" int tmx_physics_kessler(tmx_engine * e, int instance, double dt) {
	int r; if ((r = check_ready(e)) || (r = check_inst(e, instance))) return r;
	REQUIRE(!e->sw, TMX_ERR_UNSUPPORTED, "Kessler physics with the shallow-water equation set is not supported");
	REQUIRE(e->nt >= 3, TMX_ERR_INVALID, "Kessler physics needs the tracers RhoQv, RhoQc, RhoQr (n_tracers >= 3)");
	REQUIRE(dt > 0.0, TMX_ERR_INVALID, "tmx_physics_kessler: dt must be positive");
	for (int lp : e->local_patches)
		REQUIRE(e->patches[lp].zlev_set, TMX_ERR_INVALID, "tmx_set_patch_level_heights was not called for patch %d", lp);
	const size_t NS = e->NS; const int L = e->L;
	if (!e->d_zlev) {
		HIPCHK(hipMalloc((void **)&e->d_zlev, (size_t)L * NS * sizeof(double)));
		HIPCHK(hipMalloc((void **)&e->d_prect, NS * sizeof(double)));
		HIPCHK(hipMemset(e->d_prect, 0, NS * sizeof(double)));
		HIPCHK(hipMalloc((void **)&e->d_kes, (size_t)8 * L * NS * sizeof(double)));
		e->hbm_bytes += (size_t)(9 * L + 1) * NS * sizeof(double);
	}
	if (e->zlev_dirty) {
		HIPCHK(hipStreamSynchronize(e->stream));
		HIPCHK(hipMemcpy(e->d_zlev, e->h_zlev.data(), (size_t)L * NS * sizeof(double), hipMemcpyHostToDevice));
		e->zlev_dirty = false;
	}
	ProfScope ps(e, TMX_K_LINCOMB);
	tmxk_kessler(e, make_params(e), inst(e, instance), dt);
	return launch_check("physics_kessler");
}

extern "C" int tmx_download_precipitation(tmx_engine * e, int patch, double * prect, int reset) {
	int r; if ((r = check_ready(e))) return r;
	REQUIRE(prect, TMX_ERR_INVALID, "tmx_download_precipitation: null argument");
	REQUIRE(patch >= 0 && patch < e->cfg.n_patches, TMX_ERR_INVALID, "patch index out of range");
	PatchInfo & P = e->patches[patch];
	REQUIRE(P.owner == e->cfg.rank, TMX_ERR_INVALID, "patch %d is not owned by rank %d", patch, e->cfg.rank);
	std::vector<double> v((size_t)P.nea * P.neb * TMX_NQ, 0.0);
	const int c0 = P.elem_base * TMX_NQ;
	if (e->d_prect) {
		HIPCHK(hipStreamSynchronize(e->stream));
		HIPCHK(hipMemcpy(v.data(), e->d_prect + c0, v.size() * sizeof(double), hipMemcpyDeviceToHost));
		if (reset) HIPCHK(hipMemset(e->d_prect + c0, 0, v.size() * sizeof(double)));
	}
	for (int i = 1; i < P.na - 1; i++)
	for (int j = 1; j < P.nb - 1; j++) prect[(size_t)i * P.nb + j] = v[col_of(P, i, j) - c0];
	return TMX_OK;
}

extern "C" int tmx_sync(tmx_engine * e) {
	int r; if ((r = check_ready(e))) return r;
	HIPCHK(hipStreamSynchronize(e->stream));
	prof_collect(e);
	int flag = 0;
	HIPCHK(hipMemcpy(&flag, e->d_flag, sizeof(int), hipMemcpyDeviceToHost));
	if (flag) HIPCHK(hipMemset(e->d_flag, 0, sizeof(int)));
	if (flag & TMX_FLAG_COMM) {
		// (a singular matrix reported in the same interval is a consequence: the columns were solved on stale ghost data)
		tmx_set_error("halo exchange: a neighbour rank's message did not arrive within the time-out (TMX_P2P_TIMEOUT_S = %d s) (peer-to-peer transport); "
			"the state of this engine is no longer valid: upload it again after tmx_halo_p2p_reset on every rank", e->p2p_timeout_s);
		return TMX_ERR_COMM;
	}
	if (flag & TMX_FLAG_SINGULAR) {
		tmx_set_error("column solve failed: exactly singular band matrix (LAPACK dgbsv info > 0)");
		return TMX_ERR_SINGULAR;
	}
	return TMX_OK;
}

