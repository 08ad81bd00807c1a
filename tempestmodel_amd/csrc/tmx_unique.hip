// tmx_unique.hip -- the node-unique state layout (UniqueLayout, tmx_internal.h): host-side tables, the
// conversions between the element-major layout with duplicated seam nodes (D) and the per-patch node-unique slabs (U), and
// the DSS kernel restricted to the seam nodes that a producing wavefront could not average itself (tile and patch edges).
//
// Reference behaviour restated: GridCSGLL::ApplyDSS (src/atm/GridCSGLL.cpp:435-781) -- inside a patch alpha seams are
// averaged before beta seams and every copy of a node receives the same bits; the copies of a node on DIFFERENT patches may
// differ in the last bit (each patch averages in its own frame), which is why the slabs are unique per patch, not per grid.
// VerticalDynamicsFEM::StepImplicit (src/atm/VerticalDynamicsFEM.cpp:1315-1337, 1543-1633) already works on one copy per
// patch node (the copy with local node index 0 of the upper element) and copies its result to the others.
#include "tmx_device.h"
#include <algorithm>
#include <cstring>
#include <map>

// Generalised Hilbert curve over a w x h rectangle (the construction published as "gilbert": recursive halving along the longer
// side, odd halves rounded so that the curve stays continuous; a diagonal step can occur only on odd x odd rectangles).  Any run of
// consecutive cells is a compact region at ANY offset -- which is what the thread order needs: 4 consecutive elements (a wavefront)
// are edge-connected, 16 consecutive ones (a workgroup's block) have the perimeter of a 4 x 4 block give or take a cell.
static int floor_half(int v) { return (v >= 0) ? v / 2 : -((-v + 1) / 2); }
static int sign_of(int v) { return (v > 0) - (v < 0); }
static void gilbert_rec(int x, int y, int ax, int ay, int bx, int by, std::vector<std::pair<int, int>> & out) {
	const int w = std::abs(ax + ay), h = std::abs(bx + by);
	const int dax = sign_of(ax), day = sign_of(ay), dbx = sign_of(bx), dby = sign_of(by);
	if (h == 1) { for (int q = 0; q < w; q++) { out.push_back({ x, y }); x += dax; y += day; } return; }
	if (w == 1) { for (int q = 0; q < h; q++) { out.push_back({ x, y }); x += dbx; y += dby; } return; }
	int ax2 = floor_half(ax), ay2 = floor_half(ay), bx2 = floor_half(bx), by2 = floor_half(by);
	const int w2 = std::abs(ax2 + ay2), h2 = std::abs(bx2 + by2);
	if (2 * w > 3 * h) {
		if ((w2 % 2) && w > 2) { ax2 += dax; ay2 += day; }
		gilbert_rec(x, y, ax2, ay2, bx, by, out);
		gilbert_rec(x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by, out);
	} else {
		if ((h2 % 2) && h > 2) { bx2 += dbx; by2 += dby; }
		gilbert_rec(x, y, bx2, by2, ax2, ay2, out);
		gilbert_rec(x + bx2, y + by2, ax, ay, bx - bx2, by - by2, out);
		gilbert_rec(x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), -bx2, -by2, -(ax - ax2), -(ay - ay2), out);
	}
}
static std::vector<std::pair<int, int>> gilbert_order(int w, int h) {
	std::vector<std::pair<int, int>> out;
	if (w >= h) gilbert_rec(0, 0, w, 0, 0, h, out); else gilbert_rec(0, 0, 0, h, w, 0, out);
	return out;
}

template <class T> static int up(T ** d, const std::vector<T> & h, size_t * bytes) {
	const size_t n = h.size() ? h.size() : 1;
	HIPCHK(hipMalloc((void **)d, n * sizeof(T)));
	if (h.size()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
	*bytes += n * sizeof(T);
	return TMX_OK;
}

void tmxu_free(tmx_engine * e) {
	UniqueLayout & u = e->u;
	void * ptrs[] = { u.d_ustate, u.d_part, u.d_g2d_t, u.d_g2d_u, u.d_t_ucol, u.d_t_sdst, u.d_t_sred, u.d_t_tinfo, u.d_t_dcol, u.d_ucol_of_dcol, u.d_u_rep, u.d_slot_ucol,
		u.d_gsrc, u.d_gdst, u.d_gn, u.d_gx, u.d_gtype, u.d_send_slots, u.d_quads_early, u.d_quads_late, u.d_check, u.d_surf_u,
		u.d_b_sdst, u.d_b_sred, u.d_blk_info, u.d_b_gsrc, u.d_b_gdst, u.d_b_gn, u.d_b_gx, u.d_b_gtype, u.d_blks_early, u.d_blks_late };
	for (void * q : ptrs) if (q) hipFree(q);
	u = UniqueLayout();
}

// Tables of the layout: host part (no device; also what tmx_debug_unique_tables hands to the CPU tests)
int tmxu_tables(tmx_engine * e, const UniquePlanInput & in, UniqueTables & T) {
	UniqueLayout & u = e->u;
	const int NS = e->NS;
	const std::vector<int> & gc = *in.grp_cols, & gn = *in.grp_n, & gx = *in.grp_x, & gt = *in.grp_type;
	// ---- U numbering: patch by patch, row-major in (alpha node I, beta node J)
	std::vector<int> ubase(e->cfg.n_patches, -1), patch_of_elem(e->ne_local, -1);
	int nu = 0;
	for (int p : e->local_patches) {
		const PatchInfo & P = e->patches[p];
		ubase[p] = nu; nu += (3 * P.nea + 1) * (3 * P.neb + 1);
		for (int q = 0; q < P.nea * P.neb; q++) patch_of_elem[P.elem_base + q] = p;
	}
	u.NU = nu; u.NUS = (nu + 63) / 64 * 64;
	REQUIRE(u.NU == e->nunique, TMX_ERR_INVALID, "internal: %d unique nodes, %d unique columns of the column solve", u.NU, e->nunique);
	std::vector<int> ucol_of_dcol(NS, -1), u_rep(u.NUS, 0);
	for (int p : e->local_patches) {
		const PatchInfo & P = e->patches[p];
		const int NJ = 3 * P.neb + 1;
		for (int a = 0; a < P.nea; a++) for (int b = 0; b < P.neb; b++) for (int ii = 0; ii < 4; ii++) for (int jj = 0; jj < 4; jj++) {
			const int col = (P.elem_base + a * P.neb + b) * 16 + ii * 4 + jj;
			const int uu = ubase[p] + (3 * a + ii) * NJ + 3 * b + jj;
			ucol_of_dcol[col] = uu;
			// representative = the copy the reference's column solve works on (VerticalDynamicsFEM.cpp:1315-1337: the node is
			// skipped when it is the last of its element and another element follows): local index < 3, or the patch's last element
			const bool ra = (ii < 3) || (a == P.nea - 1), rb = (jj < 3) || (b == P.neb - 1);
			if (ra && rb) u_rep[uu] = col;
		}
	}
	for (int uu = u.NU; uu < u.NUS; uu++) u_rep[uu] = u_rep[u.NU - 1];
	// ---- T order: the rank's elements, four per wavefront
	std::vector<int> telem;      // D element of T element t
	if (u.tile_shape == 1) {
		std::vector<int> left;
		for (int p : e->local_patches) {
			const PatchInfo & P = e->patches[p];
			auto el = [&](int a, int b) { return P.elem_base + a * P.neb + b; };
			for (int a = 0; a + 1 < P.nea; a += 2) for (int b = 0; b + 1 < P.neb; b += 2) {
				telem.push_back(el(a, b)); telem.push_back(el(a, b + 1)); telem.push_back(el(a + 1, b)); telem.push_back(el(a + 1, b + 1));
			}
			// what 2 x 2 blocks do not cover: the last column / row of a patch with an odd element count, as strips
			if (P.neb & 1) for (int a = 0; a < (P.nea & ~1); a++) left.push_back(el(a, P.neb - 1));
			if (P.nea & 1) for (int b = 0; b < P.neb; b++) left.push_back(el(P.nea - 1, b));
		}
		telem.insert(telem.end(), left.begin(), left.end());
	} else if (u.tile_shape == 2) {
		// 1 x 4 strips along beta that never wrap around a patch row (the element-major order lets them wrap where the patch width
		// is not a multiple of four); what is left of every row goes to the end, again as runs along beta
		std::vector<int> left;
		for (int p : e->local_patches) {
			const PatchInfo & P = e->patches[p];
			for (int a = 0; a < P.nea; a++) {
				int b = 0;
				for (; b + 3 < P.neb; b += 4) for (int q = 0; q < 4; q++) telem.push_back(P.elem_base + a * P.neb + b + q);
				for (; b < P.neb; b++) left.push_back(P.elem_base + a * P.neb + b);
			}
		}
		telem.insert(telem.end(), left.begin(), left.end());
	} else if (u.tile_shape == 4) {
		// 4 x 4 element blocks of every patch, row by row inside a block: a wavefront is a 1 x 4 strip along beta as in the default order, a
		// workgroup of four wavefronts is a block, whose strip seams it averages itself (block tables below).  Blocks at the patch's ragged
		// edges (widths that are no multiple of four) are padded to whole wavefronts with padding elements (-1): 15 x 15 patches: 3 of 228.
		for (int p : e->local_patches) {
			const PatchInfo & P = e->patches[p];
			for (int a0 = 0; a0 < P.nea; a0 += 4) for (int b0 = 0; b0 < P.neb; b0 += 4) {
				const size_t first = telem.size();
				for (int a = a0; a < std::min(a0 + 4, P.nea); a++) for (int b = b0; b < std::min(b0 + 4, P.neb); b++) telem.push_back(P.elem_base + a * P.neb + b);
				while ((telem.size() - first) % 4) telem.push_back(-1);
				T.blk_first.push_back((int)first / 4); T.blk_nt.push_back((int)(telem.size() - first) / 4);
			}
		}
	} else if (u.tile_shape == 3) {
		// a space-filling curve through every patch (x = beta element, y = alpha element)
		for (int p : e->local_patches) {
			const PatchInfo & P = e->patches[p];
			for (const std::pair<int, int> & c : gilbert_order(P.neb, P.nea)) telem.push_back(P.elem_base + c.second * P.neb + c.first);
		}
	} else {
		for (int q = 0; q < e->ne_local; q++) telem.push_back(q);
	}
	{
		int real = 0;
		for (int el : telem) real += (el >= 0);
		REQUIRE(real == e->ne_local, TMX_ERR_INVALID, "internal: thread order covers %d of %d elements", real, e->ne_local);
	}
	u.NTS = ((int)telem.size() * 16 + 63) / 64 * 64;
	// (the block order pads: its thread space may be larger than the column space; every other order is a permutation of the elements)
	REQUIRE(u.NTS == NS || u.tile_shape == 4, TMX_ERR_INVALID, "internal: thread space %d != column space %d", u.NTS, NS);
	if (u.tile_shape != 4) {      // every four consecutive wavefronts form a block (the last one may be short)
		for (int t = 0; t < u.NTS / 64; t += 4) { T.blk_first.push_back(t); T.blk_nt.push_back(std::min(4, u.NTS / 64 - t)); }
	}
	std::vector<int> t_dcol(u.NTS, -1), tcol_of_dcol(NS, -1), t_ucol(u.NTS, -1), t_sdst(u.NTS, -1), t_sred(u.NTS, 0);
	for (size_t t = 0; t < telem.size(); t++) if (telem[t] >= 0) for (int q = 0; q < 16; q++) {
		t_dcol[t * 16 + q] = telem[t] * 16 + q; tcol_of_dcol[telem[t] * 16 + q] = (int)t * 16 + q;
	}
	for (int t = 0; t < u.NTS; t++) if (t_dcol[t] >= 0) { t_ucol[t] = ucol_of_dcol[t_dcol[t]]; t_sdst[t] = t_ucol[t]; }
	// ---- DSS groups: averaged inside the wavefront, or left to the DSS kernel through partial slots
	const int ng = (int)gn.size();
	std::vector<char> intile(ng, 0);
	for (int g = 0; g < ng; g++) {
		const int n = gn[g];
		bool ok = (n == 2 || n == 4);
		int tile = -1, patch = -1;
		for (int m = 0; m < n && ok; m++) {
			const int c = gc[(size_t)g * 4 + m];
			if (c < 0 || c >= NS) { ok = false; break; }
			const int tc = tcol_of_dcol[c], pp = patch_of_elem[c / 16];
			if (m == 0) { tile = tc / 64; patch = pp; }
			else if (tc / 64 != tile || pp != patch) ok = false;
		}
		// inside one patch every member pairs alike (type bits equal): the owner's combination is every member's
		for (int m = 1; m < n && ok; m++) if (((gt[g] >> (2 * m)) & 3) != (gt[g] & 3)) ok = false;
		if (ok && gx[g] >= 0) ok = false;      // (never inside a patch: a group that spans panels)
		if (!ok) continue;
		intile[g] = 1;
		int lanes[4], owner = 0;
		for (int m = 0; m < n; m++) { lanes[m] = tcol_of_dcol[gc[(size_t)g * 4 + m]] & 63; if (lanes[m] < lanes[owner]) owner = m; }
		int l1, l2 = 0, l3 = 0;
		if (n == 2) l1 = lanes[1 - owner];
		else {
			// members are stored as [m0, alpha partner, beta partner, diagonal] of m0 with the owner's pairing type
			// (0 {01|23}, 1 {02|13}, 2 {03|12}): own + alpha partner first, then the other two
			const int ty = (gt[g] >> (2 * owner)) & 3;
			const int ap = (ty == 0) ? (owner ^ 1) : ((ty == 1) ? (owner ^ 2) : (3 - owner));
			int rest[2], nr = 0;
			for (int m = 0; m < 4; m++) if (m != owner && m != ap) rest[nr++] = m;
			l1 = lanes[ap]; l2 = lanes[rest[0]]; l3 = lanes[rest[1]];
		}
		const int tb = (tcol_of_dcol[gc[(size_t)g * 4]] / 64) * 64;
		for (int m = 0; m < n; m++) t_sdst[tb + lanes[m]] = (m == owner) ? t_ucol[tb + lanes[m]] : -1;
		t_sred[tb + lanes[owner]] = n | (l1 << 4) | (l2 << 10) | (l3 << 16);
	}
	// ---- the same one level up: groups whose copies all belong to ONE BLOCK (a workgroup of up to four wavefronts) are averaged by that
	// workgroup through LDS (block kernels: k_h_walk<.., BLK>, k_hypervis_blk); position of a thread in its block = 64 * wavefront + lane
	std::vector<int> blk_of_tile(u.NTS / 64, -1);
	for (size_t b = 0; b < T.blk_first.size(); b++) for (int w = 0; w < T.blk_nt[b]; w++) blk_of_tile[T.blk_first[b] + w] = (int)b;
	std::vector<char> inblock(ng, 0);
	std::vector<int> b_sdst = t_sdst, b_sred(u.NTS, 0);      // (threads of in-tile groups and lone nodes start from the wavefront-level entries; overwritten below)
	for (int t = 0; t < u.NTS; t++) if (t_dcol[t] >= 0) b_sdst[t] = t_ucol[t];
	for (int g = 0; g < ng; g++) {
		const int n = gn[g];
		bool ok = (n == 2 || n == 4);
		int blk = -1, patch = -1;
		for (int m = 0; m < n && ok; m++) {
			const int c = gc[(size_t)g * 4 + m];
			if (c < 0 || c >= NS) { ok = false; break; }
			const int tc = tcol_of_dcol[c], pp = patch_of_elem[c / 16];
			if (m == 0) { blk = blk_of_tile[tc / 64]; patch = pp; }
			else if (blk_of_tile[tc / 64] != blk || pp != patch) ok = false;
		}
		for (int m = 1; m < n && ok; m++) if (((gt[g] >> (2 * m)) & 3) != (gt[g] & 3)) ok = false;
		if (ok && gx[g] >= 0) ok = false;
		if (!ok) continue;
		inblock[g] = 1;
		const int tb = T.blk_first[blk] * 64;
		int pos[4], owner = 0;
		for (int m = 0; m < n; m++) { pos[m] = tcol_of_dcol[gc[(size_t)g * 4 + m]] - tb; if (pos[m] < pos[owner]) owner = m; }
		int l1, l2 = 0, l3 = 0;
		if (n == 2) l1 = pos[1 - owner];
		else {
			const int ty = (gt[g] >> (2 * owner)) & 3;
			const int ap = (ty == 0) ? (owner ^ 1) : ((ty == 1) ? (owner ^ 2) : (3 - owner));
			int rest[2], nr = 0;
			for (int m = 0; m < 4; m++) if (m != owner && m != ap) rest[nr++] = m;
			l1 = pos[ap]; l2 = pos[rest[0]]; l3 = pos[rest[1]];
		}
		for (int m = 0; m < n; m++) b_sdst[tb + pos[m]] = (m == owner) ? t_ucol[tb + pos[m]] : -1;
		b_sred[tb + pos[owner]] = n | (l1 << 4) | (l2 << 12) | (l3 << 20);
	}
	// partial slots in thread order
	std::vector<char> needs_slot(u.NTS, 0);
	for (int g = 0; g < ng; g++) if (!intile[g])
		for (int m = 0; m < gn[g]; m++) { const int c = gc[(size_t)g * 4 + m]; if (c >= 0 && c < NS) needs_slot[tcol_of_dcol[c]] = 1; }
	std::vector<int> slot_of_tcol(u.NTS, -1), slot_ucol;
	for (int t = 0; t < u.NTS; t++) if (needs_slot[t]) { slot_of_tcol[t] = (int)slot_ucol.size(); slot_ucol.push_back(t_ucol[t]); t_sdst[t] = -2 - slot_of_tcol[t]; }
	// (block tables: the slots keep their numbers; copies of groups no block averages go there as before)
	for (int g = 0; g < ng; g++) if (!inblock[g])
		for (int m = 0; m < gn[g]; m++) { const int c = gc[(size_t)g * 4 + m]; if (c >= 0 && c < NS) b_sdst[tcol_of_dcol[c]] = -2 - slot_of_tcol[tcol_of_dcol[c]]; }
	u.nslots = (int)slot_ucol.size();
	u.NP = std::max(64, (u.nslots + 63) / 64 * 64);
	slot_ucol.resize(u.NP, 0);
	std::vector<int> gsrc, gdst, ugn, ugx, ugt;
	u.ngroups = 0; u.ngroups_local = 0; u.b_ngroups = 0; u.b_ngroups_local = 0;
	for (int g = 0; g < ng; g++) {
		if (intile[g]) continue;
		int dst[4] = { -1, -1, -1, -1 };
		for (int m = 0; m < 4; m++) {
			const int c = (m < gn[g]) ? gc[(size_t)g * 4 + m] : -1;
			if (c < 0) { gsrc.push_back(-1); continue; }
			if (c >= NS) { gsrc.push_back(u.NP + (c - NS)); continue; }
			gsrc.push_back(slot_of_tcol[tcol_of_dcol[c]]);
			dst[m] = ucol_of_dcol[c];
			for (int q = 0; q < m; q++) if (dst[q] == dst[m]) dst[m] = -1;      // copies in one patch receive the same bits: one store
		}
		for (int m = 0; m < 4; m++) gdst.push_back(dst[m]);
		ugn.push_back(gn[g]); ugx.push_back(gx[g]); ugt.push_back(gt[g]);
		if (g < e->ngroups_local) u.ngroups_local++;
		u.ngroups++;
		if (!inblock[g]) {      // the DSS kernel's list behind a block kernel: the same entries, fewer of them
			for (int m = 0; m < 4; m++) { T.b_gsrc.push_back(gsrc[gsrc.size() - 4 + m]); T.b_gdst.push_back(dst[m]); }
			T.b_gn.push_back(gn[g]); T.b_gx.push_back(gx[g]); T.b_gt.push_back(gt[g]);
			if (g < e->ngroups_local) u.b_ngroups_local++;
			u.b_ngroups++;
		}
	}
	REQUIRE(gsrc.size() == 4 * (size_t)u.ngroups, TMX_ERR_INVALID, "internal: DSS group table");
	// what other ranks need: the slots of the sent columns; tiles that hold one run before the exchange starts
	std::vector<int> send_slots;
	std::vector<char> early(u.NTS / 64, 0);
	for (int c : *in.send_cols) {
		const int tc = tcol_of_dcol[c];
		REQUIRE(slot_of_tcol[tc] >= 0, TMX_ERR_INVALID, "internal: a column another rank needs has no partial slot");
		send_slots.push_back(slot_of_tcol[tc]); early[tc / 64] = 1;
	}
	std::vector<int> qe, ql;
	for (int t = 0; t < (int)early.size(); t++) (early[t] ? qe : ql).push_back(t);
	u.split = e->cfg.n_ranks > 1 && !qe.empty() && !ql.empty() && e->split_stage;
	u.n_early = (int)qe.size(); u.n_late = (int)ql.size();
	// the same split for block kernels: a block runs early if one of its wavefronts does
	for (size_t b = 0; b < T.blk_first.size(); b++) {
		bool ea = false;
		for (int w = 0; w < T.blk_nt[b]; w++) ea = ea || early[T.blk_first[b] + w];
		(ea ? T.blks_early : T.blks_late).push_back((int)b);
	}
	u.nblocks = (int)T.blk_first.size(); u.nb_early = (int)T.blks_early.size(); u.nb_late = (int)T.blks_late.size();
	T.b_sdst = b_sdst; T.b_sred = b_sred;
	T.blk_info.resize(T.blk_first.size());
	for (size_t b = 0; b < T.blk_first.size(); b++) T.blk_info[b] = T.blk_first[b] * 8 + T.blk_nt[b];
	T.t_dcol = t_dcol; T.t_ucol = t_ucol; T.t_sdst = t_sdst; T.t_sred = t_sred; T.ucol_of_dcol = ucol_of_dcol; T.u_rep = u_rep; T.slot_ucol = slot_ucol;
	T.gsrc = gsrc; T.gdst = gdst; T.gn = ugn; T.gx = ugx; T.gt = ugt; T.send_slots = send_slots; T.quads_early = qe; T.quads_late = ql;
	// per tile: where node (0,0) of each of its four elements lives and its patch's row pitch (ucol = base + i * pitch + j)
	T.tinfo.assign((size_t)(u.NTS / 64) * 8, -1);
	for (size_t t = 0; t < telem.size(); t++) {
		if (telem[t] < 0) continue;
		const int p = patch_of_elem[telem[t]];
		T.tinfo[(t / 4) * 8 + (t % 4)] = t_ucol[t * 16];
		T.tinfo[(t / 4) * 8 + 4 + (t % 4)] = 3 * e->patches[p].neb + 1;
		for (int q = 0; q < 16; q++) REQUIRE(t_ucol[t * 16 + q] == t_ucol[t * 16] + (q >> 2) * (3 * e->patches[p].neb + 1) + (q & 3), TMX_ERR_INVALID, "internal: tile table");
	}
	return TMX_OK;
}

// Tables + device buffers.  Called by tmx_finalize while the host copy of the 2-D geometry still exists.
int tmxu_build(tmx_engine * e, const UniquePlanInput & in, size_t * bytes) {
	UniqueLayout & u = e->u;
	UniqueTables T;
	int r = tmxu_tables(e, in, T);
	if (r) return r;
	const int NS = e->NS;
	// geometry in thread order and per unique column (the representative copy's)
	std::vector<double> g2t((size_t)G2_COUNT * u.NTS, 0.0), g2u((size_t)G2_COUNT * u.NUS, 0.0);
	for (int f = 0; f < G2_COUNT; f++) {
		for (int t = 0; t < u.NTS; t++)
			g2t[(size_t)f * u.NTS + t] = (T.t_dcol[t] >= 0) ? e->h_g2d[(size_t)f * NS + T.t_dcol[t]] : ((f == G2_J2D || f == G2_JN || f == G2_JE) ? 1.0 : 0.0);
		for (int uu = 0; uu < u.NUS; uu++) g2u[(size_t)f * u.NUS + uu] = e->h_g2d[(size_t)f * NS + T.u_rep[uu]];
	}
	// do the copies of every node agree, bit for bit, in the per-column geometry the explicitly evaluated implicit terms read (k_vi_terms_explicit:
	// horizontal metric, Jacobians, d_xi R, the factors of the closed-form 3-D metric)?  The reference evaluates that operator per stored copy.
	{
		static const int rows[] = { G2_C2A0, G2_C2A1, G2_C2B1, G2_DRX, G2_JE, G2_JN, G2_MP, G2_MA, G2_MB, G2_MC, G2_DAZ, G2_DBZ, G2_IDX, G2_IDX2 };
		u.vite_ok = true;
		for (int c = 0; c < e->ncol && u.vite_ok; c++) {
			const int rep = T.u_rep[T.ucol_of_dcol[c]];
			if (rep == c) continue;
			for (int f : rows) if (memcmp(&e->h_g2d[(size_t)f * NS + c], &e->h_g2d[(size_t)f * NS + rep], sizeof(double)) != 0) { u.vite_ok = false; break; }
		}
	}
	if ((r = up(&u.d_t_tinfo, T.tinfo, bytes))) return r;
	if ((r = up(&u.d_g2d_t, g2t, bytes)) || (r = up(&u.d_g2d_u, g2u, bytes))) return r;
	if ((r = up(&u.d_t_ucol, T.t_ucol, bytes)) || (r = up(&u.d_t_sdst, T.t_sdst, bytes)) || (r = up(&u.d_t_sred, T.t_sred, bytes)) || (r = up(&u.d_t_dcol, T.t_dcol, bytes))) return r;
	if ((r = up(&u.d_ucol_of_dcol, T.ucol_of_dcol, bytes)) || (r = up(&u.d_u_rep, T.u_rep, bytes)) || (r = up(&u.d_slot_ucol, T.slot_ucol, bytes))) return r;
	if ((r = up(&u.d_gsrc, T.gsrc, bytes)) || (r = up(&u.d_gdst, T.gdst, bytes)) || (r = up(&u.d_gn, T.gn, bytes)) || (r = up(&u.d_gx, T.gx, bytes)) || (r = up(&u.d_gtype, T.gt, bytes))) return r;
	if ((r = up(&u.d_send_slots, T.send_slots, bytes)) || (r = up(&u.d_quads_early, T.quads_early, bytes)) || (r = up(&u.d_quads_late, T.quads_late, bytes))) return r;
	if ((r = up(&u.d_b_sdst, T.b_sdst, bytes)) || (r = up(&u.d_b_sred, T.b_sred, bytes)) || (r = up(&u.d_blk_info, T.blk_info, bytes))) return r;
	if ((r = up(&u.d_b_gsrc, T.b_gsrc, bytes)) || (r = up(&u.d_b_gdst, T.b_gdst, bytes)) || (r = up(&u.d_b_gn, T.b_gn, bytes)) || (r = up(&u.d_b_gx, T.b_gx, bytes)) || (r = up(&u.d_b_gtype, T.b_gt, bytes))) return r;
	if ((r = up(&u.d_blks_early, T.blks_early, bytes)) || (r = up(&u.d_blks_late, T.blks_late, bytes))) return r;
	u.ustride = (size_t)e->nslab * u.NUS;
	const size_t ub = (size_t)(e->cfg.n_instances + 2) * u.ustride * sizeof(double), pb = (size_t)e->nslab * u.NP * sizeof(double);
	HIPCHK(hipMalloc((void **)&u.d_ustate, ub)); HIPCHK(hipMemset(u.d_ustate, 0, ub)); *bytes += ub;
	HIPCHK(hipMalloc((void **)&u.d_part, pb)); HIPCHK(hipMemset(u.d_part, 0, pb)); *bytes += pb;
	HIPCHK(hipMalloc((void **)&u.d_check, sizeof(int))); HIPCHK(hipMemset(u.d_check, 0, sizeof(int)));
	HIPCHK(hipMalloc((void **)&u.d_surf_u, (size_t)2 * u.NUS * sizeof(double))); HIPCHK(hipMemset(u.d_surf_u, 0, (size_t)2 * u.NUS * sizeof(double)));
	*bytes += (size_t)2 * u.NUS * sizeof(double);
	u.form.assign(e->cfg.n_instances, 0);
	u.uslot.resize(e->cfg.n_instances);
	for (int k = 0; k < e->cfg.n_instances; k++) u.uslot[k] = k;
	u.uspare = e->cfg.n_instances; u.uprefix = e->cfg.n_instances + 1;
	u.n_uform = 0;
	u.built = true;
	return TMX_OK;
}

KParams tmxu_params(const tmx_engine * e, const KParams & base) {
	KParams p = base;
	const UniqueLayout & u = e->u;
	p.NS = u.NTS; p.ncol = u.NTS; p.g2d = u.d_g2d_t;
	p.NSS = u.NUS; p.t_ucol = u.d_t_ucol; p.t_tinfo = u.d_t_tinfo; p.t_sdst = u.d_t_sdst; p.t_sred = u.d_t_sred; p.part = u.d_part; p.NP = u.NP;
	p.NSD = e->NS; p.t_dcol = u.d_t_dcol; p.b_sdst = u.d_b_sdst; p.b_sred = u.d_b_sred; p.blk_info = u.d_blk_info;
	return p;
}

KParams tmxu_params_columns(const tmx_engine * e, const KParams & base) {
	KParams p = base;
	const UniqueLayout & u = e->u;
	p.NS = u.NUS; p.ncol = u.NU; p.g2d = u.d_g2d_u; p.NSS = u.NUS; p.quads = nullptr;
	return p;
}

// ---------------------------------------------------------------------------------------------
// conversions

// U <- D: every unique node takes the value of its representative copy.  check: also raise *flag when some copy of a node
// differs from the representative (bitwise) -- the slab then has no node-unique form.
__global__ __launch_bounds__(256) void k_d_to_u(int nslab, size_t NS, size_t NUS, int NU, const double * __restrict__ xd, double * __restrict__ xu,
	const int * __restrict__ u_rep)
{
	const int uu = blockIdx.x * 256 + threadIdx.x;
	if (uu >= NU) return;
	const int c = u_rep[uu];
	for (int s = blockIdx.y; s < nslab; s += gridDim.y) xu[(size_t)s * NUS + uu] = xd[(size_t)s * NS + c];
}

__global__ __launch_bounds__(256) void k_check_copies(int nslab, size_t NS, int ncol, const double * __restrict__ xd,
	const int * __restrict__ ucol_of_dcol, const int * __restrict__ u_rep, int * __restrict__ flag)
{
	const int c = blockIdx.x * 256 + threadIdx.x;
	if (c >= ncol) return;
	const int r = u_rep[ucol_of_dcol[c]];
	if (r == c) return;
	bool bad = false;
	for (int s = blockIdx.y; s < nslab; s += gridDim.y) {
		const unsigned long long a = __double_as_longlong(xd[(size_t)s * NS + c]), b = __double_as_longlong(xd[(size_t)s * NS + r]);
		bad = bad || (a != b);
	}
	if (bad) atomicOr(flag, 1);
}

// D <- U: every copy takes its node's value
__global__ __launch_bounds__(256) void k_u_to_d(int nslab, size_t NS, size_t NUS, int ncol, const double * __restrict__ xu, double * __restrict__ xd,
	const int * __restrict__ ucol_of_dcol)
{
	const int c = blockIdx.x * 256 + threadIdx.x;
	if (c >= ncol) return;
	const int uu = ucol_of_dcol[c];
	for (int s = blockIdx.y; s < nslab; s += gridDim.y) xd[(size_t)s * NS + c] = xu[(size_t)s * NUS + uu];
}

// do the copies of every node of this element-major slot agree, bit for bit?  (one pass over the slot and a host synchronisation)
int tmxu_check(tmx_engine * e, int slot, bool * consistent) {
	UniqueLayout & u = e->u;
	const double * xd = e->d_state + (size_t)slot * e->inst_stride;
	const int ny = std::min(e->nslab, 64);
	// (with tracked surface slots -- the two rows behind the slabs that the Held-Suarez forcing reads -- those are checked too: the
	// forcing on the unique columns relies on it)
	const int nrows = e->nslab + (e->track_surface ? 2 : 0);
	HIPCHK(hipMemsetAsync(u.d_check, 0, sizeof(int), e->stream));
	hipLaunchKernelGGL(k_check_copies, dim3((e->ncol + 255) / 256, ny), dim3(256), 0, e->stream, nrows, (size_t)e->NS, e->ncol, xd,
		(const int *)u.d_ucol_of_dcol, (const int *)u.d_u_rep, u.d_check);
	int h = 0;
	HIPCHK(hipMemcpyAsync(&h, u.d_check, sizeof(int), hipMemcpyDeviceToHost, e->stream));
	HIPCHK(hipStreamSynchronize(e->stream));
	*consistent = (h == 0);
	return TMX_OK;
}

int tmxu_to_u(tmx_engine * e, int slot, bool check, bool * consistent) {
	UniqueLayout & u = e->u;
	if (consistent) *consistent = true;
	if (u.form[slot] != 0) { u.form[slot] = 1; return TMX_OK; }      // (form 2: both copies were current; the caller is about to write the node-unique one)
	const double * xd = e->d_state + (size_t)slot * e->inst_stride;
	double * xu = u.d_ustate + (size_t)u.uslot[slot] * u.ustride;
	const int ny = std::min(e->nslab, 64);
	if (check) {
		bool ok = true;
		int r = tmxu_check(e, slot, &ok);
		if (r) return r;
		if (!ok) { if (consistent) *consistent = false; return TMX_OK; }
	}
	hipLaunchKernelGGL(k_d_to_u, dim3((u.NU + 255) / 256, ny), dim3(256), 0, e->stream, e->nslab, (size_t)e->NS, (size_t)u.NUS, u.NU, xd, xu, (const int *)u.d_u_rep);
	u.form[slot] = 1; u.n_uform++; u.conversions++;
	return TMX_OK;
}

// keep_u: the caller only reads -- the node-unique copy stays valid beside the element-major one (form 2)
int tmxu_to_d(tmx_engine * e, int slot, bool keep_u) {
	UniqueLayout & u = e->u;
	if (u.form[slot] == 0) return TMX_OK;
	if (u.form[slot] == 2) {      // the element-major copy is current already
		if (!keep_u) { u.form[slot] = 0; u.n_uform--; }
		return TMX_OK;
	}
	double * xd = e->d_state + (size_t)slot * e->inst_stride;
	const double * xu = u.d_ustate + (size_t)u.uslot[slot] * u.ustride;
	const int ny = std::min(e->nslab, 64);
	hipLaunchKernelGGL(k_u_to_d, dim3((e->ncol + 255) / 256, ny), dim3(256), 0, e->stream, e->nslab, (size_t)e->NS, (size_t)u.NUS, e->ncol, xu, xd, (const int *)u.d_ucol_of_dcol);
	u.conversions++;
	if (keep_u) u.form[slot] = 2;
	else { u.form[slot] = 0; u.n_uform--; }
	return TMX_OK;
}

// ---------------------------------------------------------------------------------------------
// partial slots and the DSS kernel that finishes them

// slot <- the value of its node (an instance that is consistent inside patches, e.g. after the column solve: the DSS of such an
// instance changes the nodes on patch edges only, and the general kernel below does that from the slots)
__global__ __launch_bounds__(256) void k_slots_from_u(int nslab, size_t NUS, size_t NP, int nslots, const double * __restrict__ xu, double * __restrict__ part,
	const int * __restrict__ slot_ucol)
{
	const int q = blockIdx.x * 256 + threadIdx.x;
	if (q >= nslots) return;
	const int uu = slot_ucol[q];
	for (int s = blockIdx.y; s < nslab; s += gridDim.y) part[(size_t)s * NP + q] = xu[(size_t)s * NUS + uu];
}

void tmxuk_slots_from_u(tmx_engine * e, const double * xu) {
	const UniqueLayout & u = e->u;
	if (u.nslots == 0) return;
	hipLaunchKernelGGL(k_slots_from_u, dim3((u.nslots + 255) / 256, std::min(e->nslab, 64)), dim3(256), 0, e->stream, e->nslab, (size_t)u.NUS, (size_t)u.NP, u.nslots, xu,
		u.d_part, (const int *)u.d_slot_ucol);
}

// k_dss (tmx_k_exchange.hip) with the copies read from the partial slots (or the ghost buffer) and the results written to the
// unique slabs: one thread per group, level and class of variables, every member's own combination in the reference's order (dss_combine), (U,V)
// of copies on other panels rotated into the member's frame.  Copies of a node inside one patch receive identical bits, so only the
// first of them stores (gdst == -1 for the others).
// (round 6: two / four levels per thread, the group's tables read once and all copies in flight together, cost +6 % / +40 %: the kernel
// lives on the number of threads in flight, not on the bytes its tables take.  The other direction pays a little: a wavefront = ONE class
// of variables (vc: W | rho*theta | rho | U,V) of one level -- four times the threads, 56 registers instead of 78, eight wavefronts per
// SIMD instead of six: 0.216 - 0.219 -> 0.205 - 0.212 ms per step for the six launches, profiles/r06_dss_and_column_null_results.txt)
template <int vc>
__device__ __forceinline__ void dss_u_body(int L, size_t NUS, size_t NP, double * __restrict__ xu, const double * __restrict__ part,
	const double * __restrict__ ghost, const int * __restrict__ gbase, const int * __restrict__ gstride, int ngroups, const int * __restrict__ gsrc,
	const int * __restrict__ gdst, const int * __restrict__ gn, const int * __restrict__ gx, const int * __restrict__ gtype, const double * __restrict__ xmat, int g, int kk)
{
	if (vc > 0 && kk >= L) return;
	const int n = gn[g];
	const int4 c4 = *(const int4 *)(gsrc + g * 4), d4 = *(const int4 *)(gdst + g * 4);
	const int c[4] = { c4.x, c4.y, c4.z, c4.w }, d[4] = { d4.x, d4.y, d4.z, d4.w };
	const int xi = gx[g], gt = gtype[g];
	const double * ptr[4]; size_t str[4];
#pragma unroll
	for (int m = 0; m < 4; m++) {
		if (m >= n) { ptr[m] = part; str[m] = 0; }
		else if (c[m] < (int)NP) { ptr[m] = part + c[m]; str[m] = NP; }
		else { ptr[m] = ghost + gbase[c[m] - (int)NP]; str[m] = (size_t)gstride[c[m] - (int)NP]; }
	}
	const bool lev = (kk < L);
	int slabs[5];
	slabs[0] = TMX_SLAB_W(L, kk);
	slabs[1] = lev ? TMX_SLAB_T(L, kk) : slabs[0]; slabs[2] = lev ? TMX_SLAB_R(L, kk) : slabs[0];
	slabs[3] = lev ? TMX_SLAB_U(L, kk) : slabs[0]; slabs[4] = lev ? TMX_SLAB_V(L, kk) : slabs[0];
	auto mine = [&](int t) { return vc == 3 ? t >= 3 : t == vc; };      // the slabs of this wavefront's class
	double v[5][4];
#pragma unroll
	for (int t = 0; t < 5; t++)
#pragma unroll
		for (int m = 0; m < 4; m++) v[t][m] = (m < n && (t == 0 || lev) && mine(t)) ? ptr[m][(size_t)slabs[t] * str[m]] : 0.0;
	const bool same = (gt == 0) && (n != 3);
	const int t0 = gt & 3, t1 = (gt >> 2) & 3, t2 = (gt >> 4) & 3, t3 = (gt >> 6) & 3;
#define DSS_SCALAR(W0, W1, W2, W3, O0, O1, O2, O3) do { \
		if (same) { const double a_ = dss_combine<0>(W0, W1, W2, W3, n, 0); O0 = a_; O1 = a_; O2 = a_; O3 = a_; } \
		else { O0 = dss_combine<0>(W0, W1, W2, W3, n, t0); O1 = dss_combine<1>(W0, W1, W2, W3, n, t1); \
		       O2 = dss_combine<2>(W0, W1, W2, W3, n, t2); O3 = dss_combine<3>(W0, W1, W2, W3, n, t3); } } while (0)
	double out[5][4];
#pragma unroll
	for (int t = 0; t < 3; t++) if (mine(t)) DSS_SCALAR(v[t][0], v[t][1], v[t][2], v[t][3], out[t][0], out[t][1], out[t][2], out[t][3]);
	if (!mine(3)) { }
	else if (xi < 0) {
		DSS_SCALAR(v[3][0], v[3][1], v[3][2], v[3][3], out[3][0], out[3][1], out[3][2], out[3][3]);
		DSS_SCALAR(v[4][0], v[4][1], v[4][2], v[4][3], out[4][0], out[4][1], out[4][2], out[4][3]);
	} else {
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
			if (d[m] >= 0 && (t == 0 || lev) && mine(t)) xu[(size_t)slabs[t] * NUS + d[m]] = out[t][m];
#undef DSS_SCALAR
}
__global__ __launch_bounds__(256, 8) void k_dss_u(int L, int nt, size_t NUS, size_t NP, double * __restrict__ xu, const double * __restrict__ part,
	const double * __restrict__ ghost, const int * __restrict__ gbase, const int * __restrict__ gstride, int ngroups, const int * __restrict__ gsrc,
	const int * __restrict__ gdst, const int * __restrict__ gn, const int * __restrict__ gx, const int * __restrict__ gtype, const double * __restrict__ xmat, int g0)
{
	int bx, by;
	(void)nt;
	// a workgroup = the four classes of variables of one level of 64 groups
	if (!xcd_tile((ngroups - g0 + 63) / 64, L + 1, bx, by)) return;
	const int kk = by, vc = WAVE_UNIFORM(threadIdx.y);
	const int g = g0 + bx * 64 + threadIdx.x;
	if (g >= ngroups || kk > L) return;
#define DSS_GO(VC_) dss_u_body<VC_>(L, NUS, NP, xu, part, ghost, gbase, gstride, ngroups, gsrc, gdst, gn, gx, gtype, xmat, g, kk)
	if (vc == 0) DSS_GO(0); else if (vc == 1) DSS_GO(1); else if (vc == 2) DSS_GO(2); else DSS_GO(3);
#undef DSS_GO
}

// blocks: the groups a block kernel left (UniqueLayout::b_*) instead of the groups a wavefront-level kernel leaves
void tmxuk_dss(tmx_engine * e, double * xu, int g0, int g1, bool blocks) {
	if (g1 <= g0) return;
	const UniqueLayout & u = e->u;
	const int nx = (g1 - g0 + 63) / 64, ny = e->L + 1;
	hipLaunchKernelGGL(k_dss_u, dim3(8 * ((nx + 7) / 8) * ny), dim3(64, 4), 0, e->stream, e->L, e->nt, (size_t)u.NUS, (size_t)u.NP, xu, (const double *)u.d_part,
		(const double *)e->d_ghost, (const int *)e->d_ghost_base, (const int *)e->d_ghost_stride, g1, (const int *)(blocks ? u.d_b_gsrc : u.d_gsrc), (const int *)(blocks ? u.d_b_gdst : u.d_gdst),
		(const int *)(blocks ? u.d_b_gn : u.d_gn), (const int *)(blocks ? u.d_b_gx : u.d_gx), (const int *)(blocks ? u.d_b_gtype : u.d_gtype), (const double *)e->d_xmat, g0);
}

// ---------------------------------------------------------------------------------------------
// column physics on the unique columns

void tmxuk_gather_rows(tmx_engine * e, int nrows, const double * xd, double * xu) {
	const UniqueLayout & u = e->u;
	hipLaunchKernelGGL(k_d_to_u, dim3((u.NU + 255) / 256, std::min(nrows, 64)), dim3(256), 0, e->stream, nrows, (size_t)e->NS, (size_t)u.NUS, u.NU, xd, xu, (const int *)u.d_u_rep);
}

// The per-column inputs of the Held-Suarez forcing (rows G2_LAT .. G2_COSLAT of the 2-D geometry: latitude, pinned surface
// pressure, sin and cos of the latitude) per unique column, and whether the forcing may run there at all: only if every copy
// of a node carries the same input bits -- then every copy would receive the same result, which is what one unique column holds.
int tmxu_physics_inputs(tmx_engine * e, bool * ok) {
	UniqueLayout & u = e->u;
	if (u.physics_dirty) {
		static_assert(G2_PS == G2_LAT + 1 && G2_SINLAT == G2_LAT + 2 && G2_COSLAT == G2_LAT + 3, "physics input rows are contiguous");
		const double * xd = e->d_g2d + (size_t)G2_LAT * e->NS;
		HIPCHK(hipMemsetAsync(u.d_check, 0, sizeof(int), e->stream));
		hipLaunchKernelGGL(k_check_copies, dim3((e->ncol + 255) / 256, 4), dim3(256), 0, e->stream, 4, (size_t)e->NS, e->ncol, xd,
			(const int *)u.d_ucol_of_dcol, (const int *)u.d_u_rep, u.d_check);
		tmxuk_gather_rows(e, 4, xd, u.d_g2d_u + (size_t)G2_LAT * u.NUS);
		int h = 0;
		HIPCHK(hipMemcpyAsync(&h, u.d_check, sizeof(int), hipMemcpyDeviceToHost, e->stream));
		HIPCHK(hipStreamSynchronize(e->stream));
		u.physics_ok = (h == 0);
		u.physics_dirty = false;
	}
	*ok = u.physics_ok;
	return TMX_OK;
}
