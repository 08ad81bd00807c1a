// tmx_internal.h -- engine internals shared by the host side (tmx_host.hip, tmx_step.hip, tmx_unique.hip) and the kernels
// (tmx_k_*.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/tempest_mi355x.h"

#define TMX_NP 4            // GLL nodes per element side (horizontal order 4)
#define TMX_NQ 16           // nodes per element
#define TMX_TILE 64         // columns per workgroup row = 4 elements = one wavefront
// bits of the engine's deferred-failure word (d_flag), reported by tmx_sync
#define TMX_FLAG_SINGULAR 1    // a column solve met an exactly zero pivot
#define TMX_FLAG_COMM 2        // a neighbour's halo message did not arrive (peer-to-peer transport)
#define TMX_FTOT 3          // unknowns per level of the implicit column system (rho*theta, W, rho)
#define TMX_BW 9            // band entries per row of the column Jacobian (kl = ku = 4)

// variable slabs inside one state instance: var v, level k -> slab index; every slab is NS doubles
// U: [0,L)  V: [L,2L)  T(rho*theta): [2L,3L)  R(rho): [3L,4L)  W: [4L,5L]   (5L+1 slabs)
#define TMX_SLAB_U(L,k) (k)
#define TMX_SLAB_V(L,k) ((L) + (k))
#define TMX_SLAB_T(L,k) (2 * (L) + (k))
#define TMX_SLAB_R(L,k) (3 * (L) + (k))
#define TMX_SLAB_W(L,k) (4 * (L) + (k))
// tracer c, level k (after the 5L+1 state slabs)
#define TMX_SLAB_Q(L,c,k) (5 * (L) + 1 + (c) * (L) + (k))

// 2-D (per column) geometry slabs
enum { G2_J2D = 0, G2_F, G2_C2A0, G2_C2A1, G2_C2B1, G2_JN, G2_JE, G2_DRX, G2_ZS,
	G2_LAT, G2_PS,      // column physics inputs (tmx_set_patch_physics_inputs): latitude, Held-Suarez surface pressure
	G2_SINLAT, G2_COSLAT,   // sin / cos of the latitude, evaluated ON THE HOST with the libm the reference calls (HeldSuarezPhysics.cpp:186-200)
	// factors of the closed-form 3-D metric (tmx_set_patch_metric_factors): -s/d_xi R, 1+Y^2, XY, 1+X^2,
	// d_alpha zs, d_beta zs, 1/d_xi R, 1/(d_xi R)^2
	G2_MP, G2_MA, G2_MB, G2_MC, G2_DAZ, G2_DBZ, G2_IDX, G2_IDX2,
	// element spacing of the column's patch (tmx_set_patch_element_spacing): 1 / delta_alpha, 1 / delta_beta and the local
	// hyperviscosity scale (delta_alpha / reference length)^3.2 (HorizontalDynamicsFEM.cpp:1973)
	G2_IDA, G2_IDB, G2_NUS, G2_COUNT };
// 3-D node geometry [field][L][NS]
enum { G3N_CA2 = 0, G3N_CB2, G3N_CX2, G3N_DRA, G3N_DRB, G3N_COUNT };
// 3-D interface geometry [field][L+1][NS]
enum { G3E_CX0 = 0, G3E_CX1, G3E_CX2, G3E_COUNT };

// column operators in relative-offset form: coefficient of in[k+off], off = -2..+2
#define TMX_OPW 5

struct KParams {
	int L, ncol, NS;
	const double * g2d;     // [G2_COUNT][NS]
	const double * g3n;     // [G3N_COUNT][L][NS]
	const double * g3e;     // [G3E_COUNT][L+1][NS]
	const double * ops;     // [TMX_OP_COUNT][L+1][TMX_OPW]
	const double * eta;     // closed-form metric: 1 - eta of the levels [0,L) and of the interfaces [L, 2L+1)
	int closed;             // 1: evaluate the 3-D metric from the G2_M* factors (g3n / g3e are null)
	double inv_da;
	double grav, Rd, cp, p0;
	const int * quads;      // not null: blockIdx.x indexes this list of 64-column tiles (boundary-first / interior launches of a multi-rank stage)
	// Node-unique state layout (UniqueLayout below; kernels instantiated with UQ = true).  Threads are still (tile, lane) =
	// 4 whole elements x 16 nodes, geometry is indexed by the thread's own column (stride NS); the STATE lives in per-patch
	// node-unique slabs of NSS doubles, t_ucol names the slab entry a thread's node reads, t_sdst / t_sred say what it does with
	// its result (seam_store, tmx_device.h), `part` receives the per-element values of the seam nodes on tile and patch edges.
	int u_ntiles, u_xcd;    // U kernels are launched 1-D: tiles of this launch, and whether every XCD sweeps a contiguous range of them (xcd_tile)
	int NSS;                // state slab stride in doubles (== NS in the element-major layout)
	const int * t_ucol;     // [NS] unique-slab index of the thread's node, -1: padding lane
	const int * t_tinfo;    // [NS / 64][8] per tile {unique index of node (0,0) of its four elements (-1: padding), their patches' row pitch}: t_ucol without a
	                        //   per-lane load in front of every state load (wave-uniform: scalar loads)
	const int * t_sdst;     // [NS] >= 0: store the (averaged) value there; -1: nothing (another lane of the tile stores this node); <= -2: partial slot -2 - t_sdst
	const int * t_sred;     // [NS] n | lane1 << 4 | lane2 << 10 | lane3 << 16: average with the copies held by these lanes of the wavefront (n = 0, 2, 4)
	double * part;          // [nslab][NP] partial slots
	int NP;
	// block kernels (a workgroup = up to four wavefronts = one block of the thread order averages the seams between its wavefronts through LDS):
	int NSD;                // slab stride of ELEMENT-MAJOR instances (mixed reads): the thread space may be padded beyond it
	const int * t_dcol;     // [NS] stored (element-major) column of a thread, -1: padding -- where a mixed read finds the thread's own copy
	const int * b_sdst;     // [NS] as t_sdst with the block's averaging done: fewer partial slots
	const int * b_sred;     // [NS] n | pos1 << 4 | pos2 << 12 | pos3 << 20, positions 64 * wavefront + lane inside the block
	const int * blk_info;   // [blocks] first wavefront (tile) * 8 + wavefronts
	const int * bquads;     // not null: blockIdx.x indexes this list of blocks (boundary-first launches)
	double dx[16];          // dx[s*4+i]    = dDxBasis1D(s,i)
	double stiff[16];       // stiff[i*4+s] = dStiffness1D(i,s)
};

struct PatchInfo {
	bool defined = false, halo_set = false, geom_set = false, metric_ok = false, rayleigh_set = false, physics_set = false, ref_set = false, zlev_set = false;
	int panel = -1, nea = 0, neb = 0, na = 0, nb = 0, owner = 0;
	int nbp[8];
	int elem_base = -1;                      // first local element, -1 if not owned
	double da = 0.0, db = 0.0;               // GridPatchGLL::GetElementDeltaA / B of this patch (0: cfg.element_delta_a)
	std::vector<int> hi, hj, hsp, hsi, hsj, hspanel;
	std::vector<double> htrans;
	std::vector<int> hoff;                   // host node offset i * nb + j of every device column (built on first transfer)
};

struct StepGraphKey { int scheme, first, last; double dt; };
struct StepGraph { StepGraphKey key; hipGraphExec_t exec = nullptr; };

struct ProfSlot { double ms = 0.0; long long n = 0; };

// Node-unique state layout.  Inside a patch the co-located copies of a seam node carry identical bits
// after every averaging (GridCSGLL::ApplyDSS pairs alpha seams, then beta seams: both results are symmetric in the members),
// so a state instance can be held ONCE per patch node: (3 nea + 1) x (3 neb + 1) entries per patch and slab instead of 16 per
// element (9/16 of the bytes).  Kernels keep their thread = (element, node) shape and gather their operands; a seam node's
// per-element results are averaged in the reference's order inside the producing wavefront when all its copies belong to the
// wavefront's four elements, and otherwise go to "partial slots", which a DSS kernel restricted to those nodes (tile and patch
// edges, remote copies from the ghost buffer as before) finishes.  Three index spaces: D = element-major columns (the layout
// of everything else, host transfers included), T = thread order (a permutation of the rank's elements: 2 x 2 element blocks
// per wavefront), U = unique nodes.  Slots of the instance storage are in D or U form (form[]); U-form slots exist only inside
// tmx_step of a configuration all of whose kernels have the U form, every other entry point first converts what it names.
struct UniqueLayout {
	bool built = false;
	int option = -1;                         // tmx_set_option("unique_layout"): -1 default (= on where the configuration is eligible), 0 off, 1 on where eligible
	int tile_shape = 0;                      // T order: 0 = the element-major order (1 x 4 element strips that wrap around patch rows), 1 = 2 x 2 element blocks, 2 = 1 x 4 strips inside a patch row,
	                                         // 3 = a generalised Hilbert curve, 4 = 4 x 4 element blocks of 1 x 4 strips (padded at ragged patch edges)
	int xcd_order = 1;                       // workgroup -> tile map of the U kernels: 1 = every XCD sweeps a contiguous range of tiles
	int NU = 0, NUS = 0, NTS = 0, nslots = 0, NP = 0;
	int ngroups = 0, ngroups_local = 0;      // groups left to the DSS kernel (not averaged inside a wavefront); [0, ngroups_local): no remote member
	// the same behind a block kernel (groups no workgroup averaged), the blocks themselves, and which kind of kernel filled the partial slots last
	int b_ngroups = 0, b_ngroups_local = 0, nblocks = 0, nb_early = 0, nb_late = 0;
	int * d_b_sdst = nullptr, * d_b_sred = nullptr, * d_blk_info = nullptr, * d_b_gsrc = nullptr, * d_b_gdst = nullptr, * d_b_gn = nullptr, * d_b_gx = nullptr, * d_b_gtype = nullptr;
	int * d_blks_early = nullptr, * d_blks_late = nullptr;
	bool vite_ok = true;                     // every copy of every node carries the same bits in the per-column geometry k_vi_terms_explicit reads (else programs with StepImplicitTermsExplicitly run element-major)
	bool slots_by_blocks = false;            // the partial slots in memory were written by block kernels: the DSS runs the shorter list
	int block_option = -1;                   // tmx_set_option("unique_blocks"): -1 auto (on with the block thread order), 0 off, 1 on
	size_t ustride = 0;                      // doubles per U-form instance (nslab * NUS)
	double * d_ustate = nullptr, * d_part = nullptr, * d_g2d_t = nullptr, * d_g2d_u = nullptr;
	int * d_t_ucol = nullptr, * d_t_sdst = nullptr, * d_t_sred = nullptr;      // [NTS]
	int * d_t_tinfo = nullptr;               // [NTS / 64][8]
	int * d_t_dcol = nullptr;                // [NTS] D column of a thread (-1: padding)
	int * d_ucol_of_dcol = nullptr;          // [NS]
	int * d_u_rep = nullptr;                 // [NUS] D column of the copy the reference's column solve works on
	int * d_slot_ucol = nullptr;             // [NP] unique index of a slot's node
	int * d_gsrc = nullptr, * d_gdst = nullptr, * d_gn = nullptr, * d_gx = nullptr, * d_gtype = nullptr;
	int * d_send_slots = nullptr;
	int * d_quads_early = nullptr, * d_quads_late = nullptr; int n_early = 0, n_late = 0; bool split = false;
	int * d_check = nullptr;                 // consistency flag of the D -> U conversion
	std::vector<char> form;                  // per data instance: 0 = the element-major slot is current, 1 = the node-unique one, 2 = both (converted for a reader)
	// U storage holds one slot more than there are instances: a stage whose update instance is also one of its sources (in-place
	// accumulation, a combination with a non-zero coefficient of its destination) must not overwrite nodes other threads still
	// read -- in the unique slabs a node is read by up to four elements' threads -- so it writes the spare slot, which then
	// becomes the instance's
	std::vector<int> uslot; int uspare = 0;
	// and one more for the prefix of a later stage's combination that an explicit stage forms from the instances it reads anyway
	// (EmitTerms below, prefix_plan in tmx_step.hip); prefix_option: tmx_set_option("unique_prefix"), 1 = default since round 6 -- with the
	// level-parallel kernel the store of the prefix cost the producing stage what the consuming stage saved (ARS343, ne30 L30: 186.8 + 122.2 us
	// against 144.1 + 166.7 us); the column-segment walk is bound by its streams, and 4 source instances less are 0.02-0.03 ms per step
	int uprefix = 0, prefix_option = 1;
	long long prefix_for = -1;               // operation index of the stage the stored prefix belongs to (-1: none stored)
	long long prefix_stages = 0;             // stages that started from a stored prefix (tmx_info)
	int n_uform = 0;
	long long conversions = 0;               // D <-> U conversions performed (tmx_info)
	// an input whose copies differ makes the step run element-major; when that keeps happening (column physics between the steps
	// that treats the copies of a node differently in the last bit, as the reference's does) the check itself -- a pass over the
	// instance and a host synchronisation -- is not repeated every step: from the second failure in a row on, the next `skip_steps` steps do not try
	int backoff = 0, skip_steps = 0;      // backoff: failures in a row
	// Instances the running program reads in ELEMENT-MAJOR form, copy by copy, although everything it writes is node-unique: the model state
	// behind an operation the reference applies per copy with per-copy inputs (Held-Suarez forcing: the copies of a seam node differ in the
	// last bit afterwards), or a freshly uploaded state.  Only the explicit stages read such an instance (dlive_program_ok); the bit goes
	// when the program overwrites the instance.  No consistency check, no host synchronisation, no conversion.
	unsigned dlive = 0u;
	int mixed_option = 1;                 // tmx_set_option("unique_mixed"): 0 = the old rule (check the copies, convert or run the step element-major)
	long long mixed_steps = 0;            // steps that ran that way (tmx_info)
	// column physics on the unique columns (Held-Suarez forcing): its per-column inputs (latitude, sin / cos of it, pinned surface
	// pressure) per unique node, valid only if every copy of a node carries the same bits (checked on the device when they change)
	bool physics_dirty = true, physics_ok = false;
	double * d_surf_u = nullptr;             // [2][NUS] the tracked surface slots of the instance being forced, per unique column
};

struct tmx_engine {
	tmx_config cfg;
	int L = 0;
	bool ops_set = false, finalized = false;
	std::vector<PatchInfo> patches;
	std::vector<int> local_patches;

	// layout
	int ne_local = 0, ncol = 0, NS = 0, nslab = 0;
	size_t inst_stride = 0;

	// host staging of geometry in device layout (filled by tmx_set_patch_geometry)
	std::vector<double> h_g2d, h_g3n, h_g3e;
	std::vector<double> h_ops;
	std::vector<double> h_eta;               // [2L+1] 1 - eta (levels, then interfaces); empty until metric factors are set
	int nt = 0;                              // tracers
	std::vector<double> h_area;              // [L][NS] element_area_node (tracer filters)
	double * d_area = nullptr;
	double * d_w0 = nullptr;                 // [L+1][NS] W of the initial column, saved when StepImplicit runs in place
	bool metric_closed = false;              // kernels evaluate the 3-D metric in registers
	double * d_eta = nullptr;
	// Rayleigh friction (tmx_set_patch_rayleigh): strength [2L+1][NS] (levels, then interfaces) and reference
	// state [4L+1][NS] (U, V, rho*theta on levels, W on interfaces)
	std::vector<double> h_ray_nu, h_ray_ref;
	bool rayleigh = false;
	double * d_ray_nu = nullptr, * d_ray_ref = nullptr;
	double h_dx[16], h_stiff[16];

	// device
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t xstream = nullptr;           // exchange stream (multi-rank): the wire overlaps the DSS of rank-local groups
	hipEvent_t ev_pack = nullptr, ev_recv = nullptr;
	double * d_state = nullptr;
	double * d_g2d = nullptr, * d_g3n = nullptr, * d_g3e = nullptr, * d_ops = nullptr;
	double * d_scratch = nullptr;            // [L][NS] (u x zeta)_xi hand-off between the two H kernels
	// DSS
	int ngroups = 0, nxgroups = 0, nghost = 0;
	int ngroups_local = 0;                   // groups [0, ngroups_local) have no member on another rank
	int * d_hvblocks = nullptr; int n_hvblocks = 0;   // element blocks of k_hypervis_block ({elem_base, nea, neb, a0, b0} each)
	int ngroups_inpatch = 0;                 // DSS groups [0, ngroups_inpatch) have all their copies in ONE patch (k_hypervis_block averages those itself)
	int hvis_block = 0;                      // TMX_HVIS_BLOCK=1: hyperviscosity passes fused with the in-patch DSS (k_hypervis_block; bit-identical, measured slower)
	int * d_colref = nullptr;                // [NS][4] the other copies of a stored column + packed group info (k_hypervis<PULL>: the DSS pulled into a kernel's loads)
	int hvis_pull = 0;                       // TMX_HVIS_PULL=1: the second hyperviscosity pass averages the first pass's Laplacians while loading them
	                                         // instead of a DSS pass in between (bit-identical; measured SLOWER: +99 us for the pass against 62 us saved)
	int * d_grp_cols = nullptr;              // [ngroups][4] column index (>= NS: ghost NS+idx), -1 = none
	int * d_grp_n = nullptr;                 // [ngroups] member count (2..4); high bit 8 = cross-panel, index in bits 16+
	int * d_grp_x = nullptr;                 // [ngroups] index into the cross-panel matrices or -1
	int * d_grp_type = nullptr;              // [ngroups] 2 bits per member: pairing type in the reference's averaging order
	double * d_xmat = nullptr;               // [nxgroups][4][2][4]: per member A (to frame) and B (from frame)
	double * d_ghost = nullptr;              // [nslab][nghost_pad] received remote copies
	double * d_sendbuf = nullptr;
	int * d_send_cols = nullptr, * d_send_base = nullptr, * d_send_stride = nullptr;
	int * d_ghost_base = nullptr, * d_ghost_stride = nullptr;
	int nsend = 0, nghost_pad = 0;
	std::vector<int> send_rank_off, recv_rank_off;   // per-rank offsets (size n_ranks+1)
	// peer-to-peer halo transport (tmx_halo_p2p_*): this rank's ghost columns live twice (alternating per exchange) in a
	// fine-grained block that the neighbour ranks map through HIP IPC and write into directly; [2][n_ranks] arrival counters lead it
	bool p2p = false, p2p_connected = false;
	void * p2p_block = nullptr; size_t p2p_block_bytes = 0;
	std::vector<void *> p2p_peer;             // mapped blocks of the other ranks (nullptr = not a neighbour)
	double ** d_p2p_dst = nullptr;            // [2][n_ranks] where my message to rank r starts in r's ghost buffer b
	unsigned long long ** d_p2p_flag = nullptr;   // [2][n_ranks] my arrival counter in rank r's block
	int * d_send_peer = nullptr, * d_send_within = nullptr;    // [nsend] destination rank of send entry i, its index within the message
	int * d_p2p_peers = nullptr; int p2p_npeers = 0;     // neighbour ranks
	unsigned long long p2p_seq = 0;
	double * d_ghost_own = nullptr;
	// implicit solve
	int nunique = 0, NUS = 0;
	int * d_ucol = nullptr;                  // [nunique]
	int * d_udep = nullptr;                  // [nunique][3] dependents (-1 = none)
	double * d_ab = nullptr;                 // [3(L+1)*9][NUS] band rows, overwritten by U rows
	double * d_rhs = nullptr;                // [3(L+1)][NUS]
	int * d_flag = nullptr;                  // singular-pivot flag
	bool track_surface = false; int patches_with_physics = 0;     // surface slots carried through the stage algebra (tmx_step.hip: surface_copy)
	// Kessler microphysics (tmx_physics_kessler): level heights [L][NS], accumulated precipitation [NS], column work arrays [8][L][NS]
	std::vector<double> h_zlev; bool zlev_dirty = false;
	double * d_zlev = nullptr, * d_prect = nullptr, * d_kes = nullptr;
	size_t hbm_bytes = 0;

	// comm
	struct LoopbackGroup * lb = nullptr;     // test hook: in-process group of rank engines sharing one barrier
	void * comm = nullptr;                   // ncclComm_t
	void * nccl_lib = nullptr;

	// profiling
	bool sw = false;                         // shallow-water equation set (no vertical dynamics)
	int vi_mode = 0;                         // 0 = fused column kernel, 1 = assemble + solve (two kernels)
	int vi_ring_depth = 0;                   // block rows in the LDS ring of the two-role column kernel: 0 auto (3, or 2 where that lets two workgroups onto a CU), 2, 3
	int vi_producers = 0;                    // assembly wavefronts per column group of the two-role column kernel: 0 auto (2 where every wavefront has a SIMD to itself), 1, 2
	int vi_pair = -1;                        // two-wavefront column kernel (assembly || elimination): -1 auto (small grids), 0 off, 1 on (TMX_VI_PAIR)
	// boundary-first launches (ranks > 1): 64-column tiles holding a column another rank needs, and the others
	int * d_quads_early = nullptr, * d_quads_late = nullptr; int n_quads_early = 0, n_quads_late = 0, launch_tiles = 0; bool split_stage = false, split_overlapped = false;
	int launch_blocks = 0;                   // blocks of the launch in progress (KParams::bquads), node-unique layout
	std::vector<struct tmx_interp *> interps;          // output-interpolation plans created on this engine and not yet destroyed
	double * d_image = nullptr; size_t image_n = 0;     // restart image of one patch (tmx_pack_active_state), grown on demand
	double strang_offc = 0.0;                          // TimestepSchemeStrang off-centering (tmx_set_strang_off_centering)
	unsigned long long * d_pivot_stats = nullptr;      // tmx_debug_pivot_stats: {uniform pivot steps, all pivot steps} of k_vi_pair, per wavefront
	int use_mfma = 0;                        // option "contraction_mfma" (TMX_MFMA=1): 4x4 contractions of the fused explicit kernel on the matrix unit (not bit-exact)
	int vi_group = -1, vi_group_max = 6400;     // lane-group column kernel: -1 auto (nunique <= vi_group_max; column solve per step, group / pair kernel: 0.171 / 0.281 ms at 2 116 columns, 0.237 / 0.279 at 6 075, 0.349 / 0.279 at 8 100), 0 off, 1 on (TMX_VI_GROUP, TMX_VI_GROUP_MAX)
	int vi_kernel_launched = -1;             // the column-solve kernel of the last launch: 0 one wavefront per column group, 1 pair (assembly || elimination), 2 lane group; -1 none yet
	int vi_cpw = 64, vi_stream_cols = 0;     // columns per wavefront of k_vi_pair (TMX_VI_CPW; auto: fill every SIMD evenly), stream columns allocated
	int p2p_timeout_s = 600;                 // peer-to-peer halo: how long a neighbour's message may take (TMX_P2P_TIMEOUT_S; 0 = for ever)
	const int * stage_quads = nullptr;       // tile list of the boundary-first stage in progress (make_params -> KParams.quads)
	// Instance map of the stepper programs (run_program): logical data instance -> slot of d_state.  A CopyData(a -> b) whose
	// copy stays bit-identical to its source until b is next overwritten (fully explicit vertical mode, shallow water: the
	// V.StepImplicit that follows is a stub) is not performed; b reads a's slot instead.  Identity outside tmx_step: every
	// other entry point settles the map first (settle_instance_map).
	std::vector<int> imap;
	int n_shared = 0;                        // instances that currently read another instance's slot
	// the same for the U,V slabs alone (implicit vertical dynamics: the column solve that follows the CopyData rewrites rho*theta,
	// W, rho of the copy, its U,V stay the source's): slot holding the U,V slabs of a logical instance
	std::vector<int> uvmap;
	int n_uvshared = 0;
	bool in_program = false;
	int share_copies = 1;                    // TMX_SHARE_COPIES=0: perform every CopyData
	bool dry_run = false; int dry_copies = 0, dry_shared = 0;     // tmx_debug_program_copies
	// symbolic contents of the dry run: what the reference's data instances hold (ref) and what the slots of d_state hold (slot),
	// as value ids, separately for the U,V slabs ([0]) and the rest ([1]); dry_mismatch counts reads through the instance maps
	// that would not have found the reference's value
	std::vector<unsigned long long> dry_ref[2], dry_slot[2];
	int dry_mismatch = 0;
	int xcd_vertical = 1;                    // TMX_XCD_VERTICAL=0: plain 2-D order of the vertical stencil kernels' workgroups
	int vi_split_back = 0, vi_back_sub = 1, vi_stagger = 0;   // experiments (TMX_VI_SPLIT_BACK, TMX_VI_BACK_SUB, TMX_VI_STAGGER_NS / 10): back substitution as its own launch
	int vi_pair_wg = 0;                      // pairs per workgroup of k_vi_pair: 0 auto (1 up to 256 column groups, else 2), TMX_VI_PAIR_WG
	int vi_carry = 1;                        // carry shared quantities between block rows (A/B switch TMX_VI_CARRY)
	bool fully_explicit = false, udiff = false;   // cfg.fully_explicit / cfg.uniform_diffusion
	std::vector<double> h_ref;               // reference state in slab layout until tmx_finalize
	double * d_ref = nullptr;                // [nslab][NS] reference state + reference tracers (uniform diffusion)
	double * d_refd = nullptr;               // [4 L][NS] the reference's own diffusion increments of U, V (tmx_k_hwalk.hip, k_ref_diffusion_terms), formed on first use
	bool refd_valid = false;                 // ... and again after tmx_set_patch_reference_state
	double * h_stage = nullptr; size_t h_stage_n = 0;   // pinned host staging buffer of tmx_upload_state / tmx_download_state
	bool prof = false;
	int use_graph = 0;                       // option "step_graph" (TMX_GRAPH=1): replay single-rank steps from a captured hipGraph (measured: no gain, see tmx_step)
	std::vector<StepGraph> graphs;
	// switches that used to be read from the environment at their point of use (tmx_set_option / tmx_options_from_environment)
	int opt_no_split = 0, opt_metric_stored = 0, opt_tracer_lincomb_pass = 0, opt_udv_separate = 0, opt_vx_fused = 0, opt_skip_exchange = 0,
	    opt_no_exchange_overlap = 0, opt_kessler_column = 0, opt_vi_sparse = 1, opt_vt_column = 0, opt_vt_explicit_v1 = 0, opt_vt_walk = -1000, opt_vite_walk = -1000, opt_vx_walk = -1000, opt_vt_lanes = 16,
	    opt_vt_lw8 = -1, opt_vt_nr = 0, opt_vt_rows = 1, opt_vi_cpw = 0;
	int lu_fma = 1;                          // option "lu_fma": 1 = band-LU updates as fused multiply-adds (OpenBLAS, MKL's FMA paths), 0 = multiply + subtract (tmx_lu_select.hip)
	int opt_h_walk_udiff = 2;                // the walk applies the horizontal uniform diffusion itself (element-major uniform-diffusion configurations): 1; 2 (default) V.StepExplicit's U,V part too; 0 = k_uniform_diffusion / k_v_explicit as passes of their own
	int opt_hv_walk = 1;                     // hyperviscosity pass on the node-unique layout as a walk (k_hv_walk): 1 (default), 0 = the level-parallel k_hypervis
	int opt_h_walk = -1000;                  // explicit stage on the node-unique layout: -1000 auto (column-segment walk, segments from the grid size), -n = n segments, 0 = the level-parallel kernel
	std::vector<int> opt_default;            // default of every option, in g_options order (tmx_record_option_defaults)
	std::vector<std::string> env_applied;    // "NAME=value" of every environment variable tmx_options_from_environment turned into an option
	UniqueLayout u;
	bool vi_unique = false;                  // transient: the column solve runs on the unique slabs (tmxk_vi_fused)
	ProfSlot prof_slots[TMX_K_COUNT];
	std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
};

void tmx_set_error(const char * fmt, ...);
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
	tmx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); return TMX_ERR_DEVICE; } } while (0)
#define REQUIRE(cond, code, ...) do { if (!(cond)) { tmx_set_error(__VA_ARGS__); return (code); } } while (0)

// node-unique layout (tmx_unique.hip)
struct UniquePlanInput {      // what tmx_finalize hands over: the DSS groups in the engine's order and the send list, in D columns
	const std::vector<int> * grp_cols, * grp_n, * grp_x, * grp_type, * send_cols;
};
struct UniqueTables {         // the host-side tables of the layout (tmxu_tables); thread space T = [tile][64], unique space U
	std::vector<int> t_dcol, t_ucol, t_sdst, t_sred, tinfo, ucol_of_dcol, u_rep, slot_ucol, gsrc, gdst, gn, gx, gt, send_slots, quads_early, quads_late;
	std::vector<int> blk_first, blk_nt, blk_info, b_sdst, b_sred, b_gsrc, b_gdst, b_gn, b_gx, b_gt, blks_early, blks_late;      // block level
};
int tmxu_tables(tmx_engine * e, const UniquePlanInput & in, UniqueTables & T);      // host only (works on plan-only engines)
int tmxu_build(tmx_engine * e, const UniquePlanInput & in, size_t * bytes);
void tmxu_free(tmx_engine * e);
KParams tmxu_params(const tmx_engine * e, const KParams & base);          // thread space T, state in U slabs
// block kernels (archived experiment, round 6): compiled into the experiments flavour of the library only
KParams tmxu_params_columns(const tmx_engine * e, const KParams & base);  // one thread per unique column (column solve)
int tmxu_to_u(tmx_engine * e, int slot, bool check, bool * consistent);   // D form -> U form of one slot (check: verify that the copies agree)
int tmxu_check(tmx_engine * e, int slot, bool * consistent);              // do the copies of every node of an element-major slot agree?
int tmxu_to_d(tmx_engine * e, int slot, bool keep_u = false);                                  // U form -> D form
void tmxuk_slots_from_u(tmx_engine * e, const double * xu);               // partial slots <- the values of their nodes (DSS of an instance that is already consistent inside patches)
void tmxuk_dss(tmx_engine * e, double * xu, int g0, int g1, bool blocks = false);   // finish the seam nodes left in the partial slots (blocks: the list behind block kernels)
void tmxuk_pack(tmx_engine * e);                                          // partial slots other ranks need -> send buffer
void tmxuk_pack_p2p(tmx_engine * e, int buf);
int tmxu_physics_inputs(tmx_engine * e, bool * ok);                       // refresh the per-unique-column physics inputs; *ok: the copies of every node agree
void tmxuk_gather_rows(tmx_engine * e, int nrows, const double * xd, double * xu);   // xu[r][u] = xd[r][representative column of u]

// Archived experiments and cross-check kernels (slower or not bit-exact paths kept for A/B runs and tests, and the timing aid that
// leaves the wire out) are compiled into the experiments flavour of the library only (`make exp`: libtempest_mi355x_exp.so,
// -DTMX_EXPERIMENTS); the options that select them are refused by the production library (tmx_set_option).
#ifdef TMX_EXPERIMENTS
#define TMX_EXP 1
#else
#define TMX_EXP 0
#endif
// which: 0 any, 1 the explicit stage's walk, 2 the hyperviscosity pass (option values 2 / 3 switch only that kernel to its block form: A/B and debugging)
static inline bool tmxu_blocks_on(const tmx_engine * e, int which = 0) {
	if (!TMX_EXP || !e->u.built) return false;
	const int o = e->u.block_option;
	if (o == 2) return which != 2;
	if (o == 3) return which != 1;
	return o == 1 || (o < 0 && e->u.tile_shape == 4);
}

// kernel launchers (tmx_k_*.hip)
// the prefix of a later stage's combination that an explicit stage also forms and stores (emit_value in tmx_device.h)
struct EmitTerms { double * xp; double coef[12]; int first; };
void tmxk_h_explicit(tmx_engine * e, const KParams & p, const double * xin, const double * xbase, double * xup, double dt, int fuse_v,
	int nterms = 0, const double * const * src = nullptr, const double * coef = nullptr, int premul = 0,
	const double * xin_uv = nullptr, const double * xbase_uv = nullptr, const double * const * src_uv = nullptr, const EmitTerms * emit = nullptr,
	unsigned dmask = 0u);      // dmask: LinTerms::dmask (node-unique kernels: which sources are element-major instances read per copy)
// the same stage as a column-segment walk (tmx_k_hwalk.hip): node-unique layout, vertical part fused in
bool tmxk_h_walk_ok(const tmx_engine * e, const KParams & p, int fuse_v, int nterms, bool emit);
void tmxk_h_walk_prepare(tmx_engine * e, const KParams & p);
int tmxk_h_walk_segments(const tmx_engine * e, const KParams & p, int ntiles, int nterms, int waves_per_simd = 0);
void tmxk_h_walk(tmx_engine * e, const KParams & p, const double * xin, const double * xbase, double * xup, double dt, int fuse_v,
	int nterms, const double * const * src, const double * coef, int premul, const double * xin_uv, const double * xbase_uv, const double * const * src_uv, unsigned dmask,
	const EmitTerms * emit = nullptr);
bool tmxk_h_walk_fuses_udiff(const tmx_engine * e, const KParams & p, int fuse_v, int nterms);      // the walk applies the horizontal uniform diffusion itself
bool tmxk_hv_walk_ok(const tmx_engine * e, const KParams & p);      // the hyperviscosity pass as a walk (same file)
void tmxk_hv_walk(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout, double dt, double nu_s, double nu_d, double nu_v, int scale);
void tmxk_h_walk_timing_read(unsigned long long * out);      // diagnostic builds (-DTMX_H_TIMING): [8 instantiations][7 phases + count], cleared by the read
void tmxk_h_timing_read(unsigned long long * out);      // diagnostic builds (-DTMX_H_TIMING): [16 instantiations][7 phases + count], cleared by the read
void tmxk_sw_explicit(tmx_engine * e, const KParams & p, const double * xin, const double * xbase, double * xup, double dt);
void tmxk_v_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_udiff_uv = false);
void tmxk_vi_assemble(tmx_engine * e, const KParams & p, const double * xin, double dt);
void tmxk_vi_solve(tmx_engine * e, const KParams & p, const double * xin, double * xup);
void tmxk_vi_fused(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt);
void tmxk_vi_terms_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt, bool with_uv = false);
void tmxk_dss(tmx_engine * e, const KParams & p, double * x, int g0, int g1);
void tmxk_hypervis(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout,
	double dt, double nu_s, double nu_d, double nu_v, int scale_locally, bool pull_dss = false);
void tmxk_lincomb(tmx_engine * e, size_t n, double * dst, int nterms, const double * const * src, const double * coef, int dst_is_term0);
void tmxk_pack(tmx_engine * e, const KParams & p, const double * x);
void tmxk_pack_p2p(tmx_engine * e, const KParams & p, const double * x, int buf);
void tmxk_p2p_signal_wait(tmx_engine * e, hipStream_t s, int buf, unsigned long long seq);
void tmxk_rayleigh(tmx_engine * e, const KParams & p, double * x, double dt, const double * xs = nullptr, size_t NSS = 0, const int * ucd = nullptr);
void tmxk_held_suarez(tmx_engine * e, const KParams & p, double * x, const double * surf, double dt, const double * xs = nullptr, size_t NSS = 0, const int * ucd = nullptr);
void tmxk_kessler(tmx_engine * e, const KParams & p, double * x, double dt);
// tracers (rows H8 / V8)
void tmxk_h_tracers(tmx_engine * e, const KParams & p, const double * xin, const double * xbase, double * xup, double dt,
	int nterms = 0, const double * const * src = nullptr, const double * coef = nullptr, int premul = 0, const double * xin_uv = nullptr);
void tmxk_hypervis_block(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout,
	double dt, double nu_s, double nu_d, double nu_v, int scale_locally);
void tmxk_hypervis_tracers(tmx_engine * e, const KParams & p, const double * xsrc, const double * xbase, double * xout, double dt, double nu, int filter, int scale_locally);
int  tmxk_vi_tracers(tmx_engine * e, const KParams & p, const double * xin, const double * w0, const double * xbase, double * xup, double dt);
void tmxk_v_filter_tracers(tmx_engine * e, const KParams & p, double * x);
struct InterpArgs {
	int npts, nreta;
	const int * col0;          // [npts] device column of the element's first node, -1: not on this rank
	const double * ca, * cb;   // [npts][4] Lagrange coefficients
	const double * rll;        // [npts][4] (lon; lat) x (alpha, beta), or nullptr
	const double * opn, * ope; // [nreta][L], [nreta][L+1] vertical interpolation operators
};
void tmxk_interp_state(tmx_engine * e, const KParams & p, const InterpArgs & a, const double * x, const double * xref,
	int only_at, int primitive, double earth_radius, double * out);
void tmxk_interp_tracers(tmx_engine * e, const KParams & p, const InterpArgs & a, const double * x, double * out);
void tmxk_active_state_image(tmx_engine * e, const KParams & p, bool pack, int c0, int ncp, int na, int nb, int neb, double * x, double * surf, double * img);
int tmxk_vi_tracers_all(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt);
int tmxk_vi_tracers_explicit(tmx_engine * e, const KParams & p, const double * xin, double * xup, double dt);
void tmxk_uniform_diffusion(tmx_engine * e, const KParams & p, const double * xsrc, const double * xref, double * xout, double dt, double ks, double kv);
void tmxk_v_uniform_diffusion_uv(tmx_engine * e, const KParams & p, const double * xin, const double * xref, double * xup, double dt, double cf);
