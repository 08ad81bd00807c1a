"""ctypes binding of the C ABI (include/tempest_mi355x.h) and thin host-side mirrors of the
reference plugin classes for this path (same method names, argument meaning and error behaviour
as src/atm/HorizontalDynamics.h:54-175, VerticalDynamics.h:46-129, TimestepScheme.h:54-121,
Grid.cpp:1585-1621), so parity tests read like calls on the reference objects.

There is NO CPU path here: every compute call goes to libtempest_mi355x.so (hand-written HIP,
gfx950).  If the library is missing, import fails loudly; if no GPU is present, creating an
engine raises :class:`TempestError`.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TMX_LIBRARY: developer override to A/B an alternative build of the same ABI (e.g. the -DTMX_VI_TIMING variant)
LIB_PATH = os.environ.get("TMX_LIBRARY") or os.path.join(_HERE, "libtempest_mi355x.so")

TMX_ABI_VERSION = 5
OP_NAMES = ["interp_node_to_redge", "interp_redge_to_node", "diff_node_to_node", "diff_node_to_redge",
            "diff_redge_to_node", "diff_redge_to_redge", "diffdiff_node_to_node", "diffdiff_redge_to_redge",
            "penalty_left", "penalty_right"]
GEOM_FIELDS = ["jacobian2d", "contra_metric_2d_a", "contra_metric_2d_b", "jacobian", "jacobian_redge",
               "contra_metric_a", "contra_metric_b", "contra_metric_xi",
               "contra_metric_a_redge", "contra_metric_b_redge", "contra_metric_xi_redge",
               "deriv_r_node", "deriv_r_redge", "coriolis_f", "topography", "element_area_node"]
KERNEL_NAMES = ["h_explicit", "v_explicit", "vi_assemble", "vi_solve", "dss", "hypervis", "lincomb", "exchange"]

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)


class TempestError(RuntimeError):
    """Mirrors the reference's Exception (src/base/Exception.h:25-49): raised on a non-zero return."""

    def __init__(self, code, msg):
        super().__init__("tempest_mi355x error %d: %s" % (code, msg))
        self.code = code


class TmxConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("horizontal_order", C.c_int), ("vertical_order", C.c_int),
                ("levels", C.c_int), ("n_patches", C.c_int), ("n_instances", C.c_int), ("n_tracers", C.c_int),
                ("hypervis_order", C.c_int),
                ("element_delta_a", C.c_double), ("reference_length", C.c_double),
                ("grav", C.c_double), ("Rd", C.c_double), ("cp", C.c_double), ("p0", C.c_double),
                ("nu_scalar", C.c_double), ("nu_div", C.c_double), ("nu_vort", C.c_double),
                ("rank", C.c_int), ("n_ranks", C.c_int), ("device", C.c_int), ("equation_set", C.c_int),
                ("fully_explicit", C.c_int), ("uniform_diffusion", C.c_int),
                ("uniform_diffusion_scalar", C.c_double), ("uniform_diffusion_vector", C.c_double), ("ztop", C.c_double)]


class TmxInterpPoints(C.Structure):
    _fields_ = [("n_points", C.c_int), ("n_reta", C.c_int), ("patch", PI), ("node_a", PI), ("node_b", PI),
                ("coeff_a", PD), ("coeff_b", PD), ("rll_from_abp", PD), ("op_levels", PD), ("op_interfaces", PD)]


class TmxPatchGeometry(C.Structure):
    _fields_ = [(n, PD) for n in GEOM_FIELDS]


_libs = {}


def load_library(flavour=None):
    """Load the HIP library.  Fails loudly when it has not been built (no fallback).

    ``flavour``: None = the production library; "experiments" = libtempest_mi355x_exp.so (`make -C tempestmodel_amd/csrc exp`),
    the same library plus the archived experiments and cross-check kernels whose options the production library refuses
    (TMX_FLAVOUR=experiments in the environment selects it for test plumbing that cannot pass the argument)."""
    flavour = flavour or os.environ.get("TMX_FLAVOUR") or None
    if flavour not in (None, "production", "experiments"):
        raise ValueError("unknown library flavour %r" % (flavour,))
    path = os.environ.get("TMX_LIBRARY") or LIB_PATH      # TMX_LIBRARY: an experimental build of the same library (tools/ablate.sh)
    # (TMX_LU_FMA=0 -- band-LU updates as multiply + subtract, for a reference linked to a BLAS without fused multiply-adds -- used to
    # select a second library; it is the option "lu_fma" of the one library now, applied with the other TMX_* variables)
    if flavour == "experiments" and not os.environ.get("TMX_LIBRARY"):
        path = LIB_PATH.replace(".so", "_exp.so")
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("libtempest_mi355x.so is missing at %s -- build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950); "
                          "this package has no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.tmx_last_error.restype = C.c_char_p
    lib.tmx_info.restype = C.c_longlong
    lib.tmx_destroy.restype = None
    for name in ("tmx_h_step_explicit", "tmx_v_step_explicit", "tmx_v_step_implicit"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.tmx_h_step_after_subcycle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.tmx_step_ars343.argtypes = [C.c_void_p, C.c_double]
    lib.tmx_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.tmx_v_step_implicit_terms_explicitly.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.tmx_apply_dss.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_copy_data.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.tmx_zero_data.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_linear_combine_data.argtypes = [C.c_void_p, PD, C.c_int, C.c_int]
    lib.tmx_sync.argtypes = [C.c_void_p]
    lib.tmx_destroy.argtypes = [C.c_void_p]
    lib.tmx_finalize.argtypes = [C.c_void_p]
    lib.tmx_info.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_plan_get.argtypes = [C.c_void_p, C.c_int, PI, C.c_int]
    lib.tmx_plan_get_matrices.argtypes = [C.c_void_p, PD, C.c_int]
    lib.tmx_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_profile_reset.argtypes = [C.c_void_p]
    lib.tmx_profile_get.argtypes = [C.c_void_p, C.c_int, PD, C.POINTER(C.c_longlong)]
    lib.tmx_upload_state.argtypes = [C.c_void_p, C.c_int, C.c_int, PD, PD]
    lib.tmx_download_state.argtypes = [C.c_void_p, C.c_int, C.c_int, PD, PD]
    lib.tmx_set_patch_metric_factors.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, C.c_double, PD, PD]
    lib.tmx_set_patch_rayleigh.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, PD]
    lib.tmx_set_patch_reference_state.argtypes = [C.c_void_p, C.c_int, PD, PD, PD]
    lib.tmx_v_filter_negative_tracers.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_interp_create.argtypes = [C.c_void_p, C.POINTER(TmxInterpPoints), C.POINTER(C.c_void_p)]
    lib.tmx_interp_destroy.argtypes = [C.c_void_p]
    lib.tmx_interp_destroy.restype = None
    lib.tmx_interp_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, PD]
    lib.tmx_interp_tracers.argtypes = [C.c_void_p, C.c_void_p, C.c_int, PD]
    lib.tmx_h_substep_after_subcycle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
    lib.tmx_h_substep_after_subcycle_count.argtypes = [C.c_void_p]
    lib.tmx_set_patch_physics_inputs.argtypes = [C.c_void_p, C.c_int, PD, PD]
    lib.tmx_physics_held_suarez.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.tmx_set_patch_level_heights.argtypes = [C.c_void_p, C.c_int, PD]
    lib.tmx_active_state_bytes.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_active_state_bytes.restype = C.c_longlong
    lib.tmx_pack_active_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.tmx_unpack_active_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.tmx_set_strang_off_centering.argtypes = [C.c_void_p, C.c_double]
    lib.tmx_debug_pivot_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    lib.tmx_set_patch_element_spacing.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    lib.tmx_physics_kessler.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.tmx_download_precipitation.argtypes = [C.c_void_p, C.c_int, PD, C.c_int]
    lib.tmx_upload_tracers.argtypes = [C.c_void_p, C.c_int, C.c_int, PD]
    lib.tmx_download_tracers.argtypes = [C.c_void_p, C.c_int, C.c_int, PD]
    lib.tmx_comm_init.argtypes = [C.c_void_p, C.c_char_p]
    lib.tmx_halo_p2p_blob_bytes.argtypes = [C.c_void_p]
    lib.tmx_halo_p2p_export.argtypes = [C.c_void_p, C.c_char_p]
    lib.tmx_halo_p2p_connect.argtypes = [C.c_void_p, C.c_char_p]
    lib.tmx_halo_p2p_enable.argtypes = [C.c_void_p, C.c_int]
    lib.tmx_halo_p2p_reset.argtypes = [C.c_void_p]
    lib.tmx_comm_unique_id.argtypes = [C.c_char_p]
    lib.tmx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.tmx_get_option.argtypes = [C.c_void_p, C.c_char_p, PD]
    lib.tmx_options_from_environment.argtypes = [C.c_void_p]
    lib.tmx_options_report.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    _libs[path] = lib
    return lib


def _pd(a):
    return a.ctypes.data_as(PD)


def _pi(a):
    return a.ctypes.data_as(PI)


def default_owner(npatch, n_ranks):
    """Panel-major block map of patches to ranks (results do not depend on the map)."""
    if not 1 <= n_ranks <= npatch:
        raise ValueError("%d ranks for %d patches: every rank must own at least one patch (use more patches per panel)"
                         % (n_ranks, npatch))
    return [p * n_ranks // npatch for p in range(npatch)]      # remainder spread over the ranks, none left empty


class Engine:
    """One engine per process / GPU, built from a :class:`tempestmodel_amd.cubed_sphere.CubedSphereGrid`
    (or anything exposing the same reference-layout arrays)."""

    def __init__(self, grid, n_instances=7, nu=(1.0e15, 1.0e15, 1.0e15), hypervis_order=4,
                 rank=0, n_ranks=1, owner=None, device=-1, fully_explicit=False, uniform_diffusion=None, options=None, flavour=None):
        """``fully_explicit``: VerticalDynamicsFEM's --explicitvertical mode; ``uniform_diffusion``: (K_scalar, K_vector)
        of Grid::HasUniformDiffusion (reference state and tracers are taken from ``grid.patches[*].geom``); ``options``:
        {name: value} for tmx_set_option before tmx_finalize (e.g. {"unique_layout": 1}); ``flavour``: see load_library."""
        self.options = dict(options or {})
        self.lib = load_library(flavour)
        self.grid = grid
        self.L = grid.L
        self.npatch = len(grid.patches)
        self.rank, self.n_ranks = rank, n_ranks
        self.owner = list(owner) if owner is not None else default_owner(self.npatch, n_ranks)
        cfg = TmxConfig()
        cfg.abi_version = TMX_ABI_VERSION
        cfg.horizontal_order, cfg.vertical_order = grid.np, 1
        self.ntracers = int(getattr(grid, "ntracers", 0))
        cfg.levels, cfg.n_patches, cfg.n_instances, cfg.n_tracers = grid.L, self.npatch, n_instances, self.ntracers
        cfg.hypervis_order = hypervis_order
        cfg.fully_explicit = 1 if fully_explicit else 0
        self.uniform_diffusion = uniform_diffusion
        cfg.ztop = float(getattr(grid, "ztop", 0.0))
        if uniform_diffusion is not None:
            cfg.uniform_diffusion = 1
            cfg.uniform_diffusion_scalar, cfg.uniform_diffusion_vector = uniform_diffusion
        cfg.element_delta_a, cfg.reference_length = grid.delta_a, grid.reference_length
        ph = grid.phys
        cfg.grav, cfg.Rd, cfg.cp, cfg.p0 = ph.g, ph.Rd, ph.cp, ph.p0
        cfg.nu_scalar, cfg.nu_div, cfg.nu_vort = nu
        cfg.rank, cfg.n_ranks, cfg.device = rank, n_ranks, device
        self.shallow_water = bool(getattr(grid, "shallow_water", False))
        cfg.equation_set = 1 if self.shallow_water else 0
        self.cfg = cfg
        self.plan_only = (device == -2)
        h = C.c_void_p()
        self._chk(self.lib.tmx_create(C.byref(cfg), C.byref(h)))
        self.h = h
        # test / bench plumbing keeps the TMX_* variables: turned into options here, explicitly (the library itself reads none)
        if self.lib.tmx_options_from_environment(self.h) < 0:      # a variable this flavour of the library refuses
            msg = self.lib.tmx_last_error().decode()
            self.lib.tmx_destroy(self.h)
            self.h = None
            raise TempestError(-1, msg)
        try:
            self._setup()
        except Exception:
            self.close()
            raise

    # -- plumbing ------------------------------------------------------------------------------
    def _chk(self, code):
        if code != 0:
            raise TempestError(code, self.lib.tmx_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.tmx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _setup(self):
        g = self.grid
        keep = []
        n = len(OP_NAMES)
        cs = (PD * n)(); bs = (PI * n)(); es = (PI * n)()
        nout = (C.c_int * n)(); nin = (C.c_int * n)()
        for i, nm in enumerate(OP_NAMES):
            if g.ops is None:
                break
            c, b, e = g.ops[nm]
            c = np.ascontiguousarray(c, dtype=np.float64)
            b = np.ascontiguousarray(b, dtype=np.int32); e = np.ascontiguousarray(e, dtype=np.int32)
            keep += [c, b, e]
            cs[i], bs[i], es[i] = _pd(c), _pi(b), _pi(e)
            nout[i], nin[i] = c.shape
        dx = np.ascontiguousarray(g.dx, dtype=np.float64)
        st = np.ascontiguousarray(g.stiff, dtype=np.float64)
        self._chk(self.lib.tmx_set_operators(self.h, _pd(dx), _pd(st), cs, bs, es, nout, nin))
        for P in g.patches:
            nbp = (C.c_int * 8)(*[int(v) for v in P.neighbor_panels])
            self._chk(self.lib.tmx_define_patch(self.h, P.index, P.panel, (P.ga1 - P.ga0) // g.np,
                                                (P.gb1 - P.gb0) // g.np, self.owner[P.index], nbp))
            if getattr(P, "delta_a", 0.0):      # the reference's own per-patch element spacing (fixtures that carry it)
                self._chk(self.lib.tmx_set_patch_element_spacing(self.h, P.index, float(P.delta_a), float(getattr(P, "delta_b", P.delta_a))))
        for P in g.patches:
            hh = P.halo
            arrs = [np.ascontiguousarray(hh[k], dtype=np.int32) for k in
                    ("i", "j", "src_patch", "src_i", "src_j", "src_panel")]
            tr = np.ascontiguousarray(hh["trans"], dtype=np.float64)
            self._chk(self.lib.tmx_set_patch_halo(self.h, P.index, len(arrs[0]), *[_pi(a) for a in arrs], _pd(tr)))
        if not self.plan_only:
            for P in g.patches:
                if self.owner[P.index] != self.rank:
                    continue
                pg = TmxPatchGeometry()
                for nm in GEOM_FIELDS:
                    a = np.ascontiguousarray(P.geom[nm], dtype=np.float64)
                    keep.append(a)
                    setattr(pg, nm, _pd(a))
                self._chk(self.lib.tmx_set_patch_geometry(self.h, P.index, C.byref(pg)))
                # Rayleigh friction (Grid::HasRayleighFriction): strength and reference state
                if not self.shallow_water and getattr(g, "has_rayleigh_friction", False) and "rayleigh_node" in P.geom:
                    fr = [np.ascontiguousarray(P.geom[nm], dtype=np.float64) for nm in
                          ("rayleigh_node", "rayleigh_redge", "ref_node", "ref_redge")]
                    keep += fr
                    self._chk(self.lib.tmx_set_patch_rayleigh(self.h, P.index, *[_pd(a) for a in fr]))
                # closed form of the 3-D metric, where the grid knows its factors (reference:
                # GridPatchCSGLL::m_dXNode/m_dYNode, GetTopographyDeriv, Grid::GetREtaLevels/Interfaces)
                if (not self.shallow_water and getattr(P, "X", None) is not None
                        and "topography_deriv" in P.geom and getattr(g, "reta_levels", None) is not None):
                    fx = [np.ascontiguousarray(a, dtype=np.float64) for a in
                          (P.X, P.Y, P.geom["topography_deriv"], g.reta_levels, g.reta_interfaces)]
                    keep += fx
                    self._chk(self.lib.tmx_set_patch_metric_factors(
                        self.h, P.index, _pd(fx[0]), _pd(fx[1]), _pd(fx[2]), float(g.phys.earth_radius),
                        _pd(fx[3]), _pd(fx[4])))
        for k_, v_ in self.options.items():
            self._chk(self.lib.tmx_set_option(self.h, k_.encode(), float(v_)))
        self._chk(self.lib.tmx_finalize(self.h))
        if self.uniform_diffusion is not None and not self.plan_only:
            for P in g.patches:
                if self.owner[P.index] != self.rank:
                    continue
                rf = [np.ascontiguousarray(P.geom[nm], dtype=np.float64) for nm in ("ref_node", "ref_redge")]
                rt = np.ascontiguousarray(P.geom["ref_tracers"], dtype=np.float64) if self.ntracers else None
                self._chk(self.lib.tmx_set_patch_reference_state(self.h, P.index, _pd(rf[0]), _pd(rf[1]),
                                                                 _pd(rt) if rt is not None else None))

    @property
    def local_patches(self):
        return [p for p in range(self.npatch) if self.owner[p] == self.rank]

    # -- state (GridPatch::GetDataState) ----------------------------------------------------------
    def upload_state(self, instance, states):
        """``states[p] = (node[5][na][nb][L], redge[5][na][nb][L+1])`` for every patch (remote ones ignored)."""
        for p in self.local_patches:
            if self.shallow_water:      # reference layout of the 3-component set: node [3][na][nb][1]
                n = np.ascontiguousarray(states[p][0][[0, 1, 2]], dtype=np.float64)
                self._chk(self.lib.tmx_upload_state(self.h, p, instance, _pd(n), None))
                continue
            n = np.ascontiguousarray(states[p][0], dtype=np.float64)
            e = np.ascontiguousarray(states[p][1], dtype=np.float64)
            self._chk(self.lib.tmx_upload_state(self.h, p, instance, _pd(n), _pd(e)))

    def download_state(self, instance):
        out = [None] * self.npatch
        for p in self.local_patches:
            P = self.grid.patches[p]
            n = np.zeros((5, P.na, P.nb, self.L)); e = np.zeros((5, P.na, P.nb, self.L + 1))
            if self.shallow_water:
                n3 = np.zeros((3, P.na, P.nb, 1))
                self._chk(self.lib.tmx_download_state(self.h, p, instance, _pd(n3), None))
                n[[0, 1, 2]] = n3
            else:
                self._chk(self.lib.tmx_download_state(self.h, p, instance, _pd(n), _pd(e)))
            out[p] = (n, e)
        return out

    # -- column physics (WorkflowProcess on the device) -----------------------------------------------
    def set_physics_inputs(self, surface_pressure=None):
        """Latitude comes from the grid (GridPatch::GetLatitude).  ``surface_pressure=None``: the engine tracks the
        interface slots HeldSuarezPhysics::Perform reads (HeldSuarezPhysics.cpp:113-116) through the stage algebra, as
        the reference does; ``surface_pressure[p] = [na][nb]`` pins the value instead."""
        for p in self.local_patches:
            P = self.grid.patches[p]
            la = np.ascontiguousarray(P.lat, dtype=np.float64)
            sp = None if surface_pressure is None else np.ascontiguousarray(surface_pressure[p], dtype=np.float64)
            self._chk(self.lib.tmx_set_patch_physics_inputs(self.h, p, _pd(la), None if sp is None else _pd(sp)))

    def held_suarez(self, instance, dt):
        """HeldSuarezPhysics::Perform (src/atm/HeldSuarezPhysics.cpp:60-301)."""
        self._chk(self.lib.tmx_physics_held_suarez(self.h, instance, dt))

    def set_level_heights(self, z_levels=None):
        """``z_levels[p] = [na][nb][L]`` (GridPatch::GetZLevels); default: the grid's own."""
        for p in self.local_patches:
            z = np.ascontiguousarray(self.grid.patches[p].geom["z_levels"] if z_levels is None else z_levels[p], dtype=np.float64)
            self._chk(self.lib.tmx_set_patch_level_heights(self.h, p, _pd(z)))

    def pivot_stats(self, enable):
        """(uniform, total) pivot steps of the two-wavefront column kernel counted so far; enable/disable counting."""
        out = (C.c_ulonglong * 2)()
        self._chk(self.lib.tmx_debug_pivot_stats(self.h, 1 if enable else 0, out))
        return int(out[0]), int(out[1])

    def kessler(self, instance, dt):
        """KesslerPhysics::Perform (test/dcmip2016/KesslerPhysics.cpp:83-285 + interface/kessler.f90)."""
        self._chk(self.lib.tmx_physics_kessler(self.h, instance, dt))

    def download_precipitation(self, reset=False):
        out = {}
        for p in self.local_patches:
            P = self.grid.patches[p]
            a = np.zeros((P.na, P.nb))
            self._chk(self.lib.tmx_download_precipitation(self.h, p, _pd(a), 1 if reset else 0))
            out[p] = a
        return out

    def upload_tracers(self, instance, tracers):
        """``tracers[p] = [n_tracers][na][nb][L]`` (GridPatch::GetDataTracers) for every patch (remote ones ignored)."""
        for p in self.local_patches:
            t = np.ascontiguousarray(tracers[p], dtype=np.float64)
            self._chk(self.lib.tmx_upload_tracers(self.h, p, instance, _pd(t)))

    def download_tracers(self, instance):
        out = [None] * self.npatch
        for p in self.local_patches:
            P = self.grid.patches[p]
            t = np.zeros((self.ntracers, P.na, P.nb, self.L))
            self._chk(self.lib.tmx_download_tracers(self.h, p, instance, _pd(t)))
            out[p] = t
        return out

    # -- C ABI, one method per entry point -----------------------------------------------------------
    def copy_data(self, src, dst):
        self._chk(self.lib.tmx_copy_data(self.h, src, dst))

    def zero_data(self, ix):
        self._chk(self.lib.tmx_zero_data(self.h, ix))

    def linear_combine_data(self, coeff, dst):
        c = np.ascontiguousarray(coeff, dtype=np.float64)
        self._chk(self.lib.tmx_linear_combine_data(self.h, _pd(c), len(c), dst))

    def h_step_explicit(self, iinit, iupd, dt):
        self._chk(self.lib.tmx_h_step_explicit(self.h, iinit, iupd, dt))

    def v_step_explicit(self, iinit, iupd, dt):
        self._chk(self.lib.tmx_v_step_explicit(self.h, iinit, iupd, dt))

    def v_step_implicit(self, iinit, iupd, dt):
        self._chk(self.lib.tmx_v_step_implicit(self.h, iinit, iupd, dt))

    def apply_dss(self, ix):
        self._chk(self.lib.tmx_apply_dss(self.h, ix))

    def h_step_after_subcycle(self, iinit, iupd, iwork, dt):
        self._chk(self.lib.tmx_h_step_after_subcycle(self.h, iinit, iupd, iwork, dt))

    def step_ars343(self, dt):
        self._chk(self.lib.tmx_step_ars343(self.h, dt))

    def pack_active_state(self, patch, instance=0):
        """Bytes of GridPatch::GetDataContainerActiveState() of the patch, transposed on the device (restart write)."""
        n = self.lib.tmx_active_state_bytes(self.h, patch)
        if n < 0:
            raise TempestError("tmx_active_state_bytes: no image for this engine / patch")
        buf = np.zeros(n, dtype=np.uint8)
        self._chk(self.lib.tmx_pack_active_state(self.h, patch, instance, buf.ctypes.data_as(C.c_void_p), n))
        return buf

    def unpack_active_state(self, patch, instance, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        self._chk(self.lib.tmx_unpack_active_state(self.h, patch, instance, image.ctypes.data_as(C.c_void_p), image.size))

    def set_strang_off_centering(self, oc):
        self._chk(self.lib.tmx_set_strang_off_centering(self.h, float(oc)))

    SCHEMES = {"ars343": 0, "ars232": 1, "ars222": 2, "ars443": 3, "ark232": 4, "strang": 5,
               "strang_fe": 6, "strang_rk4": 7, "strang_ssp3": 8, "strang_ssprk53": 9}

    def step(self, scheme, dt, first=False, last=False):
        """TimestepScheme::Step of the named reference scheme (tmx_step)."""
        self._chk(self.lib.tmx_step(self.h, self.SCHEMES[scheme], int(first), int(last), dt))

    @classmethod
    def scheme_instances(cls, scheme):
        return int(load_library().tmx_scheme_instances(cls.SCHEMES[scheme]))

    def h_substep_after_subcycle(self, iinit, iupd, iwork, dt, isubstep):
        """HorizontalDynamicsFEM.SubStepAfterSubCycle; returns the instance holding the output."""
        out = C.c_int(-1)
        self._chk(self.lib.tmx_h_substep_after_subcycle(self.h, iinit, iupd, iwork, dt, isubstep, C.byref(out)))
        return out.value

    # -- output interpolation (Grid::ReduceInterpolate) ---------------------------------------------
    def interp_create(self, pts):
        """pts: dict with patch, node_a, node_b, coeff_a, coeff_b, rll_from_abp, op_levels, op_interfaces (see the header)."""
        q = TmxInterpPoints()
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt); keep.append(a); return a
        pa = arr(pts["patch"], np.int32)
        q.n_points, q.n_reta = len(pa), pts["op_levels"].shape[0]
        q.patch, q.node_a, q.node_b = _pi(pa), _pi(arr(pts["node_a"], np.int32)), _pi(arr(pts["node_b"], np.int32))
        q.coeff_a, q.coeff_b = _pd(arr(pts["coeff_a"], np.float64)), _pd(arr(pts["coeff_b"], np.float64))
        if pts.get("rll_from_abp") is not None:
            q.rll_from_abp = _pd(arr(pts["rll_from_abp"], np.float64))
        q.op_levels, q.op_interfaces = _pd(arr(pts["op_levels"], np.float64)), _pd(arr(pts["op_interfaces"], np.float64))
        h = C.c_void_p()
        self._chk(self.lib.tmx_interp_create(self.h, C.byref(q), C.byref(h)))
        return (h, q.n_points, q.n_reta)

    def interp_destroy(self, plan):
        self.lib.tmx_interp_destroy(plan[0])

    def interp_state(self, plan, instance, only_at=0, include_reference_state=True, convert_to_primitive=True,
                     earth_radius=6.37122e6):
        out = np.zeros((5, plan[2], plan[1]))
        self._chk(self.lib.tmx_interp_state(self.h, plan[0], instance, int(only_at), int(include_reference_state),
                                            int(convert_to_primitive), float(earth_radius), _pd(out)))
        return out

    def interp_tracers(self, plan, instance):
        out = np.zeros((self.ntracers, plan[2], plan[1]))
        self._chk(self.lib.tmx_interp_tracers(self.h, plan[0], instance, _pd(out)))
        return out

    def set_reference_state(self):
        """tmx_set_patch_reference_state from the grid's geometry (also done by the constructor with uniform diffusion)."""
        for P in self.grid.patches:
            if self.owner[P.index] != self.rank:
                continue
            rf = [np.ascontiguousarray(P.geom[nm], dtype=np.float64) for nm in ("ref_node", "ref_redge")]
            rt = np.ascontiguousarray(P.geom["ref_tracers"], dtype=np.float64) if self.ntracers and "ref_tracers" in P.geom else None
            if self.ntracers and rt is None:
                rt = np.zeros((self.ntracers, P.na, P.nb, self.L))
            self._chk(self.lib.tmx_set_patch_reference_state(self.h, P.index, _pd(rf[0]), _pd(rf[1]),
                                                             _pd(rt) if rt is not None else None))

    def v_filter_negative_tracers(self, instance):
        self._chk(self.lib.tmx_v_filter_negative_tracers(self.h, instance))

    def v_step_implicit_terms_explicitly(self, iinit, iupd, dt):
        self._chk(self.lib.tmx_v_step_implicit_terms_explicitly(self.h, iinit, iupd, dt))

    def sync(self):
        self._chk(self.lib.tmx_sync(self.h))

    def options_report(self):
        """{name: value} of every engine option as in use (tmx_options_report)."""
        n = self.lib.tmx_options_report(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.tmx_options_report(self.h, buf, n + 1)
        out = {}
        for line in buf.value.decode().splitlines():
            k, _, v = line.partition("=")
            out[k] = v if k == "from_environment" else int(v)
        return out

    def get_option(self, name):
        v = np.zeros(1)
        self._chk(self.lib.tmx_get_option(self.h, name.encode(), _pd(v)))
        return float(v[0])

    def info(self, what):
        return int(self.lib.tmx_info(self.h, what))

    def plan(self, what):
        n = self.lib.tmx_plan_get(self.h, what, None, 0)
        if n < 0:
            raise TempestError(-1, "tmx_plan_get failed")
        a = np.zeros(max(n, 1), dtype=np.int32)
        self.lib.tmx_plan_get(self.h, what, _pi(a), n)
        return a[:n]

    def plan_matrices(self):
        n = self.lib.tmx_plan_get_matrices(self.h, None, 0)
        a = np.zeros(max(n, 1), dtype=np.float64)
        self.lib.tmx_plan_get_matrices(self.h, _pd(a), n)
        return a[:n].reshape(-1, 4, 4, 2, 2)      # [cross group][member m][partner q][2x2]

    def halo_p2p_export(self):
        """This rank's blob for the peer-to-peer halo transport (tmx_halo_p2p_export)."""
        n = self.lib.tmx_halo_p2p_blob_bytes(self.h)
        if n <= 0:
            raise RuntimeError("tmx_halo_p2p_blob_bytes failed")
        buf = C.create_string_buffer(n)
        self._chk(self.lib.tmx_halo_p2p_export(self.h, buf))
        return buf.raw

    def halo_p2p_connect(self, blobs):
        """blobs: the blobs of all ranks in rank order (all-gathered by the caller)."""
        n = self.lib.tmx_halo_p2p_blob_bytes(self.h)
        if len(blobs) != self.n_ranks or any(len(b) != n for b in blobs):
            raise ValueError("halo_p2p_connect needs one %d-byte blob per rank" % n)
        self._chk(self.lib.tmx_halo_p2p_connect(self.h, b"".join(blobs)))

    def halo_p2p_reset(self):
        """Recovery after a failed peer-to-peer exchange (tmx_halo_p2p_reset): between two barriers of all ranks."""
        self._chk(self.lib.tmx_halo_p2p_reset(self.h))

    def halo_p2p_enable(self, on):
        self._chk(self.lib.tmx_halo_p2p_enable(self.h, int(bool(on))))

    def comm_init(self, unique_id):
        self._chk(self.lib.tmx_comm_init(self.h, unique_id))

    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        code = lib.tmx_comm_unique_id(buf)
        if code != 0:
            raise TempestError(code, lib.tmx_last_error().decode())
        return buf.raw

    def comm_selftest(self):
        self._chk(self.lib.tmx_debug_comm_selftest(self.h))

    @staticmethod
    def dss_loopback(engines, instance):
        """Test hook: DSS across several single-process 'rank' engines (see tmx_debug_dss_loopback)."""
        lib = load_library()
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        code = lib.tmx_debug_dss_loopback(arr, len(engines), instance)
        if code != 0:
            raise TempestError(code, lib.tmx_last_error().decode())

    @staticmethod
    def loopback_group(engines):
        """Test hook: make ``engines`` (ranks 0..n-1, one host thread each) exchange through device copies
        (tmx_debug_loopback_group); an empty second call dissolves the group."""
        lib = load_library()
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        code = lib.tmx_debug_loopback_group(arr, len(engines))
        if code != 0:
            raise TempestError(code, lib.tmx_last_error().decode())

    @staticmethod
    def loopback_dissolve(engine):
        lib = load_library()
        arr = (C.c_void_p * 1)(engine.h)
        lib.tmx_debug_loopback_group(arr, 0)

    def profile_enable(self, on=True):
        self._chk(self.lib.tmx_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.lib.tmx_profile_reset(self.h))

    def profile(self):
        out = {}
        for i, nm in enumerate(KERNEL_NAMES):
            ms = C.c_double(); n = C.c_longlong()
            self._chk(self.lib.tmx_profile_get(self.h, i, C.byref(ms), C.byref(n)))
            out[nm] = (ms.value, n.value)
        return out


# ----------------------------------------------------------------------------------------------
# Mirrors of the reference plugin classes for this path.


class Grid:
    """Stage algebra + DSS entry points of the reference Grid (Grid.cpp:1585-1621, GridGLL.cpp:571-583)."""

    def __init__(self, engine):
        self.engine = engine

    def CopyData(self, ixSource, ixDest):
        self.engine.copy_data(ixSource, ixDest)

    def ZeroData(self, ixData):
        self.engine.zero_data(ixData)

    def LinearCombineData(self, dCoeff, ixDest):
        self.engine.linear_combine_data(dCoeff, ixDest)

    def PostProcessSubstage(self, iDataUpdate):
        self.engine.apply_dss(iDataUpdate)

    ApplyDSS = PostProcessSubstage


class HorizontalDynamicsFEM:
    """HorizontalDynamicsFEM (HorizontalDynamicsFEM.h:36-316) for the nonhydrostatic equation set."""

    def __init__(self, engine):
        self.engine = engine

    def GetHaloElements(self):
        return 1

    def StepExplicit(self, iDataInitial, iDataUpdate, time, dDeltaT):
        self.engine.h_step_explicit(iDataInitial, iDataUpdate, dDeltaT)

    def StepImplicit(self, iDataInitial, iDataUpdate, time, dDeltaT):
        pass    # no-op in the reference as well (HorizontalDynamics.h:119-126)

    def StepAfterSubCycle(self, iDataInitial, iDataUpdate, iDataWorking, time, dDeltaT):
        self.engine.h_step_after_subcycle(iDataInitial, iDataUpdate, iDataWorking, dDeltaT)


class VerticalDynamicsFEM:
    """VerticalDynamicsFEM (VerticalDynamicsFEM.h) in implicit mode."""

    def __init__(self, engine):
        self.engine = engine

    def StepExplicit(self, iDataInitial, iDataUpdate, time, dDeltaT):
        self.engine.v_step_explicit(iDataInitial, iDataUpdate, dDeltaT)

    def StepImplicit(self, iDataInitial, iDataUpdate, time, dDeltaT):
        self.engine.v_step_implicit(iDataInitial, iDataUpdate, dDeltaT)


class TimestepSchemeARS343:
    """TimestepSchemeARS343 (TimestepSchemeARS343.h:48-56, .cpp:150-236)."""

    def __init__(self, engine):
        self.engine = engine

    def GetComponentDataInstances(self):
        return 7

    def GetTracerDataInstances(self):
        return 7

    def Step(self, fFirstStep, fLastStep, time, dDeltaT):
        self.engine.step_ars343(dDeltaT)
