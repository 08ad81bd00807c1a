"""Host-side cubed-sphere GLL grid, metric terms, connectivity and test-case initialisation.

This is the *input synthesiser* for the engine: on a box without the reference it produces the
same arrays the reference's ``GridCSGLL`` / ``GridPatchCSGLL`` / ``TestCase`` objects hand to the
dynamics (geometry, column operators, halo connectivity, initial state), in the reference's own
patch layout ``[A+2][B+2][L]`` (one-node halo ring, duplicated element-edge nodes), which is the
interchange layout of the C ABI (``include/tempest_mi355x.h``).  It is init-time host code, never
on the timed path.  Every routine cites the reference lines whose behaviour it reproduces; it is
pinned against reference dumps in ``tests/test_grid_vs_reference.py``.

Reference behaviour followed (file:line under /root/reference):
  * patch layout               src/atm/GridCSGLL.cpp:92-148
  * node coordinates           src/atm/GridPatchCSGLL.cpp:181-198, src/atm/GridSpacing.cpp:180-197
  * panel orientation          src/atm/CubedSphereTrans.cpp:25-86, 200-268
  * metric terms               src/atm/GridPatchCSGLL.cpp:295-574
  * topography derivatives     src/atm/GridPatchCSGLL.cpp:255-290, src/atm/GridGLL.cpp:557-567
  * vertical coordinate        src/atm/GridGLL.cpp:456-502
  * column operators (vo=1)    src/atm/GridGLL.cpp:190-363 (stencils: SURVEY.md Appendix B)
  * GLL derivative matrices    src/atm/GridGLL.cpp:103-180
  * initial state              src/atm/GridPatchCSGLL.cpp:578-920
"""
import math
import numpy as np

# variable order of the nonhydrostatic equation set (src/atm/EquationSet.cpp:51-84)
UIX, VIX, PIX, WIX, RIX = 0, 1, 2, 3, 4

# Direction enum order used for edges (src/atm/Direction.h): Right, Top, Left, Bottom
EDGE_RIGHT, EDGE_TOP, EDGE_LEFT, EDGE_BOTTOM = 0, 1, 2, 3


class PhysicalConstants:
    """Defaults of src/atm/PhysicalConstants.h:120-135."""

    def __init__(self, **kw):
        self.earth_radius = 6.37122e6
        self.g = 9.80616
        self.omega = 7.29212e-5
        self.alpha = 0.0
        self.Rd = 287.0
        self.cp = 1004.5
        self.p0 = 1.0e5
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def cv(self):
        return self.cp - self.Rd

    @property
    def gamma(self):
        return self.cp / self.cv

    @property
    def pressure_scaling(self):
        return self.p0 * (self.Rd / self.p0) ** self.gamma

    def rhotheta_from_pressure(self, p):
        return np.exp(np.log(p / self.pressure_scaling) / self.gamma)

    def exner_from_rhotheta(self, rt):
        return self.cp * np.exp(self.Rd / (self.cp - self.Rd) * np.log(self.Rd / self.p0 * rt))


# ---------------------------------------------------------------------------------------------
# 1-D spectral-element operators


def gll_points_weights_01(npts):
    """Gauss-Lobatto points/weights on [0,1]."""
    if npts == 2:
        return np.array([0.0, 1.0]), np.array([0.5, 0.5])
    if npts == 4:
        s = 1.0 / math.sqrt(5.0)
        g = np.array([0.0, 0.5 * (1.0 - s), 0.5 * (1.0 + s), 1.0])
        w = np.array([1.0 / 12.0, 5.0 / 12.0, 5.0 / 12.0, 1.0 / 12.0])
        return g, w
    # general: roots of P'_{n-1}
    n = npts - 1
    from numpy.polynomial import legendre as Lg
    c = np.zeros(n + 1); c[n] = 1.0
    xi = np.sort(np.concatenate(([-1.0], Lg.legroots(Lg.legder(c)), [1.0])))
    Pn = Lg.legval(xi, c)
    w = 2.0 / (n * (n + 1) * Pn * Pn)
    return 0.5 * (xi + 1.0), 0.5 * w


def dx_basis_stiffness(npts):
    """``D[s][i] = phi_s'(x_i)`` and ``S[m][i] = D[m][i] w_i / w_m`` on [0,1] (GridGLL.cpp:103-180)."""
    g, w = gll_points_weights_01(npts)
    D = np.zeros((npts, npts))
    for s in range(npts):
        for i in range(npts):
            # derivative of Lagrange polynomial s at x_i
            acc = 0.0
            for m in range(npts):
                if m == s:
                    continue
                prod = 1.0 / (g[s] - g[m])
                for l in range(npts):
                    if l == s or l == m:
                        continue
                    prod *= (g[i] - g[l]) / (g[s] - g[l])
                acc += prod
            D[s, i] = acc
    S = np.zeros((npts, npts))
    for m in range(npts):
        for i in range(npts):
            S[m, i] = D[m, i] * w[i] / w[m]
    return D, S, w


def column_operators_vo1(L):
    """The ten column operators at vertical order 1, uniform levels.

    Returns ``{name: (coeff[nout][nin], begin[nout], end[nout])}``, names as in the TMXD dumps.
    """
    de = 1.0 / L
    ops = {}

    def mk(nout, nin):
        return np.zeros((nout, nin)), np.zeros(nout, dtype=np.int32), np.zeros(nout, dtype=np.int32)

    # interp node -> redge
    c, b, e = mk(L + 1, L)
    for k in range(L + 1):
        if k == 0:
            c[0, 0], c[0, 1] = 1.5, -0.5; b[k], e[k] = 0, 2
        elif k == L:
            c[L, L - 2], c[L, L - 1] = -0.5, 1.5; b[k], e[k] = L - 2, L
        else:
            c[k, k - 1] = c[k, k] = 0.5; b[k], e[k] = k - 1, k + 1
    ops["interp_node_to_redge"] = (c, b, e)
    # interp redge -> node
    c, b, e = mk(L, L + 1)
    for k in range(L):
        c[k, k] = c[k, k + 1] = 0.5; b[k], e[k] = k, k + 2
    ops["interp_redge_to_node"] = (c, b, e)
    # diff node -> node
    c, b, e = mk(L, L)
    for k in range(L):
        if k == 0:
            c[0, 0], c[0, 1] = -1.0 / de, 1.0 / de; b[k], e[k] = 0, 2
        elif k == L - 1:
            c[k, k - 1], c[k, k] = -1.0 / de, 1.0 / de; b[k], e[k] = L - 2, L
        else:
            c[k, k - 1], c[k, k + 1] = -0.5 / de, 0.5 / de; b[k], e[k] = k - 1, k + 2
    ops["diff_node_to_node"] = (c, b, e)
    # diff node -> redge (rows 0 and L identically zero)
    c, b, e = mk(L + 1, L)
    for k in range(1, L):
        # the reference's band is one entry wider to the left than the stencil (explicit zero)
        c[k, k - 1], c[k, k] = -1.0 / de, 1.0 / de; b[k], e[k] = max(k - 2, 0), k + 1
    b[0], e[0] = 0, 2
    b[L], e[L] = L - 2, L
    ops["diff_node_to_redge"] = (c, b, e)
    # diff redge -> node
    c, b, e = mk(L, L + 1)
    for k in range(L):
        c[k, k], c[k, k + 1] = -1.0 / de, 1.0 / de; b[k], e[k] = k, k + 2
    ops["diff_redge_to_node"] = (c, b, e)
    # diff redge -> redge
    c, b, e = mk(L + 1, L + 1)
    for k in range(L + 1):
        if k == 0:
            c[0, 0], c[0, 1] = -1.0 / de, 1.0 / de; b[k], e[k] = 0, 2
        elif k == L:
            c[L, L - 1], c[L, L] = -1.0 / de, 1.0 / de; b[k], e[k] = L - 1, L + 1
        else:
            c[k, k - 1], c[k, k + 1] = -0.5 / de, 0.5 / de; b[k], e[k] = k - 1, k + 2
    ops["diff_redge_to_redge"] = (c, b, e)
    # diffdiff node -> node
    c, b, e = mk(L, L)
    for k in range(L):
        if k == 0:
            c[0, 0], c[0, 1] = -1.0 / de ** 2, 1.0 / de ** 2; b[k], e[k] = 0, 2
        elif k == L - 1:
            c[k, k - 1], c[k, k] = 1.0 / de ** 2, -1.0 / de ** 2; b[k], e[k] = L - 2, L
        else:
            c[k, k - 1], c[k, k], c[k, k + 1] = 1.0 / de ** 2, -2.0 / de ** 2, 1.0 / de ** 2
            b[k], e[k] = k - 1, k + 2
    ops["diffdiff_node_to_node"] = (c, b, e)
    # diffdiff redge -> redge
    c, b, e = mk(L + 1, L + 1)
    for k in range(L + 1):
        if k == 0:
            c[0, 0], c[0, 1] = -2.0 / de ** 2, 2.0 / de ** 2; b[k], e[k] = 0, 2
        elif k == L:
            c[L, L - 1], c[L, L] = 2.0 / de ** 2, -2.0 / de ** 2; b[k], e[k] = L - 1, L + 1
        else:
            c[k, k - 1], c[k, k], c[k, k + 1] = 1.0 / de ** 2, -2.0 / de ** 2, 1.0 / de ** 2
            b[k], e[k] = k - 1, k + 2
    ops["diffdiff_redge_to_redge"] = (c, b, e)
    # discontinuous penalty (left / right)
    c, b, e = mk(L, L)
    for k in range(L - 1):
        c[k, k], c[k, k + 1] = -0.5 / de, 0.5 / de; b[k], e[k] = k, k + 2
    ops["penalty_left"] = (c, b, e)
    c, b, e = mk(L, L)
    for k in range(1, L):
        c[k, k - 1], c[k, k] = 0.5 / de, -0.5 / de; b[k], e[k] = k - 1, k + 1
    ops["penalty_right"] = (c, b, e)
    return ops


# ---------------------------------------------------------------------------------------------
# panel geometry


def xyz_from_xyp(X, Y, p):
    """Unit-sphere Cartesian point of gnomonic (X,Y) on panel p (CubedSphereTrans.cpp:25-86)."""
    X = np.asarray(X, dtype=float); Y = np.asarray(Y, dtype=float)
    sz = 1.0 / np.sqrt(1.0 + X * X + Y * Y)
    sx = sz * X
    sy = sz * Y
    if p == 0:
        return np.stack([sz, sx, sy], -1)
    if p == 1:
        return np.stack([-sx, sz, sy], -1)
    if p == 2:
        return np.stack([-sz, -sx, sy], -1)
    if p == 3:
        return np.stack([sx, -sz, sy], -1)
    if p == 4:
        return np.stack([-sy, sx, sz], -1)
    if p == 5:
        return np.stack([sy, sx, -sz], -1)
    raise ValueError("panel")


def rll_from_xyp(X, Y, p):
    """lon/lat of gnomonic (X,Y) on panel p (CubedSphereTrans.cpp:200-268)."""
    X = np.asarray(X, dtype=float); Y = np.asarray(Y, dtype=float)
    X, Y = np.broadcast_arrays(X, Y)
    if p < 4:
        lon = np.arctan(X) + 0.5 * math.pi * p
        lat = np.arctan(Y / np.sqrt(1.0 + X * X))
    elif p == 4:
        lon = np.where(np.abs(X) > np.finfo(float).eps, np.arctan2(X, -Y),
                       np.where(Y <= 0.0, 0.0, math.pi))
        lat = 0.5 * math.pi - np.arctan(np.sqrt(X * X + Y * Y))
    else:
        lon = np.where(np.abs(X) > np.finfo(float).eps, np.arctan2(X, Y),
                       np.where(Y > 0.0, 0.0, math.pi))
        lat = -0.5 * math.pi + np.arctan(np.sqrt(X * X + Y * Y))
    lon = np.where(lon < 0.0, lon + 2.0 * math.pi, lon)
    return lon, lat


def _covariant_basis(X, Y, p):
    """d(xyz)/d(alpha), d(xyz)/d(beta) on the unit sphere, shape (...,3) each."""
    X = np.asarray(X, dtype=float); Y = np.asarray(Y, dtype=float)
    d2 = 1.0 + X * X + Y * Y
    d = np.sqrt(d2)
    # local panel-centric s = (X, Y, 1)/d ; derivatives wrt X and Y
    dsx_dX = 1.0 / d - X * X / (d * d2)
    dsy_dX = -X * Y / (d * d2)
    dsz_dX = -X / (d * d2)
    dsx_dY = -X * Y / (d * d2)
    dsy_dY = 1.0 / d - Y * Y / (d * d2)
    dsz_dY = -Y / (d * d2)
    fa = 1.0 + X * X   # dX/dalpha
    fb = 1.0 + Y * Y

    def rot(sx, sy, sz):
        if p == 0:
            return np.stack([sz, sx, sy], -1)
        if p == 1:
            return np.stack([-sx, sz, sy], -1)
        if p == 2:
            return np.stack([-sz, -sx, sy], -1)
        if p == 3:
            return np.stack([sx, -sz, sy], -1)
        if p == 4:
            return np.stack([-sy, sx, sz], -1)
        return np.stack([sy, sx, -sz], -1)

    ea = rot(dsx_dX * fa, dsy_dX * fa, dsz_dX * fa)
    eb = rot(dsx_dY * fb, dsy_dY * fb, dsz_dY * fb)
    return ea, eb


def covector_panel_transform(p_src, Xs, Ys, p_dst, Xd, Yd):
    """2x2 matrix M with (u_a,u_b)_dst = M (u_a,u_b)_src for covariant components of the same
    tangent vector at one point seen from two panels.  Equivalent to the twenty closed forms of
    ``CubedSphereTrans::CoVecPanelTrans`` (CubedSphereTrans.h:1751-1823); derived here from the
    3-D embedding: u = u_a e^a + u_b e^b (dual basis of the source panel), then dotted with the
    destination panel's covariant basis."""
    ea_s, eb_s = _covariant_basis(Xs, Ys, p_src)
    ea_d, eb_d = _covariant_basis(Xd, Yd, p_dst)
    gaa = np.sum(ea_s * ea_s, -1); gab = np.sum(ea_s * eb_s, -1); gbb = np.sum(eb_s * eb_s, -1)
    det = gaa * gbb - gab * gab
    # dual basis of source
    da = (gbb[..., None] * ea_s - gab[..., None] * eb_s) / det[..., None]
    db = (-gab[..., None] * ea_s + gaa[..., None] * eb_s) / det[..., None]
    M = np.empty(np.shape(det) + (2, 2))
    M[..., 0, 0] = np.sum(ea_d * da, -1)
    M[..., 0, 1] = np.sum(ea_d * db, -1)
    M[..., 1, 0] = np.sum(eb_d * da, -1)
    M[..., 1, 1] = np.sum(eb_d * db, -1)
    return M


def covec_abp_from_rll(X, Y, p, ulon, ulat):
    """Covariant (alpha,beta) components (unit sphere) of a vector given by its physical zonal /
    meridional components (CubedSphereTrans.cpp:549-638).  Closed forms as in the reference."""
    X = np.asarray(X, dtype=float); Y = np.asarray(Y, dtype=float)
    d2 = 1.0 + X * X + Y * Y
    if p < 4:
        lat = np.arctan(Y / np.sqrt(1.0 + X * X))
        ul = ulon / np.cos(lat)
        ua = (1.0 + X * X) / d2 * ul - X * Y * np.sqrt(1.0 + X * X) / d2 * ulat
        ub = np.sqrt(1.0 + X * X) * (1.0 + Y * Y) / d2 * ulat
        return ua, ub
    rad = np.sqrt(X * X + Y * Y)
    small = (np.abs(X) < 1.0e-13) & (np.abs(Y) < 1.0e-13)
    rads = np.where(small, 1.0, rad)
    if p == 4:
        lat = 0.5 * math.pi - np.arctan(rad)
        ul = ulon / np.where(small, 1.0, np.cos(lat))
        ua = -Y * (1.0 + X * X) / d2 * ul - X * (1.0 + X * X) / (d2 * rads) * ulat
        ub = X * (1.0 + Y * Y) / d2 * ul - Y * (1.0 + Y * Y) / (d2 * rads) * ulat
        ua = np.where(small, ulon, ua)
    else:
        lat = -0.5 * math.pi + np.arctan(rad)
        ul = ulon / np.where(small, 1.0, np.cos(lat))
        ua = Y * (1.0 + X * X) / d2 * ul + X * (1.0 + X * X) / (d2 * rads) * ulat
        ub = -X * (1.0 + Y * Y) / d2 * ul + Y * (1.0 + Y * Y) / (d2 * rads) * ulat
        ua = np.where(small, -ulon, ua)
    ub = np.where(small, ulat, ub)
    return ua, ub


# ---------------------------------------------------------------------------------------------


class Patch:
    """One rectangular block of elements on one panel, reference layout (PatchBox.h)."""

    def __init__(self, index, panel, ga0, ga1, gb0, gb1):
        self.index = index
        self.panel = panel
        self.ga0, self.ga1, self.gb0, self.gb1 = ga0, ga1, gb0, gb1   # global interior [begin,end)
        self.na = ga1 - ga0 + 2   # total widths incl. one-node halo
        self.nb = gb1 - gb0 + 2
        self.geom = {}
        self.halo = None


class CubedSphereGrid:
    """GLL cubed-sphere grid in the reference's patch layout.

    ``ppd`` = patches per panel direction (``6*ppd**2`` patches, GridCSGLL.cpp:101-145).
    """

    def __init__(self, ne, levels, ztop, ppd=1, npts=4, phys=None, shallow_water=False):
        self.shallow_water = shallow_water     # 2-D equation set: one level, ztop = 1, no topography in the metric
        if shallow_water:
            levels, ztop = 1, 1.0
        if ne % ppd != 0:
            raise ValueError("elements must divide equally among patches (GridCSGLL.cpp:115-118)")
        self.ne, self.L, self.ztop, self.ppd, self.np = ne, levels, float(ztop), ppd, npts
        self.phys = phys or PhysicalConstants()
        self.Nn = ne * npts
        self.delta_a = 0.5 * math.pi / ne
        self.reference_length = 0.5 * math.pi / 30.0   # GridCSGLL.cpp:87
        self.gll, self.gllw = gll_points_weights_01(npts)
        self.dx, self.stiff, _ = dx_basis_stiffness(npts)
        L = levels
        self.reta_levels = (0.5 + np.arange(L)) * (1.0 / L)
        self.reta_interfaces = np.arange(L + 1) * (1.0 / L)
        self.reta_levels_normarea = np.full(L, 1.0 / L)
        wi = np.full(L + 1, 0.5 / L)
        wi[1:L] *= 2.0
        self.reta_interfaces_normarea = wi
        self.ops = None if shallow_water else column_operators_vo1(L)

        self.patches = []
        epp = ne // ppd
        ix = 0
        for p in range(6):
            for i in range(ppd):
                for j in range(ppd):
                    self.patches.append(Patch(ix, p, npts * epp * i, npts * epp * (i + 1),
                                              npts * epp * j, npts * epp * (j + 1)))
                    ix += 1
        self._init_coordinates()
        self._init_edge_maps()
        self._init_halos()

    # -- coordinates -------------------------------------------------------------------------
    def node_coord(self, g):
        """alpha of global node index g (GridSpacing.cpp:180-197); works for the halo (-1, Nn)."""
        g = np.asarray(g)
        e = np.floor_divide(g, self.np)
        s = np.mod(g, self.np)
        return -0.25 * math.pi + self.delta_a * e.astype(float) + self.delta_a * self.gll[s]

    def _init_coordinates(self):
        for P in self.patches:
            ga = np.arange(P.ga0 - 1, P.ga1 + 1)
            gb = np.arange(P.gb0 - 1, P.gb1 + 1)
            P.a_nodes = self.node_coord(ga)
            P.b_nodes = self.node_coord(gb)
            P.X = np.tan(P.a_nodes)
            P.Y = np.tan(P.b_nodes)
            lon, lat = rll_from_xyp(P.X[:, None], P.Y[None, :], P.panel)
            P.lon, P.lat = lon, lat

    def patch_of(self, panel, gA, gB):
        epp_nodes = self.Nn // self.ppd
        return panel * self.ppd * self.ppd + (gA // epp_nodes) * self.ppd + (gB // epp_nodes)

    # -- connectivity --------------------------------------------------------------------------
    def _edge_nodes(self, p, e):
        """Global (gA,gB) of the interior boundary nodes along edge e of panel p, along index t."""
        t = np.arange(self.Nn)
        N1 = self.Nn - 1
        if e == EDGE_RIGHT:
            return np.full_like(t, N1), t
        if e == EDGE_TOP:
            return t, np.full_like(t, N1)
        if e == EDGE_LEFT:
            return np.zeros_like(t), t
        return t, np.zeros_like(t)

    def _init_edge_maps(self):
        """For each (panel, edge): neighbour panel, its edge, and whether the along-edge index is
        reversed.  Found geometrically (coincident boundary points) instead of restating the
        reference's RelativeCoord tables (GridCSGLL.cpp:272-406)."""
        self.edge_map = {}
        pos = {}
        for p in range(6):
            for e in range(4):
                gA, gB = self._edge_nodes(p, e)
                pos[(p, e)] = xyz_from_xyp(np.tan(self.node_coord(gA)), np.tan(self.node_coord(gB)), p)
        for p in range(6):
            for e in range(4):
                x1 = pos[(p, e)][1]   # t=1: an element-interior GLL node, unique position
                found = None
                for q in range(6):
                    if q == p:
                        continue
                    for f in range(4):
                        d = np.linalg.norm(pos[(q, f)] - x1, axis=-1)
                        k = int(np.argmin(d))
                        if d[k] < 1e-10:
                            if k == 1:
                                found = (q, f, False)
                            elif k == self.Nn - 2:
                                found = (q, f, True)
                            else:
                                raise RuntimeError("unexpected edge match")
                if found is None:
                    raise RuntimeError("no neighbour for panel %d edge %d" % (p, e))
                q, f, rev = found
                # verify all nodes coincide
                t = np.arange(self.Nn)
                tt = (self.Nn - 1 - t) if rev else t
                if np.max(np.abs(pos[(p, e)] - pos[(q, f)][tt])) > 1e-10:
                    raise RuntimeError("edge map verification failed")
                self.edge_map[(p, e)] = found

    def halo_source(self, panel, gA, gB):
        """Where the value of (possibly out-of-range) global node (gA,gB) of ``panel`` lives:
        returns (src_panel, src_gA, src_gB) or None at a cube corner."""
        Nn = self.Nn
        outA = (gA < 0) or (gA >= Nn)
        outB = (gB < 0) or (gB >= Nn)
        if outA and outB:
            return None
        if not outA and not outB:
            return panel, gA, gB
        if outA:
            e = EDGE_LEFT if gA < 0 else EDGE_RIGHT
            t = gB
        else:
            e = EDGE_BOTTOM if gB < 0 else EDGE_TOP
            t = gA
        q, f, rev = self.edge_map[(panel, e)]
        tt = (Nn - 1 - t) if rev else t
        if f == EDGE_RIGHT:
            return q, Nn - 1, tt
        if f == EDGE_TOP:
            return q, tt, Nn - 1
        if f == EDGE_LEFT:
            return q, 0, tt
        return q, tt, 0

    def _init_halos(self):
        """Halo-ring table of every patch: for each ring node its source patch / local indices and
        the covector transform (identity on the same panel).  Mirrors what Grid::Exchange +
        TransformHaloVelocities deliver (Grid.cpp:627-685, GridPatchCSGLL.cpp:1783-1924)."""
        for P in self.patches:
            ring = []
            for i in range(P.na):
                for j in range(P.nb):
                    if i == 0 or j == 0 or i == P.na - 1 or j == P.nb - 1:
                        ring.append((i, j))
            hi = np.array([r[0] for r in ring], dtype=np.int32)
            hj = np.array([r[1] for r in ring], dtype=np.int32)
            n = len(ring)
            sp = np.full(n, -1, dtype=np.int32)
            si = np.zeros(n, dtype=np.int32)
            sj = np.zeros(n, dtype=np.int32)
            spanel = np.full(n, -1, dtype=np.int32)
            tr = np.zeros((n, 2, 2))
            tr[:, 0, 0] = 1.0
            tr[:, 1, 1] = 1.0
            for m, (i, j) in enumerate(ring):
                gA = P.ga0 - 1 + i
                gB = P.gb0 - 1 + j
                src = self.halo_source(P.panel, gA, gB)
                if src is None:
                    continue
                q, sgA, sgB = src
                Q = self.patches[self.patch_of(q, sgA, sgB)]
                sp[m] = Q.index
                spanel[m] = q
                si[m] = sgA - Q.ga0 + 1
                sj[m] = sgB - Q.gb0 + 1
                if q != P.panel:
                    Xs = math.tan(float(self.node_coord(sgA)))
                    Ys = math.tan(float(self.node_coord(sgB)))
                    tr[m] = covector_panel_transform(q, Xs, Ys, P.panel, P.X[i], P.Y[j])
            P.halo = dict(i=hi, j=hj, src_patch=sp, src_i=si, src_j=sj, src_panel=spanel, trans=tr)
            # neighbour panels in Direction order: Right, Top, Left, Bottom, TR, TL, BL, BR
            nbp = []
            for (da, db) in [(1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1)]:
                gA = (P.ga1 if da > 0 else (P.ga0 - 1 if da < 0 else P.ga0))
                gB = (P.gb1 if db > 0 else (P.gb0 - 1 if db < 0 else P.gb0))
                src = self.halo_source(P.panel, gA, gB)
                nbp.append(-1 if src is None else src[0])
            P.neighbor_panels = np.array(nbp, dtype=np.int32)

    # -- halo fill + reference-style DSS on host arrays (used for topography derivatives) --------
    def exchange(self, arrays, vector_pair=None):
        """Fill the halo ring of ``arrays[p]`` (shape [A][B][...]) from the source patches' interior
        nodes.  ``vector_pair=(c0,c1)``: last-axis components forming a covector to be rotated."""
        snap = [a.copy() for a in arrays]
        for P in self.patches:
            h = P.halo
            ok = h["src_patch"] >= 0
            for m in np.nonzero(ok)[0]:
                v = snap[h["src_patch"][m]][h["src_i"][m], h["src_j"][m]].copy()
                if vector_pair is not None and h["src_panel"][m] != P.panel:
                    c0, c1 = vector_pair
                    M = h["trans"][m]
                    a, b = v[..., c0].copy(), v[..., c1].copy()
                    v[..., c0] = M[0, 0] * a + M[0, 1] * b
                    v[..., c1] = M[1, 0] * a + M[1, 1] * b
                arrays[P.index][h["i"][m], h["j"][m]] = v

    def dss_average(self, arrays):
        """Pairwise alpha-then-beta averaging with the 1/3 rule at cube corners on halo-filled
        arrays (GridCSGLL.cpp:560-777)."""
        npts = self.np
        for P in self.patches:
            x = arrays[P.index]
            nea = (P.ga1 - P.ga0) // npts
            neb = (P.gb1 - P.gb0) // npts
            nb = P.neighbor_panels
            TR, TL, BL, BR = nb[4], nb[5], nb[6], nb[7]
            aib, aie = 1, P.na - 1
            bib, bie = 1, P.nb - 1

            def avg_a(iA, jb, je):
                s = 0.5 * (x[iA, jb:je] + x[iA - 1, jb:je])
                x[iA, jb:je] = s
                x[iA - 1, jb:je] = s

            def avg_b(iB, ib, ie):
                s = 0.5 * (x[ib:ie, iB] + x[ib:ie, iB - 1])
                x[ib:ie, iB] = s
                x[ib:ie, iB - 1] = s

            for a in range(1, nea):
                avg_a(a * npts + 1, bib - 1, bie + 1)
            jb, je = bib - 1, bie + 1
            if TL < 0: je -= 2
            if BL < 0: jb += 2
            avg_a(1, jb, je)
            jb, je = bib - 1, bie + 1
            if TR < 0: je -= 2
            if BR < 0: jb += 2
            avg_a(nea * npts + 1, jb, je)
            for b in range(1, neb):
                avg_b(b * npts + 1, aib - 1, aie + 1)
            ib, ie = aib - 1, aie + 1
            if BL < 0: ib += 2
            if BR < 0: ie -= 2
            avg_b(1, ib, ie)
            ib, ie = aib - 1, aie + 1
            if TL < 0: ib += 2
            if TR < 0: ie -= 2
            avg_b(neb * npts + 1, ib, ie)
            third = 1.0 / 3.0
            if TR < 0:
                iA, iB = aie - 1, bie - 1
                x[iA, iB] = third * (x[iA, iB] + x[iA + 1, iB] + x[iA, iB + 1])
            if TL < 0:
                iA, iB = aib, bie - 1
                x[iA, iB] = third * (x[iA, iB] + x[iA - 1, iB] + x[iA, iB + 1])
            if BL < 0:
                iA, iB = aib, bib
                x[iA, iB] = third * (x[iA, iB] + x[iA - 1, iB] + x[iA, iB - 1])
            if BR < 0:
                iA, iB = aie - 1, bib
                x[iA, iB] = third * (x[iA, iB] + x[iA + 1, iB] + x[iA, iB - 1])

    # -- metric ----------------------------------------------------------------------------------
    def set_topography(self, zs_func):
        """Evaluate topography on all nodes (incl. halo), its element-wise derivatives, and DSS them
        (GridPatchCSGLL.cpp:226-290, GridGLL.cpp:557-567)."""
        npts = self.np
        derivs = []
        for P in self.patches:
            P.geom["topography"] = zs_func(self.phys, P.lon, P.lat)
            zs = P.geom["topography"]
            d = np.zeros((P.na, P.nb, 2))
            nea = (P.ga1 - P.ga0) // npts
            neb = (P.gb1 - P.gb0) // npts
            for a in range(nea):
                for b in range(neb):
                    ia, ib = 1 + a * npts, 1 + b * npts
                    blk = zs[ia:ia + npts, ib:ib + npts]
                    # dDaZs(i,j) = sum_s D[s][i] zs(s,j)
                    d[ia:ia + npts, ib:ib + npts, 0] = (self.dx.T @ blk) / self.delta_a
                    d[ia:ia + npts, ib:ib + npts, 1] = (blk @ self.dx) / self.delta_a
            derivs.append(d)
        self.exchange(derivs, vector_pair=(0, 1))
        self.dss_average(derivs)
        for P in self.patches:
            P.geom["topography_deriv"] = derivs[P.index]

    def evaluate_geometric_terms(self):
        """Metric arrays on interior nodes, zero elsewhere (GridPatchCSGLL.cpp:295-574)."""
        a = self.phys.earth_radius
        L = self.L
        for P in self.patches:
            g = P.geom
            na, nb = P.na, P.nb
            X = P.X[:, None] * np.ones((1, nb))
            Y = np.ones((na, 1)) * P.Y[None, :]
            mask = np.zeros((na, nb), dtype=bool)
            mask[1:-1, 1:-1] = True
            d2 = 1.0 + X * X + Y * Y
            d = np.sqrt(d2)
            zs = g["topography"]
            dazs = g["topography_deriv"][..., 0]
            dbzs = g["topography_deriv"][..., 1]
            if self.shallow_water:     # fIs2DEquationSet (GridPatchCSGLL.cpp:388-393)
                zs = np.zeros_like(zs); dazs = np.zeros_like(dazs); dbzs = np.zeros_like(dbzs)
            g["coriolis_f"] = 2.0 * self.phys.omega * np.sin(P.lat)
            J2 = (1.0 + X * X) * (1.0 + Y * Y) / (d * d * d)
            J2 = J2 * a * a
            g["jacobian2d"] = np.where(mask, J2, 0.0)
            sc = d2 / (1.0 + X * X) / (1.0 + Y * Y) / (a * a)
            cma = np.zeros((na, nb, 2)); cmb = np.zeros((na, nb, 2))
            cma[..., 0] = sc * (1.0 + Y * Y); cma[..., 1] = sc * X * Y
            cmb[..., 0] = sc * X * Y; cmb[..., 1] = sc * (1.0 + X * X)
            cma[~mask] = 0.0; cmb[~mask] = 0.0
            g["contra_metric_2d_a"] = cma
            g["contra_metric_2d_b"] = cmb
            wi = self.gllw
            li = (np.arange(na) - 1) % self.np
            lj = (np.arange(nb) - 1) % self.np
            area2 = (wi[li] * self.delta_a)[:, None] * (wi[lj] * self.delta_a)[None, :]
            for loc, reta, warea in (("node", self.reta_levels, self.reta_levels_normarea),
                                     ("redge", self.reta_interfaces, self.reta_interfaces_normarea)):
                n = len(reta)
                dxr = (self.ztop - zs)[..., None] * np.ones(n)
                dar = (1.0 - reta)[None, None, :] * dazs[..., None]
                dbr = (1.0 - reta)[None, None, :] * dbzs[..., None]
                if loc == "node":
                    J = dxr * J2[..., None]
                else:
                    J = ((1.0 + X * X) * (1.0 + Y * Y) / (d * d * d))[..., None] * (dxr * a * a)
                sc3 = sc[..., None]
                X3, Y3 = X[..., None], Y[..., None]
                A = np.zeros((na, nb, n, 3)); B = np.zeros((na, nb, n, 3)); Xi = np.zeros((na, nb, n, 3))
                A[..., 0] = cma[..., 0][..., None]; A[..., 1] = cma[..., 1][..., None]
                A[..., 2] = -sc3 / dxr * ((1.0 + Y3 * Y3) * dar + X3 * Y3 * dbr)
                B[..., 0] = cmb[..., 0][..., None]; B[..., 1] = cmb[..., 1][..., None]
                B[..., 2] = -sc3 / dxr * (X3 * Y3 * dar + (1.0 + X3 * X3) * dbr)
                Xi[..., 0] = A[..., 2]; Xi[..., 1] = B[..., 2]
                Xi[..., 2] = 1.0 / (dxr * dxr) - 1.0 / dxr * (Xi[..., 0] * dar + Xi[..., 1] * dbr)
                dr = np.stack([dar, dbr, dxr], -1)
                area = J * area2[..., None] * warea[None, None, :]
                for arr in (J, A, B, Xi, dr, area):
                    arr[~mask] = 0.0
                sfx = "" if loc == "node" else "_redge"
                g["jacobian" + sfx] = J
                g["contra_metric_a" + sfx] = A
                g["contra_metric_b" + sfx] = B
                g["contra_metric_xi" + sfx] = Xi
                g["deriv_r_" + loc] = dr
                g["element_area_" + loc] = area
            g["z_levels"] = zs[..., None] + self.reta_levels[None, None, :] * (self.ztop - zs)[..., None]
            g["z_interfaces"] = zs[..., None] + self.reta_interfaces[None, None, :] * (self.ztop - zs)[..., None]

    # -- state -----------------------------------------------------------------------------------
    def evaluate_test_case(self, test):
        """Topography, geometry, initial and reference state on every node incl. halo
        (GridPatchCSGLL.cpp:578-920; Model::SetTestCase, Model.cpp:225-258)."""
        test.evaluate_physical_constants(self.phys)
        if getattr(test, "reference_length", None) is not None:
            self.reference_length = test.reference_length
        self.set_topography(test.evaluate_topography)
        self.evaluate_geometric_terms()
        a = self.phys.earth_radius
        if self.shallow_water:
            return self._evaluate_shallow_water(test)
        states = []
        for P in self.patches:
            g = P.geom
            out = {}
            for loc, z in (("node", g["z_levels"]), ("redge", g["z_interfaces"])):
                lon = P.lon[..., None] * np.ones(z.shape[-1])
                lat = P.lat[..., None] * np.ones(z.shape[-1])
                for kind in ("state", "ref"):
                    st = test.evaluate_pointwise(self.phys, z, lon, lat, reference=(kind == "ref"))
                    # ConvertComponents: theta -> rho*theta (EquationSet.cpp:153-155)
                    st[PIX] = st[PIX] * st[RIX]
                    ua, ub = covec_abp_from_rll(P.X[:, None, None], P.Y[None, :, None], P.panel,
                                                st[UIX] * a, st[VIX] * a)
                    st[UIX], st[VIX] = ua, ub
                    out[kind + "_" + loc] = np.stack([np.broadcast_to(s, z.shape) for s in st], 0).copy()
            g["ref_node"] = out["ref_node"]
            g["ref_redge"] = out["ref_redge"]
            # tracer densities on levels (GridPatch::GetDataTracers), every node incl. halo
            nt = int(getattr(test, "ntracers", 0))
            if nt > 0:
                z = g["z_levels"]
                lon = P.lon[..., None] * np.ones(z.shape[-1]); lat = P.lat[..., None] * np.ones(z.shape[-1])
                tr = test.evaluate_tracers(self.phys, z, lon, lat, out["state_node"][RIX])
                self.initial_tracers = getattr(self, "initial_tracers", {})
                self.initial_tracers[P.index] = np.stack([np.broadcast_to(t, z.shape) for t in tr], 0).copy()
                if hasattr(test, "evaluate_reference_tracers"):      # GridPatch::GetReferenceTracers
                    rt = test.evaluate_reference_tracers(self.phys, z, lon, lat, out["state_node"][RIX])
                    g["ref_tracers"] = np.stack([np.broadcast_to(t, z.shape) for t in rt], 0).copy()
            # Rayleigh friction strength on levels and interfaces, every node incl. halo (GridPatchCSGLL.cpp:663-683)
            if getattr(test, "has_rayleigh_friction", False):
                g["rayleigh_node"] = test.evaluate_rayleigh_strength(g["z_levels"])
                g["rayleigh_redge"] = test.evaluate_rayleigh_strength(g["z_interfaces"])
            states.append((out["state_node"], out["state_redge"]))
        self.has_rayleigh_friction = bool(getattr(test, "has_rayleigh_friction", False))
        self.ntracers = int(getattr(test, "ntracers", 0))
        return states

    def _evaluate_shallow_water(self, test):
        """Shallow-water state (U, V, H) on one level; stored in the 5-slot containers of this package with
        H in slot 2 and slots 3,4 zero (GridPatchCSGLL.cpp:578-920 with a 3-component equation set)."""
        a = self.phys.earth_radius
        states = []
        for P in self.patches:
            u, v, h = test.evaluate_pointwise(self.phys, P.lon, P.lat)
            ua, ub = covec_abp_from_rll(P.X[:, None], P.Y[None, :], P.panel, u * a, v * a)
            node = np.zeros((5, P.na, P.nb, 1)); redge = np.zeros((5, P.na, P.nb, 2))
            node[UIX, ..., 0] = ua; node[VIX, ..., 0] = ub; node[PIX, ..., 0] = h
            P.geom["ref_node"] = np.zeros_like(node); P.geom["ref_redge"] = np.zeros_like(redge)
            states.append((node, redge))
        return states

    # -- diagnostics ------------------------------------------------------------------------------
    def checksum(self, states):
        """Area-weighted sums over interior nodes (GridPatch.cpp:811-835): node vars U,V,rhotheta,rho
        with element_area_node, W with element_area_redge."""
        out = np.zeros(5)
        for P, (node, redge) in zip(self.patches, states):
            an = P.geom["element_area_node"][1:-1, 1:-1]
            ae = P.geom["element_area_redge"][1:-1, 1:-1]
            for c in (UIX, VIX, PIX, RIX):
                out[c] += np.sum(node[c, 1:-1, 1:-1] * an)
            out[WIX] += np.sum(redge[WIX, 1:-1, 1:-1] * ae)
        return out


# ---------------------------------------------------------------------------------------------


class BaroclinicWaveJW:
    """Jablonowski-Williamson (2006) baroclinic wave, as evaluated by the reference test driver
    (test/nonhydro_sphere/BaroclinicWaveJWTest.cpp:170-413)."""

    eta0 = 0.252
    tropopause_eta = 0.2
    T0 = 288.0
    deltaT = 4.8e5
    lapse = 0.005
    U0 = 35.0
    Up = 1.0
    pert_lon = math.pi / 9.0
    pert_lat = 2.0 * math.pi / 9.0
    pert_r = 0.1

    def __init__(self, alpha=0.0, ztop=30000.0, pert="exp", ntracers=0):
        self.alpha = alpha
        self.ztop = ztop
        self.pert = pert
        self.ntracers = ntracers      # closed-form tracer densities of oracle/ref_dump.cpp's JWWithTracers (a test input)

    def evaluate_tracers(self, phys, z, lon, lat, rho):
        """Tracer densities rho*q, same formulas as JWWithTracers in oracle/ref_dump.cpp."""
        out = []
        if self.ntracers > 0:
            out.append(rho * (0.5 * (1.0 + np.sin(lat) * np.cos(lon)) * np.exp(-z / 8000.0) + 0.01))
        if self.ntracers > 1:
            r = np.arccos(math.sin(0.6) * np.sin(lat) + math.cos(0.6) * np.cos(lat) * np.cos(lon - 0.9)) / 0.5
            q = np.where((r < 1.0) & (z < 12000.0), 0.5 * (1.0 + np.cos(math.pi * r)), 0.0)
            out.append(rho * q)
        for c in range(2, self.ntracers):
            out.append(rho * 0.001 * (c + 1))
        return out

    def evaluate_physical_constants(self, phys):
        phys.alpha = self.alpha

    def _profiles(self, phys, aux_eta, lat):
        s = np.sin(lat); c = np.cos(lat)
        s2 = s * s; s3 = s * s2; s4 = s * s3; s5 = s * s4; s6 = s * s5
        c2 = c * c; c3 = c * c2
        r1 = self.U0 * np.cos(aux_eta) ** 1.5 * (-2.0 * s6 * (c2 + 1.0 / 3.0) + 10.0 / 63.0)
        r2 = phys.earth_radius * phys.omega * (8.0 / 5.0 * c3 * (s2 + 2.0 / 3.0) - 0.25 * math.pi)
        return r1, r2

    def evaluate_topography(self, phys, lon, lat):
        aux = 0.5 * math.pi * (1.0 - self.eta0)
        r1, r2 = self._profiles(phys, aux, lat)
        return self.U0 * math.cos(aux) ** 1.5 * (r1 + r2) / phys.g

    def geopotential_temperature(self, phys, eta, lat):
        aux = 0.5 * math.pi * (eta - self.eta0)
        expo = phys.Rd * self.lapse / phys.g
        avgT = self.T0 * eta ** expo
        below = eta < self.tropopause_eta
        avgT = avgT + np.where(below, self.deltaT * np.abs(self.tropopause_eta - eta) ** 5.0, 0.0)
        r1, r2 = self._profiles(phys, aux, lat)
        T = 2.0 * r1 + r2
        T = avgT + 0.75 * eta * math.pi * self.U0 / phys.Rd * np.sin(aux) * np.sqrt(np.cos(aux)) * T
        avgPhi = self.T0 * phys.g / self.lapse * (1.0 - eta ** expo)
        te = self.tropopause_eta
        e2 = eta * eta; e3 = eta * e2; e4 = eta * e3; e5 = eta * e4
        t2 = te * te; t3 = te * t2; t4 = te * t3; t5 = te * t4
        corr = phys.Rd * self.deltaT * (
            (np.log(eta / te) + 137.0 / 60.0) * t5 - 5.0 * t4 * eta + 5.0 * t3 * e2
            - (10.0 / 3.0) * t2 * e3 + 5.0 / 4.0 * te * e4 - 1.0 / 5.0 * e5)
        avgPhi = avgPhi - np.where(below, corr, 0.0)
        Phi = avgPhi + self.U0 * np.cos(aux) ** 1.5 * (r1 + r2)
        return Phi, T

    def eta_from_z(self, phys, z, lat):
        """Newton iteration of EtaFromRLL (BaroclinicWaveJWTest.cpp:318-366), vectorised with a
        per-point freeze at convergence so each point follows the scalar algorithm."""
        eta = np.full(z.shape, 1.0e-7)
        done = np.zeros(z.shape, dtype=bool)
        Phi = np.zeros(z.shape); T = np.zeros(z.shape)
        for _ in range(25):
            with np.errstate(all="ignore"):
                Ph, Tt = self.geopotential_temperature(phys, eta, lat)
            Phi = np.where(done, Phi, Ph)
            T = np.where(done, T, Tt)
            F = -phys.g * z + Ph
            dF = -phys.Rd / eta * Tt
            new = eta - F / dF
            conv = np.abs(eta - new) < 1.0e-13
            upd = ~done
            eta = np.where(upd, new, eta)
            done = done | (upd & conv)
            if done.all():
                break
        if not done.all():
            raise RuntimeError("EtaFromRLL: maximum number of iterations exceeded")
        return eta, Phi, T

    def evaluate_pointwise(self, phys, z, lon, lat, reference=False):
        eta, Phi, T = self.eta_from_z(phys, z, lat)
        ulon = self.U0 * np.cos(0.5 * math.pi * (eta - self.eta0)) ** 1.5 * np.sin(2.0 * lat) * np.sin(2.0 * lat)
        p = phys.p0 * eta
        rho = p / (phys.Rd * T)
        rhotheta = phys.rhotheta_from_pressure(p)
        theta = rhotheta / rho
        if (not reference) and self.pert == "exp":
            gc = np.arccos(np.clip(math.sin(self.pert_lat) * np.sin(lat)
                                   + math.cos(self.pert_lat) * np.cos(lat) * np.cos(lon - self.pert_lon), -1.0, 1.0))
            gc = gc / self.pert_r
            ulon = ulon + np.where(gc < 1.0, self.Up * np.exp(-gc * gc), 0.0)
        zero = np.zeros_like(ulon)
        return [ulon, zero, theta, zero.copy(), rho]


class BaroclinicWaveJWSmallPlanet(BaroclinicWaveJW):
    """The configuration of the DCMIP2016 supercell (test/dcmip2016/SupercellTest.cpp:99-113: planet reduced by X, no
    rotation, uniform diffusion K_scalar / K_vector, --explicitvertical) driven by the Jablonowski-Williamson state:
    mirror of oracle/test_inputs.h's JWSmallPlanetDiffusion (a test input).  The reference state is 0.9 x the
    pointwise state (tracers 0.8 x), so that "state minus reference state" is non-trivial everywhere."""

    def __init__(self, ztop=30000.0, pert="exp", ntracers=0, X=120.0, udiff=(1500.0, 500.0)):
        BaroclinicWaveJW.__init__(self, 0.0, ztop, pert, ntracers=ntracers)
        self.X = X
        self.uniform_diffusion = tuple(udiff)

    @property
    def reference_length(self):
        return 0.5 * math.pi / 30.0 * self.X

    def evaluate_physical_constants(self, phys):
        BaroclinicWaveJW.evaluate_physical_constants(self, phys)
        phys.omega = 0.0
        phys.earth_radius = phys.earth_radius / self.X

    def evaluate_pointwise(self, phys, z, lon, lat, reference=False):
        st = BaroclinicWaveJW.evaluate_pointwise(self, phys, z, lon, lat, reference=False)
        return [0.9 * v for v in st] if reference else st

    def evaluate_reference_tracers(self, phys, z, lon, lat, rho):
        return [0.8 * t for t in self.evaluate_tracers(phys, z, lon, lat, rho)]


class ScharMountainSphere:
    """Schar-type mountain on a reduced-radius planet with a Rayleigh damping layer, as evaluated by the
    reference test driver (test/nonhydro_sphere/ScharMountainSphereTest.cpp:120-250, defaults :417-441).
    The one stock cubed-sphere test family whose Rayleigh strength is not identically zero."""

    has_rayleigh_friction = True

    def __init__(self, ztop=30000.0, X=500.0, omega=0.0, lonc=45.0, latc=0.0, h0=250.0, d=5000.0, xi=4000.0,
                 teq=300.0, ueq=20.0, cs=0.0, zh=20000.0, tau0=25.0):
        self.ztop, self.X, self.omega = ztop, X, omega
        self.lonc, self.latc = lonc * math.pi / 180.0, latc * math.pi / 180.0
        self.h0, self.d, self.xi, self.teq, self.ueq, self.cs, self.zh, self.tau0 = h0, d, xi, teq, ueq, cs, zh, tau0

    @property
    def reference_length(self):
        """model.GetGrid()->SetReferenceLength(0.5 * M_PI / 30.0 * dEarthScaling) (:463)."""
        return 0.5 * math.pi / 30.0 * self.X

    def evaluate_physical_constants(self, phys):
        phys.omega = self.omega * self.X
        phys.earth_radius = phys.earth_radius / self.X

    def evaluate_topography(self, phys, lon, lat):
        r = phys.earth_radius * np.arccos(math.sin(self.latc) * np.sin(lat)
                                          + math.cos(self.latc) * np.cos(lat) * np.cos(lon - self.lonc))
        cos_term = np.cos(math.pi * r / self.xi)
        exp_term = np.exp(-r * r / (self.d * self.d))
        return self.h0 * exp_term * cos_term * cos_term

    def evaluate_rayleigh_strength(self, z):
        nz = (z - self.zh) / (self.ztop - self.zh)
        depth = np.sin(math.pi / 2.0 * nz)
        depth = depth * depth
        return np.where(z > self.zh, depth / self.tau0, 0.0)

    def evaluate_pointwise(self, phys, z, lon, lat, reference=False):
        s2 = np.sin(lat) * np.sin(lat)
        T = self.teq * (1.0 - self.cs * self.ueq * self.ueq / phys.g * s2)
        p = phys.p0 * np.exp(-self.ueq * self.ueq / (2.0 * phys.Rd * self.teq) * s2 - phys.g * z / (phys.Rd * T))
        rho = p / (phys.Rd * T)
        u = self.ueq * np.cos(lat) * np.sqrt(2.0 * self.teq / T * self.cs * z + T / self.teq)
        zero = np.zeros_like(z)
        return [u, zero.copy(), phys.rhotheta_from_pressure(p) / rho, zero.copy(), rho]


class FlatPlanetCase:
    """Grid set-up of the stock tests whose STATE is taken from reference dumps rather than synthesised: flat topography,
    planet reduced by X, rotation on or off, optional uniform diffusion / Rayleigh layer flags.  Used for the DCMIP2016
    supercell (test/dcmip2016/SupercellTest.cpp:99-113, 285-286: X = 120, no rotation, K = 1500 / 500 m^2/s, reference
    length / X; its initial state comes out of Fortran, interface/supercell_test.f90) and for the Held-Suarez test
    (test/nonhydro_sphere/HeldSuarezTest.cpp: X = 1, ztop 35 km, Rayleigh strength 0.0 x profile (:319), initial state
    perturbed with rand()).  evaluate_pointwise returns a resting isothermal placeholder."""

    def __init__(self, ztop, X=1.0, rotating=True, ntracers=0, udiff=None, rayleigh=False, scale_reference_length=None):
        self.ztop, self.X, self.rotating, self.ntracers = ztop, X, rotating, ntracers
        if udiff is not None:
            self.uniform_diffusion = tuple(udiff)
        self.has_rayleigh_friction = bool(rayleigh)
        self._ref_scale = scale_reference_length

    @property
    def reference_length(self):
        return None if self._ref_scale is None else 0.5 * math.pi / 30.0 * self._ref_scale

    def evaluate_physical_constants(self, phys):
        if not self.rotating:
            phys.omega = 0.0
        phys.earth_radius = phys.earth_radius / self.X

    def evaluate_topography(self, phys, lon, lat):
        return np.zeros_like(lon)

    def evaluate_rayleigh_strength(self, z):
        return np.zeros_like(z)

    def evaluate_tracers(self, phys, z, lon, lat, rho):
        return [np.zeros_like(z) for _ in range(self.ntracers)]

    def evaluate_pointwise(self, phys, z, lon, lat, reference=False):
        H = phys.Rd * 280.0 / phys.g
        p = phys.p0 * np.exp(-z / H)
        rho = p / phys.g / H
        zero = np.zeros_like(z)
        return [zero.copy(), zero.copy(), phys.rhotheta_from_pressure(p) / rho, zero.copy(), rho]


class ShallowWaterTest2:
    """Williamson et al. (1992) test 2, steady geostrophic flow, as evaluated by
    test/shallowwater_sphere/SWTest2.cpp:59-118 (defaults of its command line: h0, u0, alpha = 0)."""

    def __init__(self, h0=2998.104995, u0=38.61068277, alpha_deg=0.0):
        self.h0, self.u0, self.alpha = h0, u0, alpha_deg * math.pi / 180.0

    def evaluate_physical_constants(self, phys):
        phys.alpha = self.alpha

    def evaluate_topography(self, phys, lon, lat):
        return np.zeros_like(lon)

    def evaluate_pointwise(self, phys, lon, lat):
        lat = np.where(np.abs(lat - 0.5 * math.pi) < 1.0e-12, lat - 1.0e-12, lat)
        lat = np.where(np.abs(lat + 0.5 * math.pi) < 1.0e-12, lat + 1.0e-12, lat)
        u = self.u0 * np.cos(lat) * (math.cos(self.alpha) + np.cos(lon) * np.tan(lat) * math.sin(self.alpha))
        v = -self.u0 * np.sin(lon) * math.sin(self.alpha)
        htrig = -np.cos(lon) * np.cos(lat) * math.sin(self.alpha) + np.sin(lat) * math.cos(self.alpha)
        h = self.h0 - (phys.earth_radius * phys.omega + 0.5 * self.u0) * self.u0 * htrig * htrig / phys.g
        return u, v, h
