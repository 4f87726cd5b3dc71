"""Host-side template plugins (scarplet_amd/WindowedTemplate.py) against the
reference's classes (fixtures captured by oracle/gen_golden.py)."""
import numpy as np
import pytest

from scarplet_amd import WindowedTemplate as WT
from conftest import golden, load_cases

CLS = {"scarp": WT.Scarp, "ricker": WT.Ricker,
       "right_upper_break": WT.RightFacingUpperBreakScarp,
       "left_upper_break": WT.LeftFacingUpperBreakScarp}


def test_reference_golden_templates():
    assert np.allclose(WT.Scarp(100, 10, 0, 100, 100, 1).template(),
                       np.load(golden("ref_scarp_template.npy")))
    assert np.allclose(WT.Channel(100, 0.1, 0, 100, 100, 1).template(),
                       np.load(golden("ref_channel_template.npy")))


def test_plugin_methods_match_reference():
    for c in load_cases("ref_templates.npz"):
        t = CLS[str(c["kind"])](float(c["d"]), float(c["p"]), float(c["ang"]),
                                int(c["nx"]), int(c["ny"]), float(c["de"]))
        W = t.template()
        assert np.allclose(W, c["W"], rtol=1e-13, atol=0)
        assert np.array_equal(W != 0, c["W"] != 0)
        assert np.array_equal(t.get_window_limits(), c["lim"])
        if c["err"].size:
            assert np.array_equal(t.get_err_mask(), c["err"])


def test_descriptor_bounds_are_the_masks():
    """The index bounds handed to the device describe exactly the reference
    masks: kept rectangle == ~get_window_limits(), support inside the bbox."""
    for c in load_cases("ref_templates.npz"):
        nx, ny = int(c["nx"]), int(c["ny"])
        t = CLS[str(c["kind"])](float(c["d"]), float(c["p"]), float(c["ang"]), nx, ny, float(c["de"]))
        d = t._device_descriptor()
        ilo, ihi, jlo, jhi = d["limits"]
        keep = np.zeros((ny, nx), bool)
        keep[ilo:ihi + 1, jlo:jhi + 1] = True
        assert np.array_equal(~keep, c["lim"])
        pmin, pmax, qmin, qmax = d["bbox"]
        nz = np.nonzero(c["W"])
        if nz[0].size:
            assert nz[0].min() - ny // 2 >= pmin and nz[0].max() - ny // 2 <= pmax
            assert nz[1].min() - nx // 2 >= qmin and nz[1].max() - nx // 2 <= qmax
        # the box stays on the template grid
        assert -(ny // 2) <= pmin and pmax <= ny - 1 - ny // 2
        assert -(nx // 2) <= qmin and qmax <= nx - 1 - nx // 2


def test_ricker_support_is_bounded_by_exp_underflow():
    t = WT.Ricker(10, 0.1, 0.3, 512, 512, 1.0)
    W = t.template()
    assert np.count_nonzero(W) == 3476          # SURVEY.md section 7 [probed]
    pmin, pmax, qmin, qmax = t._device_descriptor()["bbox"]
    nz = np.nonzero(W)
    assert nz[1].min() - 256 >= qmin and nz[1].max() - 256 <= qmax
    assert qmax - qmin < 200                    # not the whole 512-wide band
    assert np.exp(-WT.EXP_UNDERFLOW) == 0.0 and np.exp(-np.nextafter(WT.EXP_UNDERFLOW, 0)) > 0


def test_shifted_template_quirks():
    W = np.arange(20.).reshape(4, 5) + 1
    t = WT.ShiftedLeftFacingUpperBreakScarp(5, 2., 0.1, 5, 4, 1., dx=1, dy=1)
    out = t.shift_template(W, 2, 1)
    assert np.array_equal(out[:3, 2:], W[:3, :3]) and not out[3].any() and not out[:, :2].any()
    out = t.shift_template(W, -1, -1)
    assert np.array_equal(out[1:, :4], W[1:, 1:]) and not out[0].any() and not out[:, 4].any()


SHIFTED = {"shifted_left": WT.ShiftedLeftFacingUpperBreakScarp,
           "shifted_right": WT.ShiftedRightFacingUpperBreakScarp}


def test_shifted_templates_match_reference_captures():
    """WindowedTemplate.py:307-431: template(), window limits and error mask of the
    Shifted classes against arrays captured from the reference's own objects."""
    for c in load_cases("ref_shifted.npz"):
        ny, nx = c["z"].shape
        t = SHIFTED[str(c["name"])](float(c["scale"]), float(c["age"]), float(c["ang"]), nx, ny,
                                    float(c["de"]), dx=int(c["sdx"]), dy=int(c["sdy"]))
        W = t.template()
        assert np.allclose(W, c["W"], rtol=1e-13, atol=0)
        assert np.array_equal(W != 0, c["W"] != 0)
        assert np.array_equal(t.get_window_limits(), c["lim"])
        assert np.array_equal(t.get_err_mask(), c["err"])


def test_grid_descriptors_equal_the_per_template_descriptors():
    """WindowedTemplate.grid_descriptors (arrays for a whole (angle, parameter) grid) against
    _device_descriptor() of every single template: every field equal, bit for bit."""
    from scarplet_amd import _plan
    angs = _plan.angle_grid()
    cases = [(WT.Scarp, 100, _plan.age_grid()[::3], angs[::4], 1000, 901, 1.0),
             (WT.Scarp, 30.5, _plan.age_grid()[::5], angs[::7], 333, 400, 2.0),
             (WT.RightFacingUpperBreakScarp, 20, [3.0, 40.0], angs[::20], 128, 90, 0.5),
             (WT.LeftFacingUpperBreakScarp, 12, [1.0, 7.5], angs[::25], 90, 128, 1.0),
             (WT.Channel, 10, [0.05, 0.1, 0.3, 1.0], angs[::9], 512, 512, 1.0),
             (WT.Ricker, 8, [0.0, 0.2], angs[::30], 63, 64, 1.5)]
    for (cls, scale, params, angles, nx, ny, de) in cases:
        g = WT.grid_descriptors(cls, scale, params, angles, nx, ny, de)
        assert g is not None
        for ib, a in enumerate(angles):
            for ia, p in enumerate(params):
                d = cls(scale, p, a, nx, ny, de)._device_descriptor()
                one = dict(cos_a=d["cos_a"], sin_a=d["sin_a"], c=d["c"], d=d["d"], p0=d["p0"], p1=d["p1"],
                           ilo=d["limits"][0], ihi=d["limits"][1], jlo=d["limits"][2], jhi=d["limits"][3],
                           pmin=d["bbox"][0], pmax=d["bbox"][1], qmin=d["bbox"][2], qmax=d["bbox"][3])
                for k, v in one.items():
                    assert g[k][ib, ia] == v, (cls.__name__, k, ib, ia)
                assert g["kind"] == d["kind"] and g["flags"] == d["flags"]

    class Mine(WT.Scarp):               # a subclass may override anything: no shortcut
        pass
    assert WT.grid_descriptors(Mine, 10, [1.0], [0.1], 64, 64, 1.0) is None


def test_describe_fast_path_fills_the_same_struct_array(monkeypatch):
    """core.Matcher.describe through grid_descriptors and through one object per template:
    the same bytes in the sc_template array, the same support box and tap count."""
    import ctypes
    import types
    from scarplet_amd import _plan, core
    fake = types.SimpleNamespace(nx=700, ny=520, de=2.0)
    fake._describe_grid = lambda *a: core.Matcher._describe_grid(fake, *a)
    params, angles = _plan.age_grid()[::6], _plan.angle_grid()[::11]
    for cls, scale, pars in [(WT.Scarp, 60, params), (WT.LeftFacingUpperBreakScarp, 40, params[:3]),
                             (WT.Channel, 25, [0.02, 0.05])]:
        for id_of in (None, lambda ia, ib: ib * len(pars) + ia + 7):
            fast = core.Matcher.describe(fake, cls, scale, np.asarray(pars, float), angles, id_base=3, id_of=id_of)
            with monkeypatch.context() as m:
                m.setattr(WT, "grid_descriptors", lambda *a, **k: None)
                slow = core.Matcher.describe(fake, cls, scale, np.asarray(pars, float), angles, id_base=3, id_of=id_of)
            assert fast[1:] == slow[1:], (fast[1:], slow[1:])
            assert ctypes.string_at(fast[0], ctypes.sizeof(fast[0])) == ctypes.string_at(slow[0], ctypes.sizeof(slow[0]))


@pytest.mark.parametrize("cls", ["Scarp", "RightFacingUpperBreakScarp", "LeftFacingUpperBreakScarp", "Ricker"])
def test_nan_dem_fold_without_template_objects(cls):
    """A DEM with NaNs gets the reference's degenerate maps (core.py:228-240, 348-375).  For the
    built-in classes they come from the window-limit rectangles and the error half-planes
    directly (Matcher._nan_fold_builtin) - the same maps as folding every template's numpy
    masks one by one, without 2 x (ny, nx) temporaries per (age, orientation)."""
    import scarplet_amd as sl
    from scarplet_amd.core import Matcher
    m = object.__new__(Matcher)                     # no device: only the host-side fold is exercised
    m.ny, m.nx, m.de = 61, 84, 2.0
    T = getattr(sl, cls)
    params = [0.05, 0.1] if cls == "Ricker" else [1.0, 10.0, 100.0]
    angles = np.linspace(-np.pi / 2, np.pi / 2, 7)
    fast = m._nan_fold_builtin(T, 12, params, angles)
    assert fast is not None
    amp = np.zeros((m.ny, m.nx))
    snr = np.zeros((m.ny, m.nx))
    for ang in angles:
        for par in params:
            a, s = m._nan_maps(T(12, par, ang, m.nx, m.ny, m.de))
            amp[np.isnan(a)] = np.nan
            snr[np.isnan(s)] = np.nan
    assert np.array_equal(fast[0], amp, equal_nan=True) and np.array_equal(fast[3], snr, equal_nan=True)
    assert not fast[1].any() and not fast[2].any()


# ---- the reference's own built-in classes handed to match() (builtin_twin) ---------------------
def _standin_module(edit=None):
    """A module called ``<pkg>.WindowedTemplate`` with plugin classes written the way a user's
    own copy would be - meshgrid coordinates, masks as full arrays - NOT this package's classes
    and not the reference's text: what builtin_twin has to recognise by behaviour.  ``edit``
    changes one class so that it no longer behaves like the built-in of its name."""
    import types
    from scipy.special import erfinv
    mod = types.ModuleType("userpkg.WindowedTemplate")

    class Base(object):
        def _xy(self):
            gx = self.de * np.linspace(1, self.nx, num=self.nx)
            gy = self.de * np.linspace(1, self.ny, num=self.ny)
            return gx - np.mean(gx), gy - np.mean(gy)

        def get_coordinates(self):
            gx, gy = self._xy()
            X, Y = np.meshgrid(gx, gy)
            return X * np.cos(self.alpha) + Y * np.sin(self.alpha), -X * np.sin(self.alpha) + Y * np.cos(self.alpha)

        def get_mask(self):
            xr, yr = self.get_coordinates()
            return (abs(xr) < self.c) & (abs(yr) < self.d)

        def get_window_limits(self):
            a, q = self.alpha, self.alpha - np.pi / 2
            an_y = abs((self.d * np.cos(q) - self.d * np.cos(a)) + 2 * self.c * np.cos(q))
            an_x = abs((self.d * np.sin(a) - self.d * np.sin(q)) + 2 * self.c * np.sin(q))
            gx, gy = self._xy()
            X, Y = np.meshgrid(gx, gy)
            return (X < (min(gx) + an_x)) | (X > (max(gx) - an_x)) | (Y < (min(gy) + an_y)) | (Y > (max(gy) - an_y))

    class Scarp(Base):
        def __init__(self, d, kt, alpha, nx, ny, de):
            self.d, self.kt, self.alpha, self.nx, self.ny, self.de = d, kt, -alpha, nx, ny, de
            self.c = abs(2 * np.sqrt(kt) * erfinv(0.9))

        def template(self):
            xr, _ = self.get_coordinates()
            W = (-xr / (2. * self.kt ** (3 / 2.) * np.sqrt(np.pi))) * np.exp(-xr ** 2. / (4. * self.kt))
            if edit == "scarp_value":
                W = W * 1.0001
            return W * self.get_mask()

    class LeftFacingUpperBreakScarp(Scarp):
        def get_err_mask(self):
            xr, _ = self.get_coordinates()
            return (xr > 0) if edit == "err_mask" else (xr >= 0)

    class RightFacingUpperBreakScarp(Scarp):
        def template(self):
            return -Scarp.template(self)

        def get_err_mask(self):
            xr, _ = self.get_coordinates()
            return xr <= 0

    class Ricker(Base):
        def __init__(self, d, f, alpha, nx, ny, de):
            self.d, self.f, self.alpha, self.nx, self.ny, self.de = d, f, -alpha, nx, ny, de
            self.c = nx

        def get_window_limits(self):
            return np.zeros((self.ny, self.nx), dtype=bool)

        def template(self):
            xr, _ = self.get_coordinates()
            u = (np.pi * self.f * xr) ** 2.
            return (1. - 2. * u) * np.exp(-u) * self.get_mask()

    class Channel(Ricker):
        pass

    class Crater(Base):                       # a name this package has no device form for
        def __init__(self, d, kt, alpha, nx, ny, de):
            self.d, self.alpha, self.nx, self.ny, self.de, self.c = d, -alpha, nx, ny, de, 1.0

        def template(self):
            return np.zeros((self.ny, self.nx))

    for cls in (Scarp, LeftFacingUpperBreakScarp, RightFacingUpperBreakScarp, Ricker, Channel, Crater):
        cls.__module__ = mod.__name__
        setattr(mod, cls.__name__, cls)
    return mod


def test_foreign_builtin_classes_are_recognised_by_behaviour():
    """A script written against the reference hands ITS Scarp / Ricker / ... to match()
    (WT.py:87-215, 434-525).  Such a class gets the device descriptors of this package's class of
    that name when it behaves like it; an edited copy, a subclass or an unknown name does not."""
    mod = _standin_module()
    for name in ("Scarp", "LeftFacingUpperBreakScarp", "RightFacingUpperBreakScarp", "Ricker", "Channel"):
        assert WT.builtin_twin(getattr(mod, name)) is getattr(WT, name), name
    assert WT.builtin_twin(mod.Crater) is None
    # the same descriptors as for this package's class, for a whole grid
    ages, angles = [1.0, 12.5, 300.0], np.linspace(-np.pi / 2, np.pi / 2, 7)
    a = WT.grid_descriptors(mod.Scarp, 20, ages, angles, 120, 90, 2.0)
    b = WT.grid_descriptors(WT.Scarp, 20, ages, angles, 120, 90, 2.0)
    assert a is not None and all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in b)
    # own classes: themselves; a subclass of an own class may override anything: generic
    assert WT.builtin_twin(WT.Scarp) is WT.Scarp

    class Mine(WT.Scarp):
        pass
    assert WT.builtin_twin(Mine) is None and WT.grid_descriptors(Mine, 20, ages, angles, 120, 90, 2.0) is None
    # edited copies keep the name but not the behaviour
    assert WT.builtin_twin(_standin_module("scarp_value").Scarp) is None
    assert WT.builtin_twin(_standin_module("err_mask").LeftFacingUpperBreakScarp) is None
    # the right name in a module of another name is not trusted either
    other = _standin_module()
    other.Scarp.__module__ = "userpkg.templates"
    assert WT.builtin_twin(other.Scarp) is None


def test_describe_uses_the_device_descriptors_for_foreign_builtins(monkeypatch):
    """Matcher.describe() with the stand-in Scarp fills the struct array the built-in fills - no
    template object, no full-grid template(): the stand-in's template() must not be called for
    the search grid."""
    from scarplet_amd import core, _lib
    mod = _standin_module()
    assert WT.builtin_twin(mod.Scarp) is WT.Scarp          # (probed on the small grids here)
    calls = []
    monkeypatch.setattr(mod.Scarp, "template", lambda self: calls.append(1) or np.zeros((self.ny, self.nx)))
    m = core.Matcher.__new__(core.Matcher)
    m.nx, m.ny, m.de = 400, 300, 1.0
    ages, angles = [1.0, 10.0, 100.0], np.linspace(-0.5, 0.5, 9)
    arr_f, bbox_f, area_f = m.describe(mod.Scarp, 30, ages, angles)
    arr_o, bbox_o, area_o = m.describe(WT.Scarp, 30, ages, angles)
    assert not calls
    assert bytes(arr_f) == bytes(arr_o) and bbox_f == bbox_o and area_f == area_o
    assert all(t.kind == WT.KIND_SCARP and t.window == -1 for t in arr_f)


def test_generic_plugins_say_what_they_will_cost():
    from scarplet_amd import core
    m = core.Matcher.__new__(core.Matcher)
    m.nx, m.ny = 10000, 10000
    with pytest.warns(UserWarning, match="evaluated on the host"):
        m._warn_generic_cost(2.5, 6335)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m._warn_generic_cost(0.001, 35)
