"""Host-side template plugins (scarplet_amd/WindowedTemplate.py) against the
reference's classes (fixtures captured by oracle/gen_golden.py)."""
import numpy as np

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
