"""Downstream consumers of a search result: the
hillshade and the four-panel result plot of the reference.  Host side only;
matplotlib is imported when a plot is asked for."""

import numpy as np


def hillshade(data, az=315, elev=45):
    """Shaded relief of a DEMGrid the way the reference computes it
    (core.py:402-406, dem.py:455-458): matplotlib's LightSource with
    vert_exag = 1 and the grid's dx, dy.  Returns an (ny, nx) array in [0, 1]."""
    import matplotlib.colors
    ls = matplotlib.colors.LightSource(azdeg=az, altdeg=elev)
    gi = data._georef_info
    return ls.hillshade(np.asarray(data._griddata, dtype=float), vert_exag=1, dx=gi.dx, dy=gi.dy)


class Hillshade(object):
    """Hillshade of a DEM (dem.py:433-460)."""

    def __init__(self, dem):
        self._georef_info = dem._georef_info
        self._griddata = dem._griddata
        self._hillshade = None

    def plot(self, az=315, elev=45):
        import matplotlib.pyplot as plt
        ax = plt.gca()
        self._hillshade = hillshade(self, az, elev)
        ax.imshow(self._hillshade, alpha=1, cmap='gray', origin='lower')
        return ax


# One entry per plane of a search result, in the order ``match`` returns them.  The reference
# draws the same four quantities (core.py:380-420); its orientation panel shows the plane as
# returned (radians) under a degree label, which is kept so that figures compare.
_PANELS = (
    dict(plane=0, title="Amplitude [m]", cmap="Reds"),
    dict(plane=1, title="Relative age [m$^2$]", cmap="viridis"),
    dict(plane=2, title="Orientation [deg.]", cmap="RdBu_r"),
    dict(plane=3, title="Signal-to-noise ratio", cmap="Reds"),
)


def _draw_panel(fig, axis, shade, values, title, cmap, opacity=0.5, ticks=3):
    """One result plane, half transparent, over the grey relief, with a short horizontal
    colour bar that carries the quantity's name."""
    from matplotlib.ticker import MaxNLocator
    axis.imshow(shade, cmap="gray", alpha=1)
    mappable = axis.imshow(values, cmap=cmap, alpha=opacity)
    bar = fig.colorbar(mappable, ax=axis, orientation="horizontal", shrink=0.5, label=title)
    bar.locator = MaxNLocator(nbins=ticks)
    bar.update_ticks()
    return bar


def plot_results(data, results, az=315, elev=45, figsize=(4, 16)):
    """The four maps of a search result over the DEM's hillshade - the figure the reference's
    ``plot_results`` makes (core.py:380-420).  ``results``: the (4, ny, nx) array or the
    4-tuple ``match`` returns.  Returns the figure (2 x 2 panels, a colour bar each)."""
    import matplotlib.pyplot as plt
    planes = [np.asarray(p) for p in results]
    if len(planes) != len(_PANELS):
        raise ValueError("a search result has %d planes: amplitude, age, orientation, SNR" % len(_PANELS))
    shade = hillshade(data, az, elev)
    fig = plt.figure(figsize=figsize)
    for k, spec in enumerate(_PANELS):
        _draw_panel(fig, fig.add_subplot(2, 2, k + 1), shade, planes[spec["plane"]], spec["title"], spec["cmap"])
    return fig
