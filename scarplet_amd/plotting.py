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


def plot_results(data, results, az=315, elev=45, figsize=(4, 16)):
    """Maps of a search result over the hillshade (core.py:380-420): amplitude,
    relative age, orientation (as returned, radians) and signal-to-noise ratio.
    ``results``: the (4, ny, nx) array or 4-tuple ``match`` returns.  Returns the
    figure."""
    import matplotlib
    import matplotlib.pyplot as plt
    import matplotlib.ticker
    fig, ax = plt.subplots(2, 2, figsize=figsize)
    ax = ax.ravel()
    shade = hillshade(data, az, elev)
    labels = ['Amplitude [m]', 'Relative age [m$^2$]',
              'Orientation [deg.]', 'Signal-to-noise ratio']
    cmaps = ['Reds', 'viridis', 'RdBu_r', 'Reds']
    for i, (axis, label, cmap) in enumerate(zip(ax, labels, cmaps)):
        axis.imshow(shade, alpha=1, cmap='gray')
        im = axis.imshow(np.asarray(results[i]), alpha=0.5, cmap=cmap)
        cb = plt.colorbar(im, ax=axis, shrink=0.5, orientation='horizontal', label=label)
        cb.locator = matplotlib.ticker.MaxNLocator(nbins=3)
        cb.update_ticks()
    return fig
