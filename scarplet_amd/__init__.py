"""scarplet_amd: MI355X-native template matching for digital elevation models.

Drop-in for the hot path of stgl/scarplet: ``match``, ``match_template``,
``compare``, ``calculate_best_fit_parameters`` and the ``WindowedTemplate``
plugin classes, with the work done by hand-written HIP kernels reached through
a C-ABI shared library (include/scarplet_hip.h).
"""

from scarplet_amd.core import (match, match_scales, match_template, compare, load,  # noqa: F401
                               calculate_best_fit_parameters,
                               calculate_best_fit_parameters_serial, Matcher)
from scarplet_amd import WindowedTemplate, dem  # noqa: F401
from scarplet_amd.WindowedTemplate import (Scarp, Ricker, Channel,  # noqa: F401
                                           RightFacingUpperBreakScarp,
                                           LeftFacingUpperBreakScarp)
from scarplet_amd.dem import DEMGrid  # noqa: F401
from scarplet_amd.plotting import plot_results, Hillshade, hillshade  # noqa: F401
from scarplet_amd._hostpool import release as release_host_buffers  # noqa: F401
