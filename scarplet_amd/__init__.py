"""scarplet_amd: MI355X-native template matching for digital elevation models.

Drop-in for the hot path of stgl/scarplet: ``match``, ``match_template``,
``compare``, ``calculate_best_fit_parameters`` and the ``WindowedTemplate``
plugin classes, with the work done by hand-written HIP kernels reached through
a C-ABI shared library (include/scarplet_hip.h).
"""
