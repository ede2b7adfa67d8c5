"""numpy post-processing helpers with the reference's names (unpolarized_visualization.py)."""
from __future__ import annotations

import numpy as np


def tonemap_transient(transient, scaling=1.0):
    """Linear tonemap by the 99th percentile of |transient| (unpolarized_visualization.py:14-18)."""
    tnp = np.array(transient)
    channel_top = np.quantile(np.abs(tnp), 0.99)
    return tnp / channel_top * scaling
