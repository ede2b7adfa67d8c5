from . import common, transientpath, transientnlospath
from .common import TransientADIntegrator
from .transientpath import TransientPath
from .transientnlospath import TransientNLOSPath

transientpath.register()
transientnlospath.register()

__all__ = ["TransientADIntegrator", "TransientPath", "TransientNLOSPath", "common", "transientpath", "transientnlospath"]
