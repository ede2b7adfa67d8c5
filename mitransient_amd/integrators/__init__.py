from . import common, transientpath
from .common import TransientADIntegrator
from .transientpath import TransientPath

transientpath.register()

__all__ = ["TransientADIntegrator", "TransientPath", "common", "transientpath"]
