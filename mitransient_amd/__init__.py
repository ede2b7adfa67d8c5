"""mitransient_amd — MI355X-native drop-in for mitransient's ``transient_path`` +
``transient_hdr_film`` hot path (see DESIGN.md / INTEGRATION.md).

    import mitransient_amd.mi as mi; mi.set_variant('llvm_ad_rgb')
    import mitransient_amd as mitr
    scene = mi.load_dict(mitr.cornell_box())
    steady, transient = mi.render(scene, spp=1024)
"""
from .version import __version__                     # noqa: F401
from .utils import speed_of_light, cornell_box       # noqa: F401
from . import mi, vis, nlos                          # noqa: F401
from . import integrators, films, render, sensors    # noqa: F401
from .integrators import TransientADIntegrator, TransientPath, TransientNLOSPath   # noqa: F401
from .films import TransientHDRFilm, PhasorHDRFilm   # noqa: F401
from .render import TransientImageBlock              # noqa: F401
from .mi import load_dict                            # noqa: F401
from .mi import render as render_scene               # noqa: F401
