from .transient_hdr_film import TransientHDRFilm
from .phasor_hdr_film import PhasorHDRFilm
from ..plugins import register_film

register_film("transient_hdr_film", lambda props: TransientHDRFilm(props))
register_film("phasor_hdr_film", lambda props: PhasorHDRFilm(props))

__all__ = ["TransientHDRFilm", "PhasorHDRFilm"]
