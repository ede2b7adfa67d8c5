from .transient_hdr_film import TransientHDRFilm
from ..plugins import register_film

register_film("transient_hdr_film", lambda props: TransientHDRFilm(props))

__all__ = ["TransientHDRFilm"]
