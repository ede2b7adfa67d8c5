"""The active Mitsuba-style variant string (``mi.set_variant``).  Kept in its own module so that ``scene.py`` (colour
parsing) and ``mi.py`` can both see it."""
_variant = None


def get():
    return _variant


def set(name):
    global _variant
    _variant = name


def is_monochromatic() -> bool:
    """``*_mono`` variants: every colour is replaced by its luminance at load time and the path arithmetic runs on three
    identical channels — bit for bit what a one-channel implementation computes, since the channels never mix (Russian
    roulette takes the max of equal values)."""
    return bool(_variant) and _variant.endswith("_mono")
