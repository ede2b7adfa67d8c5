"""Counterpart of mitransient/utils.py: ``cornell_box()``, ``speed_of_light``, ``β_init``."""
from __future__ import annotations

from .transform import ScalarTransform4f

speed_of_light = 299792458.0
"""Speed of light in meters/second (mitransient/utils.py:5)."""


def β_init(sensor=None, ray=None):
    """Initial path throughput: 1 for the unpolarized RGB variant (mitransient/utils.py:9-21)."""
    return (1.0, 1.0, 1.0)


def indent(obj, amount=2):
    output = str(obj)
    lines = output.splitlines(keepends=True)
    if len(lines) == 1:
        return lines[0]
    return "".join(line + " " * amount for line in lines)


def cornell_box():
    """Dictionary description of the Cornell box for transient rendering; same objects, ids and values as the
    reference helper (mitransient/utils.py:78-220), assembled here from small tables."""
    T = ScalarTransform4f
    X, Y = [1, 0, 0], [0, 1, 0]

    def rgb(*v):
        return dict(type="rgb", value=list(v))

    def ref(name):
        return dict(type="ref", id=name)

    def place(at, axis=None, deg=0.0, scale=None):
        t = T().translate(at)
        if axis is not None:
            t = t.rotate(axis, deg)
        return t if scale is None else t.scale(scale)

    scene = dict(type="scene")
    scene["integrator"] = dict(type="transient_path", camera_unwarp=False, max_depth=8,
                               temporal_filter="box", gaussian_stddev=2.0)
    film = dict(type="transient_hdr_film", width=256, height=256, rfilter=dict(type="box"),
                temporal_bins=300, start_opl=3.5, bin_width_opl=0.02)
    scene["sensor"] = dict(type="perspective", fov_axis="smaller", near_clip=0.001, far_clip=100.0,
                           focus_distance=1000, fov=39.3077,
                           to_world=T().look_at(origin=[0, 0, 3.90], target=[0, 0, 0], up=Y),
                           sampler=dict(type="independent", sample_count=256), film=film)

    albedo = {"white": (0.885809, 0.698859, 0.666422),
              "green": (0.105421, 0.37798, 0.076425),
              "red": (0.570068, 0.0430135, 0.0443706)}
    for name, value in albedo.items():
        scene[name] = dict(type="diffuse", reflectance=rgb(*value))

    scene["light"] = dict(type="rectangle", to_world=place([0.0, 0.99, 0.01], X, 90, [0.23, 0.19, 0.19]),
                          bsdf=ref("white"), emitter=dict(type="area", radiance=rgb(18.387, 13.9873, 6.75357)))

    walls = [("floor", [0.0, -1.0, 0.0], X, -90, "white"),
             ("ceiling", [0.0, 1.0, 0.0], X, 90, "white"),
             ("back", [0.0, 0.0, -1.0], None, 0, "white"),
             ("green-wall", [1.0, 0.0, 0.0], Y, -90, "green"),
             ("red-wall", [-1.0, 0.0, 0.0], Y, 90, "red")]
    for name, at, axis, deg, material in walls:
        scene[name] = dict(type="rectangle", to_world=place(at, axis, deg), bsdf=ref(material))

    boxes = [("small-box", [0.335, -0.7, 0.38], -17, 0.3),
             ("large-box", [-0.33, -0.4, -0.28], 18.25, [0.3, 0.61, 0.3])]
    for name, at, deg, scale in boxes:
        scene[name] = dict(type="cube", to_world=place(at, Y, deg, scale), bsdf=ref("white"))
    return scene
