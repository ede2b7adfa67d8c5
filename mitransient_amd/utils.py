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
    '''
    Returns a dictionary containing a description of the Cornell Box scene for Transient Rendering.
    (Values of mitransient/utils.py:78-220.)
    '''
    T = ScalarTransform4f
    white = {'type': 'ref', 'id': 'white'}

    def rgb(r, g, b):
        return {'type': 'rgb', 'value': [r, g, b]}

    def wall(to_world, bsdf_id):
        return {'type': 'rectangle', 'to_world': to_world, 'bsdf': {'type': 'ref', 'id': bsdf_id}}

    return {
        'type': 'scene',
        'integrator': {
            'type': 'transient_path',
            'camera_unwarp': False,
            'max_depth': 8,
            'temporal_filter': 'box',
            'gaussian_stddev': 2.0,
        },
        'sensor': {
            'type': 'perspective',
            'fov_axis': 'smaller',
            'near_clip': 0.001,
            'far_clip': 100.0,
            'focus_distance': 1000,
            'fov': 39.3077,
            'to_world': T().look_at(origin=[0, 0, 3.90], target=[0, 0, 0], up=[0, 1, 0]),
            'sampler': {'type': 'independent', 'sample_count': 256},
            'film': {
                'type': 'transient_hdr_film',
                'width': 256,
                'height': 256,
                'rfilter': {'type': 'box'},
                'temporal_bins': 300,
                'start_opl': 3.5,
                'bin_width_opl': 0.02,
            },
        },
        'white': {'type': 'diffuse', 'reflectance': rgb(0.885809, 0.698859, 0.666422)},
        'green': {'type': 'diffuse', 'reflectance': rgb(0.105421, 0.37798, 0.076425)},
        'red': {'type': 'diffuse', 'reflectance': rgb(0.570068, 0.0430135, 0.0443706)},
        'light': {
            'type': 'rectangle',
            'to_world': T().translate([0.0, 0.99, 0.01]).rotate([1, 0, 0], 90).scale([0.23, 0.19, 0.19]),
            'bsdf': white,
            'emitter': {'type': 'area', 'radiance': rgb(18.387, 13.9873, 6.75357)},
        },
        'floor': wall(T().translate([0.0, -1.0, 0.0]).rotate([1, 0, 0], -90), 'white'),
        'ceiling': wall(T().translate([0.0, 1.0, 0.0]).rotate([1, 0, 0], 90), 'white'),
        'back': wall(T().translate([0.0, 0.0, -1.0]), 'white'),
        'green-wall': wall(T().translate([1.0, 0.0, 0.0]).rotate([0, 1, 0], -90), 'green'),
        'red-wall': wall(T().translate([-1.0, 0.0, 0.0]).rotate([0, 1, 0], 90), 'red'),
        'small-box': {
            'type': 'cube',
            'to_world': T().translate([0.335, -0.7, 0.38]).rotate([0, 1, 0], -17).scale(0.3),
            'bsdf': white,
        },
        'large-box': {
            'type': 'cube',
            'to_world': T().translate([-0.33, -0.4, -0.28]).rotate([0, 1, 0], 18.25).scale([0.3, 0.61, 0.3]),
            'bsdf': white,
        },
    }
