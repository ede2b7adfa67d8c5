"""``mi.load_file``: Mitsuba 3 XML scene descriptions -> the dictionary form ``mi.load_dict`` takes.

The reference's example scenes are XML files (examples/transient/cornell-box/cbox_diffuse.xml,
examples/transient-nlos/nlos_Z.xml, examples/diff-transient/staircase/scene.xml = BASELINE config 5), loaded with
``mi.load_file(path, spp=..., res=...)``.  This module implements the part of the format those files use
[mitsuba3: src/core/xml.cpp, docs "Scene XML file format"]:

* ``<default name value>`` and ``$name`` substitution, overridable by keyword arguments of ``load_file``;
* property tags ``integer float string boolean rgb spectrum point vector transform`` (with
  ``translate rotate scale lookat matrix`` children, composed left to right = applied in document order);
* plugin tags ``scene integrator sensor sampler film rfilter shape bsdf emitter texture`` with ``type``, ``id``, ``name``;
* ``<ref id name>`` (ids are global, whatever the nesting depth) and ``<include filename>``.

The result is a plain dictionary; plugin availability is decided by ``load_dict`` (unknown plugins raise there).
"""
from __future__ import annotations

import os
import re
import xml.etree.ElementTree as ET
from typing import Any, Dict

import numpy as np

from .transform import ScalarTransform4f

_PLUGIN_TAGS = {"scene", "integrator", "sensor", "sampler", "film", "rfilter", "shape", "bsdf", "emitter", "texture",
                "medium", "phase", "volume"}
_VAR = re.compile(r"\$(\w+)|\$\{(\w+)\}")


class _Ctx:
    def __init__(self, params: Dict[str, str], base_dir: str):
        self.params = params
        self.base_dir = base_dir
        self.ids: Dict[str, Any] = {}
        self.id_tags: Dict[str, str] = {}
        self.counter = 0

    def sub(self, s: str) -> str:
        def rep(m):
            k = m.group(1) or m.group(2)
            if k not in self.params:
                raise ValueError(f"XML: undefined parameter '${k}'")
            return str(self.params[k])
        return _VAR.sub(rep, s) if "$" in s else s


def _floats(s: str):
    return [float(x) for x in re.split(r"[\s,]+", s.strip()) if x]


def _vec3(el, ctx, default=0.0):
    if "value" in el.attrib:
        v = _floats(ctx.sub(el.attrib["value"]))
        if len(v) == 1:
            v = v * 3
        if len(v) != 3:
            raise ValueError(f"XML: <{el.tag}> expects 1 or 3 values")
        return v
    return [float(ctx.sub(el.attrib.get(a, str(default)))) for a in "xyz"]


def _transform(el, ctx) -> ScalarTransform4f:
    m = np.eye(4)
    for op in el:
        if op.tag == "translate":
            t = ScalarTransform4f().translate(_vec3(op, ctx)).matrix
        elif op.tag == "scale":
            t = ScalarTransform4f().scale(_vec3(op, ctx, 1.0)).matrix
        elif op.tag == "rotate":
            axis = _vec3(op, ctx)
            t = ScalarTransform4f().rotate(axis, float(ctx.sub(op.attrib["angle"]))).matrix
        elif op.tag == "lookat":
            t = ScalarTransform4f().look_at(_floats(ctx.sub(op.attrib["origin"])), _floats(ctx.sub(op.attrib["target"])),
                                            _floats(ctx.sub(op.attrib.get("up", "0, 1, 0")))).matrix
        elif op.tag == "matrix":
            v = _floats(ctx.sub(op.attrib["value"]))
            if len(v) == 16:
                t = np.asarray(v, dtype=np.float64).reshape(4, 4)
            elif len(v) == 9:
                t = np.eye(4)
                t[:3, :3] = np.asarray(v, dtype=np.float64).reshape(3, 3)
            else:
                raise ValueError("XML: <matrix> expects 9 or 16 values")
        else:
            raise ValueError(f"XML: unknown transform operation <{op.tag}>")
        m = t @ m                                  # later operations are applied after earlier ones
    return ScalarTransform4f(m)


def _bool(s: str) -> bool:
    v = s.strip().lower()                      # the reference's own nlos-z-simple.xml writes "False"
    if v not in ("true", "false"):
        raise ValueError(f"XML: boolean value must be 'true' or 'false', got '{s}'")
    return v == "true"


def _property(el, ctx):
    t = el.tag
    if t == "integer":
        return int(ctx.sub(el.attrib["value"]))
    if t == "float":
        return float(ctx.sub(el.attrib["value"]))
    if t == "string":
        return ctx.sub(el.attrib["value"])
    if t == "boolean":
        return _bool(ctx.sub(el.attrib["value"]))
    if t == "rgb":
        v = _floats(ctx.sub(el.attrib["value"]))
        return {"type": "rgb", "value": v * 3 if len(v) == 1 else v}
    if t == "spectrum":
        v = _floats(ctx.sub(el.attrib["value"]))
        if len(v) != 1:
            raise ValueError("XML: only uniform <spectrum value=\"x\"/> is supported (RGB variants)")
        return {"type": "spectrum", "value": v[0]}
    if t in ("point", "vector"):
        return _vec3(el, ctx)
    if t == "transform":
        return _transform(el, ctx)
    raise ValueError(f"XML: unknown tag <{t}>")


def _plugin(el, ctx) -> Dict[str, Any]:
    d: Dict[str, Any] = {}
    if el.tag != "scene":
        if "type" not in el.attrib:
            raise ValueError(f"XML: <{el.tag}> needs a type")
        d["type"] = ctx.sub(el.attrib["type"])
    else:
        d["type"] = "scene"
    if "id" in el.attrib:
        ctx.ids[el.attrib["id"]] = d               # ids are global; registered before the children are parsed
        ctx.id_tags[el.attrib["id"]] = el.tag
    for ch in el:
        tag = ch.tag
        if tag in ("default", "alias", "path"):
            if tag == "default":
                ctx.params.setdefault(ch.attrib["name"], ctx.sub(ch.attrib["value"]))
            continue
        if tag == "include":
            sub = _parse_file(os.path.join(ctx.base_dir, ctx.sub(ch.attrib["filename"])), ctx)
            for k, v in sub.items():
                if k != "type":
                    d[k] = v
            continue
        name = ch.attrib.get("name")
        if tag == "ref":
            rid = ch.attrib["id"]
            if rid not in ctx.ids:
                raise ValueError(f"XML: reference to unknown id '{rid}'")
            val = ctx.ids[rid]                     # the SAME dictionary object: load_dict dedups materials by identity
            tag = ctx.id_tags[rid]
        elif tag in _PLUGIN_TAGS:
            val = _plugin(ch, ctx)
            if name is None and el.tag == "scene":
                name = ch.attrib.get("id")
        else:
            val = _property(ch, ctx)
            if name is None:
                raise ValueError(f"XML: <{tag}> needs a name")
        if name is None:                           # the conventional key of the dictionary form: the tag itself
            name = tag
            while name in d:
                name = f"{tag}_{ctx.counter}"
                ctx.counter += 1
        if name in d:
            raise ValueError(f"XML: duplicate property '{name}' in <{el.tag}>")
        d[name] = val
    return d


def _parse_file(path: str, ctx: _Ctx) -> Dict[str, Any]:
    root = ET.parse(path).getroot()
    if root.tag != "scene":
        raise ValueError(f"XML: root element must be <scene>, got <{root.tag}>")
    return _plugin(root, ctx)


def xml_to_dict(path: str, **params) -> Dict[str, Any]:
    """parse a Mitsuba XML scene into the dictionary form; keyword arguments override ``<default>`` values"""
    path = os.fspath(path)
    ctx = _Ctx({k: (str(v).lower() if isinstance(v, bool) else str(v)) for k, v in params.items()},
               os.path.dirname(os.path.abspath(path)))
    return _parse_file(path, ctx)


def load_file(path, approximate_materials: bool = False, **params):
    """``mi.load_file(path, **params)``.  ``approximate_materials=True`` is an extension: BSDF / texture plugins
    outside the hot path's material model are mapped to the nearest supported one instead of raising (see
    ``scene.py``: roughconductor -> conductor, roughplastic -> diffuse, bumpmap -> its nested BSDF, bitmap -> mean
    colour); the render is then NOT comparable with the reference's for those surfaces."""
    from . import mi
    d = xml_to_dict(path, **params)
    return mi.load_dict(d, base_dir=os.path.dirname(os.path.abspath(os.fspath(path))),
                        approximate_materials=approximate_materials)
