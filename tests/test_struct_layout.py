"""The structs of include/orbx.h against their ctypes / numpy mirrors in the Python host side: a C program compiled from the header prints
sizeof and every field's offset and size; each mirror has to have the same fields, in the same order, at the same offsets.  (A field added to
the header and not to a mirror shifts everything behind it silently - the calls would still "work" on whatever bytes they find.)"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from orb_slam3_detailed_comments_amd import _lib, extractor, matcher, views

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MIRRORS = {
    "OrbmKB8Stereo": matcher.KB8Stereo, "OrbmFrameView": views.FrameView, "OrbmMapPointView": views.MapPointView, "OrbmLastFrameView": views.LastFrameView,
    "OrbmFrustumView": matcher._FrustumView, "OrbmWorldPointView": matcher._WorldPointView, "OrbmTrackOut": matcher._TrackOut,
    "OrbmLastFrameBatch": matcher._LastFrameBatch, "OrbmKeyFramePointBatch": matcher._KeyFramePointBatch, "OrbmKeyFrameView": views.KeyFrameView,
    "OrbxInputSpec": extractor.InputSpec, "OrbmFisheyeFrameView": views.FisheyeFrameView, "OrbmMapPointRightView": views.MapPointRightView,
    "OrbmFrustumRigView": matcher._FrustumRigView, "OrbmTrackOutRight": matcher._TrackOutRight, "OrbmProjection": matcher._Projection,
    "OrbmProjectIn": matcher._ProjectIn, "OrbmProjectOut": matcher._ProjectOut, "OrbmProjectedPointView": views.ProjectedPointView,
}
NUMPY_MIRRORS = {"OrbxKeyPoint": _lib.KP_DTYPE}
NO_MIRROR = {"OrbmKB8Pair", "OrbmAreaQuery"}          # passed as packed float / int arrays by the Python side (checked below by size)


def header_structs():
    src = open(os.path.join(ROOT, "include", "orbx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name, body = m.group(1), m.group(2)
        assert name == m.group(3)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for d in decl.split(","):
                ident = re.findall(r"[A-Za-z_]\w*", re.sub(r"\[[^\]]*\]", "", d))
                fields.append(ident[-1])
        out[name] = fields
    return out


def c_layout(tmp_path, structs):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "orbx.h"', 'int main(void) {']
    for name, fields in structs.items():
        lines.append('printf("%s . %%zu 0\\n", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('printf("%s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (name, f, name, f, name, f))
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    out = {}
    for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines():
        n, f, off, size = l.split()
        out.setdefault(n, {})[f] = (int(off), int(size))
    return out


def test_every_header_struct_has_a_mirror_with_the_same_layout(tmp_path):
    structs = header_structs()
    assert len(structs) >= 22 and set(MIRRORS) | set(NUMPY_MIRRORS) | NO_MIRROR == set(structs), sorted(set(structs) ^ (set(MIRRORS) | set(NUMPY_MIRRORS) | NO_MIRROR))
    lay = c_layout(tmp_path, structs)
    for name, cls in MIRRORS.items():
        assert C.sizeof(cls) == lay[name]["."][0], "%s: sizeof %d in C, %d in %s" % (name, lay[name]["."][0], C.sizeof(cls), cls.__name__)
        assert [f[0] for f in cls._fields_] == structs[name], "%s: fields %s in the header, %s in %s" % (name, structs[name], [f[0] for f in cls._fields_], cls.__name__)
        for f in structs[name]:
            d = getattr(cls, f)
            assert (d.offset, d.size) == lay[name][f], "%s.%s: (offset, size) %s in C, %s in %s" % (name, f, lay[name][f], (d.offset, d.size), cls.__name__)
    for name, dt in NUMPY_MIRRORS.items():
        assert dt.itemsize == lay[name]["."][0] and list(dt.names) == structs[name]
        for f in structs[name]:
            assert (dt.fields[f][1], dt.fields[f][0].itemsize) == lay[name][f], (name, f)
    assert lay["OrbmAreaQuery"]["."][0] == 20 and lay["OrbmKB8Pair"]["."][0] % 4 == 0
