"""Minimal NRRD reader/writer for the files the reference's tools exchange.

On-disk format (ref: code/HeaderOnly/NRRD/nrrd.hxx:133-395): magic line "NRRD000x", "key: value"
fields, "key:=value" meta entries, a blank line, then raw little-endian data (x fastest).
Only what the hot path needs: 2-D/3-D raw float/short/uchar images and the meta dictionary that
carries "Projection Matrix" (images, ref: config/example_data/proj000.nrrd:8) and the dtr keys
written by RadonIntermediate::writePropertiesToMeta (ref: code/LibEpipolarConsistency/
RadonIntermediate.cpp:95-103).
"""
import re

import numpy as np

_TYPES = {
    "float": np.float32, "double": np.float64,
    "short": np.int16, "unsigned short": np.uint16, "ushort": np.uint16,
    "int": np.int32, "unsigned int": np.uint32, "uint": np.uint32,
    "char": np.int8, "signed char": np.int8, "unsigned char": np.uint8, "uchar": np.uint8,
    "uint8": np.uint8, "int8": np.int8, "int16": np.int16, "uint16": np.uint16,
    "int32": np.int32, "uint32": np.uint32,
}


def read(path):
    """Returns (array, fields, meta).  array has numpy shape sizes[::-1] (x is the last axis)."""
    with open(path, "rb") as f:
        raw = f.read()
    if not raw.startswith(b"NRRD"):
        raise ValueError("%s: not a NRRD file" % path)
    end = raw.find(b"\n\n")
    crlf = raw.find(b"\r\n\r\n")
    if crlf != -1 and (end == -1 or crlf < end):
        end, skip = crlf, 4
    else:
        skip = 2
    if end == -1:
        raise ValueError("%s: header not terminated by a blank line" % path)
    header = raw[:end].decode("latin-1").splitlines()
    fields, meta = {}, {}
    for line in header[1:]:
        if not line or line.startswith("#"):
            continue
        if ":=" in line:
            k, v = line.split(":=", 1)
            meta[k] = v.strip()
        elif ":" in line:
            k, v = line.split(":", 1)
            fields[k.strip()] = v.strip()
    if fields.get("encoding", "raw") != "raw":
        raise ValueError("%s: only raw encoding is supported" % path)
    if fields.get("endian", "little") != "little":
        raise ValueError("%s: only little-endian data is supported" % path)
    sizes = [int(s) for s in fields["sizes"].split()]
    dtype = np.dtype(_TYPES[fields["type"]]).newbyteorder("<")
    count = int(np.prod(sizes))
    data = np.frombuffer(raw, dtype=dtype, count=count, offset=end + skip)
    return data.reshape(sizes[::-1]).copy(), fields, meta


def write(path, array, meta=None, spacings=None):
    array = np.ascontiguousarray(array)
    tname = {np.dtype(np.float32): "float", np.dtype(np.float64): "double",
             np.dtype(np.int16): "short", np.dtype(np.uint16): "unsigned short",
             np.dtype(np.uint8): "unsigned char", np.dtype(np.int32): "int"}[array.dtype]
    lines = ["NRRD0004", "dimension: %d" % array.ndim, "encoding: raw", "endian: little",
             "sizes: " + " ".join(str(s) for s in array.shape[::-1])]
    if spacings is not None:
        lines.append("spacings: " + " ".join(repr(float(s)) for s in spacings))
    lines.append("type: " + tname)
    for k, v in (meta or {}).items():
        lines.append("%s:=%s" % (k, v))
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n\n").encode("latin-1"))
        f.write(array.astype(array.dtype.newbyteorder("<"), copy=False).tobytes())


def parse_matrix(text):
    """"[a b c d; e f g h; i j k l]" -> numpy array (the reference's stringTo<Eigen::Matrix>
    format, ref: code/LibProjectiveGeometry/EigenToStr.hxx)."""
    rows = [r for r in re.sub(r"[\[\]]", "", text).split(";") if r.strip()]
    return np.array([[float(x) for x in r.split()] for r in rows], dtype=np.float64)


def format_matrix(M):
    M = np.asarray(M, dtype=np.float64)
    return "[" + "; ".join(" ".join("%.12g" % x for x in row) for row in M) + "]"


# ---------------------------------------------------------------------------------------------------
# Radon-intermediate files and projection tables exchanged with the reference's tools
# ---------------------------------------------------------------------------------------------------

FILTER_NAMES = {0: "Derivative", 1: "Ramp", 2: "None"}


def write_dtr(path, data, n_u, n_v, filter=0, projection_matrix=None):
    """dtr as the reference stores it: n_t x n_alpha float32 (alpha fastest) with the meta keys of
    RadonIntermediate::writePropertiesToMeta (ref: code/LibEpipolarConsistency/RadonIntermediate.cpp:95-103)
    and, optionally, "Original Image/Projection Matrix" (ref: Gui/ComputeRadonIntermediate.hxx:79)."""
    data = np.ascontiguousarray(data, np.float32)
    n_t, n_alpha = data.shape
    meta = {
        "Bin Size/Angle": repr(float(np.pi / n_alpha)),
        "Bin Size/Distance": repr(float(np.sqrt(float(n_u) ** 2 + float(n_v) ** 2) / n_t)),
        "Original Image/Width": str(int(n_u)),
        "Original Image/Height": str(int(n_v)),
        "Filter": FILTER_NAMES[int(filter)],
    }
    if projection_matrix is not None:
        meta["Original Image/Projection Matrix"] = format_matrix(projection_matrix)
    write(path, data, meta=meta)


def read_dtr(path):
    """Inverse of write_dtr, ref: RadonIntermediate(const std::string path) + readPropertiesFromMeta
    (RadonIntermediate.cpp:47-67,82-93): unknown/missing "Filter" means None, like the reference."""
    data, _, meta = read(path)
    if data.ndim != 2:
        raise ValueError("%s: a Radon intermediate is a 2-D image" % path)
    flt = {"Derivative": 0, "Ramp": 1}.get(meta.get("Filter", ""), 2)
    info = dict(n_u=int(meta.get("Original Image/Width", 0)), n_v=int(meta.get("Original Image/Height", 0)),
                filter=flt, bin_size_angle=float(meta.get("Bin Size/Angle", 0) or 0),
                bin_size_distance=float(meta.get("Bin Size/Distance", 0) or 0))
    if "Original Image/Projection Matrix" in meta:
        info["projection_matrix"] = parse_matrix(meta["Original Image/Projection Matrix"])
    return np.ascontiguousarray(data, np.float32), info


def write_ompl(path, Ps, comment="", spacing=0.0, detector_size_px=None):
    """One projection matrix per line ("[a b c d; e f g h; i j k l]"), '#' comments, optional "#> key="value""
    attributes (ref: code/HeaderOnly/Utils/Projtable.hxx:168-220, saveProjectionsOneMatrixPerLine)."""
    with open(path, "w") as f:
        if comment:
            f.write("#%s\n" % comment)
        if spacing:
            line = '#> spacing="%s"' % repr(float(spacing))
            if detector_size_px is not None:
                line += ' detector_size_px="[%d %d]"' % tuple(detector_size_px)
            f.write(line + "\n")
        for P in Ps:
            f.write(format_matrix(np.asarray(P, np.float64).reshape(3, 4)) + "\n")


def read_ompl(path):
    """Returns (list of 3x4 arrays, meta dict), ref: loadProjectionsOneMatrixPerLine (Projtable.hxx:168-190)."""
    Ps, meta = [], {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line[0] == "#":
                if len(line) > 1 and line[1] == ">":
                    for k, v in re.findall(r'(\S+?)="([^"]*)"', line[2:]):
                        meta[k] = v
                elif "comment" not in meta:
                    meta["comment"] = line[1:]
                continue
            Ps.append(parse_matrix(line))
    return Ps, meta
