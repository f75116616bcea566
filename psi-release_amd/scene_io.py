"""On-disk formats of a PROX-E / MP3D-R scene as the fitting entry points read them.

* ``{scene_sdf_path}.json`` (``min``, ``max``, ``dim``) + ``{scene_sdf_path}_sdf.npy`` (flat D^3 fp32, C order x,y,z)
  — fitting_proxe.py:80-85
* ``scenes_downsampled/{scene}.ply`` vertices — fitting_proxe.py:93-96 (the reference uses open3d; only the vertex
  positions are consumed, so a small PLY vertex reader replaces that dependency)
* ``body_segments/{part}.json`` — cvae.py:99-115 (see geometry.GeometryTransformer.get_contact_id)
"""
from __future__ import annotations

import json

import numpy as np


def read_sdf(scene_sdf_path: str):
    with open(scene_sdf_path + '.json') as f:
        d = json.load(f)
    grid_min = np.array(d['min'], dtype=np.float32)
    grid_max = np.array(d['max'], dtype=np.float32)
    dim = int(d['dim'])
    sdf = np.load(scene_sdf_path + '_sdf.npy').reshape(dim, dim, dim).astype(np.float32)
    return sdf, grid_min, grid_max, dim


_PLY_TYPES = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1',
              'char': 'i1', 'int8': 'i1', 'short': '<i2', 'int16': '<i2', 'ushort': '<u2', 'uint16': '<u2',
              'int': '<i4', 'int32': '<i4', 'uint': '<u4', 'uint32': '<u4'}


def read_ply_vertices(path: str) -> np.ndarray:
    """Vertex positions [m,3] fp32 of an ASCII or binary-little-endian PLY."""
    with open(path, 'rb') as f:
        header = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError('PLY header not terminated: %s' % path)
            header.append(line.decode('ascii', 'replace').strip())
            if header[-1] == 'end_header':
                break
        fmt = [h for h in header if h.startswith('format')][0].split()[1]
        nvert, props, in_vertex = 0, [], False
        for h in header:
            t = h.split()
            if t[:2] == ['element', 'vertex']:
                nvert, in_vertex = int(t[2]), True
            elif t and t[0] == 'element':
                in_vertex = False
            elif t and t[0] == 'property' and in_vertex:
                props.append((t[1], t[2]))
        names = [p[1] for p in props]
        if fmt == 'ascii':
            ix = [names.index(c) for c in 'xyz']
            rows = np.array([f.readline().split() for _ in range(nvert)], dtype=np.float64)
            return np.ascontiguousarray(rows[:, ix], dtype=np.float32)
        if fmt != 'binary_little_endian':
            raise ValueError('unsupported PLY format %s' % fmt)
        dt = np.dtype([(n, _PLY_TYPES[t]) for t, n in props])
        arr = np.frombuffer(f.read(nvert * dt.itemsize), dtype=dt, count=nvert)
        return np.ascontiguousarray(np.stack([arr['x'], arr['y'], arr['z']], -1), dtype=np.float32)


def write_ply_vertices(path: str, verts: np.ndarray) -> None:
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    with open(path, 'wb') as f:
        f.write(b'ply\nformat binary_little_endian 1.0\n')
        f.write(('element vertex %d\n' % len(verts)).encode())
        f.write(b'property float x\nproperty float y\nproperty float z\nelement face 0\n'
                b'property list uchar int vertex_indices\nend_header\n')
        f.write(verts.tobytes())
