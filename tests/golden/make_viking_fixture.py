"""Generates tests/golden/viking_room_mesh.npz from the one model file the reference ships
(/root/reference/assets/models/viking_room.glb, glTF 2.0 binary: one node, one mesh, one double-sided material).
Run in the build container only (the reference does not exist on the GPU box); the .npz is the committed fixture.
Decoding follows the reference's loader where it matters for this path: POSITION / NORMAL as stored, TEXCOORD_0 with
V flipped (GltfModelLoader.cpp:649), uint16 indices widened, node TRS -> column-major local matrix
(Node::getLocalMatrix)."""
import json
import os
import struct
import sys

import numpy as np

SRC = "/root/reference/assets/models/viking_room.glb"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "viking_room_mesh.npz")

CTYPE = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def main():
    blob = open(SRC, "rb").read()
    magic, version, _ = struct.unpack("<III", blob[:12])
    assert magic == 0x46546C67 and version == 2
    jlen, jtype = struct.unpack("<II", blob[12:20])
    assert jtype == 0x4E4F534A
    doc = json.loads(blob[20:20 + jlen])
    off = 20 + jlen
    blen, btype = struct.unpack("<II", blob[off:off + 8])
    assert btype == 0x004E4942
    binary = blob[off + 8:off + 8 + blen]

    def accessor(i):
        a = doc["accessors"][i]
        bv = doc["bufferViews"][a["bufferView"]]
        dt, nc = CTYPE[a["componentType"]], NCOMP[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        assert "byteStride" not in bv
        return np.frombuffer(binary, dtype=dt, count=a["count"] * nc, offset=start).reshape(a["count"], nc).copy()

    prim = doc["meshes"][0]["primitives"][0]
    pos = accessor(prim["attributes"]["POSITION"]).astype(np.float32)
    nrm = accessor(prim["attributes"]["NORMAL"]).astype(np.float32)
    uv = accessor(prim["attributes"]["TEXCOORD_0"]).astype(np.float32)
    uv[:, 1] = 1.0 - uv[:, 1]
    idx = accessor(prim["indices"]).astype(np.uint32).reshape(-1)
    node = doc["nodes"][0]
    q = np.array(node.get("rotation", [0, 0, 0, 1]), dtype=np.float64)      # x y z w
    t = np.array(node.get("translation", [0, 0, 0]), dtype=np.float64)
    s = np.array(node.get("scale", [1, 1, 1]), dtype=np.float64)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4)
    M[:3, :3] = R * s[None, :]
    M[:3, 3] = t
    mat = doc["materials"][prim["material"]]
    np.savez_compressed(DST, positions=pos, normals=nrm, uvs=uv, indices=idx,
                        node_local_col_major=M.T.astype(np.float32).reshape(16),
                        double_sided=np.uint32(1 if mat.get("doubleSided") else 0))
    print(f"wrote {DST}: {len(pos)} vertices, {len(idx) // 3} triangles, {os.path.getsize(DST)} bytes")


if __name__ == "__main__":
    sys.exit(main())
