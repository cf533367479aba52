"""Host side of the derived node order (idkengine_amd/csrc/node_layout.hpp): the permutation of a BLAS's node pairs that k_trace2 fetches.
It must be a permutation that keeps pairs 0 and 1 in place, the rewritten child indices must describe the same tree (same boxes, same leaf
ranges, same left/right roles), couples must sit on 128-B aligned slots, and it must put more parent/child pairs into one 128-B line than the
reference order does — for every mode, with either parity of the BLAS's first pair, on real meshes and on soups.  CPU only."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def layout_lib(tmp_path_factory):
    so = tmp_path_factory.mktemp("layout") / "liblayoutcheck.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Werror", os.path.join(HERE, "layout_check.cpp"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.layout_compute.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_void_p]
    L.layout_compute.restype = C.c_int
    return L


def _slots(L, nodes, base_pair, mode, depth=3):
    nodes = np.ascontiguousarray(nodes)
    out = np.zeros(len(nodes) // 2, np.uint32)
    ok = L.layout_compute(nodes.ctypes.data, len(nodes), base_pair, mode, depth, out.ctypes.data)
    return bool(ok), out


def _derive(nodes, slot):
    """numpy statement of k_derive_nodes (kernels_frame.hpp)."""
    out = nodes.copy()
    pairs = len(nodes) // 2
    child = nodes["TriStartOrChild"].copy()
    internal = (nodes["TriCount"] == 0) & (child != 0)
    internal[0] = False
    child[internal] = 2 * slot[child[internal] // 2]
    moved = nodes.copy(); moved["TriStartOrChild"] = child
    for i in (0, 1):
        out[2 * slot[np.arange(pairs)] + i] = moved[2 * np.arange(pairs) + i]
    return out


def _walk(nodes):
    """(box bytes, leaf range or None) of every node in depth-first left-to-right order from the root."""
    seq = []
    stack = [1]
    while stack:
        n = stack.pop()
        nd = nodes[n]
        leaf = int(nd["TriCount"]) > 0
        seq.append((nd["Min"].tobytes() + nd["Max"].tobytes(), (int(nd["TriStartOrChild"]), int(nd["TriCount"])) if leaf else None))
        if not leaf and int(nd["TriStartOrChild"]) != 0:
            c = int(nd["TriStartOrChild"])
            stack.append(c + 1); stack.append(c)
    return seq


def _same_line_edges(nodes, slot, base_pair):
    k = np.arange(len(nodes))
    internal = (nodes["TriCount"] == 0) & (nodes["TriStartOrChild"] != 0) & (k >= 2)
    parent_line = (base_pair + slot[k[internal] // 2]) // 2
    child_line = (base_pair + slot[nodes["TriStartOrChild"][internal] // 2]) // 2
    return float((parent_line == child_line).mean())


@pytest.mark.parametrize("scene", ["soup20k", "lucy", "helmet", "cornell"])
def test_derived_order_is_the_same_tree(scene, layout_lib, native_builder):
    from idkengine_amd import scenes as S
    if scene == "soup20k":
        sc = S.soup_scene(20000, native_builder, seed=3)
    elif scene == "cornell":
        sc = S.cornell_scene(native_builder, variant="mixed")
    else:
        sc = S.mesh_scene(os.path.join(HERE, "golden", "models", scene + ".npz"), native_builder)
    for d in sc.blas_descs:
        nodes = sc.blas_nodes[d["NodeOffset"]: d["NodeOffset"] + d["NodeCount"]]
        ref_walk = _walk(nodes)
        base_frac = None
        for base_pair in (0, 1):
            for mode, depth in ((0, 3), (1, 3), (2, 2), (2, 4)):
                ok, slot = _slots(layout_lib, nodes, base_pair, mode, depth)
                assert ok
                assert sorted(slot.tolist()) == list(range(len(slot))), "not a permutation"
                assert slot[0] == 0 and (len(slot) < 2 or slot[1] == 1)
                if mode == 0:
                    assert (slot == np.arange(len(slot))).all()
                    base_frac = _same_line_edges(nodes, slot, base_pair)
                    continue
                derived = _derive(nodes, slot)
                assert _walk(derived) == ref_walk, "the derived array is not the same tree"
                if len(slot) > 64:
                    assert _same_line_edges(nodes, slot, base_pair) > base_frac + 0.05, (scene, mode, base_frac)


def test_unaligned_child_index_keeps_the_reference_order(layout_lib, native_builder):
    from idkengine_amd import scenes as S
    sc = S.soup_scene(500, native_builder, seed=1)
    nodes = sc.blas_nodes.copy()
    k = int(np.nonzero((nodes["TriCount"] == 0) & (np.arange(len(nodes)) >= 2))[0][0])
    nodes["TriStartOrChild"][k] += 1            # an odd child index: legal for the traversal (any two consecutive nodes), never built by the reference
    ok, slot = _slots(layout_lib, nodes, 0, 1)
    assert not ok and (slot == np.arange(len(slot))).all()


def test_shared_child_pair_keeps_the_reference_order(layout_lib, native_builder):
    """Two internal nodes that point at the same child pair (a DAG): legal for the traversal — the upload validation only asks for children behind their
    parent — but not a tree; the layout would emit the shared pair once per path and run past the BLAS's pair range.  It must fall back to the identity."""
    from idkengine_amd import scenes as S
    sc = S.soup_scene(800, native_builder, seed=2)
    nodes = sc.blas_nodes.copy()
    inner = np.nonzero((nodes["TriCount"] == 0) & (nodes["TriStartOrChild"] != 0) & (np.arange(len(nodes)) >= 2))[0]
    a, b = int(inner[0]), int(inner[1])                     # b lies behind a; both get the child pair of the LATER one (still behind both parents)
    later = max(int(nodes["TriStartOrChild"][a]), int(nodes["TriStartOrChild"][b]))
    nodes["TriStartOrChild"][a] = later; nodes["TriStartOrChild"][b] = later
    for mode in (1, 2):
        ok, slot = _slots(layout_lib, nodes, 0, mode)
        assert not ok and (slot == np.arange(len(slot))).all()
