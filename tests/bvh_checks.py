"""Structure checks of the acceleration structures the HIP builder (csrc/lbvh.hip) hands to the traversal kernel -- and to the
oracle, which re-traces the same arrays.  A builder bug that lost, duplicated or mis-boxed triangles would otherwise pass
every parity test (both sides would walk the same broken tree), so every tree an oracle is given goes through bvh_check
(tests/test_gpu_parity.py: Pair).  Vectorised per tree level: 100 000 triangles check in a fraction of a second."""
import numpy as np


def bvh_check(nodes_u32, tris_u32, root, num_tris):
    """Binary tree (RTGGX_BUF_BVH_NODES*, 64-byte nodes: child-0 box min/max, child-1 box min/max, child refs at words 12, 13;
    ref >= 0: node, < 0: ~leaf slot) over the leaf triangles (RTGGX_BUF_BVH_TRIS*, 64 bytes: v0 v1 v2, primitive id at word 12).
    Asserts: every primitive in exactly one leaf slot, every leaf slot and every node reachable exactly once from the root, every
    child box contains -- and is tight around -- everything below it.  Returns the depth of the deepest leaf."""
    prims = tris_u32.reshape(-1, 16)[:, 12]
    assert prims.size == num_tris and np.array_equal(np.sort(prims), np.arange(num_tris, dtype=prims.dtype)), "every primitive in exactly one leaf"
    if num_tris == 1:
        assert root == -1                                  # ~0: the root is the only leaf
        return 1
    nodes = nodes_u32.reshape(-1, 16)
    f = nodes.view(np.float32)
    child = nodes[:, 12:14].view(np.int32)
    n = nodes.shape[0]
    assert n == num_tris - 1 and 0 <= root < n
    tv = tris_u32.reshape(-1, 16).view(np.float32)[:, :9].reshape(-1, 3, 3)
    tmin, tmax = tv.min(axis=1).astype(np.float64), tv.max(axis=1).astype(np.float64)
    # levels by breadth-first search; each node and each leaf slot must be reached exactly once
    seen_node = np.zeros(n, np.int32); seen_leaf = np.zeros(num_tris, np.int32)
    levels, frontier = [], np.array([root], np.int64)
    while frontier.size:
        np.add.at(seen_node, frontier, 1)
        levels.append(frontier)
        c = child[frontier].reshape(-1)
        leaf = ~c[c < 0]
        assert (leaf < num_tris).all()
        np.add.at(seen_leaf, leaf, 1)
        frontier = c[c >= 0].astype(np.int64)
        assert (frontier < n).all()
        assert len(levels) <= n + 1, "cycle"
    assert (seen_node == 1).all(), "every node reachable exactly once"
    assert (seen_leaf == 1).all(), "every leaf slot referenced exactly once"
    # subtree bounds bottom-up, level by level
    bmin = np.zeros((n, 3)); bmax = np.zeros((n, 3))
    for nodes_l in reversed(levels):
        lo, hi = [], []
        for side, off in ((0, 0), (1, 6)):
            c = child[nodes_l, side]
            isleaf = c < 0
            cm = np.where(isleaf[:, None], tmin[np.where(isleaf, ~c, 0)], bmin[np.where(isleaf, 0, c)])
            cM = np.where(isleaf[:, None], tmax[np.where(isleaf, ~c, 0)], bmax[np.where(isleaf, 0, c)])
            box_min, box_max = f[nodes_l, off:off + 3].astype(np.float64), f[nodes_l, off + 3:off + 6].astype(np.float64)
            assert (box_min <= cm).all() and (box_max >= cM).all(), "child box must contain the child"
            assert np.allclose(box_min, cm) and np.allclose(box_max, cM), "child box is tight"
            lo.append(cm); hi.append(cM)
        bmin[nodes_l], bmax[nodes_l] = np.minimum(lo[0], lo[1]), np.maximum(hi[0], hi[1])
    return len(levels) + 1


def bvh4_check(nodes_u32, nodes4_u32, root):
    """The 4-wide nodes the trace kernel walks (RTGGX_BUF_BVH4_NODES*: minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] ref[4] pad[4])
    are the even-depth binary nodes with their internal children folded in: same references, same boxes, bit for bit; odd-depth
    slots stay zero.  Returns the number of 4-wide nodes in use."""
    if root < 0:
        assert nodes4_u32.size == 0 or not nodes4_u32.any()
        return 0
    nodes, nodes4 = nodes_u32.reshape(-1, 16), nodes4_u32.reshape(-1, 32)
    f = nodes.view(np.float32)
    child = nodes[:, 12:14].view(np.int32)
    f4, r4 = nodes4.view(np.float32), nodes4[:, 24:28].view(np.int32)
    depth = np.full(len(nodes), -1)
    frontier, d = np.array([root], np.int64), 0
    while frontier.size:
        depth[frontier] = d
        c = child[frontier].reshape(-1)
        frontier, d = c[c >= 0].astype(np.int64), d + 1
    assert (depth >= 0).all()
    assert not nodes4[depth % 2 == 1].any(), "odd-depth slots stay unused"
    even = np.nonzero(depth % 2 == 0)[0]
    EMPTY = 0x7FFFFFFF
    # expected entries per even node: for each of its two children, the child itself if it is a leaf, else that child's two children
    want_ref = np.full((len(even), 4), EMPTY, np.int64); want_box = np.zeros((len(even), 4, 6), np.float32)
    for side, off in ((0, 0), (1, 6)):
        c = child[even, side]
        leaf = c < 0
        ci = np.where(leaf, 0, c)
        want_ref[:, 2 * side] = np.where(leaf, c, child[ci, 0]); want_box[:, 2 * side] = np.where(leaf[:, None], f[even, off:off + 6], f[ci, 0:6])
        want_ref[:, 2 * side + 1] = np.where(leaf, EMPTY, child[ci, 1]); want_box[:, 2 * side + 1] = np.where(leaf[:, None], 0.0, f[ci, 6:12])
    got_ref = r4[even].astype(np.int64)
    got_box = np.stack([f4[even][:, [k, 4 + k, 8 + k, 12 + k, 16 + k, 20 + k]] for k in range(4)], axis=1)
    # compare as sets per node: sort both by reference
    og, ow = np.argsort(got_ref, axis=1, kind="stable"), np.argsort(want_ref, axis=1, kind="stable")
    gr, wr = np.take_along_axis(got_ref, og, 1), np.take_along_axis(want_ref, ow, 1)
    assert np.array_equal(gr, wr), "4-wide node references"
    gb, wb = np.take_along_axis(got_box, og[:, :, None], 1), np.take_along_axis(want_box, ow[:, :, None], 1)
    used = wr != EMPTY
    assert np.array_equal(gb[used].view(np.uint32), wb[used].view(np.uint32)), "4-wide node boxes"
    inner = wr[used & (wr >= 0)]
    assert (depth[inner] % 2 == 0).all()
    return len(even)


TOP_FLAG = 0x40000000


def bvh4_top_check(nodes4_u32, top_u32, root, capacity):
    """The table of the tree's top the trace kernel keeps in LDS (RTGGX_BUF_BVH4_TOP*): the first min(capacity, all) 4-wide nodes
    in breadth-first order, bit-identical to their records in the node array except that a reference to a node inside the table
    reads TOP_FLAG | position.  Walks table and tree side by side from the root; returns the number of table entries."""
    top = top_u32.reshape(-1, 32)
    if root < 0:
        assert len(top) == 0
        return 0
    nodes4 = nodes4_u32.reshape(-1, 32)
    assert 1 <= len(top) <= capacity
    pair = {0: root}                      # table position -> node
    order = [root]                        # breadth-first order of the tree, as far as needed
    head = 0
    while head < len(order) and len(order) < len(top) + 4:
        refs = nodes4[order[head], 24:28].view(np.int32)
        order.extend(int(r) for r in refs if 0 <= r != 0x7FFFFFFF)
        head += 1
    assert len(order) >= len(top)
    in_tree = int((nodes4.any(axis=1)).sum())
    assert len(top) == min(capacity, in_tree), "the table holds %d of %d nodes (capacity %d)" % (len(top), in_tree, capacity)
    for k in range(len(top)):
        node = order[k]
        assert np.array_equal(top[k, :24], nodes4[node, :24]), "boxes of table entry %d" % k
        want = nodes4[node, 24:28].view(np.int32).astype(np.int64)
        got = top[k, 24:28].view(np.int32).astype(np.int64)
        for e in range(4):
            if 0 <= want[e] != 0x7FFFFFFF and want[e] in order[:len(top)]:
                assert got[e] == (TOP_FLAG | order.index(int(want[e]))), "entry %d reference %d" % (k, e)
            else:
                assert got[e] == want[e], "entry %d reference %d" % (k, e)
    return len(top)
