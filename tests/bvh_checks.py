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


def bvh4_check(nodes_u32, nodes4_u32, root, tris_u32=None):
    """The 4-wide nodes the trace kernel walks (RTGGX_BUF_BVH4_NODES*: minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] ref[4] pad[4]) against
    the binary tree: a 4-wide node is an even-depth binary node whose entries are its grandchildren (or its children where those are
    leaves).  Since round 3 an entry may also be a MULTI-LEAF: a whole binary subtree of at most four triangles, referenced as
    ~(first slot | (count - 1) << 28) -- possible because the leaf slots are in depth-first order of the tree, so that every subtree's
    triangles are consecutive.  Checked: walking the 4-wide nodes from the root reaches every leaf slot exactly once; every entry's
    box is, bit for bit, the binary tree's box of the subtree (or leaf) it stands for; slots nobody references stay zero.
    Returns the number of 4-wide nodes in use."""
    if root < 0:
        assert nodes4_u32.size == 0 or not nodes4_u32.any()
        return 0
    nodes, nodes4 = nodes_u32.reshape(-1, 16), nodes4_u32.reshape(-1, 32)
    f = nodes.view(np.float32)
    child = nodes[:, 12:14].view(np.int32)
    n = len(nodes)
    u4, r4 = nodes4[:, :24], nodes4[:, 24:28].view(np.int32)
    # per binary node: depth, parent side (where its box is stored), leaf range [first, first + count) of its subtree
    depth = np.full(n, -1); parent = np.full(n, -1); side = np.zeros(n, np.int64)
    levels, frontier, d = [], np.array([root], np.int64), 0
    while frontier.size:
        depth[frontier] = d; levels.append(frontier)
        for sd in (0, 1):
            c = child[frontier, sd]
            inner = c >= 0
            parent[c[inner]] = frontier[inner]; side[c[inner]] = sd
        c = child[frontier].reshape(-1)
        frontier, d = c[c >= 0].astype(np.int64), d + 1
    assert (depth >= 0).all()
    first = np.zeros(n, np.int64); count = np.zeros(n, np.int64); contiguous = np.ones(n, bool)
    for nodes_l in reversed(levels):
        c0, c1 = child[nodes_l, 0].astype(np.int64), child[nodes_l, 1].astype(np.int64)
        f0 = np.where(c0 < 0, ~c0, first[np.where(c0 < 0, 0, c0)]); n0 = np.where(c0 < 0, 1, count[np.where(c0 < 0, 0, c0)])
        f1 = np.where(c1 < 0, ~c1, first[np.where(c1 < 0, 0, c1)]); n1 = np.where(c1 < 0, 1, count[np.where(c1 < 0, 0, c1)])
        ok0 = np.where(c0 < 0, True, contiguous[np.where(c0 < 0, 0, c0)]); ok1 = np.where(c1 < 0, True, contiguous[np.where(c1 < 0, 0, c1)])
        first[nodes_l] = np.minimum(f0, f1); count[nodes_l] = n0 + n1
        contiguous[nodes_l] = ok0 & ok1 & ((f0 + n0 == f1) | (f1 + n1 == f0))
    # the box the binary tree stores for a node: in its parent's child fields (the root has none)
    pr = np.where(parent >= 0, parent, 0)
    nbox = np.where((side == 0)[:, None], f[pr, 0:6], f[pr, 6:12]).view(np.uint32)
    leaf_box = {}            # leaf slot -> box (from the parent's child fields)
    for sd, off in ((0, 0), (1, 6)):
        c = child[:, sd]
        for i in np.nonzero(c < 0)[0]:
            leaf_box[int(~c[i])] = nodes[i, off:off + 6]
    by_range = {(int(first[i]), int(count[i])): i for i in range(n) if contiguous[i] and i != root}
    EMPTY = 0x7FFFFFFF
    num_leaves = int(count[root])
    seen_leaf = np.zeros(num_leaves, np.int32)
    used = np.zeros(n, bool)
    stack = [int(root)]
    while stack:
        v = stack.pop()
        assert not used[v] and depth[v] % 2 == 0, "4-wide node %d reached twice or at odd depth" % v
        used[v] = True
        got = 0
        for e in range(4):
            r = int(r4[v, e])
            if r == EMPTY:
                continue
            box = u4[v, e::4]
            if r >= 0:
                assert r < n and parent[r] >= 0 and (parent[r] == v or parent[parent[r]] == v), "entry %d of node %d is not a (grand)child" % (e, v)
                want = nbox[r]
                stack.append(r); got += int(count[r])
            else:
                lr = ~r
                slot, cnt = lr & 0x0FFFFFFF, (lr >> 28) + 1
                assert 1 <= cnt <= 4 and slot + cnt <= num_leaves
                seen_leaf[slot:slot + cnt] += 1
                if cnt == 1:
                    want = leaf_box[slot]
                else:
                    assert (slot, cnt) in by_range, "multi-leaf (%d, %d) of node %d is not a subtree of the binary tree" % (slot, cnt, v)
                    sub = by_range[(slot, cnt)]
                    assert parent[sub] == v or parent[parent[sub]] == v
                    want = nbox[sub]
                got += cnt
            assert np.array_equal(box, want), "box of entry %d of node %d" % (e, v)
        assert got == count[v], "node %d covers %d of its %d triangles" % (v, got, count[v])
    assert (seen_leaf == 1).all(), "every leaf slot reached exactly once through the 4-wide nodes"
    assert not nodes4[~used].any(), "slots of nodes nobody references stay unused"
    return int(used.sum())


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
