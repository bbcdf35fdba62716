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


def bvh4_check(nodes_u32, nodes4_u32, root, tris_u32=None, built_shape=True, weights=(1.0, 0.0)):
    """The 4-wide nodes the trace kernel walks (RTGGX_BUF_BVH4_NODES*: minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] ref[4] pad[4]) against
    the binary tree.  Round 4: WHICH binary nodes are 4-wide nodes, and with which entries, is chosen by dynamic programming over the
    binary tree so that the summed half-area of the 4-wide nodes -- the expected node steps of a random ray -- is least (lbvh.hip "the
    4-wide collapse"; `weights`: a node costs weights[0] x its half-area / the root's + weights[1] x its triangles / all triangles,
    rtggx_debug_collapse_weights):  F(leaf, i) = 0;  F(n, 1) = cost(n) + min_k F(left, k) + F(right, 4 - k);  F(n, i) = min(F(n, 1), min_k F(left, k) +
    F(right, i - k)) for i = 2, 3; ties go to the first candidate.  Checked: the choice itself, re-derived here in the same fp32 arithmetic
    from the binary nodes' boxes for every 4-wide node (built_shape=False skips this for a REFITTED tree: it keeps the choice its build
    made for another shape; every entry must then still be a descendant at most three levels down); walking the 4-wide nodes from the
    root reaches every leaf slot exactly once; every entry's box is, bit for bit, the binary tree's box of the subtree (or leaf) it
    stands for; slots nobody references stay zero; the level recorded in pad[0].  Returns the number of 4-wide nodes in use."""
    if root < 0:
        assert nodes4_u32.size == 0 or not nodes4_u32.any()
        return 0
    nodes, nodes4 = nodes_u32.reshape(-1, 16), nodes4_u32.reshape(-1, 32)
    f = nodes.view(np.float32)
    child = nodes[:, 12:14].view(np.int32)
    n = len(nodes)
    u4, r4 = nodes4[:, :24], nodes4[:, 24:28].view(np.int32)
    # per binary node: parent and side (where its box is stored), number of leaves below it
    parent = np.full(n, -1); side = np.zeros(n, np.int64)
    levels, frontier = [], np.array([root], np.int64)
    while frontier.size:
        levels.append(frontier)
        for sd in (0, 1):
            c = child[frontier, sd]
            inner = c >= 0
            parent[c[inner]] = frontier[inner]; side[c[inner]] = sd
        c = child[frontier].reshape(-1)
        frontier = c[c >= 0].astype(np.int64)
    count = np.zeros(n, np.int64)
    for nodes_l in reversed(levels):
        c0, c1 = child[nodes_l, 0].astype(np.int64), child[nodes_l, 1].astype(np.int64)
        count[nodes_l] = np.where(c0 < 0, 1, count[np.where(c0 < 0, 0, c0)]) + np.where(c1 < 0, 1, count[np.where(c1 < 0, 0, c1)])
    # the box the binary tree stores for a node: in its parent's child fields (the root has none); its half-area as the builder adds it up
    pr = np.where(parent >= 0, parent, 0)
    nboxf = np.where((side == 0)[:, None], f[pr, 0:6], f[pr, 6:12]).astype(np.float32)
    nbox = nboxf.view(np.uint32)
    ex, ey, ez = nboxf[:, 3] - nboxf[:, 0], nboxf[:, 4] - nboxf[:, 1], nboxf[:, 5] - nboxf[:, 2]
    area = ((ex * ey + ey * ez) + ez * ex).astype(np.float32)
    leaf_box = {}            # leaf slot -> box (from the parent's child fields)
    for sd, off in ((0, 0), (1, 6)):
        c = child[:, sd]
        for i in np.nonzero(c < 0)[0]:
            leaf_box[int(~c[i])] = nodes[i, off:off + 6]
    EMPTY = 0x7FFFFFFF

    # the dynamic programme, level by level from the bottom (fp32, the builder's order of operations)
    F = np.zeros((n, 3), np.float32); K1 = np.zeros(n, np.int64); D2 = np.zeros(n, np.int64); D3 = np.zeros(n, np.int64)
    if built_shape:
        zero = np.zeros(3, np.float32)
        # the root has no box in the binary tree's records: its box is the union of its children's
        lo = np.minimum(f[root, 0:3], f[root, 6:9]); hi = np.maximum(f[root, 3:6], f[root, 9:12])
        e_ = (hi - lo).astype(np.float32)
        root_area = np.float32((e_[0] * e_[1] + e_[1] * e_[2]) + e_[2] * e_[0])
        ka, kt = np.float32(weights[0]) / root_area, np.float32(weights[1]) / np.float32(count[root])
        for nodes_l in reversed(levels):
            c0, c1 = child[nodes_l, 0].astype(np.int64), child[nodes_l, 1].astype(np.int64)
            f0 = np.where((c0 < 0)[:, None], zero, F[np.where(c0 < 0, 0, c0)]).astype(np.float32)
            f1 = np.where((c1 < 0)[:, None], zero, F[np.where(c1 < 0, 0, c1)]).astype(np.float32)
            best, k1 = f0[:, 0] + f1[:, 2], np.ones(len(nodes_l), np.int64)
            for k, v in ((2, f0[:, 1] + f1[:, 1]), (3, f0[:, 2] + f1[:, 0])):
                take = v < best
                best, k1 = np.where(take, v, best), np.where(take, k, k1)
            cost = (area[nodes_l] * ka + count[nodes_l].astype(np.float32) * kt).astype(np.float32)
            if nodes_l.size == 1 and nodes_l[0] == root:
                cost = (np.array([root_area], np.float32) * ka + np.float32(count[root]) * kt).astype(np.float32)
            f1_ = (cost + best).astype(np.float32)
            v = f0[:, 0] + f1[:, 0]
            d2 = (v < f1_).astype(np.int64); f2_ = np.where(v < f1_, v, f1_)
            f3_, d3 = f1_.copy(), np.zeros(len(nodes_l), np.int64)
            for k, v in ((1, f0[:, 0] + f1[:, 1]), (2, f0[:, 1] + f1[:, 0])):
                take = v < f3_
                f3_, d3 = np.where(take, v, f3_), np.where(take, k, d3)
            F[nodes_l, 0], F[nodes_l, 1], F[nodes_l, 2] = f1_, f2_, f3_
            K1[nodes_l], D2[nodes_l], D3[nodes_l] = k1, d2, d3

    def rule(v):
        e, todo = [], [(int(child[v, 1]), 4 - int(K1[v])), (int(child[v, 0]), int(K1[v]))]
        while todo:
            x, allowed = todo.pop()
            k = 0 if (x < 0 or allowed == 1) else int(D2[x]) if allowed == 2 else int(D3[x])
            if k == 0:
                e.append(x)
            else:
                todo.append((int(child[x, 1]), allowed - k)); todo.append((int(child[x, 0]), k))
        return e + [EMPTY] * (4 - len(e))

    def within_three(v, r):
        for _ in range(3):
            r = int(parent[r])
            if r == v:
                return True
            if r < 0:
                return False
        return False

    num_leaves = int(count[root])
    seen_leaf = np.zeros(num_leaves, np.int32)
    used = np.zeros(n, bool)
    stack = [(int(root), 0)]
    while stack:
        v, lvl = stack.pop()
        assert not used[v], "4-wide node %d reached twice" % v
        used[v] = True
        if built_shape:
            assert [int(x) for x in r4[v]] == rule(v), "entries of 4-wide node %d: %s, the surface-area rule gives %s" % (v, list(r4[v]), rule(v))
        assert int(nodes4[v, 28]) == lvl, "level of node %d" % v
        got = 0
        for e in range(4):
            r = int(r4[v, e])
            if r == EMPTY:
                assert (u4[v, e:12:4].view(np.float32) == np.inf).all() and (u4[v, 12 + e::4].view(np.float32) == -np.inf).all(), "an unused entry's box is empty"
                continue
            box = u4[v, e::4]
            if r >= 0:
                assert r < n and within_three(v, r), "entry %d of node %d is not a descendant within three levels" % (e, v)
                want = nbox[r]
                stack.append((r, lvl + 1)); got += int(count[r])
            else:
                slot = ~r
                assert 0 <= slot < num_leaves
                seen_leaf[slot] += 1
                want = leaf_box[slot]
                got += 1
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
