"""CPU: hand-computed known-answer tests for the restated third-party primitives
(oracle/pyg_primitives.py). These are the only pin on PyG / torch_scatter / torch_cluster semantics
(their sources are not in /root/reference) -- SURVEY 8(c) 'Known-answer tests'."""
import math

import torch

from oracle import pyg_primitives as P


def test_self_loop_strip_and_add_with_duplicates():
    ei = torch.tensor([[0, 1, 1, 2, 2, 0], [1, 1, 2, 2, 0, 1]])      # loops (1,1),(2,2); duplicate (0,1)
    stripped, _ = P.remove_self_loops(ei)
    assert stripped.tolist() == [[0, 1, 2, 0], [1, 2, 0, 1]]
    added, _ = P.add_self_loops(stripped, num_nodes=4)
    assert added.tolist() == [[0, 1, 2, 0, 0, 1, 2, 3], [1, 2, 0, 1, 0, 1, 2, 3]]


def test_message_direction_source_to_target():
    """asymmetric edge 0->1 only: node 1 sees node 0, node 0 sees nothing but its self loop."""
    class Conv(P.MessagePassing):
        def message(self, x_i, x_j):
            return x_j - x_i
    x = torch.tensor([[1.0], [5.0], [9.0]])
    ei = torch.tensor([[0, 0, 1, 2], [1, 0, 1, 2]])
    out = Conv().propagate(ei, x=x)
    assert out.view(-1).tolist() == [0.0, 0.0, 0.0]          # max(x0-x1, 0) at node 1 = max(-4, 0)
    ei2 = torch.tensor([[1, 0, 1, 2], [0, 0, 1, 2]])          # 1 -> 0
    assert Conv().propagate(ei2, x=x).view(-1).tolist() == [4.0, 0.0, 0.0]


def test_scatter_max_empty_segment_is_zero_and_negative_values_survive():
    src = torch.tensor([[-3.0, 2.0], [-1.0, -7.0], [4.0, 0.5]])
    idx = torch.tensor([0, 0, 2])
    out, _ = P.scatter_max(src, idx, dim=0, dim_size=4)
    assert out.tolist() == [[-1.0, 2.0], [0.0, 0.0], [4.0, 0.5], [0.0, 0.0]]


def test_fps_on_a_line():
    pos = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [10.0, 0, 0], [4.0, 0, 0], [5.0, 0, 0]])
    # start at 0, farthest = 10, then the point maximising min-distance: 5 (d=5) vs 4 (d=4)
    assert P.fps(pos, None, ratio=0.5, random_start=False).tolist() == [0, 3, 5]
    # two clouds, ceil(0.5 * 3) = 2 each, indices are global
    batch = torch.tensor([0, 0, 0, 1, 1, 1])
    # cloud 1 = x in {10, 4, 5}: start 10, farthest from it is 4 (d=6)
    assert P.fps(pos, batch, ratio=0.5, random_start=False).tolist() == [0, 2, 3, 4]


def test_radius_first_hits_in_index_order_strict():
    x = torch.tensor([[0.0, 0, 0], [0.5, 0, 0], [1.0, 0, 0], [0.2, 0, 0], [0.1, 0, 0]])
    y = torch.tensor([[0.0, 0, 0]])
    row, col = P.radius(x, y, 1.0, max_num_neighbors=3)
    assert row.tolist() == [0, 0, 0] and col.tolist() == [0, 1, 3]     # x2 at exactly r excluded (strict <)
    row, col = P.radius(x, y, 1.0, max_num_neighbors=64)
    assert col.tolist() == [0, 1, 3, 4]


def test_knn_and_interpolate_with_coincident_point():
    px = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [3.0, 0, 0]])
    fx = torch.tensor([[10.0], [20.0], [40.0]])
    py = torch.tensor([[1.0, 0, 0], [2.0, 0, 0]])
    ai = P.knn(px, py, 2)
    assert ai.tolist() == [[0, 0, 1, 1], [1, 0, 1, 2]]                # y1: x1 (d=1) then x2 (d=1)? no: stable -> x1, x2
    out = P.knn_interpolate(fx, px, py, k=2)
    # y0 coincides with x1: w = 1e16 vs 1 -> 20 (to fp32 precision)
    assert abs(out[0, 0].item() - 20.0) < 1e-4
    # y1: d2 = 1 to both x1 and x2 -> mean(20, 40)
    assert abs(out[1, 0].item() - 30.0) < 1e-5


def test_knn_cosine_picks_max_similarity_same_cloud_only():
    x = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.05]])
    bx = torch.tensor([0, 0, 0, 1])
    y = torch.tensor([[2.0, 0.1], [0.1, 3.0], [5.0, 0.0]])
    by = torch.tensor([0, 0, 1])
    ai = P.knn(x, y, 1, bx, by, cosine=True)
    assert ai.tolist() == [[0, 1, 2], [0, 1, 3]]


def test_pointconv_bipartite_self_loop_quirk():
    """PointConv(add_self_loops=True) on a bipartite graph: pairs with equal raw indices are dropped
    and (k, k) for k < min(N_src, N_dst) appended -- target k also hears SOURCE k."""
    conv = P.PointConv(local_nn=None)
    pos_src = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [7.0, 0, 0]])
    pos_dst = torch.tensor([[2.0, 0, 0], [7.0, 0, 0]])                 # = sources 2 and 3
    ei = torch.tensor([[2, 1, 3, 1], [0, 0, 1, 1]])                    # (src, dst); (1,1) is a "self loop"
    out = conv((None, None), (pos_src, pos_dst), ei)
    # dst0: from src2 (0), src1 (-1), + appended (0,0): src0 (-2) -> max = 0
    # dst1: from src3 (0); (1,1) dropped then re-appended: src1 (1-7=-6) -> max = 0
    assert out[:, 0].tolist() == [0.0, 0.0]
    # drop the true self edges: now only the quirk edges + src1->dst0 remain
    ei2 = torch.tensor([[1], [0]])
    out2 = conv((None, None), (pos_src, pos_dst), ei2)
    assert out2[:, 0].tolist() == [-1.0, -6.0]


def test_global_max_pool():
    x = torch.tensor([[1.0, -1.0], [0.0, 5.0], [-2.0, -3.0]])
    assert P.global_max_pool(x, torch.tensor([0, 0, 1])).tolist() == [[1.0, 5.0], [-2.0, -3.0]]


# ---- VERDICT r3 #4c: the three [PyG-recall] behaviours that had no known-answer test yet --------------------------------------
def test_radius_over_full_ball_with_duplicates_at_equal_distance():
    """torch_cluster.radius keeps the FIRST max_num_neighbors hits in index order -- not the nearest ones -- and duplicates /
    equidistant points are ordinary hits. 70 points inside the ball of one centre: 6 far-ish ones first, then 32 pairs of
    coincident points at the same distance; a point at exactly r in between must be skipped (strict <)."""
    far = [[0.9, 0.0, 0.0]] * 3 + [[0.0, 0.9, 0.0]] * 3                   # indices 0..5, d = 0.9 (inside r = 1), two triples of duplicates
    edge = [[1.0, 0.0, 0.0]]                                               # index 6: d == r exactly -> excluded
    ring = [[0.1, 0.0, 0.0], [0.0, 0.1, 0.0]] * 32                         # indices 7..70: all at d = 0.1, 32 coincident pairs
    x = torch.tensor(far + edge + ring)
    y = torch.tensor([[0.0, 0.0, 0.0]])
    row, col = P.radius(x, y, 1.0, max_num_neighbors=64)
    assert row.tolist() == [0] * 64
    assert col.tolist() == [0, 1, 2, 3, 4, 5] + list(range(7, 65))         # 6 far ones kept although 64 nearer ones exist; 65..70 dropped
    # a cap larger than the ball keeps all 70, still without the point at r
    row, col = P.radius(x, y, 1.0, max_num_neighbors=128)
    assert col.tolist() == [i for i in range(71) if i != 6]


def test_fps_arg_max_ties_take_the_lowest_index():
    """symmetric cloud: after the start point the two extremes tie exactly; torch's argmax (and torch_cluster's serial scan)
    keep the FIRST maximum. Then the two mid points tie again."""
    pos = torch.tensor([[0.0, 0, 0], [-2.0, 0, 0], [2.0, 0, 0], [1.0, 0, 0], [-1.0, 0, 0], [0.0, 0.5, 0]])
    # start 0; d^2 = [0, 4, 4, 1, 1, .25] -> tie (1, 2) -> 1. min-d^2 = [0, 0, 4, 1, 1, .25] -> 2.
    # then [0, 0, 0, 1, 1, .25] -> tie (3, 4) -> 3.
    assert P.fps(pos, None, ratio=4 / 6, random_start=False).tolist() == [0, 1, 2, 3]
    # the same cloud with the tied points listed in the other order: the other member of each pair wins
    perm = torch.tensor([0, 2, 1, 4, 3, 5])
    assert P.fps(pos[perm], None, ratio=4 / 6, random_start=False).tolist() == [0, 1, 2, 3]
    assert pos[perm][[1, 3]].tolist() == [[2.0, 0, 0], [-1.0, 0, 0]]


def test_add_self_loops_on_bipartite_index_both_ways():
    """PointConv's self-loop step on a bipartite (source, target) index works on the RAW index pairs with
    num_nodes = min(N_src, N_dst): with more sources than targets, and with more targets than sources."""
    conv = P.PointConv(local_nn=None)
    # N_src = 5 > N_dst = 2: appended (0, 0), (1, 1); the raw pair (1, 1) is removed first, then comes back as the appended one
    ps = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [3.0, 0, 0], [4.0, 0, 0]])
    pd = torch.tensor([[10.0, 0, 0], [20.0, 0, 0]])
    ei = torch.tensor([[4, 1, 3], [0, 1, 1]])
    out = conv((None, None), (ps, pd), ei)
    # dst0: src4 (4-10 = -6), appended src0 (-10) -> -6; dst1: src3 (3-20 = -17), appended src1 (1-20 = -19) -> -17
    assert out[:, 0].tolist() == [-6.0, -17.0]
    # N_src = 2 < N_dst = 4: loops only for k < 2; targets 2 and 3 get what their real edges give; target 3 has none -> 0 (empty max)
    ps2 = torch.tensor([[1.0, 0, 0], [2.0, 0, 0]])
    pd2 = torch.tensor([[0.0, 0, 0], [0.0, 0, 0], [5.0, 0, 0], [9.0, 0, 0]])
    ei2 = torch.tensor([[1, 0, 1], [0, 2, 2]])
    out2 = conv((None, None), (ps2, pd2), ei2)
    # dst0: src1 (2), appended src0 (1) -> 2; dst1: appended src1 only (2); dst2: src0 (-4), src1 (-3) -> -3; dst3: nothing -> 0
    assert out2[:, 0].tolist() == [2.0, 2.0, -3.0, 0.0]
    idx, _ = P.add_self_loops(P.remove_self_loops(ei2)[0], num_nodes=min(2, 4))
    assert idx.tolist() == [[1, 0, 1, 0, 1], [0, 2, 2, 0, 1]]
