"""CPU checks of bench.py's bookkeeping (no GPU): the algorithmic-byte formulas against SURVEY.md §8d, the strong-split
shard workload, the per-step median and the baseline metric string."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from singa_amd import dp, graph as G  # noqa: E402


def test_kernel_bytes_follow_the_survey_formulas():
    """k10 forward E (KR HV 4 + heads 4 + frame) + N K HV 4 + (N + 1) 4 with the 36-byte frame replaced by
    the streamed reduced Wigner rows (WSZ 4 bytes: 36 / 116 / 236 floats at L = 2 / 4 / 6)."""
    E, N = 1000, 300
    for L, K, KR, WSZ in ((2, 9, 9, 36), (4, 25, 19, 116), (6, 49, 29, 236)):
        want = E * (KR * 112 * 4 + 7 * 4 + WSZ * 4) + N * K * 112 * 4 + (N + 1) * 4
        assert bench.kernel_bytes("k10_fwd", E, N, L) == want
        # the survey's literal figures differ only by the frame term
        survey = {2: E * 4100 + N * 4032, 4: E * 8580 + N * 11200, 6: E * 13060 + N * 21952}[L]
        assert bench.kernel_bytes("k10_fwd", E, N, L) - survey == E * (WSZ * 4 - 36 - 4) + (N + 1) * 4
    assert bench.kernel_bytes("k4_fwd", E, N, 6) == 2 * N * 49 * 16 * 4 + E * (29 * 32 * 4 + 576 * 4 + 236 * 4 + 8)


def test_strong_split_shard_workload():
    """cfg4_shard_r0of8 = the graphs rank 0 of dp.shard_ranges_by_cost owns of the config-3 batch."""
    L, kw, ids, n_parent = G.resolve_workload("cfg4_shard_r0of8")
    Lp, kwp, idsp, n = G.resolve_workload("cfg3_b128_l4")
    assert (L, kw, n_parent) == (Lp, kwp, n) and n == 128 and idsp == list(range(128))
    costs = [G.graph_cost(G.graph_sizes(i, **kwp)) for i in idsp]
    ranges = dp.shard_ranges_by_cost(costs, 8)
    assert ids == list(range(*ranges[0]))
    assert ranges[0][0] == 0 and ranges[-1][1] == 128 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    shard_costs = [sum(costs[lo:hi]) for lo, hi in ranges]
    assert max(shard_costs) < 1.15 * sum(costs) / 8                     # balanced by edge count, not by graph count


def test_median_and_metric_string():
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5 and bench.median([]) is None
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert bench.baseline_metric() == json.load(f)["metric"]
    assert bench.host_cores() >= 1


def test_weight_gradient_split_policy():
    """Host logic of the split-reduction launches (ops._tn_splits / ops._splits_for): never more than 4,096 rows per split
    (the flat cap of 64 splits made a 20 -> 64 Linear over 2.9 M rows 64 workgroups of 45 k rows), at least 256 rows per
    split, at most 64 splits for 128 x 128 outputs and up unless the row bound asks for more, and - for the conv launches - a
    split count whose workgroups fill their last round over the 256 CUs."""
    from singa_amd import ops
    for rows, out, cin in ((2_900_000, 64, 20), (186_000, 128, 32), (49_267, 256, 256), (49_267, 1024, 256), (6_499, 256, 256),
                           (300, 256, 256), (49_267, 256, 8)):
        S = ops._tn_splits(rows, out, cin)
        assert 1 <= S <= 1024
        assert S == 1 or rows // S >= 256 or S == rows // 256, (rows, out, cin, S)
        assert -(-rows // S) <= 4096 or S == 1024, (rows, out, cin, S)
        if out * cin >= 128 * 128 and rows <= 64 * 4096:
            assert S <= 64, (rows, out, cin, S)
    assert ops._tn_splits(2_900_000, 64, 20) > 64 and ops._tn_splits(186_000, 128, 32) > 64
    assert ops._tn_splits(49_267, 256, 256) == 64 and ops._tn_splits(49_267, 1024, 256) == 32
    # conv weight gradients at the 17-graph shard: 117 tiles x 9 splits = 4.11 rounds -> 13 splits = 5.94 rounds
    assert ops._splits_for(16_500) == 9 and ops._splits_for(16_500, 117) == 13
    for rows, tiles in ((15_000, 117), (16_500, 117), (119_782, 117), (15_000, 44), (3_000, 117)):
        S = ops._splits_for(rows, tiles)
        base = ops._splits_for(rows)
        assert base <= S <= max(base, 2 * base) and (S == base or rows // S >= 512)
        if 512 <= tiles * base < 2048:
            w, w0 = tiles * S, tiles * base
            assert w / (-(-w // 256) * 256) >= w0 / (-(-w0 // 256) * 256)


def test_weight_gradient_split_model():
    """ops._splits_few (host logic, no GPU): split counts of the SO(2) convolution's two weight-gradient launches - between 1
    and 64, at least 128 rows per split, one or two full rounds of 512 workgroups where the row count allows, never the
    1.07-round case that ran as two rounds (50 tiles x 11 splits)."""
    from singa_amd import ops
    for rows, tiles in ((13337, 50), (13337, 25), (13337, 14), (101632, 50), (101632, 14), (5003, 10), (777, 14), (100, 14), (0, 5)):
        S = ops._splits_few(rows, tiles)
        assert 1 <= S <= 64 and (S == 1 or rows // S >= 128), (rows, tiles, S)
        if rows >= 10000:
            w = tiles * S
            fill = w / (-(-w // 512) * 512)
            assert fill > 0.9, (rows, tiles, S, fill)
