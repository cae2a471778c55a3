"""Host logic of train._FusedAdam (no GPU): which queued weight-gradient products the optimiser launch may take.

A matrix group of the arena is taken only when its rows are covered completely by products that (a) are the only
contribution to their gradient in the launch, (b) overwrite (no accumulate bit) and (c) consist of whole 128 x 128 tiles;
what is left -- and not dead -- goes to the separate Adam launch."""
from types import SimpleNamespace

import torch

from openvivqa_amd.train import _FusedAdam


def _harness():
    # three matrix groups: [0] fc_q | fc_k | fc_v (3 x 128 rows, 256 cols), [1] a 128 x 128 matrix, [2] a ragged 360 x 128 one
    groups = [(0, 384, 256), (384 * 256, 128, 128), (384 * 256 + 128 * 128, 360, 128)]
    numel = groups[-1][0] + 360 * 128 + 64
    arena = SimpleNamespace(_groups2d=list(groups), grad=torch.zeros(numel), master=torch.zeros(numel),
                            shadow=torch.zeros(numel, dtype=torch.bfloat16), shadow_t=torch.zeros(numel, dtype=torch.bfloat16),
                            small_lo=groups[-1][0] + 360 * 128, numel=numel)
    optim = SimpleNamespace(exp_avg=torch.zeros(numel), exp_avg_sq=torch.zeros(numel))
    ts = SimpleNamespace(arena=arena, optim=optim, _dead=[])
    return _FusedAdam(ts), arena, groups


def _item(arena, off, N, K, acc=0):
    dw = arena.grad[off:off + N * K].view(N, K)
    return (None, None, dw, 0, 0, 64, N, K, acc, None)


def test_whole_groups_only_and_single_overwriting_products():
    fa, arena, groups = _harness()
    g0, g1, g2 = (g[0] for g in groups)
    # the packed q | k | v product covers group 0; group 1 by one product; the ragged group is not made of whole tiles
    items = [_item(arena, g0, 384, 256), _item(arena, g1, 128, 128), _item(arena, g2, 360, 128)]
    out = fa.targets(items)
    assert out[0] is not None and out[1] is not None and out[2] is None
    assert fa.ranges == [(g0, g1 + 128 * 128)]  # (adjacent groups merge)
    assert out[0].ld_transposed == 384 and out[1].ld_transposed == 128
    assert out[0].param == arena.master.data_ptr() and out[1].shadow == arena.shadow.data_ptr() + 2 * g1
    assert fa.rest_groups() == [g2] and fa.rest()[-1] == (arena.small_lo, arena.numel)


def test_partial_coverage_accumulation_and_duplicates_keep_the_separate_update():
    fa, arena, groups = _harness()
    g0, g1, _ = (g[0] for g in groups)
    # only fc_q of the q | k | v group has a product: the group stays whole with the separate launch
    assert fa.targets([_item(arena, g0, 128, 256)]) == [None] and fa.ranges == []
    # q and (k | v) as two products cover the group: taken, the transposed copies at their row offsets
    out = fa.targets([_item(arena, g0, 128, 256), _item(arena, g0 + 128 * 256, 256, 256)])
    assert all(t is not None for t in out)
    assert out[1].transposed == arena.shadow_t.data_ptr() + 2 * (g0 + 128) and out[1].ld_transposed == 384
    # an accumulating product, or two products into the same gradient: not taken
    assert fa.targets([_item(arena, g1, 128, 128, acc=1)]) == [None]
    assert fa.targets([_item(arena, g1, 128, 128), _item(arena, g1, 128, 128)]) == [None, None]
    # a dry (planning) pass works the ranges out and launches the plain form
    fa.dry = True
    assert fa.targets([_item(arena, g1, 128, 128)]) == [None] and fa.ranges == [(g1, g1 + 128 * 128)]


def test_dead_groups_are_left_out_of_the_rest():
    fa, arena, groups = _harness()
    g0, g1, g2 = (g[0] for g in groups)
    fa.ts._dead = [(g1, g1 + 128 * 128)]
    fa.targets([_item(arena, g0, 384, 256)])
    assert fa.rest_groups() == [g2]
    assert fa.rest() == [(g2, g2 + 360 * 128), (arena.small_lo, arena.numel)]
