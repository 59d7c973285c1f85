"""The C-level multi-device layer (include/ldpc_erasure_amd_multi.h), CPU part: its shard arithmetic is the one of sharding.py
(contiguous blocks in rank order), for every total / world size the jobs use and the ragged and empty corner cases."""
import numpy as np

from ldpc_erasure_codes_amd import api, sharding


def test_c_shard_arithmetic_equals_sharding_py():
    for total in (0, 1, 2, 7, 8, 9, 4095, 4096, 4097, 65536, 1000003):
        for world in (1, 2, 3, 4, 8, 13, 64):
            covered = 0
            for rank in range(world):
                f0, cnt = api.shard_frames(total, world, rank)
                assert (f0, cnt) == sharding.shard_frames(total, rank, world), (total, world, rank)
                assert f0 == covered
                covered += cnt
            assert covered == total


def test_shards_are_balanced_and_in_rank_order():
    counts = [api.shard_frames(65536 + 5, 8, r)[1] for r in range(8)]
    assert max(counts) - min(counts) == 1 and counts == sorted(counts, reverse=True)
    assert np.cumsum([0] + counts[:-1]).tolist() == [api.shard_frames(65536 + 5, 8, r)[0] for r in range(8)]
