"""Expert-parallel load balancing: placement properties and weight movement (modelled on
tests/distributed/test_eplb_algo.py and test_eplb_execute.py)."""
import numpy as np
import torch

from vllm_distributed_amd.parallel.eplb import (EplbState, balanced_packing, rearrange_expert_weights,
                                                rebalance_experts, replicate_experts)


def test_balanced_packing_and_replication():
    w = np.array([[9.0, 1, 8, 2, 7, 3, 6, 4]])
    pack, slot = balanced_packing(w, 4)
    loads = [w[0][pack[0] == p].sum() for p in range(4)]
    assert all(l == 10 for l in loads) and sorted(np.bincount(pack[0])) == [2, 2, 2, 2]
    p2l, rank, cnt = replicate_experts(np.array([[100.0, 10, 10, 10]]), 6)
    assert cnt[0].tolist() == [3, 1, 1, 1] and p2l[0].tolist()[:4] == [0, 1, 2, 3] and p2l[0].tolist()[4:] == [0, 0]


def test_rebalance_improves_rank_balance_and_is_consistent():
    rng = np.random.default_rng(0)
    layers, E, R, gpus = 3, 32, 40, 8
    load = rng.pareto(1.2, size=(layers, E)) * 100 + 1
    p2l, l2p, cnt = rebalance_experts(load, R, num_groups=4, num_nodes=2, num_gpus=gpus)
    assert p2l.shape == (layers, R) and cnt.sum(1).tolist() == [R] * layers
    for l in range(layers):
        assert set(p2l[l]) == set(range(E))                   # every logical expert is hosted
        for e in range(E):
            slots = l2p[l, e][l2p[l, e] >= 0]
            assert len(slots) == cnt[l, e] and all(p2l[l, s] == e for s in slots)
    st = EplbState(layers, E, R, gpus, num_groups=4, num_nodes=2)
    before = st.rank_imbalance(load)
    st.phy2log = p2l
    st._index_maps()
    after = st.rank_imbalance(load)
    assert after < 0.5 * before


def test_weight_rearrangement_single_rank_and_topk_mapping():
    layers, E, R = 2, 4, 6
    st = EplbState(layers, E, R, ep_size=1, window=4, step_interval=2)
    w = [[torch.stack([torch.full((3,), float(st.phy2log[l][s])) for s in range(R)])] for l in range(layers)]
    ids = torch.tensor([[0, 1], [0, 2], [0, 3], [0, 1]], dtype=torch.int32)
    for step in range(2):
        for l in range(layers):
            st.record(l, st.map_topk(l, ids))
        changed = st.step_done(None, w)
    assert changed and st.rearrangements == 1
    for l in range(layers):
        # weights follow the placement: slot s holds the weights of its logical expert
        assert [int(w[l][0][s, 0]) for s in range(R)] == st.phy2log[l].tolist()
        assert st.logcnt[l][0] == 3                            # expert 0 got both spare slots
        phys = st.map_topk(l, ids)
        assert torch.equal(torch.as_tensor(st.phy2log[l])[phys.long()].int(), ids)
        assert len(set(phys[:, 0].tolist())) > 1               # its replicas share the tokens


def test_rearrange_between_two_simulated_ranks():
    old = np.array([[0, 1, 2, 3]])
    new = np.array([[0, 0, 2, 1]])
    # world = 1 but per-rank views emulate the algorithm's local-copy branch
    t = [[torch.tensor([[0.0], [1.0], [2.0], [3.0]])]]
    rearrange_expert_weights(old, new, t, None)
    assert t[0][0].flatten().tolist() == [0.0, 0.0, 2.0, 1.0]
