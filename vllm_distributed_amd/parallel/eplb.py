"""Expert-parallel load balancing (EPLB).

Routed experts are *logical*; each rank of the EP group hosts a fixed number of *physical* expert
slots.  With ``num_redundant_experts`` extra slots, hot experts get replicas.  Every
``step_interval`` engine steps the observed token counts (sliding window, all-reduced over the EP
group) are turned into a new slot assignment that (1) replicates the hottest experts and (2) packs
slots onto ranks so that per-rank load is even, hierarchically (expert groups -> nodes -> GPUs) when
the model routes by groups, so that group-limited routing keeps its traffic inside a node; then
expert weights move between ranks point-to-point (one batched isend/irecv per layer -- on xGMI every
pair of GPUs has its own link, so the exchange is contention-free) and the layers' logical->physical
maps are swapped.
Reference: vllm/distributed/eplb/{eplb_state.py, rebalance_algo.py, rebalance_execute.py}.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from ..logger import init_logger

logger = init_logger(__name__)


# ---- placement algorithm (numpy, deterministic: every rank computes the same answer) ----------------
def balanced_packing(weight: np.ndarray, num_packs: int) -> tuple[np.ndarray, np.ndarray]:
    """Each row: n items into ``num_packs`` packs of exactly n/num_packs items with even total weight
    (heaviest first into the lightest pack that still has room).  Returns (pack index, rank inside pack)."""
    rows, n = weight.shape
    assert n % num_packs == 0
    cap = n // num_packs
    pack = np.empty((rows, n), dtype=np.int64)
    slot = np.empty((rows, n), dtype=np.int64)
    if cap == 1:
        pack[:] = np.arange(n)
        slot[:] = 0
        return pack, slot
    order = np.argsort(-weight, axis=1, kind="stable")
    for r in range(rows):
        load = np.zeros(num_packs)
        count = np.zeros(num_packs, dtype=np.int64)
        for item in order[r]:
            open_packs = np.nonzero(count < cap)[0]
            p = open_packs[np.argmin(load[open_packs])]
            pack[r, item], slot[r, item] = p, count[p]
            load[p] += weight[r, item]
            count[p] += 1
    return pack, slot


def replicate_experts(weight: np.ndarray, num_phy: int) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Each row: give the ``num_phy - num_log`` spare slots, one at a time, to the expert whose
    per-replica load is highest.  Returns (phy -> log, replica rank of each phy, replica count per log)."""
    rows, num_log = weight.shape
    assert num_phy >= num_log
    phy2log = np.tile(np.arange(num_phy), (rows, 1))
    rank = np.zeros((rows, num_phy), dtype=np.int64)
    count = np.ones((rows, num_log), dtype=np.int64)
    ar = np.arange(rows)
    for i in range(num_log, num_phy):
        hot = np.argmax(weight / count, axis=1)
        phy2log[:, i] = hot
        rank[:, i] = count[ar, hot]
        count[ar, hot] += 1
    return phy2log, rank, count


def _inverse(perm: np.ndarray) -> np.ndarray:
    inv = np.empty_like(perm)
    np.put_along_axis(inv, perm, np.tile(np.arange(perm.shape[1]), (perm.shape[0], 1)), axis=1)
    return inv


def _hierarchical(weight: np.ndarray, num_phy: int, num_groups: int, num_nodes: int, num_gpus: int):
    rows, num_log = weight.shape
    gsize = num_log // num_groups
    groups_per_node = num_groups // num_nodes
    gpus_per_node = num_gpus // num_nodes
    phy_per_gpu = num_phy // num_gpus
    # 1. expert groups -> nodes
    gw = weight.reshape(rows, num_groups, gsize).sum(-1)
    g_pack, g_slot = balanced_packing(gw, num_nodes)
    log2mlog = (((g_pack * groups_per_node + g_slot) * gsize)[:, :, None] + np.arange(gsize)).reshape(rows, num_log)
    mlog2log = _inverse(log2mlog)
    # 2. replicate inside each node
    w_node = np.take_along_axis(weight, mlog2log, axis=1).reshape(rows * num_nodes, num_log // num_nodes)
    phy2mlog, phyrank, mlogcnt = replicate_experts(w_node, num_phy // num_nodes)
    # 3. physical slots -> GPUs inside each node
    w_phy = np.take_along_axis(w_node / mlogcnt, phy2mlog, axis=1)
    p_pack, p_slot = balanced_packing(w_phy, gpus_per_node)
    phy2pphy = p_pack * phy_per_gpu + p_slot
    pphy2phy = _inverse(phy2pphy)
    pphy2mlog = np.take_along_axis(phy2mlog, pphy2phy, axis=1)
    pphy2mlog = (pphy2mlog.reshape(rows, num_nodes, -1) +
                 (np.arange(num_nodes) * (num_log // num_nodes))[None, :, None]).reshape(rows, -1)
    pphy2log = np.take_along_axis(mlog2log, pphy2mlog, axis=1)
    pphyrank = np.take_along_axis(phyrank, pphy2phy, axis=1).reshape(rows, -1)
    logcnt = np.take_along_axis(mlogcnt.reshape(rows, -1), log2mlog, axis=1)
    return pphy2log, pphyrank, logcnt


def rebalance_experts(weight, num_replicas: int, num_groups: int, num_nodes: int, num_gpus: int):
    """weight [layers, logical experts] -> (phy2log [layers, num_replicas], log2phy [layers, logical, max
    replicas] padded with -1, replica count per logical expert)."""
    w = np.asarray(weight, dtype=np.float64)
    rows, num_log = w.shape
    if num_groups % num_nodes == 0 and num_groups > 1:
        phy2log, phyrank, logcnt = _hierarchical(w, num_replicas, num_groups, num_nodes, num_gpus)
    else:
        phy2log, phyrank, logcnt = _hierarchical(w, num_replicas, 1, 1, num_gpus)
    max_rep = int(logcnt.max())
    log2phy = -np.ones((rows, num_log, max_rep), dtype=np.int64)
    for r in range(rows):
        log2phy[r, phy2log[r], phyrank[r]] = np.arange(num_replicas)
    return phy2log, log2phy, logcnt


# ---- weight movement --------------------------------------------------------------------------------
def rearrange_expert_weights(old_phy2log: np.ndarray, new_phy2log: np.ndarray, weights: Sequence[Sequence[torch.Tensor]],
                             group, rank: Optional[int] = None) -> None:
    """Move expert weights in place.  ``weights[layer]`` is a list of tensors whose first dimension
    indexes the LOCAL physical slots of this rank; ``*_phy2log [layers, global slots]``.  A slot whose new
    logical expert already sits somewhere on this rank is copied locally; the rest is received from the
    first rank that holds the expert under the old assignment."""
    world = dist.get_world_size(group) if group is not None else 1
    rank = (dist.get_rank(group) if group is not None else 0) if rank is None else rank
    layers, num_phy = old_phy2log.shape
    per = num_phy // world
    lo, hi = rank * per, (rank + 1) * per
    for layer in range(layers):
        old, new = old_phy2log[layer], new_phy2log[layer]
        if np.array_equal(old, new):
            continue
        tensors = weights[layer]
        snapshot = [t.clone() for t in tensors]          # sources are read from the pre-move state
        first_holder: dict[int, int] = {}
        for slot in range(num_phy):
            first_holder.setdefault(int(old[slot]), slot)
        ops, pending = [], []
        # receives / local copies for my new slots
        for slot in range(lo, hi):
            e = int(new[slot])
            if int(old[slot]) == e:
                continue
            local_src = next((s for s in range(lo, hi) if int(old[s]) == e), None)
            if local_src is not None:
                for t, snap in zip(tensors, snapshot):
                    t[slot - lo].copy_(snap[local_src - lo])
                continue
            src_slot = first_holder[e]
            src_rank = src_slot // per
            for t in tensors:
                buf = torch.empty_like(t[slot - lo])
                ops.append(dist.P2POp(dist.irecv, buf, dist.get_global_rank(group, src_rank), group))
                pending.append((t, slot - lo, buf))
        # sends: for every remote slot that will fetch from one of my old slots
        for slot in range(num_phy):
            if lo <= slot < hi:
                continue
            e = int(new[slot])
            if int(old[slot]) == e:
                continue
            dst_rank = slot // per
            dlo, dhi = dst_rank * per, (dst_rank + 1) * per
            if any(int(old[s]) == e for s in range(dlo, dhi)):
                continue                                   # the destination copies locally
            src_slot = first_holder[e]
            if lo <= src_slot < hi:
                for snap in snapshot:
                    ops.append(dist.P2POp(dist.isend, snap[src_slot - lo].contiguous(),
                                          dist.get_global_rank(group, dst_rank), group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for t, idx, buf in pending:
            t[idx].copy_(buf)


# ---- run-time state ------------------------------------------------------------------------------------
@dataclass
class EplbState:
    num_layers: int
    num_logical: int
    num_physical: int
    ep_size: int
    window: int = 1000
    step_interval: int = 3000
    num_groups: int = 1
    num_nodes: int = 1
    phy2log: np.ndarray = field(init=False)
    log2phy: np.ndarray = field(init=False)
    logcnt: np.ndarray = field(init=False)

    def __post_init__(self):
        assert self.num_physical % self.ep_size == 0
        base = np.arange(self.num_physical) % self.num_logical
        self.phy2log = np.tile(base, (self.num_layers, 1))
        self._index_maps()
        self.load_window = np.zeros((self.window, self.num_layers, self.num_physical), dtype=np.int64)
        self.pass_load = torch.zeros((self.num_layers, self.num_physical), dtype=torch.int64)
        self._dev, self._maps_stale = None, True
        self.step = 0
        self.rearrangements = 0

    def _index_maps(self) -> None:
        cnt = np.zeros((self.num_layers, self.num_logical), dtype=np.int64)
        for layer in range(self.num_layers):
            np.add.at(cnt[layer], self.phy2log[layer], 1)
        mx = int(cnt.max())
        self.log2phy = -np.ones((self.num_layers, self.num_logical, mx), dtype=np.int64)
        seen = np.zeros((self.num_layers, self.num_logical), dtype=np.int64)
        for layer in range(self.num_layers):
            for slot, e in enumerate(self.phy2log[layer]):
                self.log2phy[layer, e, seen[layer, e]] = slot
                seen[layer, e] += 1
        self.logcnt = cnt

    def _device_maps(self, device: torch.device) -> None:
        """Routing tables and the load counters as persistent tensors on the model's device: updated in
        place after a rearrangement, so captured hipGraphs keep reading the current placement and the
        per-step bookkeeping needs no host round trip."""
        width = self.num_physical - self.num_logical + 1          # most replicas one expert can have
        if self._dev is None or self._dev != device:
            self._dev = device
            self._l2p_dev = torch.zeros((self.num_layers, self.num_logical, width), dtype=torch.int64, device=device)
            self._cnt_dev = torch.ones((self.num_layers, self.num_logical), dtype=torch.int64, device=device)
            self.pass_load = self.pass_load.to(device)
            self._maps_stale = True
        if self._maps_stale:
            l2p = np.zeros((self.num_layers, self.num_logical, width), dtype=np.int64)
            l2p[:, :, :self.log2phy.shape[2]] = np.maximum(self.log2phy, 0)
            self._l2p_dev.copy_(torch.from_numpy(l2p))
            self._cnt_dev.copy_(torch.from_numpy(self.logcnt))
            self._maps_stale = False

    def map_topk(self, layer: int, topk_ids: torch.Tensor) -> torch.Tensor:
        """Logical expert ids -> physical slot ids; replicas are chosen round-robin by token index."""
        self._device_maps(topk_ids.device)
        ids = topk_ids.long()
        tok = torch.arange(ids.shape[0], device=ids.device).unsqueeze(1)
        rep = tok % self._cnt_dev[layer][ids]
        return self._l2p_dev[layer][ids, rep].to(topk_ids.dtype)

    def record(self, layer: int, phys_ids: torch.Tensor) -> None:
        """Count the routed pairs per physical slot (negative ids = padding / not routed) on the device."""
        self._device_maps(phys_ids.device)
        flat = phys_ids.reshape(-1).long()
        self.pass_load[layer].index_add_(0, flat.clamp(min=0), (flat >= 0).to(torch.int64))

    def step_done(self, group=None, weights=None, p2p_group=None) -> bool:
        """Call once per engine step; returns True when a rearrangement happened.  ``group`` carries the
        (host) load all-reduce, ``p2p_group`` the weight exchange (defaults to ``group``)."""
        load = self.pass_load.cpu().clone()
        if group is not None and dist.get_world_size(group) > 1:
            dist.all_reduce(load, group=group)
        self.load_window[self.step % self.window] = load.numpy()
        self.pass_load.zero_()
        self.step += 1
        if self.step % self.step_interval:
            return False
        self.rearrange(p2p_group if p2p_group is not None else group, weights)
        return True

    def rearrange(self, group=None, weights=None) -> None:
        phys = self.load_window.sum(0)                                   # [layers, physical]
        logical = np.zeros((self.num_layers, self.num_logical))
        for layer in range(self.num_layers):
            np.add.at(logical[layer], self.phy2log[layer], phys[layer])
        new_p2l, _, _ = rebalance_experts(logical, self.num_physical, self.num_groups, self.num_nodes, self.ep_size)
        if weights is not None:
            rearrange_expert_weights(self.phy2log, new_p2l, weights, group)
        self.phy2log = new_p2l
        self._index_maps()
        self._maps_stale = True
        self.rearrangements += 1
        logger.info("EPLB: experts rearranged (#%d)", self.rearrangements)

    def rank_imbalance(self, logical_load: np.ndarray) -> float:
        """max / mean load over EP ranks for the current placement, averaged over layers."""
        per = self.num_physical // self.ep_size
        out = []
        for layer in range(self.num_layers):
            w = logical_load[layer][self.phy2log[layer]] / self.logcnt[layer][self.phy2log[layer]]
            ranks = w.reshape(self.ep_size, per).sum(1)
            out.append(ranks.max() / max(ranks.mean(), 1e-9))
        return float(np.mean(out))
