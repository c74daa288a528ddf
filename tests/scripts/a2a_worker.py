"""Rank program: peer-memory all-to-all-v between processes that share GPU 0 (HIP IPC), checked
bit for bit against the same exchange through gloo, eagerly and under hipGraph replay with changing
counts, plus the expert-parallel dispatch / combine round trip on top of it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

from vllm_distributed_amd import ops
from vllm_distributed_amd.parallel.all2all import ExpertDispatcher, RowExchange
from vllm_distributed_amd.parallel.state import GroupCoordinator


def host_reference(world, rank, cap, x, sorted_ids, offsets, k_div, meta):
    """The contract, spelled out with gloo object exchange."""
    off = offsets.cpu().tolist()
    xs, so = x.cpu(), sorted_ids.cpu().long()
    mine = []
    for d in range(world):
        idx = so[off[d]:off[d + 1]][:cap]
        mine.append((xs[idx // k_div].clone(), meta.cpu()[idx].clone() if meta is not None else None))
    box = [None] * world
    dist.all_gather_object(box, mine)
    recv = torch.zeros((world * cap, x.shape[1]), dtype=x.dtype)
    rmeta = torch.zeros(world * cap, dtype=torch.int32)
    rc = torch.zeros(world, dtype=torch.int32)
    for s in range(world):
        rows, m = box[s][rank]
        n = rows.shape[0]
        recv[s * cap:s * cap + n] = rows
        if m is not None:
            rmeta[s * cap:s * cap + n] = m
        rc[s] = n
    return recv, rmeta, rc


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = GroupCoordinator(list(range(world)), rank, 0, "gloo+hip", "ep_test")
    g._attach(dist.group.WORLD, dist.group.WORLD)
    W, cap = 1024, 96
    rx = RowExchange(g, cap, W * 2)
    assert rx.peer is not None, "peer-memory all-to-all self-test failed"
    gen = torch.Generator().manual_seed(100 + rank)
    for it in range(6):
        n_rows = int(torch.randint(1, 200, (1,), generator=gen))
        k_div = 1 + it % 2
        x = torch.randn((n_rows, W), generator=gen).to(torch.bfloat16).to(dev)
        dest = torch.randint(-1, world, (n_rows * k_div,), generator=gen, dtype=torch.int32).to(dev)
        meta = torch.randint(0, 1000, (n_rows * k_div,), generator=gen, dtype=torch.int32).to(dev)
        sorted_ids, offsets = ops.sort_by_key(dest, world)
        c = cap if it % 3 else 64
        recv, rmeta, rc = rx.exchange(x, sorted_ids, offsets, k_div, meta, c)
        torch.cuda.synchronize()
        ref, ref_meta, ref_rc = host_reference(world, rank, c, x, sorted_ids, offsets, k_div, meta)
        assert torch.equal(rc.cpu(), ref_rc), (it, rc, ref_rc)
        for s in range(world):
            n = int(ref_rc[s])
            assert torch.equal(recv[s * c:s * c + n].cpu(), ref[s * c:s * c + n]), (it, s)
            assert torch.equal(rmeta[s * c:s * c + n].cpu(), ref_meta[s * c:s * c + n]), (it, s)
        # the way back: every row returns to its sender, doubled
        back, bc = rx.exchange_regions(recv * 2, rc, c)
        torch.cuda.synchronize()
        off = offsets.cpu().tolist()
        so = sorted_ids.cpu().long()
        for d in range(world):
            idx = so[off[d]:off[d + 1]][:c]
            assert int(bc[d]) == idx.numel()
            assert torch.equal(back[d * c:d * c + idx.numel()].cpu(), (x.cpu()[idx // k_div] * 2)), (it, d)
    assert rx.check_error() == 0

    # hipGraph: capture one exchange + return trip, replay with other rows and counts
    n_rows = 64
    x = torch.zeros((n_rows, W), dtype=torch.bfloat16, device=dev)
    dest = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        so, of = ops.sort_by_key(dest, world)
        rx.exchange(x, so, of, 1, None, 64)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        so, of = ops.sort_by_key(dest, world)
        recv, _, rc = rx.exchange(x, so, of, 1, None, 64)
        back, bc = rx.exchange_regions(recv, rc, 64)
    for it in range(4):
        x.copy_(torch.randn((n_rows, W), generator=gen).to(torch.bfloat16))
        dest.copy_(torch.randint(0, world, (n_rows,), generator=gen, dtype=torch.int32))
        dist.barrier()
        graph.replay()
        torch.cuda.synchronize()
        offs, srt = of.cpu().tolist(), so.cpu().long()
        for d in range(world):
            idx = srt[offs[d]:offs[d + 1]]
            assert torch.equal(back[d * 64:d * 64 + idx.numel()].cpu(), x.cpu()[idx]), (it, d)
    assert rx.check_error() == 0

    # expert-parallel round trip: each rank routes its own tokens, owners run their experts
    E, k, H, I = 4 * world, 2, 256, 128
    gw = torch.Generator().manual_seed(7)
    w13 = (torch.randn((E, 2 * I, H), generator=gw) * 0.05).to(torch.bfloat16).to(dev)
    w2 = (torch.randn((E, H, I), generator=gw) * 0.05).to(torch.bfloat16).to(dev)
    epr = E // world
    disp = ExpertDispatcher(g, epr, k, H * 2, 80)
    for T in (1, 37, 80):
        xt = torch.randn((T, H), generator=gen).to(torch.bfloat16).to(dev)
        tw, ids = ops.moe_topk_softmax(torch.randn((T, E), generator=gen).to(dev), k, True)
        if T == 37:
            ids[3, 1] = -1                      # a padded pair and an id outside the table go nowhere
            ids[5, 0] = E + 7
        # the bookkeeping kernel against its PyTorch form
        got = ops.ep_prep(ids.reshape(-1), world, epr)
        want = ops.ep_prep(ids.reshape(-1).cpu(), world, epr)
        n_live = int(want[3][world])
        for name, a, b_ in zip(("ids", "dest", "sorted", "offsets", "slot"), got, want):
            a, b_ = a.cpu(), b_.cpu()
            if name == "sorted":
                a, b_ = a[:n_live], b_[:n_live]
            assert torch.equal(a, b_), (T, name, a, b_)
        recv, local, st = disp.dispatch(xt, ids)
        # local expert ids arrive ready-made: -1 past every source's count
        rc_ = st.recv_counts.cpu().tolist()
        lv = local.view(world, st.cap).cpu()
        for s_ in range(world):
            assert bool((lv[s_, rc_[s_]:] == -1).all()) and bool(((lv[s_, :rc_[s_]] >= 0) & (lv[s_, :rc_[s_]] < epr)).all())
        ones = torch.ones((recv.shape[0], 1), dtype=torch.float32, device=dev)
        y = ops.fused_moe(recv, w13[rank * epr:(rank + 1) * epr].contiguous(), w2[rank * epr:(rank + 1) * epr].contiguous(),
                          ones, local)
        out = disp.combine(y, tw, st)
        # the fused return + weighted sum against the separate exchange + PyTorch gather: the same fp32 sum up to the
        # fused multiply-add (one rounding instead of two), i.e. at most one step of the 16-bit result
        peer, disp.rx.peer = disp.rx.peer, None
        try:
            ret, _ = peer.exchange_regions(y, st.recv_counts, st.cap)
            d = st.dest.long()
            pos = (d.clamp(min=0) * st.cap + st.slot.long().clamp(min=0))
            rows = torch.where((d >= 0).view(T, k, 1), ret[pos].view(T, k, -1).float(), torch.zeros((), device=dev))
            plain = (rows * torch.where((d >= 0).view(T, k), tw.float(), torch.zeros((), device=dev)).unsqueeze(-1)).sum(1)
        finally:
            disp.rx.peer = peer
        torch.cuda.synchronize()
        plain = plain.to(out.dtype).float()
        gap = (out.float() - plain).abs()
        assert bool((gap <= plain.abs() * 2 ** -7 + 1e-6).all()), (T, gap.max())
        assert float((gap > 0).float().mean()) < 0.02, T
        ids = torch.where((ids >= 0) & (ids < E), ids, torch.full_like(ids, -1))
        ref = ops.reference.fused_moe(xt, w13, w2, tw, ids)
        err = float((out.float() - ref.float()).norm() / ref.float().norm())
        assert err < 2e-2, (T, err)
    assert disp.rx.check_error() == 0

    # timing (informative): 64 rows of 8 KiB to every peer
    rows = 64 * world
    x = torch.randn((rows, 4096), device=dev, dtype=torch.bfloat16)
    rx2 = RowExchange(g, 64, 8192)
    dest = (torch.arange(rows, device=dev, dtype=torch.int32) % world)
    so, of = ops.sort_by_key(dest, world)
    for _ in range(3):
        rx2.exchange(x, so, of)
    torch.cuda.synchronize(); dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rx2.exchange(x, so, of)
    e1.record(); torch.cuda.synchronize()
    if rank == 0:
        print(f"all-to-all {rows} rows x 8 KiB: {e0.elapsed_time(e1) / 20 * 1000:.1f} us (same-GPU peers)", flush=True)
    dist.barrier()
    rx.close(); rx2.close(); disp.rx.close()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
