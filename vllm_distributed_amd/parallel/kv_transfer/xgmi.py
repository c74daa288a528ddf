"""Pull-based GPU-to-GPU KV transfer for disaggregated prefill / decode.

Producer (prefill instance): when a request marked ``do_remote_decode`` finishes, its blocks are
kept and the response carries ``kv_transfer_params`` = {remote_block_ids, remote_engine_id,
remote_host, remote_port}.  Each producer worker runs a small side-channel server.
Consumer (decode instance): a request arriving with those params is parked in
WAITING_FOR_REMOTE_KVS while its worker pulls the blocks, then a DONE message lets the producer
free them.

Data planes, chosen per peer at handshake:
  * peer memory (same node): the producer's KV arena is one IPC-shareable allocation; the consumer
    maps it once and copies blocks with the block-transfer kernel straight over xGMI on a side
    stream -- one launch per layer half, no staging, no per-transfer registration;
  * socket (other node, or CPU plumbing tests): the producer reads the blocks and streams the bytes.
Same role and wire contract as the reference's NixlConnector / P2pNcclConnector
(vllm/distributed/kv_transfer/kv_connector/v1/nixl_connector.py, p2p/p2p_nccl_connector.py).
"""
from __future__ import annotations

import os
import socket
import threading
import time
from dataclasses import dataclass, field
from typing import Any, Optional

import torch

from ... import envs
from ...engine.transport import Connection, Listener, connect
from ...logger import init_logger
from .base import KVConnectorBase, KVConnectorMetadata, KVConnectorRole

logger = init_logger(__name__)


@dataclass
class RecvSpec:
    local_block_ids: list[int]
    remote_block_ids: list[int]
    remote_host: str
    remote_port: int
    remote_engine_id: str


@dataclass
class XgmiMetadata(KVConnectorMetadata):
    reqs_to_recv: dict[str, RecvSpec] = field(default_factory=dict)
    reqs_to_send: dict[str, float] = field(default_factory=dict)     # req id -> expiry (producer)


class _Peer:
    def __init__(self, conn: Connection, info: dict, region=None):
        self.conn, self.info, self.region = conn, info, region
        self.lock = threading.Lock()


class XgmiConnector(KVConnectorBase):
    def __init__(self, vllm_config, role: KVConnectorRole):
        super().__init__(vllm_config, role)
        kt = vllm_config.kv_transfer_config
        self.engine_id = kt.engine_id
        self.block_size = vllm_config.cache_config.block_size
        self.host = kt.kv_connector_extra_config.get("side_channel_host", envs.VDA_KV_SIDE_CHANNEL_HOST)
        self.base_port = int(kt.kv_connector_extra_config.get("side_channel_port", kt.kv_port))
        self.timeout = float(envs.VLLM_NIXL_ABORT_REQUEST_TIMEOUT)
        self.tp_size = vllm_config.parallel_config.tensor_parallel_size
        # scheduler side
        self._recv_pending: dict[str, RecvSpec] = {}
        self._send_pending: dict[str, float] = {}
        # worker side
        self.runner = None
        self.listener: Optional[Listener] = None
        self._peers: dict[tuple[str, int], _Peer] = {}
        self._lock = threading.Lock()
        self._done_sending: set[str] = set()
        self._done_recving: set[str] = set()
        self._expiry: dict[str, float] = {}
        self._inflight: list[tuple[str, Any, _Peer]] = []
        self._stream = None
        self._closed = False

    # =========================== scheduler side ===========================
    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> tuple[int, bool]:
        p = request.kv_transfer_params
        if p and p.get("do_remote_prefill"):
            n = len(request.prompt_token_ids) - num_computed_tokens
            if n > 0:
                return n, True
        return 0, False

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int) -> None:
        p = request.kv_transfer_params
        if not p or not p.get("do_remote_prefill"):
            return
        if num_external_tokens > 0 and p.get("remote_block_ids"):
            ids = list(blocks.get_block_ids()[0]) if hasattr(blocks, "get_block_ids") else list(blocks)
            remote = list(p["remote_block_ids"])
            # block i of the request here corresponds to remote block i; the locally cached prefix
            # (whole blocks) needs no transfer
            h = (len(request.prompt_token_ids) - num_external_tokens) // self.block_size
            n = min(len(ids), len(remote))
            self._recv_pending[request.request_id] = RecvSpec(
                local_block_ids=ids[h:n], remote_block_ids=remote[h:n], remote_host=p["remote_host"],
                remote_port=int(p["remote_port"]), remote_engine_id=p.get("remote_engine_id", ""))
        p["do_remote_prefill"] = False

    def build_connector_meta(self, so) -> XgmiMetadata:
        meta = XgmiMetadata(dict(self._recv_pending), dict(self._send_pending))
        self._recv_pending.clear()
        self._send_pending.clear()
        return meta

    def request_finished(self, request, block_ids: list[int]) -> tuple[bool, Optional[dict]]:
        from ...request import RequestStatus
        p = request.kv_transfer_params
        if not p or not p.get("do_remote_decode") or request.status != RequestStatus.FINISHED_LENGTH_CAPPED:
            return False, None
        n_full = request.num_computed_tokens // self.block_size
        n_blocks = -(-request.num_computed_tokens // self.block_size)
        ids = list(block_ids[:n_blocks])
        if not ids:
            return False, None
        self._send_pending[request.request_id] = time.time() + self.timeout
        return True, dict(do_remote_prefill=True, do_remote_decode=False, remote_block_ids=ids,
                          remote_engine_id=self.engine_id, remote_host=self.host, remote_port=self.base_port,
                          tp_size=self.tp_size, num_full_blocks=n_full)

    # ============================= worker side =============================
    def register_kv_caches(self, runner) -> None:
        self.runner = runner
        from ...parallel.state import get_tensor_model_parallel_rank
        self.tp_rank = get_tensor_model_parallel_rank()
        kt = self.vllm_config.kv_transfer_config
        if kt.is_kv_producer and self.listener is None:
            self.listener = Listener(f"tcp://{self.host}:{self.base_port + self.tp_rank}")
            threading.Thread(target=self._serve, daemon=True, name="kv-side-channel").start()
            logger.info("KV side channel listening on %s (engine %s)", self.listener.address, self.engine_id)
        if runner.is_gpu:
            self._stream = torch.cuda.Stream(device=runner.device)

    def _layout(self) -> dict:
        r = self.runner
        regions = r.kv_regions
        return {"engine_id": self.engine_id, "chunk_bytes": [int(c.numel()) for c in r.kv_chunks],      # full region sizes (may be padded)
                "num_blocks": int(r.kv_cache_config.num_blocks),
                "layers": {n: [int(ci), int(off), int(half)] for n, (ci, off, half) in r.kv_layer_loc.items()},
                "ipc_handles": [x.handle for x in regions] if regions else None,
                "device_index": r.device.index if r.is_gpu else -1, "hostname": socket.gethostname(),
                "pid": os.getpid(), "block_size": self.block_size}

    # ---- producer: side channel ------------------------------------------------------------------
    def _serve(self) -> None:
        while not self._closed:
            try:
                c = self.listener.accept(timeout=0.5)
            except OSError:
                return
            if c is not None:
                threading.Thread(target=self._serve_conn, args=(c,), daemon=True, name="kv-side-conn").start()

    def _serve_conn(self, c: Connection) -> None:
        try:
            while True:
                msg = c.recv()
                kind = msg[0]
                if kind == "HANDSHAKE":
                    c.send(self._layout())
                elif kind == "READ":
                    c.send(self._read_blocks(list(msg[1])))
                elif kind == "DONE":
                    with self._lock:
                        self._done_sending.add(msg[1])
                        self._expiry.pop(msg[1], None)
        except (EOFError, OSError):
            pass

    @staticmethod
    def _block_views(chunks: list, layout_layers: dict, num_blocks: int):
        """Per layer: (k rows [num_blocks, bytes_per_block], v rows [...]) as uint8 views of the chunks."""
        out = {}
        for n, (ci, off, half) in layout_layers.items():
            per = half // num_blocks
            c = chunks[ci]
            out[n] = (c[off: off + half].view(num_blocks, per), c[off + half: off + 2 * half].view(num_blocks, per))
        return out

    def _my_views(self):
        r = self.runner
        return self._block_views(r.kv_chunks, r.kv_layer_loc, r.kv_cache_config.num_blocks)

    def _read_blocks(self, block_ids: list[int]) -> dict:
        r = self.runner
        idx = torch.tensor(block_ids, dtype=torch.int64, device=r.device)
        views = self._my_views()
        return {n: [k.index_select(0, idx).cpu(), v.index_select(0, idx).cpu()] for n, (k, v) in views.items()}

    # ---- consumer: pulling ------------------------------------------------------------------------
    def _peer(self, host: str, port: int) -> _Peer:
        key = (host, port + self.tp_rank)
        p = self._peers.get(key)
        if p is not None:
            return p
        conn = connect(f"tcp://{key[0]}:{key[1]}", timeout=30)
        conn.send(("HANDSHAKE",))
        info = conn.recv()
        region = None
        r = self.runner
        same_node = info["hostname"] == socket.gethostname()
        if r.is_gpu and same_node and info.get("ipc_handles") and info["pid"] != os.getpid():
            from ...device_allocator import open_shared
            region = [open_shared(h, n, r.device) for h, n in zip(info["ipc_handles"], info["chunk_bytes"])]
            logger.info("mapped the KV cache of engine %s (GPU %s, %d chunks) over xGMI", info["engine_id"],
                        info["device_index"], len(region))
        p = self._peers[key] = _Peer(conn, info, region)
        return p

    def start_load_kv(self) -> None:
        meta = self._meta
        if not isinstance(meta, XgmiMetadata):
            return
        now = time.time()
        with self._lock:
            for rid, exp in meta.reqs_to_send.items():
                if rid not in self._done_sending:
                    self._expiry[rid] = exp
        for rid, spec in meta.reqs_to_recv.items():
            if not spec.local_block_ids:
                with self._lock:
                    self._done_recving.add(rid)
                continue
            peer = self._peer(spec.remote_host, spec.remote_port)
            if peer.region is not None:
                self._pull_peer_memory(rid, spec, peer)
            else:
                threading.Thread(target=self._pull_socket, args=(rid, spec, peer), daemon=True,
                                 name="kv-pull").start()
        del now

    def _pull_peer_memory(self, rid: str, spec: RecvSpec, peer: _Peer) -> None:
        from ... import ops
        r = self.runner
        dev = r.device
        dst_ids = torch.tensor(spec.local_block_ids, dtype=torch.int64, device=dev)
        src_ids = torch.tensor(spec.remote_block_ids, dtype=torch.int64, device=dev)
        mine = self._my_views()
        theirs = self._block_views([x.tensor for x in peer.region], peer.info["layers"], peer.info["num_blocks"])
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            for n, (k, v) in mine.items():
                rk, rv = theirs[n]
                ops.transfer_blocks(k, rk, dst_ids, src_ids)
                ops.transfer_blocks(v, rv, dst_ids, src_ids)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._inflight.append((rid, ev, peer))

    def _pull_socket(self, rid: str, spec: RecvSpec, peer: _Peer) -> None:
        try:
            r = self.runner
            with peer.lock:
                peer.conn.send(("READ", spec.remote_block_ids))
                data = peer.conn.recv()
            idx = torch.tensor(spec.local_block_ids, dtype=torch.int64, device=r.device)
            mine = self._my_views()
            for n, (k, v) in mine.items():
                sk, sv = data[n]
                k.index_copy_(0, idx, sk.to(k.device))
                v.index_copy_(0, idx, sv.to(v.device))
            if r.is_gpu:
                torch.cuda.synchronize(r.device)
            with peer.lock:
                peer.conn.send(("DONE", rid))
        except Exception:   # noqa: BLE001
            logger.exception("pulling KV of request %s failed", rid)
        with self._lock:
            self._done_recving.add(rid)

    def get_finished(self, finished_req_ids: set[str]):
        still = []
        for rid, ev, peer in self._inflight:
            if ev.query():
                with peer.lock:
                    peer.conn.send(("DONE", rid))
                with self._lock:
                    self._done_recving.add(rid)
            else:
                still.append((rid, ev, peer))
        self._inflight = still
        now = time.time()
        with self._lock:
            for rid, exp in list(self._expiry.items()):
                if now > exp:
                    logger.warning("KV blocks of request %s were never fetched; releasing them", rid)
                    self._done_sending.add(rid)
                    del self._expiry[rid]
            sending, self._done_sending = self._done_sending, set()
            recving, self._done_recving = self._done_recving, set()
        if self.tp_size > 1:
            sending, recving = self._all_ranks_agree(sending, recving)
        return sending or None, recving or None

    def _all_ranks_agree(self, sending: set[str], recving: set[str]):
        """With TP every rank moves its own shard: report a request only once all ranks are done."""
        from ...parallel.state import get_tp_group
        g = get_tp_group()
        if not hasattr(self, "_partial"):
            self._partial = ({}, {})
        outs = []
        for mine, counts in ((sending, self._partial[0]), (recving, self._partial[1])):
            everyone = g.all_gather_object(sorted(mine))
            for ids in everyone:
                for rid in ids:
                    counts[rid] = counts.get(rid, 0) + 1
            done = {rid for rid, c in counts.items() if c >= g.world_size}
            for rid in done:
                del counts[rid]
            outs.append(done)
        return outs[0], outs[1]

    def shutdown(self) -> None:
        self._closed = True
        if self.listener is not None:
            self.listener.close()
        for p in self._peers.values():
            p.conn.close()
            for x in p.region or ():
                x.close()


def _register_wire_types() -> None:
    from ...engine import serial
    from . import multi, shared_storage
    for i, cls in enumerate((RecvSpec, XgmiMetadata, shared_storage._Req, shared_storage.SharedStorageMetadata,
                             multi.MultiMetadata), start=60):
        serial.wire_struct(i)(cls)


_register_wire_types()
