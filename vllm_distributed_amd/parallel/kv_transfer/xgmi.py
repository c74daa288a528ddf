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
Producer and consumer may run different tensor-parallel degrees: KV heads are the outer dimension
of a cache block, so a consumer rank copies exactly the byte range of its heads out of every block of
the producer rank(s) that hold them (``head_segments``); tensors not sharded by head (latent caches,
recurrent states) are copied whole from one producer rank.
Everything a client can influence is checked before it reaches a raw-pointer kernel: the handshake
compares block size, layer set and bytes per head, block ids must lie inside both caches and come in
equal numbers, and the producer only serves (or approves, for one-sided pulls) ids it exported for that
request.
Same role and wire contract as the reference's NixlConnector / P2pNcclConnector
(vllm/distributed/kv_transfer/kv_connector/v1/nixl_connector.py :500-501, :695-810 heterogeneous TP,
:936-1030 reads; p2p/p2p_nccl_connector.py).
"""
from __future__ import annotations

import hashlib
import hmac
import secrets
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
    remote_token: str = ""        # the producer's signature over the block list it exported
    remote_export: list[int] = field(default_factory=list)     # that whole list (remote_block_ids is a slice)
    remote_first: int = 0                                      # ... starting at this position


@dataclass
class XgmiMetadata(KVConnectorMetadata):
    reqs_to_recv: dict[str, RecvSpec] = field(default_factory=dict)
    reqs_to_send: dict[str, float] = field(default_factory=dict)     # req id -> expiry (producer)
    send_nonces: dict[str, str] = field(default_factory=dict)        # req id -> nonce of its export token


def head_range(rank: int, tp: int, total_heads: int) -> tuple[int, int]:
    """Global KV heads [lo, hi) held by tensor-parallel rank ``rank`` (heads are replicated once tp
    exceeds their number)."""
    if total_heads >= tp:
        per = total_heads // tp
        return rank * per, (rank + 1) * per
    h = rank // (tp // total_heads)
    return h, h + 1


def head_segments(my_rank: int, my_tp: int, peer_tp: int, total_heads: int) -> list[tuple[int, int, int, int]]:
    """Which producer ranks hold this consumer rank's KV heads: [(peer rank, first head inside the peer's
    block, first head inside my block, number of heads)].  Among replicas of a head the peer is chosen by
    my rank, so pulls spread over the replicas."""
    lo, hi = head_range(my_rank, my_tp, total_heads)
    out = []
    h = lo
    while h < hi:
        if total_heads >= peer_tp:
            per = total_heads // peer_tp
            p = h // per
            plo, phi = p * per, (p + 1) * per
        else:
            rep = peer_tp // total_heads
            p = h * rep + my_rank % rep
            plo, phi = h, h + 1
        n = min(hi, phi) - h
        out.append((p, h - plo, h - lo, n))
        h += n
    return out


class _Peer:
    def __init__(self, conn: Connection, info: dict, region=None):
        self.conn, self.info, self.region = conn, info, region
        self.views = None             # per-layer views of the mapped cache (peer-memory plane)
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
        self._send_nonces: dict[str, str] = {}                      # scheduler: nonce of each pending export
        self._nonce_of: dict[str, str] = {}                         # worker: req id -> nonce of its live export
        self._revoked: dict[str, float] = {}                        # worker: nonce -> token expiry, blocks released
        self._key = str(kt.kv_connector_extra_config.get("auth_key", "")).encode()
        # worker side
        self.runner = None
        self.listener: Optional[Listener] = None
        self._peers: dict[tuple[str, int], _Peer] = {}
        self._lock = threading.Lock()
        self._done_sending: set[str] = set()
        self._done_recving: set[str] = set()
        self._expiry: dict[str, float] = {}
        self._inflight: list[tuple[str, Any, list]] = []
        self._stream = None
        self._closed = False
        self._views = None
        self._done_votes: dict[str, int] = {}        # producer: DONE notices received so far
        self._outstanding: set[str] = set()          # transfers announced by the scheduler and not yet reported
        self._failed_recving: set[str] = set()     # set by pull threads
        self._failed_pending: set[str] = set()     # known failures not yet reported with their request
        self._failed_report: set[str] = set()      # failures among the imports the last get_finished reported
        self.tp_rank = 0
        self.total_kv_heads = 1

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
            if len(remote) < len(ids) - 1 or any((not isinstance(b, int)) or b < 0 for b in remote):
                # fewer exported blocks than the prompt needs, or ids that cannot be block ids: nothing is
                # pulled, the prompt is computed here (the request arrived with made-up parameters)
                logger.warning("request %s: kv_transfer_params do not describe its prompt; prefilling locally",
                               request.request_id)
                p["do_remote_prefill"] = False
                return
            n = min(len(ids), len(remote))
            self._recv_pending[request.request_id] = RecvSpec(
                local_block_ids=ids[h:n], remote_block_ids=remote[h:n], remote_host=p["remote_host"],
                remote_port=int(p["remote_port"]), remote_engine_id=p.get("remote_engine_id", ""),
                remote_token=str(p.get("remote_token", "")), remote_export=remote, remote_first=h)
        p["do_remote_prefill"] = False

    def _sign(self, nonce: str, expiry: int, block_ids: list) -> str:
        """Export token ``nonce.expiry.mac``: the MAC covers the nonce of this export, its expiry and the block list
        in order.  Only block lists this engine handed out are served (a request cannot read another request's blocks
        by naming their ids), a token dies with its export (the nonce is revoked when the blocks are released) and at
        its expiry at the latest, so replaying it cannot read the blocks' next owner.  The decode instance sees the
        request under the id the proxy gave it, so the request id is not part of the message; the nonce takes its
        place."""
        msg = f"{nonce}|{int(expiry)}|".encode() + ",".join(str(int(b)) for b in block_ids).encode()
        return f"{nonce}.{int(expiry)}." + hmac.new(self._key, msg, hashlib.sha256).hexdigest()

    def build_connector_meta(self, so) -> XgmiMetadata:
        meta = XgmiMetadata(dict(self._recv_pending), dict(self._send_pending), dict(self._send_nonces))
        self._recv_pending.clear()
        self._send_pending.clear()
        self._send_nonces.clear()
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
        deadline = time.time() + self.timeout
        nonce = secrets.token_hex(8)
        self._send_pending[request.request_id] = deadline
        self._send_nonces[request.request_id] = nonce
        return True, dict(do_remote_prefill=True, do_remote_decode=False, remote_block_ids=ids,
                          remote_token=self._sign(nonce, int(deadline) + 1, ids),
                          remote_engine_id=self.engine_id, remote_host=self.host, remote_port=self.base_port,
                          tp_size=self.tp_size, num_full_blocks=n_full)

    # ============================= worker side =============================
    def register_kv_caches(self, runner) -> None:
        self.runner = runner
        from ...parallel.state import get_tensor_model_parallel_rank
        self.tp_rank = get_tensor_model_parallel_rank()
        self.total_kv_heads = int(self.vllm_config.model_config.get_total_num_kv_heads())
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
        nb = int(r.kv_cache_config.num_blocks)
        return {"engine_id": self.engine_id, "chunk_bytes": [int(c.numel()) for c in r.kv_chunks],      # full region sizes (may be padded)
                "num_blocks": nb,
                "layers": {n: [[int(ci), int(off), int(nbytes), int(heads)] for ci, off, nbytes, heads in parts]
                           for n, parts in r.kv_layer_parts.items()},
                "ipc_handles": [x.handle for x in regions] if regions else None,
                "device_index": r.device.index if r.is_gpu else -1, "hostname": socket.gethostname(),
                "pid": os.getpid(), "block_size": self.block_size, "tp_size": self.tp_size, "tp_rank": self.tp_rank,
                "total_kv_heads": self.total_kv_heads}

    # ---- producer: side channel ------------------------------------------------------------------
    def _serve(self) -> None:
        while not self._closed:
            try:
                c = self.listener.accept(timeout=0.5)
            except OSError:
                return
            if c is not None:
                threading.Thread(target=self._serve_conn, args=(c,), daemon=True, name="kv-side-conn").start()

    def _owns(self, token: str, block_ids: list) -> bool:
        """True when ``block_ids`` is (a tail of) a block list this engine exported: the token is the
        signature of the full exported list, the consumer names how many leading blocks it skipped."""
        nb = int(self.runner.kv_cache_config.num_blocks)
        if not all(isinstance(b, int) and 0 <= b < nb for b in block_ids):
            return False
        parts = str(token).split(".")
        if len(parts) != 3 or not parts[1].isdigit():
            return False
        nonce, expiry = parts[0], int(parts[1])
        now = time.time()
        with self._lock:
            for k in [k for k, e in self._revoked.items() if e < now]:
                del self._revoked[k]                    # past its expiry a token is refused anyway
            if nonce in self._revoked:
                return False
        if now > expiry:
            return False
        return hmac.compare_digest(self._sign(nonce, expiry, block_ids), str(token))

    def _revoke(self, rid: str) -> None:
        """The blocks of ``rid`` are being released: its export token must not open them again (lock held)."""
        nonce = self._nonce_of.pop(rid, None)
        if nonce is not None:
            self._revoked[nonce] = time.time() + self.timeout + 2

    def _serve_conn(self, c: Connection) -> None:
        try:
            while True:
                msg = c.recv()
                kind = msg[0]
                if kind == "HANDSHAKE":
                    c.send(self._layout())
                elif kind == "CHECK":                    # one-sided pull ahead: is this an exported block list?
                    c.send(bool(self._owns(msg[1], list(msg[2]))))
                elif kind == "READ":                     # (token, exported list, positions of it to send)
                    full, take = list(msg[2]), list(msg[3])
                    ok = self._owns(msg[1], full) and all(isinstance(i, int) and 0 <= i < len(full) for i in take)
                    c.send(self._read_blocks([full[i] for i in take]) if ok else None)
                elif kind == "DONE":
                    rid, want = msg[1], int(msg[2]) if len(msg) > 2 else 1
                    with self._lock:
                        got = self._done_votes.get(rid, 0) + 1
                        if got >= want:          # every consumer rank that reads from this rank has finished
                            self._done_votes.pop(rid, None)
                            self._done_sending.add(rid)
                            self._expiry.pop(rid, None)
                            self._revoke(rid)
                        else:
                            self._done_votes[rid] = got
        except (EOFError, OSError):
            pass

    @staticmethod
    def _part_views(chunks: list, layers: dict, num_blocks: int) -> dict:
        """Per layer: one uint8 view [num_blocks, bytes per block] per per-block tensor, with its head count."""
        out = {}
        for n, parts in layers.items():
            vs = []
            for ci, off, nbytes, heads in parts:
                if nbytes % num_blocks or off < 0 or off + nbytes > chunks[ci].numel():
                    raise ValueError(f"layer {n}: cache part does not fit its region")
                vs.append((chunks[ci][off: off + nbytes].view(num_blocks, nbytes // num_blocks), int(heads)))
            out[n] = vs
        return out

    def _my_views(self):
        r = self.runner
        if self._views is None:
            self._views = self._part_views(r.kv_chunks, r.kv_layer_parts, int(r.kv_cache_config.num_blocks))
        return self._views

    def _read_blocks(self, block_ids: list[int]) -> dict:
        r = self.runner
        idx = torch.tensor(block_ids, dtype=torch.int64, device=r.device)
        return {n: [v.index_select(0, idx).cpu() for v, _ in parts] for n, parts in self._my_views().items()}

    # ---- consumer: pulling ------------------------------------------------------------------------
    def _check_layout(self, info: dict) -> None:
        """Refuse a peer whose cache does not line up with this one: a mismatch would make the block copy
        read or write the wrong bytes."""
        r = self.runner
        mine = r.kv_layer_parts
        nb = int(r.kv_cache_config.num_blocks)
        if int(info["block_size"]) != self.block_size:
            raise ValueError(f"KV peer {info['engine_id']}: block size {info['block_size']} != {self.block_size}")
        if int(info.get("total_kv_heads", self.total_kv_heads)) != self.total_kv_heads:
            raise ValueError(f"KV peer {info['engine_id']}: another model (KV heads differ)")
        if set(info["layers"]) != set(mine):
            raise ValueError(f"KV peer {info['engine_id']}: layer sets differ")
        pnb = int(info["num_blocks"])
        for n, parts in mine.items():
            theirs = info["layers"][n]
            if len(theirs) != len(parts):
                raise ValueError(f"KV peer {info['engine_id']}: layer {n} has another cache structure")
            for (_, _, nbytes, heads), (_, _, pbytes, pheads) in zip(parts, theirs):
                a, b = nbytes // nb, pbytes // pnb
                if (heads == 0) != (pheads == 0) or (heads and a // heads != b // pheads) or (not heads and a != b):
                    raise ValueError(f"KV peer {info['engine_id']}: layer {n} differs in dtype, head size or "
                                     "block layout")

    def _peer(self, host: str, port: int, rank: int) -> _Peer:
        key = (host, port + rank)
        p = self._peers.get(key)
        if p is not None:
            return p
        conn = connect(f"tcp://{key[0]}:{key[1]}", timeout=30)
        conn.send(("HANDSHAKE",))
        info = conn.recv()
        self._check_layout(info)
        region = None
        r = self.runner
        same_node = info["hostname"] == socket.gethostname()
        if r.is_gpu and same_node and info.get("ipc_handles") and info["pid"] != os.getpid():
            from ...device_allocator import open_shared
            region = [open_shared(h, n, r.device) for h, n in zip(info["ipc_handles"], info["chunk_bytes"])]
            logger.info("mapped the KV cache of engine %s rank %d (GPU %s, %d chunks) over xGMI", info["engine_id"],
                        rank, info["device_index"], len(region))
        p = self._peers[key] = _Peer(conn, info, region)
        if region is not None:
            p.views = self._part_views([x.tensor for x in region], info["layers"], int(info["num_blocks"]))
        return p

    def _plan(self, spec: RecvSpec) -> list[tuple[int, list]]:
        """Producer ranks to read from and, per rank, what: [(peer rank, [(part index or None = all head-less
        parts, src head, dst head, heads)])]."""
        first = self._peer(spec.remote_host, spec.remote_port, 0)       # rank 0 always exists: learn the peer's tp
        ptp = int(first.info.get("tp_size", 1))
        segs = head_segments(self.tp_rank, self.tp_size, ptp, self.total_kv_heads)
        plan: dict[int, list] = {}
        for p, src_h, dst_h, n in segs:
            plan.setdefault(p, []).append((src_h, dst_h, n))
        whole_from = segs[0][0]                # head-less tensors are replicated: take them from one rank
        return [(p, hs, p == whole_from) for p, hs in plan.items()]

    def start_load_kv(self) -> None:
        meta = self._meta
        if not isinstance(meta, XgmiMetadata):
            return
        with self._lock:
            for rid, exp in meta.reqs_to_send.items():
                nonce = meta.send_nonces.get(rid)
                if nonce is not None:
                    self._nonce_of[rid] = nonce
                if rid not in self._done_sending:
                    self._expiry[rid] = exp
                    self._outstanding.add(rid)
                else:
                    self._revoke(rid)                   # the consumer was done before this step saw the export
        for rid, spec in meta.reqs_to_recv.items():
            self._outstanding.add(rid)
            if not spec.local_block_ids:
                with self._lock:
                    self._done_recving.add(rid)
                continue
            try:
                self._start_pull(rid, spec)
            except Exception:   # noqa: BLE001 - a bad peer or bad parameters must not take the worker down
                logger.exception("cannot pull the KV blocks of request %s; its prompt is recomputed", rid)
                with self._lock:
                    self._failed_recving.add(rid)
                    self._done_recving.add(rid)

    def _start_pull(self, rid: str, spec: RecvSpec) -> None:
        nb = int(self.runner.kv_cache_config.num_blocks)
        if len(spec.local_block_ids) != len(spec.remote_block_ids):
            raise ValueError("local and remote block lists differ in length")
        if any((not isinstance(b, int)) or b < 0 or b >= nb for b in spec.local_block_ids):
            raise ValueError("local block id out of range")
        plan = self._plan(spec)
        peers = [(self._peer(spec.remote_host, spec.remote_port, p), hs, whole) for p, hs, whole in plan]
        for peer, _, _ in peers:
            pnb = int(peer.info["num_blocks"])
            if any((not isinstance(b, int)) or b < 0 or b >= pnb for b in spec.remote_block_ids):
                raise ValueError("remote block id out of range")
            if peer.region is not None:
                # one-sided read ahead: the producer confirms that it exported these blocks for this request
                with peer.lock:
                    peer.conn.send(("CHECK", spec.remote_token, spec.remote_export))
                    if peer.conn.recv() is not True:
                        raise ValueError("the producer did not export these blocks for this request")
        if all(peer.region is not None for peer, _, _ in peers):
            self._pull_peer_memory(rid, spec, peers)
        else:
            threading.Thread(target=self._pull_socket, args=(rid, spec, peers), daemon=True, name="kv-pull").start()

    def _votes_for(self, peer: _Peer) -> int:
        """How many consumer ranks read from one producer rank (it frees the blocks after that many DONEs)."""
        ptp = int(peer.info.get("tp_size", 1))
        prank = int(peer.info.get("tp_rank", 0))
        return sum(1 for r in range(self.tp_size)
                   if any(p == prank for p, _, _, _ in head_segments(r, self.tp_size, ptp, self.total_kv_heads)))

    def _copy_parts(self, mine, theirs, hs, whole, dst_ids, src_ids, copy) -> None:
        """One layer: my heads' byte ranges out of the peer's blocks; head-less tensors whole (if asked)."""
        for (dv, dheads), (sv, sheads) in zip(mine, theirs):
            if dheads == 0:
                if whole:
                    copy(dv, sv, dst_ids, src_ids, dv.shape[1], 0, 0)
                continue
            hb = dv.shape[1] // dheads
            for src_h, dst_h, n in hs:
                copy(dv, sv, dst_ids, src_ids, n * hb, dst_h * hb, src_h * hb)

    def _pull_peer_memory(self, rid: str, spec: RecvSpec, peers) -> None:
        from ... import ops
        dev = self.runner.device
        dst_ids = torch.tensor(spec.local_block_ids, dtype=torch.int64, device=dev)
        src_ids = torch.tensor(spec.remote_block_ids, dtype=torch.int64, device=dev)
        mine = self._my_views()
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            for n, parts in mine.items():
                for peer, hs, whole in peers:
                    self._copy_parts(parts, peer.views[n], hs, whole, dst_ids, src_ids,
                                     lambda d, s_, di, si, nbytes, doff, soff: ops.transfer_block_slices(
                                         d, s_, di, si, nbytes, doff, soff))
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._inflight.append((rid, ev, [p for p, _, _ in peers]))

    def _pull_socket(self, rid: str, spec: RecvSpec, peers) -> None:
        ok = True
        try:
            r = self.runner
            idx = torch.tensor(spec.local_block_ids, dtype=torch.int64, device=r.device)
            mine = self._my_views()
            for peer, hs, whole in peers:
                with peer.lock:
                    take = list(range(spec.remote_first, spec.remote_first + len(spec.remote_block_ids)))
                    peer.conn.send(("READ", spec.remote_token, spec.remote_export, take))
                    data = peer.conn.recv()
                if data is None:
                    raise ValueError("the producer did not export these blocks for this request")

                def copy(dv, sv, di, si, nbytes, doff, soff):
                    dv[di, doff:doff + nbytes] = sv[:, soff:soff + nbytes].to(dv.device)
                for n, parts in mine.items():
                    theirs = [(t, int(pi[3])) for t, pi in zip(data[n], peer.info["layers"][n])]
                    self._copy_parts(parts, theirs, hs, whole, idx, None, copy)
            if r.is_gpu:
                torch.cuda.synchronize(r.device)
            for peer, _, _ in peers:
                with peer.lock:
                    peer.conn.send(("DONE", rid, self._votes_for(peer)))
        except Exception:   # noqa: BLE001
            logger.exception("pulling KV of request %s failed; its prompt is recomputed", rid)
            ok = False
        with self._lock:
            if not ok:
                self._failed_recving.add(rid)
            self._done_recving.add(rid)

    def get_finished(self, finished_req_ids: set[str]):
        still = []
        for rid, ev, peers in self._inflight:
            if ev.query():
                for peer in peers:
                    with peer.lock:
                        peer.conn.send(("DONE", rid, self._votes_for(peer)))
                with self._lock:
                    self._done_recving.add(rid)
            else:
                still.append((rid, ev, peers))
        self._inflight = still
        now = time.time()
        with self._lock:
            for rid, exp in list(self._expiry.items()):
                if now > exp:
                    logger.warning("KV blocks of request %s were never fetched; releasing them", rid)
                    self._done_sending.add(rid)
                    del self._expiry[rid]
                    self._revoke(rid)
                    self._done_votes.pop(rid, None)
            sending, self._done_sending = self._done_sending, set()
            recving, self._done_recving = self._done_recving, set()
            self._failed_pending |= self._failed_recving
            self._failed_recving = set()
        if self.tp_size > 1:
            # The ranks agree through a host collective, but only while a transfer is outstanding: which
            # requests are in flight is known to every rank alike (it comes with the scheduler output), so
            # steady-state decode steps with nothing to move make no collective call at all.
            if self._outstanding:
                sending, recving = self._all_ranks_agree(sending, recving)
            else:
                sending, recving = set(), set()
        self._outstanding -= sending | recving
        self._failed_report = self._failed_pending & recving
        self._failed_pending -= self._failed_report
        return sending or None, recving or None

    def failed_recving(self) -> set[str]:
        out, self._failed_report = self._failed_report, set()
        return out

    def _all_ranks_agree(self, sending: set[str], recving: set[str]):
        """With TP every rank moves its own shard: report a request only once all ranks are done."""
        from ...parallel.state import get_tp_group
        g = get_tp_group()
        if not hasattr(self, "_partial"):
            self._partial = ({}, {})
        outs = []
        everyone = g.all_gather_object((sorted(sending), sorted(recving), sorted(self._failed_pending)))
        for both in everyone:                    # a pull that failed on any rank failed
            self._failed_pending |= set(both[2])
        for which, counts in ((0, self._partial[0]), (1, self._partial[1])):
            for both in everyone:
                for rid in both[which]:
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
                             multi.MultiMetadata), start=60):            # 65-67: push.py
        serial.wire_struct(i)(cls)


_register_wire_types()
