"""Push-mode KV transfer: the prefill instance writes a request's KV blocks into the decode instance while
the prompt is still being computed, and frees its own blocks as soon as the last copy has landed.

The decode instance does not know the request yet when the data arrives (and so has no blocks for it), so
every consumer worker owns a *landing pool*: an IPC-shareable device arena of block-sized slots, backed by a
pinned host pool that takes over when the arena is full.  A transfer runs as

    PUT_BEGIN(transfer id, blocks)   consumer reserves slots (device first, then host) and names them
    per step, per layer              the producer copies the blocks written in this step into the slots:
                                     straight over xGMI with the block-transfer kernel on a side stream that
                                     waits on an event recorded after that layer's cache write (layer
                                     pipelined: layer i travels while layer i+1 computes); blocks that landed
                                     in host slots, and peers on other nodes, go through the side channel
    PUT_END(transfer id)             after the last copy: the producer reports the request as sent and its
                                     blocks are freed at once (no delayed free, no expiry timer)

``send_type`` = ``PUT`` waits for the step's copies before the step returns, ``PUT_ASYNC`` (default) lets
them run behind the next step.  When the decode request arrives (its ``kv_transfer_params`` carry the
transfer id) the consumer moves the slots into the request's blocks and releases them.  A refused
reservation (pool exhausted) or a transfer that never completes falls back to the pull path of the base
connector, and from there to local prefill.  Producer and consumer run the same tensor-parallel degree in
push mode (rank i pushes to rank i); the pull path handles unequal degrees.

Same role as the reference's P2pNcclConnector in PUT / PUT_ASYNC mode with its TensorMemoryPool
(vllm/distributed/kv_transfer/kv_connector/v1/p2p/p2p_nccl_engine.py:192-235 send paths, :391-448
send queue; p2p/tensor_memory_pool.py host spill; p2p_nccl_connector.py:240-300 save_kv_layer).
"""
from __future__ import annotations

import os
import queue
import re
import socket
import hashlib
import hmac
import threading
import time
import uuid
from dataclasses import dataclass, field
from typing import Any, Optional

import torch

from ...engine.transport import Connection
from ...logger import init_logger
from .base import KVConnectorMetadata, KVConnectorRole
from .xgmi import RecvSpec, XgmiConnector, XgmiMetadata

logger = init_logger(__name__)


@dataclass
class PushSpec:
    transfer_id: str
    host: str
    port: int
    block_ids: list[int]          # the request's blocks, in order
    first_block: int              # blocks from here on were written (or first seen) in this step
    num_blocks_total: int         # blocks of the whole prompt
    last: bool                    # this step finishes the prompt


@dataclass
class LandSpec:
    transfer_id: str
    recv: RecvSpec                # what a pull would need (fallback), incl. the local block ids


@dataclass
class PushMetadata(XgmiMetadata):
    pushes: dict[str, PushSpec] = field(default_factory=dict)
    lands: dict[str, LandSpec] = field(default_factory=dict)
    aborted: list[str] = field(default_factory=list)        # transfers whose request went away on the producer


class LandingPool:
    """Block-sized slots for data that arrives before its request: a device arena other processes can map,
    and a (pinned) host pool behind it."""

    def __init__(self, slot_bytes: int, dev_slots: int, host_slots: int, device: torch.device, is_gpu: bool):
        self.slot_bytes = slot_bytes
        self.region = None
        self.dev = None
        if is_gpu and dev_slots > 0:
            from ...device_allocator import alloc_shareable, ipc_safe_size
            self.region = alloc_shareable(ipc_safe_size(dev_slots * slot_bytes), device)
            self.dev = self.region.tensor[: dev_slots * slot_bytes].view(dev_slots, slot_bytes)
        self.host = torch.empty((host_slots, slot_bytes), dtype=torch.uint8, pin_memory=bool(is_gpu and host_slots))
        self._free = {"dev": list(range(dev_slots if self.dev is not None else 0)), "host": list(range(host_slots))}
        self._lock = threading.Lock()

    def arena(self, kind: str) -> torch.Tensor:
        return self.dev if kind == "dev" else self.host

    def alloc(self, n: int) -> Optional[list[tuple[str, int]]]:
        with self._lock:
            if len(self._free["dev"]) + len(self._free["host"]) < n:
                return None
            out = []
            for kind in ("dev", "host"):
                take = min(n - len(out), len(self._free[kind]))
                out += [(kind, self._free[kind].pop()) for _ in range(take)]
            return out

    def free(self, slots: list[tuple[str, int]]) -> None:
        with self._lock:
            for kind, s in slots:
                self._free[kind].append(s)

    def num_free(self) -> tuple[int, int]:
        return len(self._free["dev"]), len(self._free["host"])

    def close(self) -> None:
        if self.region is not None:
            self.region.close()


class XgmiPushConnector(XgmiConnector):
    def __init__(self, vllm_config, role: KVConnectorRole):
        super().__init__(vllm_config, role)
        kt = vllm_config.kv_transfer_config
        xc = kt.kv_connector_extra_config
        self.send_type = str(xc.get("send_type", "PUT_ASYNC")).upper()
        if self.send_type not in ("PUT", "PUT_ASYNC"):
            raise ValueError(f"send_type {self.send_type!r}: the push connector knows PUT and PUT_ASYNC (GET is the "
                             "pull connector)")
        self.landing_bytes = int(xc.get("landing_buffer_bytes", kt.kv_buffer_size or 1e9))
        self.host_pool_bytes = int(float(xc.get("mem_pool_size_gb", 4)) * (1 << 30))
        self.landing_slots = xc.get("landing_slots")             # device slots by count instead of by bytes
        self.is_consumer = kt.kv_role in ("kv_consumer", "kv_both")
        pk = xc.get("push_key")
        self._push_key: Optional[bytes] = str(pk).encode() if pk else None    # shared by every instance of the deployment
        # scheduler side
        self._targets: dict[str, tuple[str, int, str]] = {}      # producer: rid -> (host, port, transfer id)
        self._blocks: dict[str, list[int]] = {}                  # producer: rid -> block ids so far
        self._pushed_upto: dict[str, int] = {}                   # producer: rid -> blocks already announced
        self._prompt_len: dict[str, int] = {}
        self._land_pending: dict[str, LandSpec] = {}
        self._abort_pending: list[str] = []
        # worker side
        self.pool: Optional[LandingPool] = None
        self._tx: dict[str, dict] = {}                           # producer: transfer id -> state
        self._tx_step: list[tuple[str, PushSpec]] = []           # pushes of the running step
        self._step_layers: set[str] = set()
        self._rx: dict[str, dict] = {}                           # consumer: transfer id -> state
        self._wanted: dict[str, tuple[LandSpec, float]] = {}     # consumer: rid -> (spec, since)
        self._push_stream = None
        self._end_q: "queue.Queue[Optional[str]]" = queue.Queue()
        self._end_thread: Optional[threading.Thread] = None
        self._part_off: Optional[dict[str, list[tuple[int, int]]]] = None
        self.slot_bytes = 0

    # =========================== scheduler side ===========================
    @staticmethod
    def _target_of(request) -> Optional[tuple[str, int]]:
        p = request.kv_transfer_params or {}
        if p.get("push_host") and p.get("push_port"):
            return str(p["push_host"]), int(p["push_port"])
        # the reference's proxy writes the decode instance's address into the request id
        m = re.search(r"___decode_addr_([^_:]+):(\d+)", request.request_id)
        return (m.group(1), int(m.group(2))) if m else None

    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> tuple[int, bool]:
        return super().get_num_new_matched_tokens(request, num_computed_tokens)

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int) -> None:
        p = request.kv_transfer_params
        if p and p.get("do_remote_decode") and request.request_id not in self._targets:
            tgt = self._target_of(request)
            if tgt is not None:
                self._targets[request.request_id] = (tgt[0], tgt[1], uuid.uuid4().hex)
                self._prompt_len[request.request_id] = len(request.prompt_token_ids)
            return
        pushed = bool(p and p.get("do_remote_prefill") and p.get("pushed") and p.get("transfer_id"))
        super().update_state_after_alloc(request, blocks, num_external_tokens)
        if pushed:
            spec = self._recv_pending.pop(request.request_id, None)
            if spec is not None:
                self._land_pending[request.request_id] = LandSpec(str(p["transfer_id"]), spec)

    def build_connector_meta(self, so) -> PushMetadata:
        meta = PushMetadata(dict(self._recv_pending), dict(self._send_pending))
        self._recv_pending.clear()
        self._send_pending.clear()
        meta.lands, self._land_pending = self._land_pending, {}
        meta.aborted, self._abort_pending = self._abort_pending, []

        def flat(ids):
            return list(ids[0]) if ids and isinstance(ids[0], (list, tuple)) else list(ids or [])
        computed = {}
        for nr in so.scheduled_new_reqs:
            if nr.req_id in self._targets:
                self._blocks[nr.req_id] = flat(nr.block_ids)
                computed[nr.req_id] = nr.num_computed_tokens
        cr = so.scheduled_cached_reqs
        for i, rid in enumerate(cr.req_ids):
            if rid not in self._targets:
                continue
            if cr.resumed_from_preemption[i]:
                self._blocks[rid] = flat(cr.new_block_ids[i])
                self._pushed_upto[rid] = 0                  # other blocks now: everything is sent again
            else:
                self._blocks.setdefault(rid, []).extend(flat(cr.new_block_ids[i]))
            computed[rid] = cr.num_computed_tokens[i]
        bs = self.block_size
        for rid, (host, port, tid) in self._targets.items():
            n = so.num_scheduled_tokens.get(rid)
            if n is None or rid not in computed:
                continue
            plen = self._prompt_len[rid]
            if computed[rid] >= plen:
                continue                                    # decode steps of a request that keeps running
            upto = min(computed[rid] + n, plen)
            first = min(self._pushed_upto.get(rid, 0), computed[rid] // bs)
            nblk = -(-upto // bs)
            meta.pushes[rid] = PushSpec(tid, host, port, list(self._blocks[rid][:nblk]), first, -(-plen // bs),
                                        upto >= plen)
            self._pushed_upto[rid] = upto // bs             # a partly filled block goes again with the next step
        return meta

    def request_finished(self, request, block_ids: list[int]) -> tuple[bool, Optional[dict]]:
        rid = request.request_id
        tgt = self._targets.pop(rid, None)
        for d in (self._blocks, self._pushed_upto, self._prompt_len):
            d.pop(rid, None)
        keep, params = super().request_finished(request, block_ids)
        if tgt is None:
            return keep, params
        if params is None:                                   # aborted / stopped early: the consumer drops the slots
            self._abort_pending.append(tgt[2])
            return keep, params
        params.update(pushed=True, transfer_id=tgt[2])
        return keep, params

    # ============================= worker side =============================
    def register_kv_caches(self, runner) -> None:
        super().register_kv_caches(runner)
        kt = self.vllm_config.kv_transfer_config
        nb = int(runner.kv_cache_config.num_blocks)
        self._part_off, off = {}, 0
        for n, parts in self._my_views().items():
            offs = []
            for v, _ in parts:
                offs.append((off, int(v.shape[1])))
                off += int(v.shape[1])
            self._part_off[n] = offs
        self.slot_bytes = off
        if runner.is_gpu:
            self._push_stream = torch.cuda.Stream(device=runner.device)
        if kt.is_kv_producer and self._end_thread is None:
            self._end_thread = threading.Thread(target=self._end_loop, daemon=True, name="kv-push-end")
            self._end_thread.start()
        if self.is_consumer:
            dev_slots = min(self.landing_bytes // max(self.slot_bytes, 1), nb) if runner.is_gpu else 0
            if self.landing_slots is not None and runner.is_gpu:
                dev_slots = int(self.landing_slots)
            host_slots = min(self.host_pool_bytes // max(self.slot_bytes, 1), 4 * nb)
            self.pool = LandingPool(self.slot_bytes, int(dev_slots), int(host_slots), runner.device, runner.is_gpu)
            if self.listener is None:                        # consumers listen too: that is where pushes arrive
                from ...engine.transport import Listener
                self.listener = Listener(f"tcp://{self.host}:{self.base_port + self.tp_rank}")
                threading.Thread(target=self._serve, daemon=True, name="kv-landing").start()
            logger.info("KV landing pool: %d device + %d host slots of %d bytes, listening on %s (%s)", dev_slots,
                        host_slots, self.slot_bytes, self.listener.address, kt.kv_role)

    def _layout(self) -> dict:
        info = super()._layout()
        if self.pool is not None:
            info["landing"] = {"slot_bytes": self.slot_bytes,
                               "dev_slots": 0 if self.pool.dev is None else int(self.pool.dev.shape[0]),
                               "handle": self.pool.region.handle if self.pool.region is not None else None,
                               "nbytes": int(self.pool.region.tensor.numel()) if self.pool.region is not None else 0}
        return info

    # ---- consumer: receiving ------------------------------------------------------------------------------
    def _put_mac(self, tid: str) -> str:
        """Proof that the sender of a PUT_BEGIN knows the deployment's ``push_key`` (connector option, the same on
        the prefill and the decode instances; the per-engine ``auth_key`` cannot serve: it is private to the
        exporting engine).  With a key configured, a peer that merely reaches the port can neither take landing
        slots nor name a transfer."""
        return hmac.new(self._push_key or b"", b"put|" + tid.encode(), hashlib.sha256).hexdigest()

    def _serve_conn(self, c: Connection) -> None:
        mine: set[str] = set()         # transfers begun on THIS connection: only it may feed, end or abort them
        try:
            while True:
                msg = c.recv()
                kind = msg[0]
                if kind == "PUT_BEGIN":
                    tid, n = str(msg[1]), int(msg[2])
                    mac = str(msg[3]) if len(msg) > 3 else ""
                    with self._lock:
                        taken = tid in self._rx and tid not in mine
                    if self._push_key is None and not getattr(self, "_warned_open", False):
                        self._warned_open = True
                        logger.warning("no push_key configured: KV pushes are accepted from any peer that can reach "
                                       "the side-channel port (set kv_connector_extra_config['push_key'] on every "
                                       "instance)")
                    authentic = self._push_key is None or hmac.compare_digest(self._put_mac(tid), mac)
                    if not authentic or taken:
                        logger.warning("refused a KV push (%s)", "transfer id in use" if taken else "bad credentials")
                        c.send(None)
                        continue
                    slots = self.pool.alloc(n) if self.pool is not None and 0 < n <= 1 << 20 else None
                    with self._lock:
                        old = self._rx.pop(tid, None)
                        if old and old.get("slots"):
                            self.pool.free(old["slots"])
                        self._rx[tid] = {"slots": slots, "complete": False, "refused": slots is None,
                                         "since": time.time()}
                    mine.add(tid)
                    c.send(slots)
                elif kind == "PUT_DATA":                     # (tid, first block index, {layer: [bytes per part]})
                    tid, b0, data = str(msg[1]), int(msg[2]), msg[3]
                    with self._lock:
                        st = self._rx.get(tid) if tid in mine else None
                    ok = st is not None and st.get("slots") is not None
                    if ok:
                        ok = self._store(st["slots"], b0, data)
                    c.send(bool(ok))
                elif kind == "PUT_END":
                    with self._lock:
                        st = self._rx.get(str(msg[1])) if str(msg[1]) in mine else None
                        if st is not None:
                            st["complete"] = True
                elif kind == "PUT_ABORT":
                    if str(msg[1]) in mine:
                        mine.discard(str(msg[1]))
                        self._drop_rx(str(msg[1]))
                else:
                    self._serve_one(c, msg)
        except (EOFError, OSError):
            pass

    def _serve_one(self, c: Connection, msg) -> None:
        """The pull protocol of the base class (a producer still serves reads: that is the fallback)."""
        kind = msg[0]
        if kind == "HANDSHAKE":
            c.send(self._layout())
        elif kind == "CHECK":
            c.send(bool(self._owns(msg[1], list(msg[2]))))
        elif kind == "READ":
            full, take = list(msg[2]), list(msg[3])
            ok = self._owns(msg[1], full) and all(isinstance(i, int) and 0 <= i < len(full) for i in take)
            c.send(self._read_blocks([full[i] for i in take]) if ok else None)
        elif kind == "DONE":
            rid, want = msg[1], int(msg[2]) if len(msg) > 2 else 1
            with self._lock:
                got = self._done_votes.get(rid, 0) + 1
                if got >= want:
                    self._done_votes.pop(rid, None)
                    self._done_sending.add(rid)
                    self._expiry.pop(rid, None)
                    self._revoke(rid)
                else:
                    self._done_votes[rid] = got

    def _store(self, slots: list, b0: int, data: dict) -> bool:
        """Bytes that came over the side channel go into their slots (host or device)."""
        for n, tensors in data.items():
            offs = self._part_off.get(n)
            if offs is None or len(tensors) != len(offs):
                return False
            for t, (off, nbytes) in zip(tensors, offs):
                if t.dim() != 2 or t.shape[1] != nbytes or b0 < 0 or b0 + t.shape[0] > len(slots):
                    return False
                for i in range(t.shape[0]):
                    kind, s = slots[b0 + i]
                    self.pool.arena(kind)[s, off:off + nbytes] = t[i].to(self.pool.arena(kind).device)
        return True

    def _drop_rx(self, tid: str) -> None:
        with self._lock:
            st = self._rx.pop(tid, None)
        if st and st.get("slots"):
            self.pool.free(st["slots"])

    # ---- producer: sending --------------------------------------------------------------------------------
    def _landing_of(self, peer) -> Optional[torch.Tensor]:
        """The peer's device landing arena, mapped once."""
        if getattr(peer, "landing", None) is not None or not self.runner.is_gpu:
            return getattr(peer, "landing", None)
        if peer.info.get("hostname") != socket.gethostname() or peer.info.get("pid") == os.getpid():
            return None                                      # another node (or this very process): side channel
        info = peer.info.get("landing") or {}
        if not info.get("handle") or not info.get("dev_slots"):
            return None
        from ...device_allocator import open_shared
        peer.landing_region = open_shared(info["handle"], info["nbytes"], self.runner.device)
        n, sb = int(info["dev_slots"]), int(info["slot_bytes"])
        peer.landing = peer.landing_region.tensor[: n * sb].view(n, sb)
        return peer.landing

    def _begin(self, rid: str, spec: PushSpec) -> Optional[dict]:
        st = self._tx.get(spec.transfer_id)
        if st is not None:
            return st
        peer = self._peer(spec.host, spec.port, self.tp_rank)
        if int(peer.info.get("tp_size", 1)) != self.tp_size:
            raise ValueError("push mode needs the same tensor-parallel degree on both sides")
        land = peer.info.get("landing")
        if not land or int(land["slot_bytes"]) != self.slot_bytes:
            raise ValueError("the peer has no landing pool for this cache layout")
        with peer.lock:
            peer.conn.send(("PUT_BEGIN", spec.transfer_id, spec.num_blocks_total, self._put_mac(spec.transfer_id)))
            slots = peer.conn.recv()
        st = self._tx[spec.transfer_id] = {"peer": peer, "slots": slots, "rid": rid, "events": [], "failed": slots is None}
        if slots is None:
            logger.warning("request %s: the decode instance has no landing slots left; it will pull", rid)
        return st

    def start_load_kv(self) -> None:
        super().start_load_kv()
        meta = self._meta
        self._tx_step, self._step_layers = [], set()
        if not isinstance(meta, PushMetadata):
            return
        for tid in meta.aborted:
            st = self._tx.pop(tid, None)
            if st is not None and not st["failed"]:
                try:
                    with st["peer"].lock:
                        st["peer"].conn.send(("PUT_ABORT", tid))
                except OSError:
                    pass
        for rid, spec in meta.pushes.items():
            try:
                st = self._begin(rid, spec)
            except Exception:   # noqa: BLE001 - an unreachable decode instance must not stop prefill
                logger.exception("cannot push the KV of request %s; the decode instance will pull it", rid)
                st = self._tx[spec.transfer_id] = {"peer": None, "slots": None, "rid": rid, "events": [], "failed": True}
            if not st["failed"]:
                self._tx_step.append((rid, spec))
        now = time.time()
        for rid, land in meta.lands.items():
            self._outstanding.add(rid)
            self._wanted[rid] = (land, now)

    def _dev_rows(self, st: dict, spec: PushSpec):
        """(cache block ids, landing slots) of the blocks of this step that have device slots."""
        key = ("rows", spec.first_block, len(spec.block_ids))
        if st.get("rows_key") != key:
            idx = [i for i in range(spec.first_block, len(spec.block_ids)) if st["slots"][i][0] == "dev"]
            dev = self.runner.device
            st["rows_key"] = key
            st["rows"] = (torch.tensor([spec.block_ids[i] for i in idx], dtype=torch.int64, device=dev),
                          torch.tensor([st["slots"][i][1] for i in idx], dtype=torch.int64, device=dev), len(idx))
        return st["rows"]

    def _push_layer(self, name: str) -> None:
        from ... import ops
        parts, offs = self._my_views()[name], self._part_off[name]
        dev = self.runner.device
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._push_stream.wait_event(ev)
        with torch.cuda.stream(self._push_stream):
            for rid, spec in self._tx_step:
                st = self._tx[spec.transfer_id]
                arena = self._landing_of(st["peer"])
                if arena is None:
                    continue
                src_ids, dst_ids, n = self._dev_rows(st, spec)
                if n == 0:
                    continue
                for (v, _), (off, nbytes) in zip(parts, offs):
                    ops.transfer_block_slices(arena, v, dst_ids, src_ids, nbytes, off, 0)

    def save_kv_layer(self, layer_name: str, kv_layer: tuple, attn_metadata) -> None:
        if not self._tx_step or self._push_stream is None or layer_name not in (self._part_off or {}):
            return
        self._push_layer(layer_name)
        self._step_layers.add(layer_name)

    def wait_for_save(self) -> None:
        if not self._tx_step:
            return
        r = self.runner
        if self._push_stream is not None:
            for name in self._part_off:                      # layers that made no call (no attention hook)
                if name not in self._step_layers:
                    self._push_layer(name)
        for rid, spec in self._tx_step:
            st = self._tx[spec.transfer_id]
            arena = self._landing_of(st["peer"]) if r.is_gpu else None
            rest = [i for i in range(spec.first_block, len(spec.block_ids))
                    if arena is None or st["slots"][i][0] != "dev"]
            # host slots and socket peers: contiguous runs of blocks through the side channel
            run_start = 0
            while run_start < len(rest):
                run_end = run_start
                while run_end + 1 < len(rest) and rest[run_end + 1] == rest[run_end] + 1:
                    run_end += 1
                ids = [spec.block_ids[i] for i in rest[run_start:run_end + 1]]
                try:
                    with st["peer"].lock:
                        st["peer"].conn.send(("PUT_DATA", spec.transfer_id, rest[run_start], self._read_blocks(ids)))
                        if st["peer"].conn.recv() is not True:
                            st["failed"] = True
                except OSError:
                    st["failed"] = True
                run_start = run_end + 1
            if self._push_stream is not None:
                ev = torch.cuda.Event()
                ev.record(self._push_stream)
                st["events"].append(ev)
            if spec.last:
                st["last"] = True
                self._end_q.put(spec.transfer_id)            # PUT_END leaves as soon as the copies are done
        if self.send_type == "PUT" and self._push_stream is not None:
            self._push_stream.synchronize()
        self._tx_step = []

    def _end_loop(self) -> None:
        """Completion thread of the producer: waits for a transfer's copy events and tells the consumer that the
        data is whole.  It does not depend on further engine steps (an idle prefill instance makes none)."""
        while True:
            tid = self._end_q.get()
            if tid is None:
                return
            st = self._tx.get(tid)
            if st is None or st["failed"]:
                continue
            try:
                for ev in list(st["events"]):
                    ev.synchronize()
                with st["peer"].lock:
                    st["peer"].conn.send(("PUT_END", tid))
                st["ended"] = True
            except Exception:   # noqa: BLE001 - the consumer is gone: it pulls or prefills on its own
                logger.exception("could not complete KV transfer %s", tid)
                st["ended"] = "lost"

    # ---- both: progress ------------------------------------------------------------------------------------
    def _finish_sends(self) -> None:
        for tid, st in list(self._tx.items()):
            if st["failed"]:
                # nothing (usable) went out: the blocks wait for a pull like any export, with the usual expiry
                if st.get("last") or st["peer"] is None:
                    self._tx.pop(tid)
                continue
            if st.get("ended") == "lost":
                self._tx.pop(tid)
                continue
            if not st.get("ended"):
                continue
            # reported once the scheduler has finished the request (it then holds the blocks for export and
            # says so with the step's metadata): they are freed with this report
            with self._lock:
                if st["rid"] in self._expiry:
                    self._done_sending.add(st["rid"])
                    self._expiry.pop(st["rid"], None)
                    self._tx.pop(tid)

    def _land(self, rid: str, land: LandSpec, st: dict) -> None:
        """Slots -> the request's blocks (the locally cached prefix is skipped)."""
        from ... import ops
        spec = land.recv
        first, ids = spec.remote_first, spec.local_block_ids
        slots = st["slots"][first:first + len(ids)]
        if len(slots) != len(ids):
            raise ValueError("pushed blocks do not cover the request's prompt")
        mine = self._my_views()
        r = self.runner
        for kind in ("dev", "host"):
            sel = [i for i, (k, _) in enumerate(slots) if k == kind]
            if not sel:
                continue
            arena = self.pool.arena(kind)
            dst = torch.tensor([ids[i] for i in sel], dtype=torch.int64, device=r.device)
            src = torch.tensor([slots[i][1] for i in sel], dtype=torch.int64, device=arena.device)
            for n, parts in mine.items():
                for (v, _), (off, nbytes) in zip(parts, self._part_off[n]):
                    if kind == "dev":
                        ops.transfer_block_slices(v, arena, dst, src, nbytes, 0, off)
                    else:
                        v[dst] = arena[src, off:off + nbytes].to(v.device, non_blocking=True)

    def get_finished(self, finished_req_ids: set[str]):
        self._finish_sends()
        now = time.time()
        for rid, (land, since) in list(self._wanted.items()):
            st = self._rx.get(land.transfer_id)
            if st is not None and st["complete"]:
                del self._wanted[rid]
                try:
                    if self._stream is not None:
                        self._stream.wait_stream(torch.cuda.current_stream(self.runner.device))
                        with torch.cuda.stream(self._stream):
                            self._land(rid, land, st)
                            ev = torch.cuda.Event()
                            ev.record(self._stream)
                        self._inflight.append((rid, ev, []))
                        st["landing_event"] = ev
                    else:
                        self._land(rid, land, st)
                        with self._lock:
                            self._done_recving.add(rid)
                        self._drop_rx(land.transfer_id)
                except Exception:   # noqa: BLE001
                    logger.exception("landing the pushed KV of request %s failed; its prompt is recomputed", rid)
                    self._drop_rx(land.transfer_id)
                    with self._lock:
                        self._failed_recving.add(rid)
                        self._done_recving.add(rid)
            elif (st is not None and st["refused"]) or now - since > min(self.timeout, 30.0):
                # nothing will land: read the blocks from the producer instead (it keeps them until DONE)
                del self._wanted[rid]
                if st is not None:
                    self._drop_rx(land.transfer_id)
                try:
                    self._start_pull(rid, land.recv)
                except Exception:   # noqa: BLE001
                    logger.exception("cannot pull the KV blocks of request %s; its prompt is recomputed", rid)
                    with self._lock:
                        self._failed_recving.add(rid)
                        self._done_recving.add(rid)
        # slots of transfers whose landing copy has finished; transfers nobody claimed
        for tid, st in list(self._rx.items()):
            ev = st.get("landing_event")
            if ev is not None and ev.query():
                self._drop_rx(tid)
            elif ev is None and now - st["since"] > self.timeout and not any(
                    l.transfer_id == tid for l, _ in self._wanted.values()):
                logger.warning("pushed KV %s was never claimed by a request; dropping it", tid)
                self._drop_rx(tid)
        return super().get_finished(finished_req_ids)

    def shutdown(self) -> None:
        self._end_q.put(None)
        super().shutdown()
        for p in self._peers.values():
            reg = getattr(p, "landing_region", None)
            if reg is not None:
                reg.close()
        if self.pool is not None:
            self.pool.close()


def p2p_connector(vllm_config, role: KVConnectorRole):
    """The reference's P2pNcclConnector by name: ``send_type`` GET is the pull connector, PUT / PUT_ASYNC
    (its default) the push connector."""
    st = str(vllm_config.kv_transfer_config.kv_connector_extra_config.get("send_type", "PUT_ASYNC")).upper()
    return XgmiConnector(vllm_config, role) if st == "GET" else XgmiPushConnector(vllm_config, role)


def _register_wire_types() -> None:
    from ...engine import serial
    for i, cls in enumerate((PushSpec, LandSpec, PushMetadata), start=65):
        serial.wire_struct(i)(cls)


_register_wire_types()
