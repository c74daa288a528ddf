"""Framed message sockets between the front end, the engine cores, the DP
coordinator and external subscribers.

A frame is ``<u32 length><payload>``; payloads are the msgpack encoding of
engine/serial.py.  Addresses are ``ipc:///path`` (Unix domain socket, same
host) or ``tcp://host:port``.  This replaces the reference's ZMQ sockets
(vllm/v1/engine/core_client.py, core.py:691-809), which are not available here;
the patterns needed are point-to-point request/response and a fan-in of outputs,
both of which map onto plain stream sockets.
"""
from __future__ import annotations

import asyncio
import os
import select
import socket
import struct
import tempfile
import threading
import time
import uuid
from typing import Any, Optional

from . import serial

_HDR = struct.Struct("<I")
_MAX_FRAME = 1 << 31


def make_ipc_address(tag: str = "vda") -> str:
    from .. import envs
    return f"ipc://{os.path.join(envs.VLLM_RPC_BASE_PATH, f'{tag}-{uuid.uuid4().hex[:12]}.sock')}"


def _parse(address: str):
    if address.startswith("ipc://"):
        return socket.AF_UNIX, address[len("ipc://"):]
    if address.startswith("tcp://"):
        host, port = address[len("tcp://"):].rsplit(":", 1)
        return socket.AF_INET, (host.strip("[]") or "0.0.0.0", int(port))
    if ":" in address and not address.startswith("/"):
        host, port = address.rsplit(":", 1)
        return socket.AF_INET, (host, int(port))
    return socket.AF_UNIX, address


class Connection:
    def __init__(self, sock: socket.socket):
        self.sock = sock
        self.sock.setblocking(True)
        if sock.family == socket.AF_INET:
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._wlock = threading.Lock()
        self._rlock = threading.Lock()
        self._closed = False

    def fileno(self) -> int:
        return self.sock.fileno()

    def send_bytes(self, data: bytes) -> None:
        if len(data) >= _MAX_FRAME:
            raise ValueError("frame too large")
        with self._wlock:
            if len(data) < 65536:
                self.sock.sendall(_HDR.pack(len(data)) + data)      # one syscall for small frames
            else:
                self.sock.sendall(_HDR.pack(len(data)))
                self.sock.sendall(data)

    def _recv_exact(self, n: int) -> bytes:
        buf = bytearray(n)
        view = memoryview(buf)
        got = 0
        while got < n:
            k = self.sock.recv_into(view[got:], n - got)
            if k == 0:
                raise EOFError("peer closed the connection")
            got += k
        return bytes(buf)

    def recv_bytes(self) -> bytes:
        with self._rlock:
            (n,) = _HDR.unpack(self._recv_exact(4))
            return self._recv_exact(n) if n else b""

    def send(self, obj: Any) -> None:
        self.send_bytes(serial.encode(obj))

    def recv(self) -> Any:
        return serial.decode(self.recv_bytes())

    def poll(self, timeout: Optional[float] = 0.0) -> bool:
        if self._closed:
            return False
        try:
            r, _, _ = select.select([self.sock], [], [], timeout)
        except (OSError, ValueError):
            return False
        return bool(r)

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            try:
                self.sock.shutdown(socket.SHUT_RDWR)
            except OSError:
                pass
            self.sock.close()


class Listener:
    def __init__(self, address: str, backlog: int = 128):
        fam, addr = _parse(address)
        self.sock = socket.socket(fam, socket.SOCK_STREAM)
        self._path = None
        if fam == socket.AF_UNIX:
            if os.path.exists(addr):
                os.unlink(addr)
            self.sock.bind(addr)
            self._path = addr
            self.address = f"ipc://{addr}"
        else:
            self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self.sock.bind(addr)
            host, port = self.sock.getsockname()[:2]
            self.address = f"tcp://{addr[0] if addr[0] not in ('', '0.0.0.0') else host}:{port}"
        self.sock.listen(backlog)

    def fileno(self) -> int:
        return self.sock.fileno()

    def accept(self, timeout: Optional[float] = None) -> Optional[Connection]:
        if timeout is not None:
            r, _, _ = select.select([self.sock], [], [], timeout)
            if not r:
                return None
        conn, _ = self.sock.accept()
        return Connection(conn)

    def close(self) -> None:
        try:
            self.sock.close()
        finally:
            if self._path and os.path.exists(self._path):
                try:
                    os.unlink(self._path)
                except OSError:
                    pass


def connect(address: str, timeout: float = 30.0) -> Connection:
    fam, addr = _parse(address)
    deadline = time.monotonic() + timeout
    last: Optional[Exception] = None
    while time.monotonic() < deadline:
        s = socket.socket(fam, socket.SOCK_STREAM)
        try:
            s.connect(addr)
            return Connection(s)
        except (ConnectionRefusedError, FileNotFoundError) as e:
            last = e
            s.close()
            time.sleep(0.05)
    raise TimeoutError(f"could not connect to {address}: {last}")


# ---- asyncio flavour -----------------------------------------------------------
class AsyncConnection:
    def __init__(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter):
        self.reader, self.writer = reader, writer
        self._wlock = asyncio.Lock()

    @classmethod
    async def open(cls, address: str, timeout: float = 30.0) -> "AsyncConnection":
        fam, addr = _parse(address)
        deadline = time.monotonic() + timeout
        while True:
            try:
                if fam == socket.AF_UNIX:
                    r, w = await asyncio.open_unix_connection(addr, limit=1 << 26)
                else:
                    r, w = await asyncio.open_connection(addr[0], addr[1], limit=1 << 26)
                return cls(r, w)
            except (ConnectionRefusedError, FileNotFoundError):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"could not connect to {address}") from None
                await asyncio.sleep(0.05)

    @classmethod
    def from_connection(cls, conn: Connection) -> "asyncio.Future":
        """Adopt an already connected blocking socket into the running loop."""
        async def _go():
            conn.sock.setblocking(False)
            r, w = await asyncio.open_connection(sock=conn.sock, limit=1 << 26) if conn.sock.family != socket.AF_UNIX \
                else await asyncio.open_unix_connection(sock=conn.sock, limit=1 << 26)
            return cls(r, w)
        return _go()

    async def send(self, obj: Any) -> None:
        data = serial.encode(obj)
        async with self._wlock:
            self.writer.write(_HDR.pack(len(data)))
            self.writer.write(data)
            await self.writer.drain()

    async def recv(self) -> Any:
        hdr = await self.reader.readexactly(4)
        (n,) = _HDR.unpack(hdr)
        data = await self.reader.readexactly(n) if n else b""
        return serial.decode(data)

    def close(self) -> None:
        try:
            self.writer.close()
        except Exception:
            pass
