"""Waiting queues: first-come-first-served and priority
(reference: vllm/v1/core/sched/request_queue.py)."""
from __future__ import annotations

import heapq
from collections import deque
from typing import Iterable, Iterator

from ..request import Request


class RequestQueue:
    def add_request(self, r: Request) -> None: raise NotImplementedError
    def pop_request(self) -> Request: raise NotImplementedError
    def peek_request(self) -> Request: raise NotImplementedError
    def prepend_request(self, r: Request) -> None: raise NotImplementedError
    def prepend_requests(self, rs: "RequestQueue") -> None: raise NotImplementedError
    def remove_request(self, r: Request) -> None: raise NotImplementedError
    def remove_requests(self, rs: Iterable[Request]) -> None: raise NotImplementedError
    def __bool__(self) -> bool: raise NotImplementedError
    def __len__(self) -> int: raise NotImplementedError
    def __iter__(self) -> Iterator[Request]: raise NotImplementedError


class FCFSQueue(RequestQueue):
    def __init__(self):
        self._q: deque[Request] = deque()

    def add_request(self, r): self._q.append(r)
    def pop_request(self): return self._q.popleft()

    def peek_request(self):
        if not self._q:
            raise IndexError("peek from an empty queue")
        return self._q[0]

    def prepend_request(self, r): self._q.appendleft(r)

    def prepend_requests(self, rs):
        self._q.extendleft(reversed(list(rs)))

    def remove_request(self, r): self._q.remove(r)

    def remove_requests(self, rs):
        drop = {id(x) for x in rs}
        self._q = deque(x for x in self._q if id(x) not in drop)

    def __bool__(self): return bool(self._q)
    def __len__(self): return len(self._q)
    def __iter__(self): return iter(self._q)
    def __reversed__(self): return reversed(self._q)


class PriorityQueue(RequestQueue):
    """Smaller ``priority`` first, ties by arrival time (Request.__lt__)."""

    def __init__(self):
        self._h: list[Request] = []

    def add_request(self, r): heapq.heappush(self._h, r)

    def pop_request(self):
        if not self._h:
            raise IndexError("pop from an empty queue")
        return heapq.heappop(self._h)

    def peek_request(self):
        if not self._h:
            raise IndexError("peek from an empty queue")
        return self._h[0]

    # in a priority queue position is decided by the key, so prepend == add
    def prepend_request(self, r): self.add_request(r)

    def prepend_requests(self, rs):
        for r in rs:
            self.add_request(r)

    def remove_request(self, r):
        self._h.remove(r)
        heapq.heapify(self._h)

    def remove_requests(self, rs):
        drop = {id(x) for x in rs}
        self._h = [x for x in self._h if id(x) not in drop]
        heapq.heapify(self._h)

    def __bool__(self): return bool(self._h)
    def __len__(self): return len(self._h)
    def __iter__(self): return iter(sorted(self._h))
    def __reversed__(self): return reversed(sorted(self._h))


def create_request_queue(policy: str) -> RequestQueue:
    if policy == "priority":
        return PriorityQueue()
    if policy == "fcfs":
        return FCFSQueue()
    raise ValueError(f"unknown scheduling policy {policy!r}")
