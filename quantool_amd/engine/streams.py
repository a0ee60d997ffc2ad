"""Per-group HIP streams for the host drivers.

The Linear groups of one decoder layer are independent once their Gram sums exist, and each group's
work is a long chain of small kernels (panel factorisations, 128-column sweeps) next to a few
chip-filling ones.  Issued on one stream the chains of the four groups simply add up (≈158 ms per
Llama-3-8B layer); on a stream each they overlap (≈113 ms) -- the schedule ``bench.py`` measures.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch


class GroupStreams:
    """Round-robin pool of side streams forked from, and joined back into, the current stream."""

    _pools: Dict[tuple, List["torch.cuda.Stream"]] = {}

    def __init__(self, device, width: int = 4):
        self.device = torch.device(device)
        key = (self.device.index, width)
        if key not in GroupStreams._pools:
            GroupStreams._pools[key] = [torch.cuda.Stream(device=self.device) for _ in range(width)]
        self.streams = GroupStreams._pools[key]
        self.main = torch.cuda.current_stream(self.device)
        self._used: List["torch.cuda.Stream"] = []
        self._next = 0

    def run(self, fn: Callable[[], object]):
        """Run ``fn`` with the next side stream current.  What ``fn`` enqueues starts after everything
        already on the main stream and is waited for by ``join``."""
        st = self.streams[self._next % len(self.streams)]
        self._next += 1
        st.wait_stream(self.main)
        with torch.cuda.stream(st):
            out = fn()
        if st not in self._used:
            self._used.append(st)
        return out

    def join(self) -> None:
        for st in self._used:
            self.main.wait_stream(st)
        self._used.clear()
