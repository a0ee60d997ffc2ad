"""Per-group HIP streams for the host drivers.

The Linear groups of one decoder layer are independent once their Gram sums exist, and each group's
work is a long chain of small kernels (panel factorisations, 128-column sweeps) next to a few
chip-filling ones.  Issued on one stream the chains of the four groups simply add up (≈158 ms per
Llama-3-8B layer); on a stream each they overlap (≈113 ms) -- the schedule ``bench.py`` measures.

The Gram sums are the exception: ``xtx_kernel`` fills the chip by itself, so four of them launched at once
only take turns on the CUs.  They go on ONE extra stream, smallest in_features first (the short groups'
chains start within milliseconds, the long pass runs last at its stand-alone rate), and each group's
chain waits for its own Gram sum only (round 3: same throughput, the Gram launches at 0.50 instead of
0.34 of the MFMA peak while the chains of the other groups run beside them).

Round 4: the groups of one in_features share ONE chain (``gptq_quantize_batched``: a batched factorisation, a stacked
sweep) on one stream behind the Gram sums of all of them -- ``run(after=[events])`` -- so a Llama layer runs two chain
streams instead of four.
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
        gkey = (self.device.index, "gram")
        if gkey not in GroupStreams._pools:
            GroupStreams._pools[gkey] = [torch.cuda.Stream(device=self.device)]
        self.gram = GroupStreams._pools[gkey][0]
        self._used: List["torch.cuda.Stream"] = []
        self._next = 0

    def run_gram(self, fn: Callable[[], object]):
        """Run ``fn`` on the Gram stream (one after the other, in call order).  Returns ``(fn(), event)``;
        pass the event to ``run(after=...)`` for the work that consumes what ``fn`` produced."""
        st = self.gram
        if st not in self._used:
            st.wait_stream(self.main)
            self._used.append(st)
        with torch.cuda.stream(st):
            out = fn()
            ev = torch.cuda.Event()
            ev.record(st)
        return out, ev

    def run(self, fn: Callable[[], object], after=None, tensors=()):
        """Run ``fn`` with the next side stream current.  What ``fn`` enqueues starts after everything
        already on the main stream (and after the event -- or list of events -- ``after``) and is waited for by ``join``.
        ``tensors``: allocated on another stream (the Gram stream) and read by ``fn`` -- recorded on this
        one so that the caching allocator does not hand their memory out while ``fn``'s kernels run."""
        st = self.streams[self._next % len(self.streams)]
        self._next += 1
        st.wait_stream(self.main)
        for ev in ([] if after is None else (after if isinstance(after, (list, tuple)) else [after])):
            st.wait_event(ev)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        with torch.cuda.stream(st):
            out = fn()
        if st not in self._used:
            self._used.append(st)
        return out

    def join(self) -> None:
        for st in self._used:
            self.main.wait_stream(st)
        self._used.clear()
