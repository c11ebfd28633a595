"""Bag feed: host bags -> HBM, overlapped with the encoder.

The reference moves one bag per iteration with a blocking ``bag.to(device)`` from pageable
memory (main.py:434, Survival/models/RRTMIL/engine.py:71-74).  At MI355X speeds that copy
(18.4 MB per N=9000 fp32 bag, ~0.3 ms over PCIe Gen5 x16) is as long as the whole forward, so
it has to overlap: ``BagFeeder`` keeps ``depth`` pinned staging buffers and a dedicated copy
stream, issues the H2D of bag i+1.. while bag i computes, and hands out device tensors that are
ordered against the consumer's stream with events (no host synchronisation on the hot path).

    for dev_bag in BagFeeder(cpu_bags, device="cuda:0", depth=3):
        y = encoder(dev_bag)                  # runs while the next bags are in flight
"""
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, Union

import torch


def load_bag(item: Union[str, torch.Tensor], dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """A bag is an (N, D) float tensor or the path of a ``.pt`` file holding one
    (dataloader.py:181,198 ``torch.load``).  A bag that is already of ``dtype`` is handed on as it is (features stored
    in 16 bits stay 16 bits); any other float type is converted."""
    if isinstance(item, str):
        item = torch.load(item, map_location="cpu")
    if item.dim() == 3 and item.size(0) == 1:
        item = item[0]
    return item.to(dtype).contiguous()


class BagFeeder:
    def __init__(self, bags: Iterable[Union[str, torch.Tensor]], device="cuda", depth: int = 3,
                 copy_threads: int = 4, stage_pageable: bool = False, dtype: torch.dtype = torch.float32):
        """copy_threads: host threads for the pageable -> pinned staging copy (one thread moves only
        ~5 GB/s, PCIe Gen5 x16 takes ~53 GB/s); bags that are already pinned skip the staging copy.

        dtype (round 6): the type the bags cross the link in and arrive as.  ``torch.bfloat16`` / ``torch.float16`` halve the
        PCIe bytes of a slide (36.9 -> 18.4 MB at N = 9000 x 1024): ``RRTMIL`` under bf16 / fp16 arithmetic takes such bags as
        patch_to_emb's 16-bit operand directly, with logits bit-identical to the fp32-fed forward (the reference's autocast
        rounds the features to the same 16-bit values in its first op, rrt.py:208-229, main.py:439).  Feature files already
        stored in that type are moved as they are; fp32 bags are converted on the host while they are staged (that costs host
        time per bag: store the features in 16 bits to get the full rate)."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"BagFeeder: dtype must be float32, bfloat16 or float16, got {dtype}")
        self.dtype = dtype
        self.device = torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, copy_threads))
        self.copy_threads = max(1, copy_threads)
        self.stage_pageable = stage_pageable
        if self.device.type != "cuda":
            raise RuntimeError("BagFeeder feeds a HIP device; there is no CPU path")
        self.bags = bags
        self.depth = max(2, int(depth))
        self.copy_stream = torch.cuda.Stream(self.device)
        self._pinned = []          # reusable pinned staging buffers (grown on demand)

    def _stage(self, cpu_bag: torch.Tensor, slot: int) -> torch.Tensor:
        n = cpu_bag.numel()
        while len(self._pinned) <= slot:
            self._pinned.append(torch.empty(0, dtype=self.dtype).pin_memory())
        if self._pinned[slot].numel() < n:
            self._pinned[slot] = torch.empty(n, dtype=self.dtype).pin_memory()
        buf = self._pinned[slot][:n]
        src = cpu_bag.reshape(-1)
        # pageable -> pinned host memcpy (converting to the feeder's dtype on the way), chunked over the pool (Tensor.copy_
        # releases the GIL)
        step = max(1 << 18, (n + self.copy_threads - 1) // self.copy_threads)
        futs = [self.pool.submit(buf[o:o + step].copy_, src[o:o + step]) for o in range(0, n, step)]
        for f in futs:
            f.result()
        return buf.view(cpu_bag.shape)

    def __iter__(self) -> Iterator[torch.Tensor]:
        inflight = deque()                       # (device tensor, ready event, pinned slot)
        free_slots = deque(range(self.depth))
        slot_events = {}                         # slot -> event after which its pinned buffer is reusable
        it = iter(self.bags)
        exhausted = False
        consumer = torch.cuda.current_stream(self.device)

        def issue():
            nonlocal exhausted
            if exhausted or not free_slots:
                return False
            try:
                item = next(it)
            except StopIteration:
                exhausted = True
                return False
            slot = free_slots.popleft()
            if slot in slot_events:
                slot_events[slot].synchronize()  # the previous H2D out of this pinned buffer is done
            raw = item if isinstance(item, torch.Tensor) else load_bag(item, self.dtype)
            if raw.dim() == 3 and raw.size(0) == 1:
                raw = raw[0]
            if raw.dtype != self.dtype and self.stage_pageable:
                cpu_bag = raw.contiguous()         # converted by the staging copy below (thread pool), not by one thread here
            else:
                cpu_bag = raw.to(self.dtype).contiguous()
            if cpu_bag.dtype == self.dtype and (cpu_bag.is_pinned() or not self.stage_pageable):
                # pinned: true async DMA (53 GB/s measured).  pageable: torch's own staged H2D -- it
                # blocks this host thread but runs on the copy stream, under the compute already queued
                # (measured 1.5 k bags/s; a Python-side pinned staging copy was slower: 0.3-0.8 k bags/s)
                host = cpu_bag
            else:
                host = self._stage(cpu_bag, slot)
            with torch.cuda.stream(self.copy_stream):
                dev = torch.empty(host.shape, dtype=self.dtype, device=self.device)
                dev.copy_(host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            slot_events[slot] = ev
            inflight.append((dev, ev, slot))
            return True

        for _ in range(self.depth):
            if not issue():
                break
        while inflight:
            dev, ev, slot = inflight.popleft()
            consumer.wait_event(ev)              # device-side ordering only
            dev.record_stream(consumer)
            free_slots.append(slot)
            issue()
            yield dev
