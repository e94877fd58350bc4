"""Host-side helpers shared by the optimizer and the fused-op context."""
import torch


class PinnedRing:
    """Step scalars (lr / betas / bias corrections, the dropout seed) reach the device as asynchronous copies from pinned
    host memory.  One pinned buffer rewritten every step would race: nothing synchronises the host in graph / flat mode,
    so the CPU could overwrite the values of step t before the DMA of step t has run.  Hence a ring of slots, each with
    the event recorded after its last copy; a slot is only rewritten once that event has completed (normally long ago)."""

    def __init__(self, shape, dtype, device, slots=32):
        self.device = torch.device(device)
        pin = self.device.type == "cuda"
        self.slots = [torch.zeros(shape, dtype=dtype).pin_memory() if pin else torch.zeros(shape, dtype=dtype) for _ in range(slots)]
        self.events = [None] * slots
        self.cur = -1

    def next(self):
        """-> the host slot to fill for this step (safe to overwrite)."""
        self.cur = (self.cur + 1) % len(self.slots)
        ev = self.events[self.cur]
        if ev is not None:
            ev.synchronize()
        return self.slots[self.cur]

    def push(self, dst):
        """Copy the current slot to the device tensor `dst` on the current stream and remember when that copy is done."""
        dst.copy_(self.slots[self.cur], non_blocking=True)
        if self.device.type == "cuda":
            ev = self.events[self.cur]
            if ev is None:
                ev = self.events[self.cur] = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
