"""Host -> device uploads that do not stall the training loop.

A copy from pageable host memory waits for everything queued on the stream (i.e. for the previous step); a fresh pinned
allocation per step costs more than that wait.  ``upload`` keeps a small ring of persistent pinned staging buffers per
(shape, dtype, device), each guarded by the event of its last use: the host only blocks when the device is ``ring`` steps
behind.  Values pass through unchanged, so host-side random draws keep the reference's generator order."""
import torch

_RINGS = {}


def _slot(shape, dtype, device, ring):
    key = (tuple(shape), dtype, device)
    st = _RINGS.get(key)
    if st is None:
        st = _RINGS[key] = {"bufs": [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(ring)],
                            "evs": [None] * ring, "i": 0}
    i = st["i"]
    st["i"] = (i + 1) % len(st["bufs"])
    if st["evs"][i] is not None:
        st["evs"][i].synchronize()
    return st, i


def upload(fill, shape, dtype, device, out=None, ring=4):
    """``fill(pinned_buffer)`` writes the host values; returns them on ``device`` (into ``out`` when given), queued on
    the current stream without blocking the host."""
    st, i = _slot(shape, dtype, device, ring)
    buf = st["bufs"][i]
    fill(buf)
    if out is None:
        out = buf.to(device, non_blocking=True)
    else:
        out.copy_(buf, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["evs"][i] = ev
    return out
