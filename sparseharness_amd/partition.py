"""Row-range sharding of a CSR matrix across GPUs.

The reference is single-device; this is the new multi-GPU layer of the hot
path.  Rows are independent, so the matrix is cut into `parts` contiguous row
ranges with ~equal non-zeros (power-law rows make equal-row cuts badly
imbalanced); every rank keeps a full replica of x.

For the iterative apps the result vector of one iteration is the next
iteration's x, so ranks exchange their slices with ONE all-gather per
iteration.  To make that a plain equal-size all-gather with no copies, vectors
live in a *slotted layout*: rank k owns slot k of `slot` elements
(slot = longest row range rounded up to 64, plus one 64-element tail whose first
word carries the rank's "changed" flag), and column indices are remapped once,
on the host, from global row ids to slotted positions.
"""
import numpy as np


def row_bounds(row_ptr, parts, cols=None):
    """Boundaries b[0..parts] of contiguous row ranges of ~equal WORK.

    Without `cols`: equal stored entries per range.  With `cols` (the x length) and a matrix big
    enough for the engine's x-tiled plan, entries are weighted by the HBM bytes that plan moves
    for them: ~18 B for an entry of a light row, ~7 B for an entry of a heavy row (rows averaging
    >= 8 entries per 32768-column tile, pre-reduced inside phase 1).  A shard
    full of heavy rows would otherwise finish early while the others still stream."""
    row_ptr = np.asarray(row_ptr)
    rows = len(row_ptr) - 1
    if cols is not None and cols > (1 << 20) and int(row_ptr[-1]) >= (1 << 22):
        deg = np.diff(row_ptr).astype(np.int64)
        thr = max(512, 8 * ((int(cols) + 32767) // 32768))
        work = np.where(deg >= thr, 7 * deg, 18 * deg)
        cum = np.concatenate([[0], np.cumsum(work)])
    else:
        cum = row_ptr.astype(np.int64)
    total = int(cum[-1])
    targets = (np.arange(1, parts, dtype=np.float64) * total / parts)
    cuts = np.searchsorted(cum, targets, side="left").astype(np.int64)
    b = np.concatenate([[0], np.clip(cuts, 0, rows), [rows]])
    return np.maximum.accumulate(b)


def take_rows(row_ptr, col_idx, val, r0, r1):
    """CSR slice of rows [r0, r1) with row_ptr rebased to 0 (views, no copy of col/val)."""
    s, e = int(row_ptr[r0]), int(row_ptr[r1])
    rp = (np.asarray(row_ptr[r0:r1 + 1], dtype=np.int64) - s).astype(np.int32)
    return rp, col_idx[s:e], val[s:e]


class SlottedLayout:
    """Mapping between global vector indices and the slotted all-gather layout."""

    FLAG_PAD = 64  # elements reserved at the end of each slot; word 0 = changed flag

    def __init__(self, bounds):
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.parts = len(self.bounds) - 1
        longest = int(np.diff(self.bounds).max()) if self.parts else 0
        self.payload = (longest + 63) // 64 * 64
        self.slot = self.payload + self.FLAG_PAD
        self.length = self.slot * self.parts

    def slot_offset(self, k):
        return k * self.slot

    def flag_index(self, k):
        return k * self.slot + self.payload

    def to_slotted_index(self, idx):
        """Global ids -> slotted positions (out-of-range ids stay out of range => identity)."""
        idx = np.asarray(idx)
        owner = np.searchsorted(self.bounds, idx, side="right") - 1
        valid = (idx >= 0) & (idx < self.bounds[-1])
        owner = np.clip(owner, 0, self.parts - 1)
        pos = owner.astype(np.int64) * self.slot + (idx - self.bounds[owner])
        return np.where(valid, pos, -1).astype(np.int32)

    def scatter(self, global_vec, fill):
        """Global vector -> slotted vector (padding = `fill`, flags = 0)."""
        out = np.full(self.length, fill, dtype=global_vec.dtype)
        for k in range(self.parts):
            r0, r1 = self.bounds[k], self.bounds[k + 1]
            out[k * self.slot:k * self.slot + (r1 - r0)] = global_vec[r0:r1]
            out[self.flag_index(k):(k + 1) * self.slot] = 0
        return out

    def gather(self, slotted_vec):
        """Slotted vector -> global vector."""
        parts = [slotted_vec[k * self.slot:k * self.slot + (self.bounds[k + 1] - self.bounds[k])]
                 for k in range(self.parts)]
        return np.concatenate(parts) if parts else slotted_vec[:0]

    def flags(self, slotted_vec):
        return np.array([slotted_vec[self.flag_index(k)] for k in range(self.parts)])
