"""Row-range sharding of a CSR matrix across GPUs.

The reference is single-device; this is the new multi-GPU layer of the hot
path.  Rows are independent, so the matrix is cut into `parts` contiguous row
ranges with ~equal non-zeros (power-law rows make equal-row cuts badly
imbalanced); every rank keeps a full replica of x.

For the iterative apps the result vector of one iteration is the next
iteration's x, so ranks exchange their slices by all-gather every iteration.
To make that a plain equal-size in-place all-gather with no copies, vectors
live in the layout of SlottedLayout (equal-size pieces per rank, a 64-element
tail carrying the rank's "changed" flag), optionally cut into chunks so that
the all-gather of a finished chunk overlaps the computation of the next one;
column indices are remapped once, on the host, from global row ids to layout
positions.
"""
import numpy as np


def row_bounds(row_ptr, parts, cols=None, shard_nnz=None):
    """Boundaries b[0..parts] of contiguous row ranges of ~equal WORK.

    Without `cols`: equal stored entries per range (pure numpy, any offset width: a global matrix with 2^31 entries or
    more -- exactly the kind one shards, since one engine refuses it -- is cut in int64).  With `cols` (the x length)
    the rows are weighted by the HBM bytes the engine expects to move for them (sh_plan_row_work of the C ABI: under
    the x-tiled plan an entry of a heavy row -- pre-reduced inside phase 1 -- costs a third of an entry of a light row;
    see DESIGN.md 3) under the plan it would choose for a matrix of `shard_nnz` entries -- what ONE rank uploads;
    default: the total over `parts` -- so that a shard full of heavy rows does not finish early while the others still
    stream.  The weights describe the x-tiled and CSR-stream plans; a caller whose shards run on another layout (the
    bit-blocked (or,and) plan: 4 B per live entry, no heavy / light split) passes cols=None."""
    row_ptr = np.asarray(row_ptr)
    rows = len(row_ptr) - 1
    if cols is not None and rows > 0:
        import ctypes as C

        from . import abi
        total_nnz = int(row_ptr[-1])
        per_shard = int(shard_nnz) if shard_nnz is not None else -(-total_nnz // max(parts, 1))
        cum = np.zeros(rows + 1, np.int64)
        # sh_plan_row_work takes int32 offsets: hand it blocks of rows whose entries fit, offsets rebased per block
        r = 0
        lib = abi.load()
        while r < rows:
            hi = int(np.searchsorted(row_ptr, int(row_ptr[r]) + (2 ** 31 - 2), side="right")) - 1
            hi = min(max(hi, r + 1), rows)
            if int(row_ptr[hi]) - int(row_ptr[r]) > 2 ** 31 - 2:
                raise ValueError(f"row {r} alone has more than 2^31 - 2 entries")
            blk = np.ascontiguousarray(row_ptr[r:hi + 1].astype(np.int64) - int(row_ptr[r]), dtype=np.int32)
            part = np.zeros(hi - r + 1, np.uint64)
            rc = lib.sh_plan_row_work(hi - r, int(cols), per_shard, blk.ctypes.data_as(C.c_void_p), None,
                                      part.ctypes.data_as(C.c_void_p))
            if rc:
                raise RuntimeError(f"sh_plan_row_work failed: {rc}")
            cum[r + 1:hi + 1] = cum[r] + part[1:].astype(np.int64)
            r = hi
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
    """Mapping between global vector indices and the all-gather layout.

    Every rank's row range is cut into `chunks` pieces of `piece` rows (the longest range / chunks,
    rounded up to 64), and the vector is laid out CHUNK-major: region c holds piece c of rank 0, of
    rank 1, ... -- so the all-gather of one chunk is an ordinary in-place equal-size all-gather over a
    contiguous region, and a rank can gather its finished chunk c while it still computes chunk c + 1.
    The pieces of the LAST region carry a 64-element tail whose first word is the rank's "changed"
    flag: it travels with the last chunk, after every chunk of the iteration has been computed.
    chunks = 1 is the plain slotted layout (rank k owns slot k)."""

    FLAG_PAD = 64  # elements reserved behind each piece of the last region; word 0 = changed flag

    def __init__(self, bounds, chunks=1):
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.parts = len(self.bounds) - 1
        self.chunks = max(1, int(chunks))
        longest = int(np.diff(self.bounds).max()) if self.parts else 0
        self.piece = (-(-longest // self.chunks) + 63) // 64 * 64     # rows per piece
        self.payload = self.piece * self.chunks
        self.slot = self.payload + self.FLAG_PAD                      # elements a rank owns in total
        self.length = self.slot * self.parts

    # ---- geometry
    def piece_len(self, c):
        """Elements one rank contributes to region c."""
        return self.piece + (self.FLAG_PAD if c == self.chunks - 1 else 0)

    def region(self, c):
        """(start, length) of region c = what one all-gather of chunk c covers."""
        return c * self.parts * self.piece, self.parts * self.piece_len(c)

    def piece_offset(self, k, c):
        """Position of row 0 of rank k's piece c."""
        return self.region(c)[0] + k * self.piece_len(c)

    def piece_rows(self, k, c):
        """(first local row, row count) of rank k's piece c (the count may be 0)."""
        n = int(self.bounds[k + 1] - self.bounds[k])
        lo = min(n, c * self.piece)
        return lo, min(n, (c + 1) * self.piece) - lo

    def slot_offset(self, k):
        """chunks == 1 only: where rank k's rows start."""
        assert self.chunks == 1
        return self.piece_offset(k, 0)

    def flag_index(self, k):
        return self.piece_offset(k, self.chunks - 1) + self.piece

    def _pos(self, owner, local):
        c = local // max(self.piece, 1)
        c = np.minimum(c, self.chunks - 1)
        plen = np.where(c == self.chunks - 1, self.piece + self.FLAG_PAD, self.piece)
        return c * self.parts * self.piece + owner * plen + (local - c * self.piece)

    def to_slotted_index(self, idx):
        """Global ids -> layout positions (out-of-range ids stay out of range => identity)."""
        idx = np.asarray(idx)
        owner = np.searchsorted(self.bounds, idx, side="right") - 1
        valid = (idx >= 0) & (idx < self.bounds[-1])
        owner = np.clip(owner, 0, self.parts - 1).astype(np.int64)
        pos = self._pos(owner, idx.astype(np.int64) - self.bounds[owner])
        return np.where(valid, pos, -1).astype(np.int32)

    def scatter(self, global_vec, fill):
        """Global vector -> layout vector (padding = `fill`, flags = 0)."""
        out = np.full(self.length, fill, dtype=global_vec.dtype)
        for k in range(self.parts):
            for c in range(self.chunks):
                lo, n = self.piece_rows(k, c)
                o = self.piece_offset(k, c)
                out[o:o + n] = global_vec[self.bounds[k] + lo:self.bounds[k] + lo + n]
            f = self.flag_index(k)
            out[f:f + self.FLAG_PAD] = 0
        return out

    def gather(self, slotted_vec):
        """Layout vector -> global vector."""
        parts = []
        for k in range(self.parts):
            for c in range(self.chunks):
                lo, n = self.piece_rows(k, c)
                o = self.piece_offset(k, c)
                parts.append(slotted_vec[o:o + n])
        return np.concatenate(parts) if parts else slotted_vec[:0]

    def flags(self, slotted_vec):
        return np.array([slotted_vec[self.flag_index(k)] for k in range(self.parts)])
