"""Row-sharded multi-GPU driver for the iterative apps (SSSP, BFS): one process
per GPU, the new vector exchanged by all-gather every iteration.

The reference iterates on a single device and copies the whole vector to the
host every iteration to test convergence (app/sssp.cpp:97-176).  Here each
rank owns a contiguous, work-balanced row range of the matrix and a full
replica of x in the layout of partition.SlottedLayout, its rows cut into
`chunks` pieces:

    iteration k on rank r:
      1. clear my "changed" word in x_next
      2. for each chunk c:
           local step: x_next[piece c of r] = kernel(A_r,c, x_cur, y = x_cur[piece c of r])
             -- the HIP kernel also raises my changed word (fused convergence test)
           all_gather_into_tensor(region c of x_next, my piece c), asynchronously
             -- RCCL over xGMI, in place; it waits for the step just enqueued and runs
                on the collective's own stream WHILE chunk c + 1 is computed, so only
                the last chunk's all-gather (1/chunks of the bytes) is exposed
      3. wait for the all-gathers; every rank now holds every rank's changed word
         (it travels behind the last chunk): stop when all are 0
      4. swap x_cur / x_next

Step 3 reads `parts` words back to the host (the only PCIe traffic per
iteration) and waits for exactly that copy.  Results are bit-identical to the
single-GPU sh_iterate for every world size and chunk count, because each row
is reduced by the same code over the same data.

`LocalStep` is the seam between this driver and the device: `HipLocalStep`
(the product) calls the C ABI on torch-owned device memory; the CPU tests plug
the oracle in instead to exercise the sharding + collective logic under gloo.
"""
import numpy as np

from . import partition
from .abi import MAX_MIN_I32, MIN_PLUS_F32, OR_AND_I32, PLUS_TIMES_F32  # noqa: F401


def _torch_dtype(semiring):
    import torch
    return torch.int32 if semiring in (OR_AND_I32, MAX_MIN_I32) else torch.float32


def _np_dtype(semiring):
    return np.int32 if semiring in (OR_AND_I32, MAX_MIN_I32) else np.float32


class ShardPlan:
    """What one rank needs: its rows of the matrix, cut into `chunks` pieces, columns remapped to layout positions."""

    def __init__(self, row_ptr, col_idx, val, rank, world, chunks=1):
        self.rank, self.world = rank, world
        self.rows_total = len(row_ptr) - 1
        self.bounds = partition.row_bounds(row_ptr, world, cols=self.rows_total)
        self.layout = partition.SlottedLayout(self.bounds, chunks)
        self.chunks = self.layout.chunks
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.rows = self.r1 - self.r0
        # per chunk: (row_ptr, col_idx in layout positions, val, rows)
        self.pieces = []
        for c in range(self.chunks):
            lo, n = self.layout.piece_rows(rank, c)
            rp, ci, va = partition.take_rows(row_ptr, col_idx, val, self.r0 + lo, self.r0 + lo + n)
            self.pieces.append((np.ascontiguousarray(rp), self.layout.to_slotted_index(ci), np.ascontiguousarray(va), n))
        # the whole shard as one matrix (chunks == 1 callers, tests)
        self.row_ptr, self.col_idx, self.val = self.pieces[0][:3] if self.chunks == 1 else (None, None, None)


class HipLocalStep:
    """Local step on the GPU through the C ABI (sh_spmv_step) on torch-owned buffers; one matrix per chunk."""

    def __init__(self, plan, semiring, device_index):
        import torch
        from .engine import Engine
        self.torch = torch
        self.plan, self.semiring = plan, semiring
        self.engine = Engine(device_index, stream=torch.cuda.current_stream().cuda_stream)
        self.mats = [self.engine.upload_csr(n, plan.layout.length, rp, ci, np.ascontiguousarray(va, _np_dtype(semiring)))
                     for rp, ci, va, n in plan.pieces]
        self.A = self.mats[0]
        self.device = torch.device("cuda", device_index)
        self._wrapped = {}   # data_ptr -> engine vector handles (the driver ping-pongs two buffers)

    def _vec(self, ptr, n):
        v = self._wrapped.get((ptr, n))
        if v is None:
            v = self._wrapped[(ptr, n)] = self.engine.wrap(ptr, n)
        return v

    def step(self, c, x_cur, y_piece, x_next, alpha, beta, delta):
        """Chunk c of this rank: x_next[piece c] = kernel(A_c, x_cur, y_piece); raises the rank's changed word."""
        lay, k = self.plan.layout, self.plan.rank
        rows = self.plan.pieces[c][3]
        flag = self._vec(x_next.data_ptr() + lay.flag_index(k) * 4, lay.FLAG_PAD)
        if c == 0:
            flag.fill(0, np.int32)   # clear my changed word (async, same stream)
        if rows == 0:
            return
        off = lay.piece_offset(k, c)
        x = self._vec(x_cur.data_ptr(), lay.length)
        y = self._vec(y_piece.data_ptr(), rows)
        out = self._vec(x_next.data_ptr() + off * 4, rows)
        self.engine.step(self.semiring, self.mats[c], x, y, alpha, beta, out, x_row_offset=off, delta=delta,
                         changed_ptr=flag.device_ptr)


class ShardedIteration:
    """The do/while of the iterative apps over `world` ranks (torch.distributed must be initialised
    when world > 1).  Tensors live wherever `local.device` says (cuda for HIP, cpu for the tests)."""

    def __init__(self, plan, semiring, local):
        self.plan, self.semiring, self.local = plan, semiring, local

    def run(self, x0, y0, alpha, beta, delta=1e-4, max_iters=10000):
        import time

        import torch
        import torch.distributed as dist
        plan, lay = self.plan, self.plan.layout
        k, world, chunks = plan.rank, plan.world, plan.chunks
        dt = _np_dtype(self.semiring)
        dev = self.local.device
        x0 = np.ascontiguousarray(x0, dt)
        y0 = np.ascontiguousarray(y0, dt)
        fill = x0.dtype.type(0)
        x_cur = torch.from_numpy(lay.scatter(x0, fill)).to(dev)
        x_next = torch.zeros_like(x_cur)
        y_first = torch.from_numpy(np.ascontiguousarray(y0[plan.r0:plan.r1])).to(dev)
        flag_i = lay.flag_index(k)
        flag_idx = torch.tensor([lay.flag_index(j) for j in range(world)], device=dev, dtype=torch.long)
        flags_dev = torch.zeros(world, dtype=torch.int32, device=dev)
        flags_host = torch.zeros(world, dtype=torch.int32)
        copied = None
        if dev.type == "cuda":
            flags_host = flags_host.pin_memory()
            copied = torch.cuda.Event()
        clears_own_flag = isinstance(self.local, HipLocalStep)
        iters, converged = 0, False
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t_loop = time.perf_counter()
        while iters < max_iters:
            if not clears_own_flag:
                x_next[flag_i:flag_i + lay.FLAG_PAD] = 0
            pending = []
            for c in range(chunks):
                lo, n = lay.piece_rows(k, c)
                off = lay.piece_offset(k, c)
                y_piece = y_first[lo:lo + max(n, 1)] if iters == 0 else x_cur[off:off + max(n, 1)]
                self.local.step(c, x_cur, y_piece, x_next, alpha, beta, delta)
                if world > 1:
                    start, length = lay.region(c)
                    region = x_next[start:start + length]
                    mine = region[k * lay.piece_len(c):(k + 1) * lay.piece_len(c)]
                    if dev.type == "cpu":
                        mine = mine.clone()   # gloo does not take an input aliasing the output
                    # asynchronous: ordered behind the step just enqueued, concurrent with the next chunk's step
                    pending.append(dist.all_gather_into_tensor(region, mine, async_op=True))
            for w in pending:
                w.wait()
            torch.index_select(x_next.view(torch.int32), 0, flag_idx, out=flags_dev)
            flags_host.copy_(flags_dev, non_blocking=True)
            if copied is not None:
                copied.record()
                copied.synchronize()   # exactly the flag words, not everything else on the device
            iters += 1
            x_cur, x_next = x_next, x_cur
            if not bool(flags_host.any()):
                converged = True
                break
        self.last_loop_seconds = time.perf_counter() - t_loop   # iterations only (setup/readback excluded)
        final = lay.gather(x_cur.cpu().numpy())
        return final, iters, converged
