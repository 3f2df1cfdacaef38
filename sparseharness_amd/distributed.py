"""Row-sharded multi-GPU driver for the iterative apps (SSSP, BFS): one process
per GPU, ONE all-gather of the new vector per iteration.

The reference iterates on a single device and copies the whole vector to the
host every iteration to test convergence (app/sssp.cpp:97-176).  Here each
rank owns a contiguous, nnz-balanced row range of the matrix and a full
replica of x in the *slotted layout* of partition.SlottedLayout:

    iteration k on rank r:
      1. clear my "changed" word in x_next
      2. local step: x_next[slot r] = kernel(A_r, x_cur, y = x_cur[slot r])
         -- the HIP kernel also raises my changed word (fused convergence test)
      3. all_gather_into_tensor(x_next, x_next[slot r])      # RCCL over xGMI, in place
      4. every rank now holds every rank's changed word: stop when all are 0
      5. swap x_cur / x_next

Step 4 reads `parts` words back to the host (the only PCIe traffic per
iteration).  Results are bit-identical to the single-GPU sh_iterate for every
world size, because each row is reduced by the same code over the same data.

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
    """What one rank needs: its rows of the matrix with columns remapped to slotted positions."""

    def __init__(self, row_ptr, col_idx, val, rank, world):
        self.rank, self.world = rank, world
        self.rows_total = len(row_ptr) - 1
        self.bounds = partition.row_bounds(row_ptr, world, cols=self.rows_total)
        self.layout = partition.SlottedLayout(self.bounds)
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        rp, ci, va = partition.take_rows(row_ptr, col_idx, val, self.r0, self.r1)
        self.row_ptr = np.ascontiguousarray(rp)
        self.col_idx = self.layout.to_slotted_index(ci)
        self.val = np.ascontiguousarray(va)
        self.rows = self.r1 - self.r0


class HipLocalStep:
    """Local step on the GPU through the C ABI (sh_spmv_step) on torch-owned buffers."""

    def __init__(self, plan, semiring, device_index):
        import torch
        from .engine import Engine
        self.torch = torch
        self.plan, self.semiring = plan, semiring
        self.engine = Engine(device_index, stream=torch.cuda.current_stream().cuda_stream)
        self.A = self.engine.upload_csr(plan.rows, plan.layout.length, plan.row_ptr, plan.col_idx,
                                        np.ascontiguousarray(plan.val, _np_dtype(semiring)))
        self.device = torch.device("cuda", device_index)
        self._wrapped = {}   # data_ptr -> engine vector handles (the driver ping-pongs two buffers)

    def _vec(self, ptr, n):
        v = self._wrapped.get((ptr, n))
        if v is None:
            v = self._wrapped[(ptr, n)] = self.engine.wrap(ptr, n)
        return v

    def step(self, x_cur, y_slot, x_next, alpha, beta, delta):
        lay, k = self.plan.layout, self.plan.rank
        off = lay.slot_offset(k)
        x = self._vec(x_cur.data_ptr(), lay.length)
        y = self._vec(y_slot.data_ptr(), self.plan.rows)
        out = self._vec(x_next.data_ptr() + off * 4, self.plan.rows)
        flag = self._vec(x_next.data_ptr() + lay.flag_index(k) * 4, lay.FLAG_PAD)
        flag.fill(0, np.int32)   # clear my changed word (async, same stream)
        self.engine.step(self.semiring, self.A, x, y, alpha, beta, out, x_row_offset=off, delta=delta,
                         changed_ptr=flag.device_ptr)


class ShardedIteration:
    """The do/while of the iterative apps over `world` ranks (torch.distributed must be initialised
    when world > 1).  Tensors live wherever `local.device` says (cuda for HIP, cpu for the tests)."""

    def __init__(self, plan, semiring, local):
        self.plan, self.semiring, self.local = plan, semiring, local

    def run(self, x0, y0, alpha, beta, delta=1e-4, max_iters=10000):
        import torch
        import torch.distributed as dist
        plan, lay = self.plan, self.plan.layout
        k, world = plan.rank, plan.world
        dt = _np_dtype(self.semiring)
        dev = self.local.device
        x0 = np.ascontiguousarray(x0, dt)
        y0 = np.ascontiguousarray(y0, dt)
        fill = x0.dtype.type(0)
        x_cur = torch.from_numpy(lay.scatter(x0, fill)).to(dev)
        x_next = torch.zeros_like(x_cur)
        y_first = torch.from_numpy(np.ascontiguousarray(y0[plan.r0:plan.r1])).to(dev)
        off, flag_i = lay.slot_offset(k), lay.flag_index(k)
        flag_idx = torch.tensor([lay.flag_index(j) for j in range(world)], device=dev, dtype=torch.long)
        flags_dev = torch.zeros(world, dtype=torch.int32, device=dev)
        flags_host = torch.zeros(world, dtype=torch.int32)
        if dev.type == "cuda":
            flags_host = flags_host.pin_memory()
        clears_own_flag = isinstance(self.local, HipLocalStep)
        iters, converged = 0, False
        import time
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t_loop = time.perf_counter()
        while iters < max_iters:
            if not clears_own_flag:
                x_next[flag_i:flag_i + lay.FLAG_PAD] = 0
            y_slot = y_first if iters == 0 else x_cur[off:off + max(plan.rows, 1)]
            self.local.step(x_cur, y_slot, x_next, alpha, beta, delta)
            if world > 1:
                mine = x_next[off:off + lay.slot]
                if dev.type == "cpu":
                    mine = mine.clone()   # gloo does not take an input aliasing the output
                dist.all_gather_into_tensor(x_next, mine)
            torch.index_select(x_next.view(torch.int32), 0, flag_idx, out=flags_dev)
            flags_host.copy_(flags_dev, non_blocking=True)
            if dev.type == "cuda":
                torch.cuda.current_stream().synchronize()
            iters += 1
            x_cur, x_next = x_next, x_cur
            if not bool(flags_host.any()):
                converged = True
                break
        self.last_loop_seconds = time.perf_counter() - t_loop   # iterations only (setup/readback excluded)
        final = lay.gather(x_cur.cpu().numpy())
        return final, iters, converged
